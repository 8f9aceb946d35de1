"""ctypes binding of libpyro_amd.so (C-ABI declared in include/pyro_amd.h).

The library is loaded lazily on first use.  There is NO fallback: if the shared library
is missing, or a tensor handed to a kernel wrapper does not live on a HIP device, a
RuntimeError is raised -- the product path never silently runs on the CPU.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_uint32,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpyro_amd.so")

PA_OK, PA_ERR_INVALID, PA_ERR_UNSUPPORTED, PA_ERR_LAUNCH = 0, -1, -2, -3
PA_F32, PA_F64 = 0, 1
ABI_VERSION = 8      # PA_ABI_VERSION of include/pyro_amd.h

DIST_NORMAL = 0
DIST_BERNOULLI_LOGITS = 1
DIST_HALF_CAUCHY = 2
DIST_LOG_NORMAL = 3
DIST_EXPONENTIAL = 4
DIST_HALF_NORMAL = 5
DIST_GAMMA = 6
DIST_BETA = 7
DIST_POISSON = 8
DIST_BINOMIAL_LOGITS = 9
DIST_KL_NORMAL_LOC = 10
DIST_KL_NORMAL_SCALE = 11

KERNEL_GLM, KERNEL_NUTS, KERNEL_LDA, KERNEL_SITE_SUM = 1, 2, 3, 4


class View2D(Structure):
    _fields_ = [("ptr", c_void_p), ("stride_row", c_int64), ("stride_col", c_int64)]


NULL_VIEW = View2D(None, 0, 0)

SITE_IDENTITY = 100
SITE_NONE = 101
NEED_VALUE, NEED_P0, NEED_P1, VALUE_BY_CHAIN = 1, 2, 4, 16
MULTI_MAX_ENTRIES = 16
MULTI_MAX_ELEMS = 65536
MF_MAX_SITES = 16
CHAIN_SYNC_BYTES = 64


class SiteEntry(Structure):
    """pa_site_entry (include/pyro_amd.h)."""
    _fields_ = [("dist", c_int32), ("need", c_int32), ("rows", c_int64), ("cols", c_int64),
                ("value", View2D), ("p0", View2D), ("p1", View2D), ("mask", View2D),
                ("coef", c_double), ("d_value", c_void_p), ("d_p0", c_void_p), ("d_p1", c_void_p),
                ("chain_next", c_int32), ("reserved", c_int32), ("extra_grad", c_void_p),
                ("extra_coef", c_double)]


LSE_MAX_TERMS, LSE_MAX_DIMS = 4, 6


class LseTerm(Structure):
    """pa_lse_term (include/pyro_amd.h)."""
    _fields_ = [("ptr", c_void_p), ("strides", c_int64 * LSE_MAX_DIMS)]


class MfSite(Structure):
    """pa_mf_site (include/pyro_amd.h)."""
    _fields_ = [("loc", c_void_p), ("rho", c_void_p), ("z", c_void_p), ("scale", c_void_p),
                ("loc_out", c_void_p), ("eps", c_void_p), ("n", c_int64), ("offset", c_uint64),
                ("accumulate", c_int32), ("reserved", c_int32), ("d_z", c_void_p), ("d_scale", c_void_p), ("d_loc_out", c_void_p),
                ("d_loc", c_void_p), ("d_rho", c_void_p)]


class Unsupported(RuntimeError):
    """Raised when a fused kernel does not cover the requested shape (PA_ERR_UNSUPPORTED)."""


_SIGNATURES = {
    "pa_abi_version": (c_int, []),
    "pa_last_error": (c_char_p, []),
    "pa_device_cu_count": (c_int, []),
    "pa_profile_bracket_next": (c_int, [c_int, c_void_p, c_void_p]),
    "pa_philox_normal": (c_int, [c_void_p, c_int64, c_int, c_uint64, c_uint64, c_void_p, c_void_p]),
    "pa_philox_uniform": (c_int, [c_void_p, c_int64, c_int, c_uint64, c_uint64, c_void_p, c_void_p]),
    "pa_counter_add": (c_int, [c_void_p, c_uint64, c_void_p]),
    "pa_publish_scalar": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p]),
    "pa_exp_site_fwd": (c_int, [c_int, c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p]),
    "pa_exp_site_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p,
                                c_void_p]),
    "pa_meanfield_score_blocks": (c_int64, [c_int64, c_int64]),
    "pa_meanfield_score": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p,
                                   c_void_p, c_void_p]),
    "pa_gate_defer": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "pa_gate_defer_stats": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pa_gate": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "pa_gate_scope": (c_int, [c_void_p]),
    "pa_gate_stats": (c_int, [c_void_p, c_void_p]),
    "pa_dist_log_prob": (c_int, [c_int, c_int, c_void_p, View2D, View2D, View2D, c_int64, c_int64,
                                 c_void_p]),
    "pa_dist_log_prob_sum_workspace": (c_size_t, [c_int64, c_int64]),
    "pa_dist_log_prob_sum": (c_int, [c_int, c_int, c_void_p, c_void_p, View2D, View2D, View2D, View2D,
                                     c_double, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "pa_dist_log_prob_grad": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, View2D, View2D,
                                      View2D, View2D, View2D, c_double, c_int64, c_int64,
                                      c_void_p]),
    "pa_normal_rsample": (c_int, [c_int, c_void_p, c_void_p, View2D, View2D, c_int64, c_int64,
                                  c_uint64, c_uint64, c_void_p, c_void_p]),
    "pa_multi_log_prob_sum": (c_int, [c_int, c_void_p, POINTER(SiteEntry), c_int, c_double, c_int,
                                      c_void_p]),
    "pa_multi_log_prob_grad": (c_int, [c_int, c_void_p, POINTER(SiteEntry), c_int, c_double,
                                       c_void_p]),
    "pa_multi_log_prob_sum_grad": (c_int, [c_int, c_void_p, c_void_p, POINTER(SiteEntry), c_int,
                                           c_double, c_int, c_void_p]),
    "pa_meanfield_normal_sample": (c_int, [c_int, POINTER(MfSite), c_int, c_int64, c_uint64,
                                           c_void_p, c_void_p]),
    "pa_meanfield_normal_sample_bwd": (c_int, [c_int, POINTER(MfSite), c_int, c_int64, c_void_p]),
    "pa_glm_set_variant": (c_int, [c_int]),
    "pa_glm_chain": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                             c_void_p]),
    "pa_glm_bernoulli_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_glm_bernoulli_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_double, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_size_t, c_void_p]),
    "pa_tall_linear": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                               c_void_p, c_void_p]),
    "pa_tall_linear_act": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                   c_void_p, c_int, c_void_p, c_void_p]),
    "pa_tall_wgrad_act": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "pa_tall_wgrad_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_tall_wgrad": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                              c_size_t, c_void_p]),
    "pa_glm_planes_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "pa_glm_grouped_planes_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "pa_glm_pack_planes_grouped": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                           c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "pa_group_rows_workspace": (c_size_t, [c_int64, c_int64]),
    "pa_group_rows_build": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "pa_glm_pack_planes_grouped_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                                c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_size_t,
                                                c_void_p]),
    "pa_glm_bernoulli_grouped_planes_workspace": (c_size_t, [c_int64, c_int64]),
    "pa_glm_bernoulli_grouped_planes_fwd_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_double, c_int64,
                                                        c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                                        c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "pa_glm_pack_planes": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "pa_glm_planes_tune": (c_int, [c_int, c_int]),
    "pa_glm_planes_finalize_mode": (c_int, [c_int]),
    "pa_glm_planes_stamps": (c_int, [c_void_p]),
    "pa_glm_bernoulli_planes_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_glm_bernoulli_planes_fwd_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                                c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pa_glm_label_moments_workspace": (c_size_t, [c_int64]),
    "pa_glm_label_moments": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "pa_glm_bernoulli_grouped_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_glm_bernoulli_grouped_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_double, c_int64, c_int64, c_int64, c_int64,
                                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pa_leapfrog_kick_drift": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "pa_leapfrog_kick": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                 c_void_p]),
    "pa_nuts_gaussian_transition": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                            c_uint64, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "pa_nuts_gaussian_run": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int64, c_int64, c_int, c_int, c_uint64, c_uint64,
                                     c_int64, c_uint64, c_void_p, c_double, c_void_p, c_int64,
                                     c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pa_nuts_gaussian_set_variant": (c_int, [c_int]),
    "pa_nuts_tree_workspace": (c_size_t, [c_int, c_int64, c_int64, c_int]),
    "pa_nuts_tree_begin": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                   c_uint64, c_uint64, c_uint64, c_void_p, c_size_t, c_void_p]),
    "pa_nuts_tree_advance": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                     c_int64, c_int, c_int, c_uint64, c_uint64, c_uint64,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "pa_nuts_tree_advance_tdev": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                          c_int64, c_int, c_int, c_uint64, c_void_p, c_uint64,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "pa_nuts_tree_run_begin": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                       c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "pa_nuts_tree_run_advance": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_int64, c_int, c_int, c_uint64, c_uint64, c_void_p,
                                         c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_size_t, c_void_p]),
    "pa_nuts_tree_compact": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64,
                                     c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pa_nuts_tree_run_advance_direct": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_uint64,
                                                c_uint64, c_void_p, c_void_p, c_double, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_int64,
                                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_size_t, c_void_p]),
    "pa_nuts_direct_potential": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "pa_rtc_compile": (c_int, [c_char_p, c_char_p, c_void_p]),
    "pa_rtc_compile_cached": (c_int, [c_char_p, c_char_p, c_char_p, c_void_p, c_void_p]),
    "pa_rtc_version": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pa_rtc_blocks_begin": (c_void_p, []),
    "pa_rtc_blocks_end": (c_int, [c_void_p, c_void_p]),
    "pa_rtc_blocks_free": (c_int, [c_void_p]),
    "pa_mixture_workspace": (c_size_t, [c_int, c_int64]),
    "pa_mixture_fwd_bwd": (c_int, [c_int, c_int, c_void_p, c_int64, c_int, c_int64, c_void_p, c_int64, c_void_p,
                                   c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p,
                                   c_void_p]),
    "pa_mixture_diag_normal_layout": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "pa_mixture_diag_normal_workspace": (c_size_t, [c_int, c_int, c_int64]),
    "pa_mixture_diag_normal_fwd_bwd": (c_int, [c_int, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_int64,
                                               c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                               c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pa_graph_direct_plan": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int)]),
    "pa_graph_direct_launch": (c_int, [c_void_p, c_void_p]),
    "pa_graph_direct_free": (c_int, [c_void_p]),
    "pa_rtc_launch": (c_int, [c_void_p, c_uint32, c_uint32, c_void_p, c_int, c_void_p]),
    "pa_nuts_gaussian_find_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int64, c_int64, c_uint64, c_uint64, c_uint64,
                                           c_double, c_double, c_double, c_void_p]),
    "pa_lda_factor_workspace": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "pa_lda_factor_fwd_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                      c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "pa_dirichlet_log_prob": (c_int, [c_int, c_void_p, View2D, View2D, c_int64, c_int64, c_void_p]),
    "pa_dirichlet_log_prob_grad": (c_int, [c_int, c_void_p, View2D, View2D, c_int64, c_int64,
                                           c_void_p, c_void_p, c_void_p]),
    "pa_logsumexp_terms": (c_int, [c_int, c_void_p, c_int, POINTER(LseTerm), c_int, POINTER(c_int64),
                                   c_int, c_void_p]),
    "pa_logsumexp_terms_grad": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, POINTER(LseTerm),
                                        c_int, POINTER(c_int64), c_int, c_void_p]),
    "pa_lda_index_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_lda_index_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_lda_build_index": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_size_t,
                                   c_void_p, c_size_t, c_void_p]),
    "pa_lda_factor_indexed_workspace": (c_size_t, [c_int, c_int64, c_int64, c_int64, c_int64]),
    "pa_lda_factor_indexed_fwd_bwd": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                              c_void_p, c_int64, c_int64, c_int64, c_int64,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                              c_void_p]),
    "pa_adam_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double,
                             c_double, c_double, c_double, c_double, c_double, c_double, c_int,
                             c_void_p, c_int, c_void_p]),
    "pa_dist_log_prob_sum_nd_workspace": (c_size_t, []),
    "pa_dist_log_prob_sum_nd": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_double, c_void_p, c_size_t, c_void_p]),
    "pa_dist_log_prob_grad_nd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_double, c_void_p]),
    "pa_sum_to_nd_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_sum_to_nd": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                             c_size_t, c_void_p]),
    "pa_sum_to_nd_pair": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                  c_void_p, c_size_t, c_void_p]),
    "pa_mvn_tril_sample": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_uint64,
                                   c_uint64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pa_mvn_tril_sample_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p]),
    "pa_logchain_workspace": (c_size_t, [c_int, c_int64, c_int64, c_int64]),
    "pa_logchain_fwd_bwd": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "pa_chain_matvec": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                c_void_p]),
    "pa_bow_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_bow_linear_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "pa_bow_linear_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "pa_bow_linear_fwd_act": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "pa_bow_linear_bwd_act": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "pa_tsgemm_tn_workspace": (c_size_t, [c_int64, c_int64, c_int64]),
    "pa_tsgemm_tn": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_size_t,
                             c_void_p]),
    "pa_gamma_rsample": (c_int, [c_int, c_void_p, c_void_p, View2D, c_int64, c_int64, c_uint64, c_uint64,
                                 c_void_p, c_void_p]),
    "pa_gamma_implicit_grad": (c_int, [c_int, c_void_p, View2D, View2D, c_int64, c_int64, c_void_p]),
    "pa_chain_begin": (c_int, [c_void_p, c_void_p, c_size_t]),
    "pa_chain_flush": (c_int, []),
    "pa_chain_end": (c_int, [POINTER(c_int), POINTER(c_int)]),
    "pa_chain_pending": (c_int, []),
    "pa_chain_debug_stamps": (c_int, [c_void_p]),
    "pa_chain_tune": (c_int, [c_int]),
    "pa_chain_fused_launches": (c_int, []),
    "pa_adam_step_publish": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_double, c_double, c_double, c_double, c_double, c_double,
                                     c_double, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_uint64, c_void_p]),
}

_lib = None


def exported_symbols():
    """Names of every entry point the binding expects (mirrors include/pyro_amd.h)."""
    return sorted(_SIGNATURES)


def load(path=None):
    """Load the shared library (idempotent). Raises RuntimeError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("PYRO_AMD_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise RuntimeError(
            "pyro_amd: HIP extension %s not found. Build it with "
            "`python -m pyro_amd.csrc.build` (hipcc --offload-arch=gfx950). There is no CPU "
            "fallback." % path)
    import torch  # noqa: F401  (loads libamdhip64.so.7 first so we share torch's HIP runtime)
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> missing symbol, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.pa_abi_version() != ABI_VERSION:
        raise RuntimeError("pyro_amd: ABI version mismatch: %d" % lib.pa_abi_version())
    if os.environ.get("PYRO_AMD_GLM_FINALIZE") == "in_kernel":     # developer A/B switch
        lib.pa_glm_planes_finalize_mode(1)
    _lib = lib
    return lib


def check(rc):
    if rc == PA_OK:
        return
    msg = load().pa_last_error().decode("utf-8", "replace")
    if rc == PA_ERR_INVALID:
        raise ValueError("pyro_amd: " + msg)
    if rc == PA_ERR_UNSUPPORTED:
        raise Unsupported("pyro_amd: " + msg)
    raise RuntimeError("pyro_amd: kernel launch failed: " + msg)
