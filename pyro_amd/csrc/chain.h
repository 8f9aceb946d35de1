// chain.h -- recording side of the CHAINED TAIL of an SVI step (chain.hip).
//
// Between pa_chain_begin() and pa_chain_end() the small dependent launches that end an
// ELBO-gradient step -- the fp64 reduction of the GLM kernel's partial records, the one-launch ELBO
// assembly (pa_multi_log_prob_sum_grad), the backward of the mean-field guide draw and the flat
// Adam update -- are not launched one by one: their entry points RECORD their arguments here and
// pa_chain_end() (or anything else that launches on the library's behalf: as_stream() flushes)
// launches ONE kernel that runs them as phases separated by device-wide barriers.  A dependent
// launch inside a captured hipGraph costs ~5 us of dispatch whatever it computes; a phase boundary
// costs one release/acquire round trip.
#pragma once
#include "common.h"

namespace pa {

struct MultiArgs;
struct MfArgs;
struct AdamPublish;

// Each returns 1 when the call was recorded (the caller must NOT launch), 0 when the caller has to
// launch itself (not recording, other stream, not eligible), < 0 = error code.
int chain_record_fin(pa_stream_t stream, int DT, int PT, const float* part, int nblocks, int npass,
                     int D, int P, double scale, float* ll, float* gw, float* gb, double ll_offset);
int chain_record_multi(pa_stream_t stream, const MultiArgs& args, float* out, const float* g,
                       double coef_all, int accumulate);
int chain_record_mf_bwd(pa_stream_t stream, const MfArgs& args, int nsites, int64_t P, int gy);

// The guide draw in FRONT of the step's big kernel: inside a recording, pa_meanfield_normal_sample (f32)
// parks its launch here; a plane-image GLM launch whose weights / bias ARE two of the parked sites'
// draws (w == z of a site with n == D, b == z of a site with n == 1, the same particle count) takes it
// (glm_take_pending_draw) and draws them in its own prologue -- one graph node and ~5 us less per step;
// anything else that launches (as_stream) or a flush launches the parked draw first, as it was.
struct GlmDraw {               // what the GLM kernel needs to draw w[P,D] and b[P] itself
  const float *loc_w, *rho_w, *loc_b, *rho_b;
  float *z_w, *eps_w, *scale_w, *lout_w, *z_b, *eps_b, *scale_b, *lout_b;
  uint64_t seed, off_w, off_b;
  const uint64_t* offset_dev;
  int have_b;
};
// 1: recorded (the caller must not launch); 0: launch as usual
int chain_park_draw(pa_stream_t stream, const MfArgs& args, int nsites, int64_t P, uint64_t seed,
                    const uint64_t* offset_dev);
// true: `out` describes the parked draw and nothing is pending any more.  When the parked launch
// holds further sites they are launched here (before the GLM kernel), without the two taken ones
bool glm_take_pending_draw(pa_stream_t stream, const float* w, const float* b, int64_t P, int64_t D,
                           GlmDraw* out);
int chain_flush_draw();

int chain_record_adam(pa_stream_t stream, float* p, float* g, float* m, float* v, int64_t n,
                      double lr, double b1, double b2, double eps, double wd, double clip,
                      double lrd, int clipped, int64_t* step_dev, int zero_grad,
                      const AdamPublish& pub);

}  // namespace pa
