// glm_planes.h -- the fused Bernoulli-logits GLM pass (forward + gradient) on the bf16 matrix cores
// with the design matrix kept in HBM as its three bf16 planes (included by glm.hip; same record
// format and finalize kernel as the other GLM kernels).
//
// The reference recomputes everything of size [P, N] every ELBO-gradient step
// (pyro/poutine/trace_struct.py:264-278 + autograd duals); X itself never changes between steps.
// pa_glm_pack_planes splits X ONCE into the exact 3-way bf16 decomposition x = x1 + x2 + x3 of
// glm_bf16.h (same rounding, same bits) and stores it as a TILE IMAGE: per 32-row tile three planes
// [32 rows][32 cols] bf16 of 2048 B each, the 16-byte slots of a row XOR-swizzled so that the image
// can be copied verbatim into LDS (buffer/global_load ... lds: 64 lanes x 16 B, lane-linear, no
// VGPRs, no ds_write) and then read conflict-free both as rows (ds_read_b128: A operand of
// GEMM1) and as columns (ds_read_b64_tr_b16: B operand of GEMM2).
//
// Work decomposition (P <= 64 particles per pass, NPT = 1 or 2 particle tiles of 32):
//   workgroup = 4 waves = NRT row tiles x NPT particle tiles of one SUPER-TILE of 32*NRT rows;
//   a wave owns 32 rows x 32 particles: 13 + 12 bf16 MFMAs and 16 accumulator elements per lane and
//   tile, <= 168 VGPRs => 3 waves per SIMD (the r01 kernel: 64 particles per wave, 2 waves per SIMD).
//   The super-tile images travel HBM -> LDS through an NB-deep ring, NB-1 tiles ahead, one raw
//   s_barrier per tile; waits are counted (s_waitcnt vmcnt(N)), never drained in the loop.
#pragma once
#include "glm_bf16.h"
#include "lds_dma.h"

namespace pa {

constexpr int GLMP_PLANE = 2048;            // bytes of one plane of a 32-row tile
constexpr int GLMP_TILE = 3 * GLMP_PLANE;   // bytes of one 32-row tile image
constexpr int GLMP_PAD_TILES = 4;           // the image is padded to whole 128-row groups
constexpr float GLMP_LOG2E = 1.44269504088896340736f;

// byte offset of (row r, 16-byte slot s = 8 columns) inside one plane
__host__ __device__ constexpr int glmp_slot_ofs(int r, int s) {
  return r * 64 + ((s ^ ((r >> 2) & 3)) << 4);
}

// one thread per (tile, row, slot): 8 consecutive features of one row -> 3 x 16 B
__global__ __launch_bounds__(256) void glm_pack_planes_kernel(const float* __restrict__ X, int64_t N,
                                                              int D, int64_t ntiles,
                                                              unsigned char* __restrict__ img) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntiles * 128) return;
  const int64_t T = idx >> 7;
  const int r = (int)(idx >> 2) & 31, s = (int)idx & 3;
  const int64_t row = T * 32 + r;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 8 * s + j;
    v[j] = (row < N && d < D) ? X[row * D + d] : 0.0f;
  }
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p1[j], p2[j], p3[j]);
  unsigned char* q = img + T * GLMP_TILE + glmp_slot_ofs(r, s);
  *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  *reinterpret_cast<uint4*>(q + GLMP_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  *reinterpret_cast<uint4*>(q + 2 * GLMP_PLANE) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}

// The hierarchical variant (BASELINE config 5): rows are sorted by group and cut into segments
// {row_begin, row_end, group} (kernels.GroupSegments); in the image every segment starts on a
// super-tile (64-row) boundary -- segment s owns the super-tiles [st_off[s], st_off[s + 1]), its last
// one padded with zero rows -- so that a workgroup streams one segment with one group's weights.
// The observations travel in the same padded row order (y_img: zeros in the padding).
__global__ __launch_bounds__(256) void glm_pack_planes_grouped_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const int64_t* __restrict__ row_of, int D,
    const int64_t* __restrict__ seg, const int64_t* __restrict__ st_off, int nseg, int64_t ntiles,
    unsigned char* __restrict__ img, float* __restrict__ y_img) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ntiles * 128) return;
  const int64_t T = idx >> 7, st = T >> 1;
  const int r = (int)(idx >> 2) & 31, sl = (int)idx & 3;
  // the segment that owns super-tile st: the last s with st_off[s] <= st
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (st_off[mid] <= st) lo = mid;
    else hi = mid - 1;
  }
  const int64_t a = seg[3 * lo], e = seg[3 * lo + 1];
  const int64_t pos = a + (T - 2 * st_off[lo]) * 32 + r;
  const bool ok = pos < e;
  // row_of (pa_group_rows_build): the image's row `pos` is the data's row row_of[pos] (unsorted ids)
  const int64_t row = (ok && row_of) ? row_of[pos] : pos;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 8 * sl + j;
    v[j] = (ok && d < D) ? X[row * D + d] : 0.0f;
  }
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p1[j], p2[j], p3[j]);
  unsigned char* q = img + T * GLMP_TILE + glmp_slot_ofs(r, sl);
  *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  *reinterpret_cast<uint4*>(q + GLMP_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  *reinterpret_cast<uint4*>(q + 2 * GLMP_PLANE) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
  if (sl == 0) y_img[T * 32 + r] = ok ? y[row] : 0.0f;
}

typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));

// split_pair of glm_bf16.h written on scalars (same roundings, same bits): two v_sub_f32 per level
// instead of one v_pk_add_f32
__device__ __forceinline__ void split_pair_scalar(float a, float b, uint32_t& p1, uint32_t& p2,
                                                  uint32_t& p3) {
  p1 = cvt_pk_bf16(a, b);
  const float ra = a - bf16_lo(p1), rb = b - bf16_hi(p1);
  p2 = cvt_pk_bf16(ra, rb);
  const float qa = ra - bf16_lo(p2), qb = rb - bf16_hi(p2);
  p3 = cvt_pk_bf16(qa, qb);
}
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int NPT, int NB>
struct GlmPlCfg {
  static constexpr int NRT = 4 / NPT;                  // row tiles per super-tile
  static constexpr int ST_BYTES = NRT * GLMP_TILE;     // super-tile image
  static constexpr int PW = NRT * 6 / 4;               // 1 KiB DMA pieces per wave and super-tile
  static constexpr int NDMA = PW + 1;                  // + the wave's 32 observations
  static constexpr int WROWS = 32 * NPT;
  static constexpr int WPL = WROWS * 64;               // one W plane
  static constexpr int OFS_WAUX = 3 * WPL;
  static constexpr int OFS_RING = OFS_WAUX + WROWS * 8;
  static constexpr int OFS_Y = OFS_RING + NB * ST_BYTES;
  static constexpr int LDS_BYTES = OFS_Y + NB * 4 * 256;
};

// Exact 3-way split of a pair by TRUNCATION: piece 1 = the upper 16 bits of the f32 (8 significant
// bits), the residual (exact in f32, <= 16 significant bits) again, the second residual (<= 8
// significant bits) is a bf16 itself.  a = a1 + a2 + a3 exactly for every finite a (subnormal
// residuals included).  Two v_and + two v_sub per level on the fast VALU path and one v_perm_b32 to
// pack a pair's upper halves; the round-to-nearest split of glm_bf16.h needs v_cvt_pk + a shift
// (both on the slow path) per level.
__device__ __forceinline__ uint32_t pack_hi16(float a, float b) {   // {bf16 bits of a, of b}
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a),
                               0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float a) {
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, a) & 0xffff0000u);
}
__device__ __forceinline__ void split_pair_trunc(float a, float b, uint32_t& p1, uint32_t& p2,
                                                 uint32_t& p3) {
  p1 = pack_hi16(a, b);
  const float ra = a - trunc_bf16(a), rb = b - trunc_bf16(b);
  p2 = pack_hi16(ra, rb);
  const float qa = ra - trunc_bf16(ra), qb = rb - trunc_bf16(rb);
  p3 = pack_hi16(qa, qb);
}

// tools/probes/glm_planes_probe: per-phase shader-clock stamps of one wave per workgroup
#ifdef PA_GLMP_STAMP
#define PA_STAMP(i)                                                              \
  do {                                                                           \
    const uint64_t now_ = __builtin_readcyclecounter();                          \
    stamp_acc[i] += now_ - stamp_last;                                           \
    stamp_last = now_;                                                           \
  } while (0)
#else
#define PA_STAMP(i) do { } while (0)
#endif

// ---- in-kernel finalize -------------------------------------------------------------------------
// The stand-alone finalize kernel (glm_finalize.h) reads the 768 partial records (6.7 MB at the
// headline size) after this kernel has ended: ~10 us of a 100 us step, most of it waiting.  Here the
// same sums are formed by the workgroups themselves as they finish, two levels, never blocking:
//   level 1: record i belongs to group i % 32; the LAST workgroup of a group to arrive (device-wide
//            ticket) sums the group's records, in increasing record order, into an fp64 partial;
//   level 2: the last of those group finishers sums the 32 partials in group order, scales, and
//            writes ll / gw / gb.
// That is exactly the arithmetic of glm_finalize_kernel (its thread (j, s) sums the records
// s, s + 32, ... in fp64 and the 32 sums are then added in order of s): bit-identical outputs.  Most
// groups complete while other workgroups are still streaming; exposed after the last workgroup's
// main loop: one group sum (209 KB) + the final sum (557 KB of partials) by one workgroup each.
struct GlmFinArgs {
  uint32_t* counters;     // NULL = off.  Per particle pass 40 words: [0..31] groups, [32] level 2;
                          // zero between launches (the final arrival re-arms them)
  double* part64;         // [npass][32][REC] level-1 partials
  float *ll, *gw, *gb;
  double scale, ll_offset;
  int D, P;
  // measurement hook (pa_glm_planes_stamps): {earliest workgroup entry, latest workgroup exit} of the
  // launch on the 100 MHz wall clock (min / max atomics), or NULL.  The kernel's duration INSIDE a
  // captured hipGraph, where HIP events do not time their node on ROCm 7.2.
  unsigned long long* tstamps;
};
constexpr int GLMF_GROUPS = 32, GLMF_CNT_STRIDE = 40;

template <int NPT>
__device__ __forceinline__ void glmp_finalize_in_kernel(const GlmFinArgs& f, const float* part,
                                                        unsigned char* smem) {
  constexpr int REC = NPT * 1024 + 2 * NPT * 32;
  const int nblocks = (int)gridDim.x, pass = (int)blockIdx.y, me = (int)blockIdx.x;
  const int ngroups = nblocks < GLMF_GROUPS ? nblocks : GLMF_GROUPS;
  const int s = me % GLMF_GROUPS;
  uint32_t* cnt = f.counters + pass * GLMF_CNT_STRIDE;
  int* flag = reinterpret_cast<int*>(smem);
  __syncthreads();                                  // every wave's record stores are at the L2
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t gsize = (uint32_t)((nblocks - s + GLMF_GROUPS - 1) / GLMF_GROUPS);
    const uint32_t t = __hip_atomic_fetch_add(&cnt[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == gsize - 1u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  // ---- level 1: the group's records, increasing order, fp64
  const float* base = part + (int64_t)pass * nblocks * REC;
  double* my64 = f.part64 + ((int64_t)pass * GLMF_GROUPS + s) * REC;
  for (int j = threadIdx.x; j < REC; j += 256) {
    double acc = 0.0;
    int blk = s;
    {
      float v[24];
      for (; blk + 23 * GLMF_GROUPS < nblocks; blk += 24 * GLMF_GROUPS) {
#pragma unroll
        for (int u = 0; u < 24; ++u) v[u] = base[(int64_t)(blk + u * GLMF_GROUPS) * REC + j];
#pragma unroll
        for (int u = 0; u < 24; ++u) acc += (double)v[u];
      }
    }
    for (; blk < nblocks; blk += GLMF_GROUPS) acc += (double)base[(int64_t)blk * REC + j];
    my64[j] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t t = __hip_atomic_fetch_add(&cnt[GLMF_GROUPS], 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == (uint32_t)ngroups - 1u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  // ---- level 2: the partials in group order -> outputs (slot arithmetic of glm_finalize_body)
  const double* p64 = f.part64 + (int64_t)pass * GLMF_GROUPS * REC;
  const int D = f.D, WROWS = 32 * NPT;
  const int nloc = WROWS * D + 2 * WROWS;
  for (int q = threadIdx.x; q < nloc; q += 256) {
    int pl, slot, kind;                       // kind 0: gw, 1: ll, 2: gb
    int d = 0;
    if (q < WROWS * D) {
      pl = q / D;
      d = q - pl * D;
      kind = 0;
      const int pt = pl >> 5, i = pl & 31, hh = (i >> 2) & 1, reg = (i & 3) + 4 * (i >> 3);
      slot = (pt * 16 + reg) * 64 + d + 32 * hh;
    } else {
      const int k = q - WROWS * D, which = k >= WROWS ? 1 : 0;
      pl = k - which * WROWS;
      kind = 1 + which;
      slot = NPT * 1024 + (2 * (pl >> 5) + which) * 32 + (pl & 31);
    }
    const int p = pass * WROWS + pl;
    if (p >= f.P) continue;
    double part_v[GLMF_GROUPS];
#pragma unroll
    for (int k = 0; k < GLMF_GROUPS; ++k) part_v[k] = k < ngroups ? p64[(int64_t)k * REC + slot] : 0.0;
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < GLMF_GROUPS; ++k) t += part_v[k];
    if (kind == 0) f.gw[(int64_t)p * D + d] = (float)(t * f.scale);
    else if (kind == 1) f.ll[p] = (float)((t + f.ll_offset) * f.scale);
    else f.gb[p] = (float)(t * f.scale);
  }
  if (threadIdx.x <= GLMF_GROUPS)              // re-arm: every arrival of this pass has happened
    __hip_atomic_store(&cnt[threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// GROUPED: blockIdx.x = a segment of one group's rows (seg / st_off as in the pack kernel, y = the
// padded observation image): the workgroup walks the segment's own super-tiles with that group's
// weights w[p, group, :] and its partial record belongs to the group.
struct GlmGroupArgs {
  const int64_t* seg;       // [nseg][3] {row_begin, row_end, group}
  const int64_t* st_off;    // [nseg + 1]
  int G;
};

template <int NPT, int NB, int OCC, bool GROUPED = false>
__global__ __launch_bounds__(256, OCC) void glm_planes_kernel(
    const unsigned char* __restrict__ img, const float* __restrict__ y,
    const float* __restrict__ w, const float* __restrict__ b, int64_t N, int D, int P,
    int64_t nst, float* __restrict__ part, int prio_cus, const GlmFinArgs fin,
    const GlmGroupArgs grp) {
  using C = GlmPlCfg<NPT, NB>;
  constexpr int NRT = C::NRT, ST_BYTES = C::ST_BYTES, PW = C::PW, WROWS = C::WROWS, WPL = C::WPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int rt = wave / NPT, pt = wave % NPT;
  const int pbase = blockIdx.y * WROWS;

#ifdef PA_GLMP_STAMP
  const uint64_t stamp_entry = wall_clock64();
#endif
  if (fin.tstamps != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_min(&fin.tstamps[0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  // the super-tiles this workgroup walks (first, first + grid, ...: my_count of them, prefetches
  // clamped below st_end), the rows they hold counted from super-tile row_st0 (n_rows of them are
  // real), and the weights' row stride
  int64_t grid = gridDim.x, first = blockIdx.x, st_end = nst, row_st0 = 0, n_rows = N;
  int64_t my_count = first < nst ? (nst - first + grid - 1) / grid : 0;
  int64_t w_stride = D;
  if constexpr (GROUPED) {
    const int64_t sg = blockIdx.x;
    first = grp.st_off[sg];
    st_end = grp.st_off[sg + 1];
    my_count = st_end - first;
    grid = 1;
    row_st0 = first;
    n_rows = grp.seg[3 * sg + 1] - grp.seg[3 * sg];
    w += grp.seg[3 * sg + 2] * D;                       // w[p, group, :]
    w_stride = (int64_t)grp.G * D;
  }

  // ---- DMA of super-tile `st` (clamped: prefetches past the end re-read the last one) into ring
  //      slot `bi`: this wave's PW pieces of the image + its 32 observations --------------------
  auto issue = [&](int64_t st, int bi) {
#ifdef PA_GLMP_ABL_NODMA
    return;
#endif
    const int64_t stc = st < st_end ? st : st_end - 1;
    const unsigned char* src = img + stc * ST_BYTES + (wave * PW) * 1024 + lane * 16;
    const uint32_t dst = lds_base + C::OFS_RING + bi * ST_BYTES + (wave * PW) * 1024;
#pragma unroll
    for (int k = 0; k < PW; ++k) dma16(src + k * 1024, dst + k * 1024);
    int64_t row = (stc * NRT + rt) * 32 + l31;
    // (GROUPED: y is the padded observation image, every row of every super-tile exists)
    if constexpr (!GROUPED) row = row < N ? row : N - 1;
    dma4(y + row, lds_base + C::OFS_Y + (bi * 4 + wave) * 256);
  };

#pragma unroll
  for (int k = 0; k < NB - 1; ++k) issue(first + k * grid, k);

  // ---- W planes (same row image as X: slot swizzle by row) and the bias pieces, once per block --
  for (int idx = threadIdx.x; idx < WROWS * 4; idx += 256) {
    const int pl = idx >> 2, s = idx & 3;
    const int p = pbase + pl;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 8 * s + j;
      // log2(e) rides in W and b (one f32 rounding each): the accumulator then holds
      // l2 = l * log2(e), the argument of the hardware exp2 / the natural scale of log2
      v[j] = (p < P && d < D) ? w[(int64_t)p * w_stride + d] * GLMP_LOG2E : 0.0f;
    }
    uint32_t p1[4], p2[4], p3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p1[j], p2[j], p3[j]);
    unsigned char* q = smem + (pl >> 5) * GLMP_PLANE + glmp_slot_ofs(pl & 31, s);
    *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    *reinterpret_cast<uint4*>(q + WPL) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    *reinterpret_cast<uint4*>(q + 2 * WPL) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
  }
  uint32_t* waux = reinterpret_cast<uint32_t*>(smem + C::OFS_WAUX);
  for (int pl = threadIdx.x; pl < WROWS; pl += 256) {
    const int p = pbase + pl;
    const float bv = (p < P && b != nullptr) ? b[p] * GLMP_LOG2E : 0.0f;
    uint32_t p1, p2, p3;
    split_pair(bv, 0.0f, p1, p2, p3);
    waux[2 * pl] = BF16_ONE | (p1 << 16);                    // k slots {0: 1.0, 1: b1}
    waux[2 * pl + 1] = (p2 & 0xffffu) | (p3 << 16);          // k slots {2: b2, 3: b3}
  }
  __syncthreads();

  const bf16x8 b_aux = as_bf16x8(h == 0 ? waux[2 * (pt * 32 + l31)] : 0u,
                                 h == 0 ? waux[2 * (pt * 32 + l31) + 1] : 0u, 0u, 0u);
  f32x16v gwacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) gwacc[r] = 0.0f;
  float s_yl[2] = {0.0f, 0.0f}, s_abs[2] = {0.0f, 0.0f}, s_g[2] = {0.0f, 0.0f};
  // sum_n log2(t_n), t in [1, 2], is kept as an exponent count plus a running product of mantissas:
  // one multiply per element instead of one v_log_f32 (8 issue cycles) + one add, and two
  // v_frexp per 8 elements; the logarithm of the two products is taken once, after the loop
  float p_t[2] = {1.0f, 1.0f};
  int e_t[2] = {0, 0};

  // lane-constant LDS offsets
  //   A operand of GEMM1 / B operand (W): row l31, K chunk c, slot 2c + h
  const int a_ofs0 = glmp_slot_ofs(l31, h), a_ofs1 = glmp_slot_ofs(l31, 2 + h);
  const unsigned char* w_row = smem + pt * GLMP_PLANE;
  //   B operand of GEMM2 (transpose read): within a 16-lane group lane q passes the address of 4
  //   consecutive bf16 of row 4h + (q >> 2), columns 16*(group & 1) + 4*(q & 3) .. +3, and receives
  //   column q of that 4 x 16 block; the rows of K half kh are 16kh + 4h + {0..3} and + 8, whose
  //   slot swizzle is h and h + 2
  const int q = lane & 15, gi1 = (lane >> 4) & 1;
  const int tr_row = 4 * h + (q >> 2);
  const int tr_slot = 2 * gi1 + ((q & 3) >> 1), tr_in = (q & 1) * 8;
  const int tr_ofs_a = tr_row * 64 + ((tr_slot ^ h) << 4) + tr_in;              // rows +0..3
  const int tr_ofs_b = (tr_row + 8) * 64 + ((tr_slot ^ ((h + 2) & 3)) << 4) + tr_in;   // rows +8..11

  constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
  constexpr int TB[6] = {0, 1, 2, 0, 1, 0};

#ifdef PA_GLMP_STAMP
  uint64_t stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t stamp_last = __builtin_readcyclecounter();
  const uint64_t stamp_t0 = stamp_last, stamp_w0 = wall_clock64();
#endif
  // ---- pieces of the per-tile work ------------------------------------------------------------
  // Rows past the end of the plate (the image holds zeros there) must not count: their aux operand
  // is all zero (no bias), so their logit is exactly 0, and their y - 1/2 is stored as 0: then
  // g = 0 - copysign(1/2 - 1/2, 0) = 0 and every running sum gets 0 except the log2(1 + e) sum,
  // which gets log2(2) = 1 per such row and particle -- a known count, taken out again by the
  // finalize step.  The wave's 32 observations of ring slot b become y - 1/2 in place.
  auto prep_rows = [&](int b, int64_t st_) -> bool {
    float* ys_ = reinterpret_cast<float*>(smem + C::OFS_Y + (b * 4 + wave) * 256);
    const int64_t rows_left = n_rows - ((st_ - row_st0) * NRT + rt) * 32;        // scalar
    const bool okr = (int64_t)l31 < rows_left;
    if (lane < 32) ys_[lane] = okr ? ys_[lane] - 0.5f : 0.0f;
    return okr;
  };
  // GEMM1: L2[n, p] = log2(e) (sum_d X[n, d] W[p, d] + b[p]); first the bias through the aux operand
  auto gemm1_aux = [&](bool okr) -> f32x16v {
    const uint32_t a0 = (h == 0 && okr) ? (BF16_ONE << 16) : 0u;              // k slots {-, 1.0}
    const uint32_t a1 = (h == 0 && okr) ? (BF16_ONE | (BF16_ONE << 16)) : 0u;  // k slots {1.0, 1.0}
    const f32x16v zero = {};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0, a1, 0u, 0u), b_aux, zero, 0, 0, 0);
  };
  auto load_ab = [&](const unsigned char* Xt, int c, bf16x8 (&xa)[3], bf16x8 (&wa)[3]) {
    const int ao = c == 0 ? a_ofs0 : a_ofs1;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      xa[pl] = *reinterpret_cast<const bf16x8*>(Xt + pl * GLMP_PLANE + ao);
      wa[pl] = *reinterpret_cast<const bf16x8*>(w_row + pl * WPL + ao);
    }
  };
  // element-wise on one accumulator element (row n = (r&3) + 8(r>>2) + 4h of the wave's tile).
  // Plain (unpacked) f32 instructions only: v_pk_*_f32 does not overlap the bf16 MFMAs on gfx950
  // and costs 2-4x a plain VALU instruction next to them (tools/probes/issue_probe: ~6 plain VALU or
  // 2 transcendentals per MFMA issue for free).  With e = exp2(-|l2|), t = 1 + e:
  //     y l - softplus(l) = ln2 ((y - 1/2) l2 - |l2|/2 - log2(t))       (three running sums)
  //     g = y - sigmoid(l) = (y - 1/2) - copysign(1/t - 1/2, l2)
  auto elem1 = [&](float l2, float yh, int par) -> float {
    const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(l2));
    const float t = e + 1.0f;
    const float inv = __builtin_amdgcn_rcpf(t);
    s_yl[par] = __builtin_fmaf(yh, l2, s_yl[par]);
    s_abs[par] += __builtin_fabsf(l2);
    p_t[par] *= t;            // sum of log2(t) = log2 of the running product (renormalised per half)
    const float g = yh - __builtin_copysignf(inv - 0.5f, l2);
    s_g[par] += g;
    return g;
  };
  auto renorm = [&]() {       // 4 factors <= 2 per chain since the last call: the product stays < 16
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      e_t[c2] += __builtin_amdgcn_frexp_expf(p_t[c2]);
      p_t[c2] = __builtin_amdgcn_frexp_mantf(p_t[c2]);
    }
  };
  // B operand of GEMM2: ds_read_b64_tr_b16 as inline asm (the builtin makes hipcc drain vmcnt(0) in
  // front of it); hipcc does not count inline-asm DS operations, tr_wait() waits for them and names
  // every destination, so that nothing consuming them moves above the wait
  auto tr_wait = [&](v2u32 (&xlo)[3], v2u32 (&xhi)[3], bf16x8 (&xb)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xlo[0]), "+v"(xhi[0]), "+v"(xlo[1]), "+v"(xhi[1]), "+v"(xlo[2]), "+v"(xhi[2])
                 :
                 : "memory");
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const u32x4v cc = {xlo[pl][0], xlo[pl][1], xhi[pl][0], xhi[pl][1]};
      xb[pl] = __builtin_bit_cast(bf16x8, cc);
    }
  };
  auto tr_issue = [&](uint32_t tr_a, uint32_t tr_b, int kh, v2u32 (&xlo)[3], v2u32 (&xhi)[3]) {
    const uint32_t a = tr_a + (kh ? 1024u : 0u), b2 = tr_b + (kh ? 1024u : 0u);
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xlo[0]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xhi[0]) : "v"(b2));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xlo[1]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xhi[1]) : "v"(b2));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(xlo[2]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(xhi[2]) : "v"(b2));
  };
  auto load_y = [&](const float* ys_, int kh, float (&yv)[8]) {
    const float4 y0 = *reinterpret_cast<const float4*>(ys_ + 16 * kh + 4 * h);
    const float4 y1 = *reinterpret_cast<const float4*>(ys_ + 16 * kh + 8 + 4 * h);
    yv[0] = y0.x; yv[1] = y0.y; yv[2] = y0.z; yv[3] = y0.w;
    yv[4] = y1.x; yv[5] = y1.y; yv[6] = y1.z; yv[7] = y1.w;
  };

  int64_t st = first;
  int bi = 0;
  const uint32_t prio_slot = (uint32_t)(blockIdx.x / prio_cus);
  uint64_t prio_clock = wall_clock64();

  // ---- software pipeline: iteration `it` runs GEMM1 of tile it+1 (matrix pipe) against the
  //      element-wise work and GEMM2 of tile it, so that every wave always has both MFMA and VALU
  //      instructions to issue (a wave issues in order: with the phases back to back each wave
  //      alternates between a pure-MFMA and a pure-VALU stretch and three waves per SIMD do not
  //      cover that -- measured 1600 cycles per tile and SIMD against ~950 of issue time).
  //      Ring: slot it % NB holds tile it (GEMM2 reads), slot (it+1) % NB tile it+1 (GEMM1 reads),
  //      the DMA of tile it+NB-1 goes into the slot tile it-1 left -------------------------------
  f32x16v acc_cur = {};
  if (my_count > 0) {
    wait_vmcnt<(NB - 2) * C::NDMA>();       // this wave's pieces of tile 0 have landed
    __builtin_amdgcn_s_barrier();           // ... and everybody else's
    const bool ok0 = prep_rows(0, st);
    acc_cur = gemm1_aux(ok0);
    const unsigned char* X0 = smem + C::OFS_RING + rt * GLMP_TILE;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 xa[3], wa[3];
      load_ab(X0, c, xa, wa);
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[TA[t]], wa[TB[t]], acc_cur, 0, 0, 0);
    }
  }
  for (int64_t it = 0; it < my_count; ++it) {
    // The SIMD arbitrates between its waves by priority, then AGE: left alone, the workgroup that
    // was dispatched first to a CU runs ahead of its co-residents and finishes ~25 us early, and the
    // youngest one runs the tail alone at a third of the CU's issue rate (measured:
    // tools/probes/glm_planes_probe).  Time slices of 2.56 us rotate the priority levels over the
    // co-resident workgroups (slot = blockIdx.x / #CUs: speed only, no correctness dependence).
#ifndef PA_GLMP_ABL_NOPRIO
    if constexpr (OCC > 1) {
      const uint32_t ph = ((uint32_t)(prio_clock >> 8) + prio_slot) % (uint32_t)OCC;
      if (ph == 0) __builtin_amdgcn_s_setprio(0);
      else if (ph == 1) __builtin_amdgcn_s_setprio(1);
      else if (ph == 2) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(3);
      prio_clock = wall_clock64();     // read now, used at the next tile: the SMEM latency hides
    }
#endif
    int bn = bi + 1 == NB ? 0 : bi + 1;     // slot of tile it+1
    wait_vmcnt<(NB - 3) * C::NDMA>();       // this wave's pieces of tile it+1 have landed
    __builtin_amdgcn_s_barrier();           // ... and everybody else's; the slot of tile it-1 is free
    {
      int bf = bi + (NB - 1);
      bf = bf >= NB ? bf - NB : bf;
      issue(st + (NB - 1) * grid, bf);
    }
    const unsigned char* Xc = smem + C::OFS_RING + bi * ST_BYTES + rt * GLMP_TILE;
    const unsigned char* Xn = smem + C::OFS_RING + bn * ST_BYTES + rt * GLMP_TILE;
    const float* ysc = reinterpret_cast<const float*>(smem + C::OFS_Y + (bi * 4 + wave) * 256);
    const uint32_t tr_a = (uint32_t)(uintptr_t)Xc + (uint32_t)tr_ofs_a;
    const uint32_t tr_b = (uint32_t)(uintptr_t)Xc + (uint32_t)tr_ofs_b;

    const bool okn = prep_rows(bn, st + grid);
    f32x16v acc_nxt = gemm1_aux(okn);
    v2u32 xlo[3], xhi[3];
    bf16x8 xa[3], wa[3], xb[3];
    float yv[8], g[8];
    uint32_t g1[4], g2[4], g3[4];

    // -- GEMM1(it+1), K chunk 0 and 1  ||  element-wise(it, K half 0) and its split
    tr_issue(tr_a, tr_b, 0, xlo, xhi);
    load_y(ysc, 0, yv);
    load_ab(Xn, 0, xa, wa);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      acc_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[TA[t]], wa[TB[t]], acc_nxt, 0, 0, 0);
      g[t] = elem1(acc_cur[t], yv[t], t & 1);
    }
    load_ab(Xn, 1, xa, wa);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      acc_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[TA[t]], wa[TB[t]], acc_nxt, 0, 0, 0);
      if (t < 2) g[6 + t] = elem1(acc_cur[6 + t], yv[6 + t], t & 1);
      else split_pair_trunc(g[2 * (t - 2)], g[2 * (t - 2) + 1], g1[t - 2], g2[t - 2], g3[t - 2]);
    }
    renorm();
    tr_wait(xlo, xhi, xb);
    // -- GEMM2(it, K half 0)  ||  element-wise(it, K half 1) and its split
    {
      const bf16x8 ga[3] = {as_bf16x8(g1[0], g1[1], g1[2], g1[3]), as_bf16x8(g2[0], g2[1], g2[2], g2[3]),
                            as_bf16x8(g3[0], g3[1], g3[2], g3[3])};
      tr_issue(tr_a, tr_b, 1, xlo, xhi);
      load_y(ysc, 1, yv);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        gwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[TA[t]], xb[TB[t]], gwacc, 0, 0, 0);
        if (t < 4) {
          g[2 * t] = elem1(acc_cur[8 + 2 * t], yv[2 * t], 0);
          g[2 * t + 1] = elem1(acc_cur[8 + 2 * t + 1], yv[2 * t + 1], 1);
        }
      }
    }
    renorm();
    uint32_t h1[4], h2[4], h3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair_trunc(g[2 * j], g[2 * j + 1], h1[j], h2[j], h3[j]);
    tr_wait(xlo, xhi, xb);
    // -- GEMM2(it, K half 1)
    {
      const bf16x8 ga[3] = {as_bf16x8(h1[0], h1[1], h1[2], h1[3]), as_bf16x8(h2[0], h2[1], h2[2], h2[3]),
                            as_bf16x8(h3[0], h3[1], h3[2], h3[3])};
#pragma unroll
      for (int t = 0; t < 6; ++t)
        gwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[TA[t]], xb[TB[t]], gwacc, 0, 0, 0);
    }
    acc_cur = acc_nxt;
    st += grid;
    bi = bn;
  }
#ifdef PA_GLMP_STAMP
  if (lane == 0 && blockIdx.y == 0) {
    uint64_t* dbg = reinterpret_cast<uint64_t*>(part) + (1 << 20) + ((int64_t)blockIdx.x * 4 + wave) * 16;
    for (int i = 0; i < 8; ++i) dbg[i] = stamp_acc[i];
    dbg[8] = __builtin_readcyclecounter() - stamp_t0;
    dbg[9] = wall_clock64() - stamp_w0;
    dbg[10] = (uint64_t)my_count;
    dbg[11] = stamp_entry;
    dbg[12] = stamp_w0;
    dbg[13] = wall_clock64();
  }
#endif
  wait_vmcnt<0>();
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();

  // ---- block reduction over the row tiles in a fixed order, then one partial record in the format
  //      of glm.hip: [pt][16 regs][64 lanes] accumulator tiles, then ll and gb per particle -------
  constexpr int REC = NPT * 1024 + 2 * NPT * 32;
  static_assert((NPT * 1024 + 2 * NPT * 64) * 4 <= C::LDS_BYTES - C::OFS_RING, "LDS too small");
  float* red = reinterpret_cast<float*>(smem + C::OFS_RING);
  float* red2 = red + NPT * 1024;
  const float s_lg = (float)(e_t[0] + e_t[1]) + (__builtin_amdgcn_logf(p_t[0]) + __builtin_amdgcn_logf(p_t[1]));
  const float ll_acc = 0.69314718055994530942f *
                       ((s_yl[0] + s_yl[1]) - 0.5f * (s_abs[0] + s_abs[1]) - s_lg);
  const float gb_acc = s_g[0] + s_g[1];
  for (int rr = 0; rr < NRT; ++rr) {
    if (rt == rr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = (pt * 16 + r) * 64 + lane;
        red[idx] = (rr == 0 ? 0.0f : red[idx]) + gwacc[r];
      }
      const int i0 = (2 * pt) * 64 + lane, i1 = (2 * pt + 1) * 64 + lane;
      red2[i0] = (rr == 0 ? 0.0f : red2[i0]) + ll_acc;
      red2[i1] = (rr == 0 ? 0.0f : red2[i1]) + gb_acc;
    }
    __syncthreads();
  }
  float* rec = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * REC;
  for (int i = threadIdx.x; i < NPT * 1024; i += 256) rec[i] = red[i];
  for (int i = threadIdx.x; i < 2 * NPT * 32; i += 256) {
    const int qq = i >> 5, j = i & 31;
    rec[NPT * 1024 + i] = red2[qq * 64 + j] + red2[qq * 64 + 32 + j];
  }
#ifdef PA_GLMP_STAMP
  if (lane == 0 && blockIdx.y == 0)
    (reinterpret_cast<uint64_t*>(part) + (1 << 20) + ((int64_t)blockIdx.x * 4 + wave) * 16)[14] = wall_clock64();
#endif
  if (fin.counters != nullptr) glmp_finalize_in_kernel<NPT>(fin, part, smem);
  if (fin.tstamps != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_max(&fin.tstamps[1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace pa
