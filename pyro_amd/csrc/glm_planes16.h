// glm_planes16.h -- the fused Bernoulli-logits GLM pass of glm_planes.h with the design matrix kept
// as TWO scaled f16 planes instead of three bf16 planes (included by glm.hip; same partial-record
// format, same finalize).
//
// Why.  glm_planes_kernel is bound by instruction issue, not by HBM: per 32x32 tile 25 bf16 MFMAs
// (six piece products per GEMM K chunk) next to ~300 VALU instructions, 2100 clocks per tile and SIMD
// (profiles/r03_pmc_summary.json), and its image is 1.5x the f32 matrix.  An f16 carries 11
// significant bits: two pieces x ~= x1 + x2 (round-to-nearest at each level, the residual exact in
// f32) represent an f32 to 2^-22 relative -- the representation error of ONE f32 rounding is 2^-24
// -- and a product needs the three piece products of order >= 2^-11:
//     x*w ~= x2*w1 + x1*w2 + x1*w1                       (dropped: x2*w2 = O(2^-22 |x||w|))
// i.e. 13 MFMAs per tile instead of 25, a two-level split of the gradient operand instead of a
// three-level one, and an image of exactly the f32 matrix's size (4 B per element: the algorithmic
// bytes of SURVEY 8d).  Error per logit <= 3 * 2^-22 sum_d |x_d w_d| against the 32 * 2^-24 bound of
// an f32 FMA chain over D = 32 -- the same class; measured against the f64 oracle next to torch's own
// f32 result in tests/test_kernels_gpu.py::test_glm_planes_f16_is_f32_class.
//
// f16 has 5 exponent bits: the pieces are SCALED by powers of two (exact) so that they sit at the top
// of the f16 range and the second pieces stay normal over 2^-13 of dynamic range below the largest
// element (smaller elements keep an ABSOLUTE error <= 2^-40 of the largest: below the f32
// accumulation's own rounding):
//   X:  one exponent kx[d] per COLUMN, max_n |X[n,d]| 2^kx[d] in [2^14, 2^15) (pa_glm_pack_planes finds
//       the column maxima on the device and stores the exponents in the image's trailer: no host round
//       trip).  Per column, not per image: a design matrix with raw columns of order 1e6 beside 0/1
//       indicators keeps 22 bits in every column (round 3 scaled the whole image by one power of two
//       and lost the second piece of columns 2^13 below the largest one);
//   W:  the kernel's prologue forms w'[p,d] = w[p,d] 2^-kx[d] (so that x'.w' = x.w) and picks one
//       exponent kw[p] per particle row from max_d |w'[p,d]| and the bias (below); the f32 accumulator
//       holds 2^kw[p] * l2 and the element-wise code starts with one multiply by the per-lane constant
//       2^-kw[p].  Columns whose w' is 2^13 below the row's largest lose w's second piece -- their
//       products are that far below the row's dominant one;
//   b:  enters the accumulator through the aux MFMA as three f16 pieces of b 2^(kw[p]-15) against
//       2^15 (rows past the end of the plate: 0, their logit is exactly 0 as in glm_planes.h);
//       kw[p] is capped so that this stays inside f16 -- when the bias dominates the row, W gives up
//       low bits that are below the rounding of (x.w + b) anyway;
//   g:  y - sigmoid(l) is formed as 2^14 g (one fma instead of a subtraction), so that its second
//       piece is normal down to |g| = 2^-17; the gradient accumulator of column d holds
//       2^(14 + kx[d]) gw[., d].
#pragma once
#include "chain.h"
#include "glm_planes.h"
#include "multisite_dev.h"

namespace pa {

constexpr int GLMH_TILE = 2 * GLMP_PLANE;   // bytes of one 32-row tile image: planes x1, x2
constexpr int GLMH_TRAILER = 256;           // after the tiles (and y_img): u32[32] column max |X| bits,
                                            // i32[32] column exponents kx[d]
constexpr int GLMH_KX = 32;                 // index of kx[0] in the trailer's words
constexpr int GLMH_KNONE = 1 << 20;         // "no constraint" in the choice of a row's exponent

constexpr float GLMH_GSCALE = 16384.0f;      // 2^14: the scale of y - 1/2 and of g
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x8 as_f16x8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4v v = {a, b, c, d};
  return __builtin_bit_cast(f16x8, v);
}

// the image exponent from max |X| (IEEE bits of a non-negative f32; 0 or non-finite: no scaling)
__host__ __device__ __forceinline__ int glmh_exponent_of(uint32_t absmax_bits) {
  const int e = (int)(absmax_bits >> 23) & 0xff;
  if (absmax_bits == 0u || e == 0xff) return 0;
  // ilogb of a normal f32 is e - 127; subnormals (e == 0) are treated as 2^-127
  return 14 - (e == 0 ? -127 : e - 127);          // -113 .. 141
}

// column maxima of |X| as the unsigned maximum of the magnitudes' bit patterns (NaN patterns order
// above +inf: a non-finite column gets kx = 0 and its NaN / inf reach the outputs as they would in
// f32).  256 threads = 8 rows x 32 columns per step; out[d], d < D
__global__ __launch_bounds__(256) void glm_absmax_kernel(const float* __restrict__ X, int64_t N, int D,
                                                         uint32_t* __restrict__ out) {
  const int d = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  uint32_t m = 0u;
  if (d < D)
    for (int64_t r = (int64_t)blockIdx.x * 8 + r0; r < N; r += (int64_t)gridDim.x * 8) {
      const uint32_t v = __builtin_bit_cast(uint32_t, X[r * D + d]) & 0x7fffffffu;
      m = v > m ? v : m;
    }
  const uint32_t t = (uint32_t)__shfl_xor((int)m, 32);           // the wave's other row of this column
  m = t > m ? t : m;
  if ((threadIdx.x & 63) < 32 && d < D && m != 0u) atomicMax(out + d, m);
}

// ---- data moments of the label-linear term (LIN above): c[d] = sum_n (y_n - 1/2) x[n,d], c[32] = sum_n
//      (y_n - 1/2); float64, fixed order: per-workgroup partials [grid][33], then one workgroup adds
//      them in index order (a pure function of (X, y): bit-reproducible) ---------------------------------
__global__ __launch_bounds__(256) void glm_label_moments_partial_kernel(const float* __restrict__ X,
                                                                        const float* __restrict__ y,
                                                                        int64_t N, int D,
                                                                        double* __restrict__ part) {
  __shared__ double sm[8][33];
  const int d = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  double acc = 0.0, acc0 = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 8 + r0; r < N; r += (int64_t)gridDim.x * 8) {
    const double yh = (double)y[r] - 0.5;
    if (d < D) acc = __builtin_fma(yh, (double)X[r * D + d], acc);
    if (d == 0) acc0 += yh;
  }
  sm[r0][d] = acc;
  if (d == 0) sm[r0][32] = acc0;
  __syncthreads();
  if (threadIdx.x < 33) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[k][threadIdx.x];
    part[(int64_t)blockIdx.x * 33 + threadIdx.x] = t;
  }
}
__global__ __launch_bounds__(64) void glm_label_moments_final_kernel(const double* __restrict__ part,
                                                                     int nblocks, double* __restrict__ out) {
  if (threadIdx.x >= 33) return;
  double t = 0.0;
  for (int k = 0; k < nblocks; ++k) t += part[(int64_t)k * 33 + threadIdx.x];
  out[threadIdx.x] = t;
}

// (a, b) -> hi and lo f16 pairs, a ~= a1 + a2 to 2^-22 |a| (RN at both levels).  The residual
// a - a1 is exact in f32 and goes straight to its f16 half: one v_fma_mixlo_f16 / v_fma_mixhi_f16 per
// element (f16 piece times -1 plus the f32 value, rounded once to f16), 3 instructions per pair
__device__ __forceinline__ void split_pair_f16(float a, float b, uint32_t& p1, uint32_t& p2) {
#ifdef PA_GLMH_ABL_NOSPLIT       // (tools/probes/glm_planes16_probe.hip: timing ablations, wrong numbers)
  p1 = __builtin_bit_cast(uint32_t, a);
  p2 = __builtin_bit_cast(uint32_t, b);
  return;
#endif
  const f32x2v v = {a, b};
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));   // v_cvt_pk_f16_f32
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p1), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p1), "v"(b));
  p2 = r;
}
// The gradient operand g = sigmoid(l) - y of GEMM2.  PA_GLMH_ABL_G1 (tools/probes/glm_planes16_probe.hip ONLY: a
// timing ablation of the precision contract, never built into the library): ONE f16 piece of g -- 2^-11 relative per
// element instead of 2^-22 -- drops the 2 x v_fma_mix per pair and the (g lo, X hi) product of every K half.
__device__ __forceinline__ void split_pair_g(float a, float b, uint32_t& p1, uint32_t& p2) {
#ifdef PA_GLMH_ABL_G1
  const f32x2v v = {a, b};
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2v));
  p2 = 0u;
#else
  split_pair_f16(a, b, p1, p2);
#endif
}
#ifdef PA_GLMH_ABL_G1
#define GLMH_G_PIECE(t) (TA[t] == 0)
#else
#define GLMH_G_PIECE(t) true
#endif
__device__ __forceinline__ float f16_lo(uint32_t p) {
  return (float)__builtin_bit_cast(f16x2v, p)[0];
}
__device__ __forceinline__ float f16_hi(uint32_t p) {
  return (float)__builtin_bit_cast(f16x2v, p)[1];
}

// 8 consecutive features of one row, already scaled -> 2 x 16 B
__device__ __forceinline__ void glmh_store_slot(const float (&v)[8], unsigned char* q) {
  uint32_t p1[4], p2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair_f16(v[2 * j], v[2 * j + 1], p1[j], p2[j]);
  *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  *reinterpret_cast<uint4*>(q + GLMP_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
}

// one thread per (tile, row, slot); `trailer` = {max |X| bits (glm_absmax_kernel), kx (written here)}
__global__ __launch_bounds__(256) void glm_pack_planes_f16_kernel(const float* __restrict__ X,
                                                                  int64_t N, int D, int64_t ntiles,
                                                                  unsigned char* __restrict__ img,
                                                                  uint32_t* __restrict__ trailer) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < 32) trailer[GLMH_KX + idx] = (uint32_t)glmh_exponent_of(trailer[idx]);
  if (idx >= ntiles * 128) return;
  const int64_t T = idx >> 7;
  const int r = (int)(idx >> 2) & 31, s = (int)idx & 3;
  const int64_t row = T * 32 + r;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 8 * s + j;
    v[j] = (row < N && d < D) ? ldexpf(X[row * D + d], glmh_exponent_of(trailer[d])) : 0.0f;
  }
  glmh_store_slot(v, img + T * GLMH_TILE + glmp_slot_ofs(r, s));
}

// the hierarchical variant: segments on super-tile boundaries, see glm_pack_planes_grouped_kernel
__global__ __launch_bounds__(256) void glm_pack_planes_f16_grouped_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const int64_t* __restrict__ row_of, int D,
    const int64_t* __restrict__ seg, const int64_t* __restrict__ st_off, int nseg, int64_t ntiles,
    unsigned char* __restrict__ img, float* __restrict__ y_img, uint32_t* __restrict__ trailer) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < 32) trailer[GLMH_KX + idx] = (uint32_t)glmh_exponent_of(trailer[idx]);
  if (idx >= ntiles * 128) return;
  const int64_t T = idx >> 7, st = T >> 1;
  const int r = (int)(idx >> 2) & 31, sl = (int)idx & 3;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (st_off[mid] <= st) lo = mid;
    else hi = mid - 1;
  }
  const int64_t a = seg[3 * lo], e = seg[3 * lo + 1];
  const int64_t pos = a + (T - 2 * st_off[lo]) * 32 + r;
  const bool ok = pos < e;
  const int64_t row = (ok && row_of) ? row_of[pos] : pos;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 8 * sl + j;
    v[j] = (ok && d < D) ? ldexpf(X[row * D + d], glmh_exponent_of(trailer[d])) : 0.0f;
  }
  glmh_store_slot(v, img + T * GLMH_TILE + glmp_slot_ofs(r, sl));
  // the observations as the kernel consumes them: 2^14 (y - 1/2), 0 in the padding
  if (sl == 0) y_img[T * 32 + r] = ok ? __builtin_fmaf(y[row], GLMH_GSCALE, -0.5f * GLMH_GSCALE) : 0.0f;
}

// PRIV: every wave keeps a PRIVATE ring of its own 32-row tile (4 KiB: the two waves of a row tile
// -- particle tiles 0 and 1 -- each fetch it, the second from the cache) instead of the workgroup
// sharing a ring of super-tiles: the tile loop then needs no workgroup barrier at all, only the wave's
// own counted s_waitcnt.
// NRT x NPT waves per workgroup: NRT row tiles of 32 rows (one super-tile of the ring) x NPT particle tiles of 32.
// 2 x 2 (256 threads, 64 particles per pass over the image) is the SVI geometry; 2 x 4 and 1 x 8 (512 threads, 128 /
// 256 particles per pass) serve many chains / particles with ONE pass over the image where 2 x 2 makes two / four
// (NUTS on a model: P = the number of chains) -- the image pieces, the observations and their LDS-DMA issue cost
// are then shared by twice / four times as many (row, particle) elements.
template <int NB, bool PRIV = false, int NRT_ = 2, int NPT_ = 2>
struct GlmHCfg {
  static constexpr int NRT = NRT_, NPT = NPT_, NW = NRT_ * NPT_;
  static constexpr int ST_BYTES = NRT * GLMH_TILE;     // super-tile image (32 NRT rows)
  static constexpr int PIECES = ST_BYTES / 1024;       // 1 KiB DMA pieces per super-tile
  // pieces per issuing wave and tile, and how many waves issue (1 x 8: four of the eight)
  static constexpr int PW = PRIV ? GLMH_TILE / 1024 : (PIECES >= NW ? PIECES / NW : 1);
  static constexpr int DMA_WAVES = PRIV ? NW : PIECES / PW;
  static constexpr int NDMA = PW + 1;                  // + the observations (wave 0; PRIV: every wave)
  static constexpr int RING_BYTES = PRIV ? NW * NB * GLMH_TILE : NB * ST_BYTES;
  static constexpr int WROWS = 32 * NPT;
  static constexpr int WPL = WROWS * 64;               // one W plane
  static constexpr int OFS_WAUX = 2 * WPL;             // per particle 16 B: {b1 | b2, b3, descale, -}
  static constexpr int OFS_RING = OFS_WAUX + WROWS * 16;
  static constexpr int OFS_Y = OFS_RING + RING_BYTES;
  static constexpr int LDS_BYTES = OFS_Y + NB * 4 * 256;
  static_assert(!PRIV || (NRT_ == 2 && NPT_ == 2), "private rings: the 2 x 2 geometry only");
  static_assert(PW * DMA_WAVES == PIECES || PRIV, "the image pieces must divide over the issuing waves");
};

constexpr uint32_t F16_2P15 = 0x7800u;        // 2^15

// the two GEMMs' MFMAs (timing ablations of tools/probes/glm_planes16_probe.hip keep the operands alive
// and drop the instruction)
__device__ __forceinline__ f32x16v glmh_keep(const f16x8& a, const f16x8& b, const f32x16v& c) {
  asm volatile("" : : "v"(a), "v"(b));
  return c;
}
#ifdef PA_GLMH_ABL_NOGEMM1
#define GLMH_MFMA1(a, b, c) glmh_keep(a, b, c)
#else
#define GLMH_MFMA1(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
#ifdef PA_GLMH_ABL_NOGEMM2
#define GLMH_MFMA2(a, b, c) glmh_keep(a, b, c)
#else
#define GLMH_MFMA2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

// PA_GLMH_SCHED (experiment): a scheduling barrier behind each MFMA + element-wise group pins the source's
// interleave (hipcc's machine scheduler otherwise gathers dependent MFMAs into back-to-back runs)
#ifndef PA_GLMH_SCHED
#define PA_GLMH_SCHED 0
#endif
// PA_GLMH_DEFER (experiment): the three GEMM2 MFMAs of a tile's SECOND K half, which have no element-wise work left
// beside them, run beside the first element-wise groups of the NEXT tile instead (their operands wait in 16
// registers; same accumulation order: bit-identical)
#ifndef PA_GLMH_DEFER
#define PA_GLMH_DEFER 0
#endif
#if PA_GLMH_SCHED
#define GLMH_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define GLMH_SB() do { } while (0)
#endif

// LIN: the label-linear part of the log-likelihood, sum_n (y_n - 1/2) l[n,p] = c . w_p + c0 b_p with the
// data moments c[d] = sum_n (y_n - 1/2) x[n,d], c0 = sum_n (y_n - 1/2) (pa_glm_label_moments: float64,
// once per (X, y)), is added by workgroup 0 of each pass in the epilogue instead of one fma per (row,
// particle) element in the loop.  Only the log-likelihood takes this route: the gradient keeps
// g = y - sigmoid(l) element by element (near the optimum g is small where sigmoid - 1/2 is not).
// DRAW: the weights and the bias ARE the draws of a mean-field Normal guide (chain.h GlmDraw): the
// prologue draws them itself -- z = loc + softplus(rho) eps with eps from the guide's Philox blocks, the
// numbers pa_meanfield_normal_sample would have written -- and workgroup 0 of each pass stores z, eps,
// scale and loc for the step's tail; `w` / `b` are not read.
template <int NB, int OCC, bool GROUPED = false, bool PRIV = false, bool LIN = false, bool DRAW = false,
          int NRT_ = 2, int NPT_ = 2>
__global__ __launch_bounds__(64 * NRT_ * NPT_, OCC) void glm_planes_f16_kernel(
    const unsigned char* __restrict__ img, const float* __restrict__ y,
    const float* __restrict__ w, const float* __restrict__ b, int64_t N, int D, int P,
    int64_t nst, float* __restrict__ part, int prio_cus, const uint32_t* __restrict__ trailer,
    unsigned long long* __restrict__ tstamps, const GlmGroupArgs grp,
    const int64_t* __restrict__ gate, const double* __restrict__ moments = nullptr,
    const GlmDraw draw = GlmDraw{}) {
  if (gate != nullptr && *gate != 0) return;        // the step gate gave this replay up (pa_gate)
  using C = GlmHCfg<NB, PRIV, NRT_, NPT_>;
  constexpr int NRT = C::NRT, NPT = C::NPT, ST_BYTES = C::ST_BYTES, PW = C::PW, WROWS = C::WROWS,
                WPL = C::WPL, NT = 64 * C::NW;
  static_assert(!GROUPED || (NRT == 2 && NPT == 2), "the grouped image: 2 x 2 only");
#ifdef PA_GLMH_Y_PER_WAVE
  constexpr bool YSHARE = !PRIV && !(NRT == 2 && NPT == 2);   // (A/B switch: in the 2 x 2 geometry every wave
                                                              //  fetches and transforms its own 32 observations)
#else
  constexpr bool YSHARE = !PRIV;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int rt = wave / NPT, pt = wave % NPT;
  const int pbase = blockIdx.y * WROWS;

  if (tstamps != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_min(&tstamps[0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
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

  auto issue = [&](int64_t st, int bi) {
#ifdef PA_GLMH_ABL_NODMA
    return;
#endif
    const int64_t stc = st < st_end ? st : st_end - 1;
    const unsigned char* src = img + stc * ST_BYTES + (PRIV ? rt * GLMH_TILE : (wave * PW) * 1024) + lane * 16;
    const uint32_t dst = lds_base + C::OFS_RING +
                         (PRIV ? (wave * NB + bi) * GLMH_TILE : bi * ST_BYTES + (wave * PW) * 1024);
    if (PRIV || wave < C::DMA_WAVES) {
#pragma unroll
      for (int k = 0; k < PW; ++k) dma16(src + k * 1024, dst + k * 1024);
    }
    if constexpr (!YSHARE) {
      int64_t row = (stc * NRT + rt) * 32 + l31;
      if constexpr (!GROUPED) row = row < N ? row : N - 1;
      dma4(y + row, lds_base + C::OFS_Y + (bi * 4 + wave) * 256);
    } else if (wave == 0) {
      // the super-tile's 64 observations are ONE piece (64 lanes x 4 B) issued by wave 0 for the workgroup
      // (every wave used to fetch its own 32: four LDS-DMA issues per super-tile instead of one; an LDS-DMA
      // instruction costs ~60-90 cycles of issue beside the compute stream)
      int64_t row = stc * (NRT * 32) + lane;
      if constexpr (!GROUPED) row = row < N ? row : N - 1;
      dma4(y + row, lds_base + C::OFS_Y + bi * 1024);
    }
  };
  // this wave's LDS-DMA instructions per tile (wave-uniform): its image pieces (+ the observations: wave 0)
  const int my_ndma = !YSHARE ? C::NDMA : (wave < C::DMA_WAVES ? PW : 0) + (wave == 0 ? 1 : 0);
  auto wait_tiles_in_flight = [&](auto kconst) {
    constexpr int K = decltype(kconst)::value;
    if (my_ndma == PW + 1) wait_vmcnt<K * (PW + 1)>();
    else if (my_ndma == PW) wait_vmcnt<K * PW>();
    else wait_vmcnt<0>();                           // (a wave that issues nothing has nothing in flight)
  };

#pragma unroll
  for (int k = 0; k < NB - 1; ++k) issue(first + k * grid, k);

  // ---- W planes and the per-particle constants, once per block: thread (pl, s) holds 8 features of
  //      particle row pl; the four threads of a row are neighbours --------------------------------
  const int kx_l = (int)trailer[GLMH_KX + l31];        // the exponent of this lane's gradient column
  float* sp_s = reinterpret_cast<float*>(smem + C::LDS_BYTES);          // DRAW: 64 floats behind the rings
  if constexpr (DRAW) {
    // softplus(rho) once per workgroup (32 + 1 values through LDS) instead of 8 per thread: the
    // draw sits on the critical path of every workgroup's start
    if (threadIdx.x < 32) sp_s[threadIdx.x] = (int)threadIdx.x < D ? softplus_t<float>(draw.rho_w[threadIdx.x]) : 0.0f;
    else if (threadIdx.x == 32) sp_s[32] = draw.have_b ? softplus_t<float>(draw.rho_b[0]) : 0.0f;
    __syncthreads();
  }
  // (NT / 4 particle rows per round: one round at 2 row tiles, two at one)
#pragma unroll 1
  for (int pl0 = 0; pl0 < WROWS; pl0 += NT / 4) {
    const int pl = pl0 + (int)(threadIdx.x >> 2), s = threadIdx.x & 3;
    const int p = pbase + pl;
    float v[8];
    float mw = 0.0f;
    float wraw[8], braw = 0.0f;          // the weights / bias in model units
    if constexpr (DRAW) {
      const uint64_t obase = draw.offset_dev ? *draw.offset_dev : 0;
      const bool store = blockIdx.x == 0 && p < P;
      float nrm[8];
      const int64_t i0 = (int64_t)p * D + 8 * s;
      if ((D & 3) == 0 && p < P) {
        // 8 consecutive elements = two whole Philox blocks (element i: lane i % 4 of block i / 4)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x4 blk = philox4x32_10(draw.seed, draw.off_w + obase + (uint64_t)(i0 >> 2) + q, 0);
          box_muller_f32(u32_to_unit_f32(blk.x), u32_to_unit_f32(blk.y), nrm[4 * q], nrm[4 * q + 1]);
          box_muller_f32(u32_to_unit_f32(blk.z), u32_to_unit_f32(blk.w), nrm[4 * q + 2], nrm[4 * q + 3]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          nrm[j] = (p < P && 8 * s + j < D)
                       ? philox_normal_f32(draw.seed, draw.off_w + obase, (uint64_t)(i0 + j)) : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = 8 * s + j;
        wraw[j] = 0.0f;
        if (p < P && d < D) {
          const float sp = sp_s[d];
          wraw[j] = __builtin_fmaf(sp, nrm[j], draw.loc_w[d]);
          if (store) {
            draw.eps_w[i0 + j] = nrm[j];
            draw.z_w[i0 + j] = wraw[j];
            if (p == 0) {
              draw.scale_w[d] = sp;
              draw.lout_w[d] = draw.loc_w[d];
            }
          }
        }
      }
      if (draw.have_b && p < P) {
        // (every thread of the row computes it: the four of them need b for the row's exponent)
        const float eb_ = philox_normal_f32(draw.seed, draw.off_b + obase, (uint64_t)p);
        const float spb = sp_s[32];
        braw = __builtin_fmaf(spb, eb_, draw.loc_b[0]);
        if (store && s == 0) {
          draw.eps_b[p] = eb_;
          draw.z_b[p] = braw;
          if (p == 0) {
            draw.scale_b[0] = spb;
            draw.lout_b[0] = draw.loc_b[0];
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = 8 * s + j;
        wraw[j] = (p < P && d < D) ? w[(int64_t)p * w_stride + d] : 0.0f;
      }
      braw = (p < P && b != nullptr) ? b[p] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 8 * s + j;
      // log2(e) rides in W and b (one f32 rounding each), as in glm_planes.h; the column's exponent
      // comes off here (exact)
      v[j] = (p < P && d < D) ? ldexpf(wraw[j] * GLMP_LOG2E, -(int)trailer[GLMH_KX + d]) : 0.0f;
      mw = __builtin_fmaxf(mw, __builtin_fabsf(v[j]));
    }
    mw = __builtin_fmaxf(mw, __shfl_xor(mw, 1));
    mw = __builtin_fmaxf(mw, __shfl_xor(mw, 2));
    const float b2 = braw * GLMP_LOG2E;
    // NaN / inf weights: fmaxf drops a NaN; the scaled pieces below carry it into the accumulator
    const uint32_t mwb = __builtin_bit_cast(uint32_t, mw), bb = __builtin_bit_cast(uint32_t, b2) & 0x7fffffffu;
    const int ew = (int)(mwb >> 23), eb = (int)(bb >> 23);
    int kw = (mwb != 0u && ew != 0xff) ? 14 - (ew == 0 ? -127 : ew - 127) : GLMH_KNONE;
    const int kb = (bb != 0u && eb != 0xff) ? 29 - (eb == 0 ? -127 : eb - 127) : GLMH_KNONE;
    kw = kw < kb ? kw : kb;
    if (kw == GLMH_KNONE) kw = 0;                   // an all-zero (or non-finite) row
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ldexpf(v[j], kw);
    uint32_t p1[4], p2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair_f16(v[2 * j], v[2 * j + 1], p1[j], p2[j]);
    unsigned char* q = smem + (pl >> 5) * GLMP_PLANE + glmp_slot_ofs(pl & 31, s);
    *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    *reinterpret_cast<uint4*>(q + WPL) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    if (s == 0) {
      // the bias in accumulator units over 2^15, three pieces (24 bits)
      const float bs = ldexpf(b2, kw - 15);
      uint32_t q1, q2, q3, dummy;
      split_pair_f16(bs, 0.0f, q1, q2);
      const float r2 = (bs - f16_lo(q1)) - f16_lo(q2);
      split_pair_f16(r2, 0.0f, q3, dummy);
      uint32_t* wx = reinterpret_cast<uint32_t*>(smem + C::OFS_WAUX) + 4 * pl;
      wx[0] = (q1 & 0xffffu) | (q2 << 16);              // k slots {0: b1, 1: b2}
      wx[1] = q3 & 0xffffu;                             // k slots {2: b3, 3: 0}
      // the descale factor stays a normal f32: beyond +-126 the logits are below / above anything
      // f32 itself could hold
      int kd = -kw;
      kd = kd > 126 ? 126 : (kd < -126 ? -126 : kd);
      wx[2] = __builtin_bit_cast(uint32_t, ldexpf(1.0f, kd));
      wx[3] = 0u;
    }
  }
  __syncthreads();

  const uint32_t* wx_l = reinterpret_cast<const uint32_t*>(smem + C::OFS_WAUX) + 4 * (pt * 32 + l31);
  const f16x8 b_aux = as_f16x8(h == 0 ? wx_l[0] : 0u, h == 0 ? wx_l[1] : 0u, 0u, 0u);
  const float dsc = __builtin_bit_cast(float, wx_l[2]);       // 2^-kw[particle of this lane]
  f32x16v gwacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) gwacc[r] = 0.0f;
  // (sum_n g stays on the vector pipe: measured, an MFMA costs ~22 cycles of SIMD time next to this
  //  loop's VALU stream -- 4 MFMAs against a ones operand are slower than the 16 v_add they replace)
  float s_yl[2] = {0.0f, 0.0f}, s_abs[2] = {0.0f, 0.0f}, s_g[2] = {0.0f, 0.0f};
  float p_t[2] = {1.0f, 1.0f};
  int e_t[2] = {0, 0};

  const int a_ofs0 = glmp_slot_ofs(l31, h), a_ofs1 = glmp_slot_ofs(l31, 2 + h);
  const unsigned char* w_row = smem + pt * GLMP_PLANE;
  const int q = lane & 15, gi1 = (lane >> 4) & 1;
  const int tr_row = 4 * h + (q >> 2);
  const int tr_slot = 2 * gi1 + ((q & 3) >> 1), tr_in = (q & 1) * 8;
  const int tr_ofs_a = tr_row * 64 + ((tr_slot ^ h) << 4) + tr_in;
  const int tr_ofs_b = (tr_row + 8) * 64 + ((tr_slot ^ ((h + 2) & 3)) << 4) + tr_in;

  // piece products of a K chunk, smallest first: x2 w1, x1 w2, x1 w1
  constexpr int TA[3] = {1, 0, 0};
  constexpr int TB[3] = {0, 1, 0};

  // W operands of both K chunks stay in registers for the whole launch (2 x 2 x 4 VGPRs)
  f16x8 wa0[2], wa1[2];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    wa0[pl] = *reinterpret_cast<const f16x8*>(w_row + pl * WPL + a_ofs0);
    wa1[pl] = *reinterpret_cast<const f16x8*>(w_row + pl * WPL + a_ofs1);
  }

  // the wave's 32 observations of ring slot b become 2^14 (y - 1/2) in place (0 past the end)
  auto prep_rows = [&](int b_, int64_t st_) -> bool {
    const int64_t rows_left = n_rows - ((st_ - row_st0) * NRT + rt) * 32;        // scalar
    const bool okr = (int64_t)l31 < rows_left;
    // (GROUPED: the image already holds the transformed observations)
    if constexpr (!GROUPED) {
      if constexpr (!YSHARE) {
        float* ys_ = reinterpret_cast<float*>(smem + C::OFS_Y + (b_ * 4 + wave) * 256);
        if (lane < 32) ys_[lane] = okr ? __builtin_fmaf(ys_[lane], GLMH_GSCALE, -0.5f * GLMH_GSCALE) : 0.0f;
      } else if (wave == 0) {
        // the shared slot, all 64 rows of the super-tile (the other waves read it behind the tile's barrier)
        float* ys_ = reinterpret_cast<float*>(smem + C::OFS_Y + b_ * 1024);
        const int64_t left64 = n_rows - (st_ - row_st0) * (NRT * 32);
        ys_[lane] = (int64_t)lane < left64 ? __builtin_fmaf(ys_[lane], GLMH_GSCALE, -0.5f * GLMH_GSCALE) : 0.0f;
      }
    }
    return okr;
  };
  auto gemm1_aux = [&](bool okr) -> f32x16v {
    const uint32_t a0 = (h == 0 && okr) ? (F16_2P15 | (F16_2P15 << 16)) : 0u;   // k slots {0, 1}
    const uint32_t a1 = (h == 0 && okr) ? F16_2P15 : 0u;                        // k slot 2
    const f32x16v zero = {};
    return GLMH_MFMA1(as_f16x8(a0, a1, 0u, 0u), b_aux, zero);
  };
  auto load_a = [&](const unsigned char* Xt, int c, f16x8 (&xa)[2]) {
    const int ao = c == 0 ? a_ofs0 : a_ofs1;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) xa[pl] = *reinterpret_cast<const f16x8*>(Xt + pl * GLMP_PLANE + ao);
  };
  // element-wise on one accumulator element, plain f32 instructions only (see glm_planes.h); returns
  // 2^14 g
  auto elem1 = [&](float acc, float yh, int par) -> float {
#ifdef PA_GLMH_ABL_NOELEM
    return acc + yh;
#endif
    const float l2 = acc * dsc;
#ifdef PA_GLMH_ABL_NOTRANS
    const float e = __builtin_fabsf(l2) * -0.001f;
    const float t = e + 1.0f;
    const float inv = t * 0.5f;
#else
    const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(l2));
    const float t = e + 1.0f;
    const float inv = __builtin_amdgcn_rcpf(t);
#endif
    if constexpr (!LIN) s_yl[par] = __builtin_fmaf(yh, l2, s_yl[par]);
    // (spelled out: left to itself the compiler sometimes materialises |l2| with a v_and first)
    asm("v_add_f32 %0, |%1|, %0" : "+v"(s_abs[par]) : "v"(l2));
    p_t[par] *= t;
    const float g = yh - __builtin_copysignf(__builtin_fmaf(inv, GLMH_GSCALE, -0.5f * GLMH_GSCALE), l2);
    s_g[par] += g;
    return g;
  };
  // two elements at once: ONE reciprocal serves both -- r = 1 / (t0 t1), 1 / t0 = r t1, 1 / t1 = r t0 --
  // and the product t0 t1 is also what the log-sum chain multiplies in: per pair one transcendental
  // and the two chain multiplies become four plain multiplies (transcendentals issue at a quarter
  // of the plain rate).  t in [1, 2]: no range issue; two more roundings per sigmoid (~2.5 ulp).
  auto elem2 = [&](float acc0, float acc1, float yh0, float yh1, int chain, float& g0, float& g1) {
#if defined(PA_GLMH_ABL_NOELEM) || defined(PA_GLMH_ABL_NOTRANS)
    g0 = elem1(acc0, yh0, 0);
    g1 = elem1(acc1, yh1, 1);
    return;
#endif
    const float l0 = acc0 * dsc, l1 = acc1 * dsc;
#ifdef PA_GLMH_OLD_PAIR
    const float t0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(l0)) + 1.0f;
    const float t1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(l1)) + 1.0f;
    const float tt = t0 * t1;
    const float r = __builtin_amdgcn_rcpf(tt);
    const float inv0 = r * t1, inv1 = r * t0;
#else
    // t0 = 1 + e0 is never formed: t0 t1 = e0 t1 + t1 and 1 / t1 = r t0 = r e0 + r (one instruction fewer per
    // pair, and one rounding fewer on each of the two)
    const float e0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(l0));
    const float t1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(l1)) + 1.0f;
    const float tt = __builtin_fmaf(e0, t1, t1);
    const float r = __builtin_amdgcn_rcpf(tt);
    const float inv0 = r * t1, inv1 = __builtin_fmaf(r, e0, r);
#endif
    if constexpr (!LIN) {
      s_yl[0] = __builtin_fmaf(yh0, l0, s_yl[0]);
      s_yl[1] = __builtin_fmaf(yh1, l1, s_yl[1]);
    }
    asm("v_add_f32 %0, |%1|, %0" : "+v"(s_abs[0]) : "v"(l0));
    asm("v_add_f32 %0, |%1|, %0" : "+v"(s_abs[1]) : "v"(l1));
    p_t[chain] *= tt;
    g0 = yh0 - __builtin_copysignf(__builtin_fmaf(inv0, GLMH_GSCALE, -0.5f * GLMH_GSCALE), l0);
    g1 = yh1 - __builtin_copysignf(__builtin_fmaf(inv1, GLMH_GSCALE, -0.5f * GLMH_GSCALE), l1);
    s_g[0] += g0;
    s_g[1] += g1;
  };
  // (1 + e <= 2: the mantissa product of 8 tiles x 8 factors per chain stays below 2^64; its
  //  exponent is harvested every 8th tile and after the loop -- the rounding of the product is
  //  relative whatever its magnitude)
  auto renorm = [&]() {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      e_t[c2] += __builtin_amdgcn_frexp_expf(p_t[c2]);
      p_t[c2] = __builtin_amdgcn_frexp_mantf(p_t[c2]);
    }
  };
  auto tr_wait = [&](v2u32 (&xlo)[2], v2u32 (&xhi)[2], f16x8 (&xb)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xlo[0]), "+v"(xhi[0]), "+v"(xlo[1]), "+v"(xhi[1])
                 :
                 : "memory");
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const u32x4v cc = {xlo[pl][0], xlo[pl][1], xhi[pl][0], xhi[pl][1]};
      xb[pl] = __builtin_bit_cast(f16x8, cc);
    }
  };
  auto tr_issue = [&](uint32_t tr_a, uint32_t tr_b, int kh, v2u32 (&xlo)[2], v2u32 (&xhi)[2]) {
#ifdef PA_GLMH_ABL_NOTR
    xlo[0] = xhi[0] = xlo[1] = xhi[1] = v2u32{tr_a, tr_b};
    return;
#endif
    const uint32_t a = tr_a + (kh ? 1024u : 0u), b2 = tr_b + (kh ? 1024u : 0u);
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xlo[0]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xhi[0]) : "v"(b2));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xlo[1]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xhi[1]) : "v"(b2));
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

  // ---- software pipeline as in glm_planes.h: GEMM1 of tile it+1 runs against the element-wise work
  //      and GEMM2 of tile it ---------------------------------------------------------------------
  f32x16v acc_cur = {};
  if (my_count > 0) {
    wait_tiles_in_flight(std::integral_constant<int, NB - 2>{});
    if constexpr (!PRIV) __builtin_amdgcn_s_barrier();
    const bool ok0 = prep_rows(0, st);
    acc_cur = gemm1_aux(ok0);
    const unsigned char* X0 = smem + C::OFS_RING + (PRIV ? wave * NB * GLMH_TILE : rt * GLMH_TILE);
    f16x8 xa[2];
    load_a(X0, 0, xa);
#pragma unroll
    for (int t = 0; t < 3; ++t)
      acc_cur = GLMH_MFMA1(xa[TA[t]], wa0[TB[t]], acc_cur);
    load_a(X0, 1, xa);
#pragma unroll
    for (int t = 0; t < 3; ++t)
      acc_cur = GLMH_MFMA1(xa[TA[t]], wa1[TB[t]], acc_cur);
  }
  // one tile; the accumulator of the tile after it is built in `acc_nxt` while `acc_cur` is consumed
  // -- the loop below calls it with the two accumulators swapped every other tile, so that the
  // hand-over costs no register copies
  int64_t it = 0;
#if PA_GLMH_DEFER
  f16x8 pend_ga[2] = {}, pend_xb[2] = {};       // (zeros the first time: those MFMAs add nothing)
#endif
  auto tile = [&](const f32x16v& acc_cur, f32x16v& acc_nxt) {
#if defined(PA_GLMH_ABL_NOPRIO)
    if constexpr (false) {
      const uint32_t ph = 0;
#elif defined(PA_GLMH_PRIO_BY_TILE)
    if constexpr (OCC > 1) {
      const uint32_t ph = ((uint32_t)(it >> 2) + prio_slot) % (uint32_t)OCC;
#else
    if constexpr (OCC > 1) {
      const uint32_t ph = ((uint32_t)(prio_clock >> 8) + prio_slot) % (uint32_t)OCC;
#endif
      if (ph == 0) __builtin_amdgcn_s_setprio(0);
      else if (ph == 1) __builtin_amdgcn_s_setprio(1);
      else if (ph == 2) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(3);
#if !defined(PA_GLMH_PRIO_BY_TILE)
      prio_clock = wall_clock64();
#endif
    }
    if ((it & 7) == 7) renorm();
    int bn = bi + 1 == NB ? 0 : bi + 1;
    wait_tiles_in_flight(std::integral_constant<int, NB - 3>{});
    if constexpr (!PRIV) __builtin_amdgcn_s_barrier();
    {
      // (issued right behind the barrier by every wave.  Moving the issue into the element-wise stream of half
      //  or all of the waves -- so that the ~100 issue cycles of a piece do not fall on every wave of a SIMD at
      //  once -- was measured at 76 / 79 us against 54: profiles/r06_glm16_cycle_budget.txt)
      int bf = bi + (NB - 1);
      bf = bf >= NB ? bf - NB : bf;
      issue(st + (NB - 1) * grid, bf);
    }
    const unsigned char* Xc = smem + C::OFS_RING + (PRIV ? (wave * NB + bi) * GLMH_TILE : bi * ST_BYTES + rt * GLMH_TILE);
    const unsigned char* Xn = smem + C::OFS_RING + (PRIV ? (wave * NB + bn) * GLMH_TILE : bn * ST_BYTES + rt * GLMH_TILE);
    const float* ysc = reinterpret_cast<const float*>(smem + C::OFS_Y +
                                                      (!YSHARE ? (bi * 4 + wave) * 256 : bi * 1024 + rt * 128));
    const uint32_t tr_a = (uint32_t)(uintptr_t)Xc + (uint32_t)tr_ofs_a;
    const uint32_t tr_b = (uint32_t)(uintptr_t)Xc + (uint32_t)tr_ofs_b;

    const bool okn = prep_rows(bn, st + grid);
    acc_nxt = gemm1_aux(okn);
    v2u32 xlo[2], xhi[2];
    f16x8 xa[2], xb[2];
    float yv[8], g[8];
    uint32_t g1[4], g2[4];

    // -- GEMM1(it+1)  ||  element-wise(it, K half 0) and its split
    tr_issue(tr_a, tr_b, 0, xlo, xhi);
    load_y(ysc, 0, yv);
    load_a(Xn, 0, xa);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc_nxt = GLMH_MFMA1(xa[TA[t]], wa0[TB[t]], acc_nxt);
#if PA_GLMH_DEFER
      gwacc = GLMH_MFMA2(pend_ga[TA[t]], pend_xb[TB[t]], gwacc);      // GEMM2(it - 1, K half 1)
#endif
      elem2(acc_cur[2 * t], acc_cur[2 * t + 1], yv[2 * t], yv[2 * t + 1], t & 1, g[2 * t], g[2 * t + 1]);
      GLMH_SB();
    }
    load_a(Xn, 1, xa);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      acc_nxt = GLMH_MFMA1(xa[TA[t]], wa1[TB[t]], acc_nxt);
      if (t == 0) {
        elem2(acc_cur[6], acc_cur[7], yv[6], yv[7], 1, g[6], g[7]);
      } else {
        split_pair_g(g[4 * (t - 1)], g[4 * (t - 1) + 1], g1[2 * (t - 1)], g2[2 * (t - 1)]);
        split_pair_g(g[4 * (t - 1) + 2], g[4 * (t - 1) + 3], g1[2 * (t - 1) + 1], g2[2 * (t - 1) + 1]);
      }
      GLMH_SB();
    }
    tr_wait(xlo, xhi, xb);
    // -- GEMM2(it, K half 0)  ||  element-wise(it, K half 1)
    {
      const f16x8 ga[2] = {as_f16x8(g1[0], g1[1], g1[2], g1[3]), as_f16x8(g2[0], g2[1], g2[2], g2[3])};
      tr_issue(tr_a, tr_b, 1, xlo, xhi);
      load_y(ysc, 1, yv);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (GLMH_G_PIECE(t)) gwacc = GLMH_MFMA2(ga[TA[t]], xb[TB[t]], gwacc);
        // (pairs 0, 1 | 2 | 3 of the K half beside the three MFMAs)
        const int q0 = t == 0 ? 0 : t + 1, q1 = t == 0 ? 2 : t + 2;
#pragma unroll
        for (int qp = q0; qp < q1; ++qp)
          elem2(acc_cur[8 + 2 * qp], acc_cur[9 + 2 * qp], yv[2 * qp], yv[2 * qp + 1], qp & 1, g[2 * qp],
                g[2 * qp + 1]);
        GLMH_SB();
      }
    }
    uint32_t h1[4], h2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair_g(g[2 * j], g[2 * j + 1], h1[j], h2[j]);
    tr_wait(xlo, xhi, xb);
    // -- GEMM2(it, K half 1)
    {
      const f16x8 ga[2] = {as_f16x8(h1[0], h1[1], h1[2], h1[3]), as_f16x8(h2[0], h2[1], h2[2], h2[3])};
#if PA_GLMH_DEFER
      pend_ga[0] = ga[0]; pend_ga[1] = ga[1];
      pend_xb[0] = xb[0]; pend_xb[1] = xb[1];
#else
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (GLMH_G_PIECE(t)) gwacc = GLMH_MFMA2(ga[TA[t]], xb[TB[t]], gwacc);
#endif
    }
    st += grid;
    bi = bn;
    ++it;
  };
  f32x16v acc_alt = {};
  if constexpr (OCC >= 4) {
    // (128 registers per wave: the second accumulator set of the unrolled form would spill)
    while (it < my_count) {
      tile(acc_cur, acc_alt);
      acc_cur = acc_alt;
    }
  } else {
    while (it + 1 < my_count) {
      tile(acc_cur, acc_alt);
      tile(acc_alt, acc_cur);
    }
    if (it < my_count) tile(acc_cur, acc_alt);
  }
#if PA_GLMH_DEFER
#pragma unroll
  for (int t = 0; t < 3; ++t) gwacc = GLMH_MFMA2(pend_ga[TA[t]], pend_xb[TB[t]], gwacc);
#endif
  wait_vmcnt<0>();
  __builtin_amdgcn_s_setprio(0);
  renorm();
  __syncthreads();

  // ---- block reduction over the row tiles in a fixed order, one partial record in the format of
  //      glm.hip; the power-of-two scales come out here (exact) --------------------------------------
  // (records in the 64-particle format of glm_finalize.h whatever NPT: one per PAIR of particle tiles)
  constexpr int NPG = NPT / 2, REC2 = 2 * 1024 + 2 * 2 * 32;
  constexpr int RED_OFS = NPT > 2 ? 0 : C::OFS_RING;      // (the W planes are not read any more: NPT > 2 needs the room)
  static_assert((NPT * 1024 + 2 * NPT * 64) * 4 <= C::LDS_BYTES - RED_OFS, "LDS too small");
  float* red = reinterpret_cast<float*>(smem + RED_OFS);
  float* red2 = red + NPT * 1024;
  const float g_dsc = 1.0f / GLMH_GSCALE;
  const float s_lg = (float)(e_t[0] + e_t[1]) + (__builtin_amdgcn_logf(p_t[0]) + __builtin_amdgcn_logf(p_t[1]));
  const float ll_acc = 0.69314718055994530942f *
                       ((s_yl[0] + s_yl[1]) * g_dsc - 0.5f * (s_abs[0] + s_abs[1]) - s_lg);
  const float gb_acc = (s_g[0] + s_g[1]) * g_dsc;
  for (int rr = 0; rr < NRT; ++rr) {
    if (rt == rr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = (pt * 16 + r) * 64 + lane;
        red[idx] = (rr == 0 ? 0.0f : red[idx]) + ldexpf(gwacc[r], -(14 + kx_l));
      }
      const int i0 = (2 * pt) * 64 + lane, i1 = (2 * pt + 1) * 64 + lane;
      red2[i0] = (rr == 0 ? 0.0f : red2[i0]) + ll_acc;
      red2[i1] = (rr == 0 ? 0.0f : red2[i1]) + gb_acc;
    }
    __syncthreads();
  }
  for (int gi = 0; gi < NPG; ++gi) {
    const int pass = (int)blockIdx.y * NPG + gi;
    if ((int64_t)pass * 64 >= P) continue;              // (a pair of particle tiles past the end: no record)
    float* rec = part + ((int64_t)pass * gridDim.x + blockIdx.x) * REC2;
    for (int i = threadIdx.x; i < 2 * 1024; i += NT) rec[i] = red[gi * 2048 + i];
    for (int i = threadIdx.x; i < 2 * 2 * 32; i += NT) {
      const int qq = i >> 5, j = i & 31, ptl = qq >> 1, which = qq & 1;
      const int src = (2 * (2 * gi + ptl) + which) * 64;
      float v = red2[src + j] + red2[src + 32 + j];
      if constexpr (LIN) {
        // workgroup 0 of the pass: + c . w_p + c0 b_p (natural-log units, float64) on the ll slots
        const int p = pass * 64 + ptl * 32 + j;
        if (blockIdx.x == 0 && which == 0 && p < P) {
          // (DRAW: this workgroup wrote z_w / z_b itself in its prologue, many barriers ago)
          const float* wq = DRAW ? draw.z_w : w;
          const float* bq = DRAW ? (draw.have_b ? draw.z_b : nullptr) : b;
          double lin = bq != nullptr ? moments[32] * (double)bq[p] : 0.0;
          for (int d = 0; d < D; ++d) lin = __builtin_fma(moments[d], (double)wq[(int64_t)p * w_stride + d], lin);
          v += (float)lin;
        }
      }
      rec[2 * 1024 + i] = v;
    }
  }
  if (tstamps != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_max(&tstamps[1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace pa
