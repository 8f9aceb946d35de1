// bow.hip -- the first layer of an amortised guide over word histograms (examples/lda.py) without the
// per-step histogram.
//
// Reference path replaced (examples/lda.py:113-121 + 76-92, every ELBO-gradient step):
//   counts = torch.zeros(V, B).scatter_add(0, data, torch.ones(data.shape))     410 MB at V=1024, B=1e5
//   h = nn.Linear(V, H)(counts.transpose(0, 1))                                 + its backward
// i.e. a fill, a scatter, transposing copies and four f32 rocBLAS products over a dense matrix that
// is a pure function of the corpus.  The corpus does not change between steps, so its histogram is
// built ONCE and kept as two bf16 images (counts <= words per document: exact in bf16) already in
// MFMA-operand order; per step only the weight W[H, V] (forward) and the upstream gradient d[B, H]
// (backward) are split exactly into three bf16 pieces, and the two products
//   h[b, j]  = bias[j] + sum_v C[b, v] W[j, v]                 (K = V)
//   dW[j, v] = sum_b d[b, j] C[b, v]                            (K = B: split over workgroups,
//                                                                 partials reduced in a fixed order)
// run on the bf16 matrix cores with f32 accumulation: every piece product is exact in f32, the sums
// round like an f32 GEMM (f32-class result, 3 MFMA products per f32 product).
//
// Operand images (all blocks are 64 lanes x 16 B, lane-linear: ONE coalesced 1-KiB load feeds an MFMA):
//   image A  [B/32][V/16] blocks: lane l = doc (l & 31) of the tile, k group l >> 5: C[doc][16 kt + 8 kg .. +7]
//   image B  [V/32][B/16] blocks: lane l = word (l & 31) of the tile, k group: C[16 kt + 8 kg .. +7 docs][word]
//   W planes [3][V/16][H/32] blocks: lane l = hidden unit (l & 31) of the tile, k group: W[j][16 kt + 8 kg .. +7]
//   d planes [3][H/32][B/16] blocks: lane l = hidden unit (l & 31), k group: d[16 kt + 8 kg .. +7 docs][j]
#include "common.h"
#include "glm_bf16.h"
#include "lds_dma.h"

namespace pa {

constexpr int BOW_HT = 4;                 // hidden units are padded to 128 = 4 tiles of 32

// exact 3-way bf16 split of 8 floats -> three 16-byte operand chunks
__device__ __forceinline__ void split8(const float (&v)[8], uint4& c1, uint4& c2, uint4& c3) {
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p1[j], p2[j], p3[j]);
  c1 = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  c2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  c3 = make_uint4(p3[0], p3[1], p3[2], p3[3]);
}

// W[H, V] f32 -> planes[3][V/16][BOW_HT] blocks (B operand of the forward product)
__global__ __launch_bounds__(256) void bow_split_w_kernel(const float* __restrict__ W, int H, int V,
                                                          uint4* __restrict__ planes) {
  const int64_t nblk = (int64_t)(V / 16) * BOW_HT;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nblk * 64) return;
  const int64_t blk = idx >> 6;
  const int lane = (int)(idx & 63);
  const int kt = (int)(blk / BOW_HT), nt = (int)(blk % BOW_HT);
  const int j = nt * 32 + (lane & 31), k0 = 16 * kt + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = j < H ? W[(int64_t)j * V + k0 + q] : 0.0f;
  uint4 c1, c2, c3;
  split8(v, c1, c2, c3);
  planes[idx] = c1;
  planes[nblk * 64 + idx] = c2;
  planes[2 * nblk * 64 + idx] = c3;
}

// d[B, H] f32 -> planes[3][BOW_HT][Bp/16] blocks (A operand of the backward product); rows >= B are zero.
// With `ymul` (the layer's sigmoid output y[B, H]) d is the gradient of the ACTIVATION and what is split is
// d * (1 - y) * y (torch's sigmoid_backward, never written); `db_part` [BOW_HT * Bp/16][32] receives, per
// operand block, the sums of its 16 rows for the block's 32 hidden units (the bias gradient's partial sums:
// summed over the k-steps by the caller, in a fixed order).
__global__ __launch_bounds__(256) void bow_split_d_kernel(const float* __restrict__ d,
                                                          const float* __restrict__ ymul, int64_t B,
                                                          int64_t Bp, int H, uint4* __restrict__ planes,
                                                          float* __restrict__ db_part) {
  const int64_t nkt = Bp / 16, nblk = (int64_t)BOW_HT * nkt;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nblk * 64) return;
  const int64_t blk = idx >> 6;
  const int lane = (int)(idx & 63);
  const int mt = (int)(blk / nkt);
  const int64_t kt = blk % nkt;
  const int j = mt * 32 + (lane & 31);
  const int64_t b0 = 16 * kt + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = (j < H && b0 + q < B) ? d[(b0 + q) * H + j] : 0.0f;
  if (ymul != nullptr) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float y = (j < H && b0 + q < B) ? ymul[(b0 + q) * H + j] : 0.0f;
      v[q] = v[q] * (1.0f - y) * y;
    }
  }
  if (db_part != nullptr) {               // (whole waves are inside nblk * 64: every lane takes part)
    float sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    sum += __shfl_xor(sum, 32, 64);
    if (lane < 32) db_part[blk * 32 + lane] = sum;
  }
  uint4 c1, c2, c3;
  split8(v, c1, c2, c3);
  planes[idx] = c1;
  planes[nblk * 64 + idx] = c2;
  planes[2 * nblk * 64 + idx] = c3;
}

typedef float f32x16b __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16x8 as_op(const uint4& c) { return as_bf16x8(c.x, c.y, c.z, c.w); }

// h[B, H] = bias + C W^T.  Work unit = 256 documents = 4 waves x two 32-row tiles x 128 hidden units,
// two workgroups per CU.  The W planes travel in chunks of two k-steps (2 x 12 operand blocks = 24
// KiB) L2 -> LDS by DMA (global_load_lds: no registers, each wave copies a quarter, lane-linear) into
// a DOUBLE buffer one chunk ahead of the MFMAs, all four waves read their B operands from there; the
// histogram operands (A) stream from HBM straight into registers one chunk ahead (two register sets:
// a third one, two chunks ahead, does not fit next to the 128 accumulator registers at two waves per
// SIMD).  Every VMEM operation of the loop is inline asm with hand-counted s_waitcnt: hipcc counts
// only the loads it can see, so a compiler-placed wait for an A operand also drained the younger W
// DMAs of the next chunk (147 us; before that: one LDS buffer filled through registers between two
// barriers 243 us; every wave reading the planes itself 250 us -- for 26 us of MFMA work and 205 MB
// of image at B = 1e5).
constexpr int BOW_KC = 2;                                   // k-steps per LDS chunk
constexpr int BOW_CHUNK_BLOCKS = BOW_KC * BOW_HT * 3;       // 24 blocks of 1 KiB
constexpr int BOW_FWD_LDS = 2 * BOW_CHUNK_BLOCKS * 1024;    // 48 KiB
constexpr int BOW_NA = 2 * BOW_KC;                          // A loads per chunk and lane

struct BowASet { u32x4v t[2][BOW_KC]; };                    // [tile][k-step]

__device__ __forceinline__ void bow_gload(u32x4v& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
// everything but the youngest N VMEM operations has landed; names the registers of the set about to be
// consumed so that nothing reading them moves above the wait
template <int N>
__device__ __forceinline__ void bow_wait(BowASet& a) {
  static_assert(BOW_KC == 2, "operand list");
  asm volatile("s_waitcnt vmcnt(%4)"
               : "+v"(a.t[0][0]), "+v"(a.t[0][1]), "+v"(a.t[1][0]), "+v"(a.t[1][1])
               : "n"(N)
               : "memory");
}

__global__ __launch_bounds__(256, 2) void bow_linear_fwd_kernel(const uint4* __restrict__ imgA,
                                                                const uint4* __restrict__ wpl,
                                                                const float* __restrict__ bias,
                                                                int64_t B, int V, int H, int act,
                                                                float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bow_smem[];
  const uint4* wsm = reinterpret_cast<const uint4*>(bow_smem);   // [buffer][k-step][tile][plane][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)bow_smem);
  const int64_t nmt = (B + 31) / 32;                                  // document tiles
  const int64_t nunits = (nmt + 7) / 8;
  const int nkt = V / 16, nch = nkt / BOW_KC;                         // (nch is a multiple of 4)
  const int64_t wblk = (int64_t)nkt * BOW_HT * 64;                    // uint4 per W plane
  // this wave's quarter of chunk c: (k-step, tile, plane) = block index
  auto issue_w = [&](int c, int buf) {
#pragma unroll
    for (int q = 0; q < BOW_CHUNK_BLOCKS / 4; ++q) {
      const int blk = wave * (BOW_CHUNK_BLOCKS / 4) + q;
      const int ks = blk / (BOW_HT * 3), n = (blk / 3) % BOW_HT, pl = blk % 3;
      dma16_cached(wpl + ((int64_t)(c * BOW_KC + ks) * BOW_HT + n) * 64 + lane + pl * wblk,
                   lds_base + (uint32_t)((buf * BOW_CHUNK_BLOCKS + blk) * 1024));
    }
  };
  for (int64_t unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const int64_t mt0 = (unit * 4 + wave) * 2;
    const bool live = mt0 < nmt, two = mt0 + 1 < nmt;
    f32x16b acc[2][BOW_HT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < BOW_HT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const uint4* a0p = imgA + ((live ? mt0 : 0) * nkt) * 64 + lane;
    const uint4* a1p = imgA + ((two ? mt0 + 1 : (live ? mt0 : 0)) * nkt) * 64 + lane;
    auto issue_a = [&](int c, BowASet& s) {
#pragma unroll
      for (int q = 0; q < BOW_KC; ++q) {
        bow_gload(s.t[0][q], a0p + (int64_t)(c * BOW_KC + q) * 64);
        bow_gload(s.t[1][q], a1p + (int64_t)(c * BOW_KC + q) * 64);
      }
    };
    // One chunk: set `cur` holds A(c) (requested one chunk ago together with W(c)), `nxt` receives
    // A(c + 1).  Nothing of this wave is in flight across the barrier.
    auto chunk = [&](int c, BowASet& cur, BowASet& nxt) {
      const int buf = c & 1;
      bow_wait<0>(cur);
      __syncthreads();                       // everybody's share of W(c); the other buffer is free
      if (c + 1 < nch) {
        issue_w(c + 1, buf ^ 1);
        // (never past the end: the registers of a load nobody waits for are free for the compiler
        // to reuse -- data landing later would overwrite whatever lives there, e.g. the epilogue's
        // store addresses)
        issue_a(c + 1, nxt);
      }
      const uint4* wb = wsm + (int64_t)buf * BOW_CHUNK_BLOCKS * 64;
#pragma unroll
      for (int ks = 0; ks < BOW_KC; ++ks) {
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, cur.t[0][ks]), a1 = __builtin_bit_cast(bf16x8, cur.t[1][ks]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int n = 0; n < BOW_HT; ++n) {
            const bf16x8 bb = as_op(wb[((ks * BOW_HT + n) * 3 + pl) * 64 + lane]);
            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1][n], 0, 0, 0);
          }
      }
    };
    // (the previous unit's last chunk sits in buffer 1 -- nch is even -- and this wave is past the
    // barrier that ended the reads of buffer 0)
    BowASet s0, s1;
    issue_w(0, 0);
    issue_a(0, s0);
    for (int c = 0; c < nch; c += 2) {         // (nch is even)
      chunk(c, s0, s1);
      chunk(c + 1, s1, s0);
    }
    wait_vmcnt<0>();                           // nothing asynchronous is left when registers change hands
    if (!live) continue;
    // C/D layout: lane = column (hidden unit), register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int jl = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m == 1 && !two) break;
#pragma unroll
      for (int n = 0; n < BOW_HT; ++n) {
        const int j = n * 32 + jl;
        if (j >= H) continue;
        const float bj = bias != nullptr ? bias[j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t doc = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float h = acc[m][n][r] + bj;
          if (doc < B) out[doc * H + j] = act ? fast_sigmoid(h) : h;
        }
      }
    }
  }
}

// partial[s][H_pad = 128][V] = sum over the document k-steps of chunk s of d^T C: workgroup (vb, s),
// wave = one tile of 32 hidden units x the workgroup's 128 words
__global__ __launch_bounds__(256) void bow_linear_bwd_kernel(const uint4* __restrict__ dpl,
                                                             const uint4* __restrict__ imgB,
                                                             int64_t Bp, int V, int ksplit,
                                                             float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // wave = hidden tile
  // The V / 128 workgroups of one document chunk s all read the chunk's d planes (77 MB in total at
  // B = 1e5, H = 128): dispatched to different XCDs -- consecutive workgroup ids go round-robin over
  // the 8 XCDs -- every L2 fetched its own copy from HBM (820 MB of traffic, 130 us).  The linear id is
  // decoded so that the workgroups of a chunk share an XCD and run back to back: the planes come from
  // HBM once and from that L2 afterwards.
  const int nvb = V / 128;
  const int64_t L = blockIdx.x;
  const int xcd = (int)(L & 7);
  const int64_t slot = L >> 3;
  const int vb = (int)(slot % nvb);
  const int s = (int)((slot / nvb) * 8 + xcd);
  if (s >= ksplit) return;
  const int64_t nkt = Bp / 16;
  const int64_t per = (nkt + ksplit - 1) / ksplit;
  const int64_t k_lo = (int64_t)s * per, k_hi = (k_lo + per < nkt) ? k_lo + per : nkt;
  const int64_t dblk = (int64_t)BOW_HT * nkt * 64;                    // chunks per d plane
  f32x16b acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  const uint4* ap = dpl + ((int64_t)wave * nkt) * 64 + lane;
  const uint4* bp = imgB + ((int64_t)vb * 4 * nkt) * 64 + lane;
  // The operands of a k-step are 7 independent 16-byte loads per lane and feed 12 MFMAs (384 pipe
  // clocks): issued in the iteration that uses them the loop runs at the memory latency (~2000
  // clocks per k-step and wave; measured 134 us for 40 us of matrix work at B = 1e5).  Three register
  // sets, loads two k-steps ahead (clamped at the end: a re-read of the last step, never consumed).
  struct Ops { uint4 a[3], b[4]; };
  auto load = [&](int64_t kt, Ops& o) {
    const int64_t k = kt < k_hi ? kt : k_hi - 1;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) o.a[pl] = ap[k * 64 + pl * dblk];
#pragma unroll
    for (int n = 0; n < 4; ++n) o.b[n] = bp[((int64_t)n * nkt + k) * 64];
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_op(o.a[pl]), as_op(o.b[n]), acc[n], 0, 0, 0);
  };
  if (k_lo < k_hi) {
    Ops o0, o1, o2;
    load(k_lo, o0);
    load(k_lo + 1, o1);
    int64_t kt = k_lo;
    for (; kt + 2 < k_hi; kt += 3) {
      load(kt + 2, o2);
      mma(o0);
      load(kt + 3, o0);
      mma(o1);
      load(kt + 4, o1);
      mma(o2);
    }
    if (kt < k_hi) mma(o0);
    if (kt + 1 < k_hi) mma(o1);
  }
  float* dst = partial + (int64_t)s * 128 * V;
  const int wl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int v = (vb * 4 + n) * 32 + wl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      dst[(int64_t)j * V + v] = acc[n][r];
    }
  }
}

// out[M <= 128, N <= 128] = A^T X for tall operands A[B, M], X[B, N] (f32, both split exactly into three
// bf16 pieces: the six piece products of order >= 2^-16): the weight gradient of a Linear layer over a
// large batch (examples/lda.py:76-92: dW2 = d2^T h1 with B = 1e5 documents), which rocBLAS runs as a
// 100 x 100 x 1e5 product without splitting the long dimension.  a_pl / x_pl: planes[3][4][Bp/16]
// blocks from bow_split_d_kernel (lane l = column (l & 31) of the tile, k group: rows 16 kt + 8 kg ..).
// Workgroup s sums its chunk of k-steps; wave = one 32-row tile of the output, all four column tiles.
__global__ __launch_bounds__(256) void tsgemm_tn_kernel(const uint4* __restrict__ a_pl,
                                                        const uint4* __restrict__ x_pl, int64_t Bp,
                                                        int ksplit, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x;
  const int64_t nkt = Bp / 16;
  const int64_t per = (nkt + ksplit - 1) / ksplit;
  const int64_t k_lo = (int64_t)s * per, k_hi = (k_lo + per < nkt) ? k_lo + per : nkt;
  const int64_t blk = (int64_t)BOW_HT * nkt * 64;                     // chunks per plane
  f32x16b acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
  constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
  // 15 independent operand loads per k-step for 24 MFMAs: two register sets, one k-step ahead (see
  // bow_linear_bwd_kernel)
  struct Ops { uint4 a[3], b[4][3]; };
  auto load = [&](int64_t kt, Ops& o) {
    const int64_t k = kt < k_hi ? kt : k_hi - 1;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) o.a[pl] = a_pl[((int64_t)wave * nkt + k) * 64 + lane + pl * blk];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) o.b[n][pl] = x_pl[((int64_t)n * nkt + k) * 64 + lane + pl * blk];
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int t = 0; t < 6; ++t)          // smallest products first
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_op(o.a[TA[t]]), as_op(o.b[n][TB[t]]), acc[n], 0, 0, 0);
  };
  if (k_lo < k_hi) {
    Ops o0, o1;
    load(k_lo, o0);
    int64_t kt = k_lo;
    for (; kt + 1 < k_hi; kt += 2) {
      load(kt + 1, o1);
      mma(o0);
      load(kt + 2, o0);
      mma(o1);
    }
    if (kt < k_hi) mma(o0);
  }
  float* dst = partial + (int64_t)s * 128 * 128;
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      dst[m * 128 + n * 32 + cl] = acc[n][r];
    }
}

// out[m, n] = sum_s partial[s][m][n]: 8 outputs x 32 chunk groups per workgroup (thread (j, g) sums the
// chunks g, g + 32, ... in fp64, the 32 group sums are added in order): fixed order, whole chip
__global__ __launch_bounds__(256) void tsgemm_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                            int M, int N, float* __restrict__ out) {
  __shared__ double sm[32][8];
  const int jj = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int idx = blockIdx.x * 8 + jj;
  double acc = 0.0;
  if (idx < M * N) {
    const int m = idx / N, n = idx - m * N;
    for (int s = g; s < ksplit; s += 32) acc += (double)partial[(int64_t)s * 128 * 128 + m * 128 + n];
  }
  sm[g][jj] = acc;
  __syncthreads();
  if (g == 0 && idx < M * N) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sm[k][jj];
    out[idx] = (float)t;
  }
}

// dW[H, V] = sum_s partial[s] in increasing s (fp64), one thread per element
__global__ __launch_bounds__(256) void bow_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                         int H, int V, float* __restrict__ dW) {
  // 64 consecutive outputs x 4 chunk groups per workgroup (coalesced 256-byte rows of every partial)
  __shared__ double sm[4][64];
  const int jj = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + jj;
  double acc = 0.0;
  if (idx < (int64_t)H * V)
    for (int s = g; s < ksplit; s += 4) acc += (double)partial[(int64_t)s * 128 * V + idx];
  sm[g][jj] = acc;
  __syncthreads();
  if (g == 0 && idx < (int64_t)H * V) dW[idx] = (float)(((sm[0][jj] + sm[1][jj]) + sm[2][jj]) + sm[3][jj]);
}

static int bow_ksplit(int64_t Bp, int V) {
  // (V / 128) x ksplit workgroups: about three per CU
  int64_t ks = (int64_t)cu_count() * 3 / (V / 128 > 0 ? V / 128 : 1);
  const int64_t nkt = Bp / 16;
  if (ks > nkt) ks = nkt;
  if (ks < 1) ks = 1;
  if (ks > 1024) ks = 1024;
  return (int)ks;
}

}  // namespace pa

extern "C" {

size_t pa_bow_workspace(int64_t B, int64_t V, int64_t H) {
  if (B < 0 || V < 128 || V % 128 != 0 || H < 1 || H > 128) return 0;
  const int64_t Bp = (B + 31) / 32 * 32;
  const size_t wplanes = (size_t)3 * (V / 16) * pa::BOW_HT * 64 * 16;
  const size_t dplanes = (size_t)3 * pa::BOW_HT * (Bp / 16) * 64 * 16;
  const size_t part = (size_t)pa::bow_ksplit(Bp, (int)V) * 128 * V * sizeof(float);
  const size_t big = dplanes + part;
  return wplanes > big ? wplanes : big;
}

static int pa_ts_ksplit(int64_t Bp) {
  int64_t ks = (int64_t)pa::cu_count();
  const int64_t nkt = Bp / 16;
  if (ks > nkt) ks = nkt;
  return (int)(ks < 1 ? 1 : ks);
}

size_t pa_tsgemm_tn_workspace(int64_t B, int64_t M, int64_t N) {
  if (B < 0 || M < 1 || M > 128 || N < 1 || N > 128) return 0;
  const int64_t Bp = (B + 31) / 32 * 32;
  const size_t planes = (size_t)3 * pa::BOW_HT * (Bp / 16) * 64 * 16;
  return 2 * planes + (size_t)pa_ts_ksplit(Bp) * 128 * 128 * sizeof(float);
}

int pa_tsgemm_tn(const float* A, const float* X, int64_t B, int64_t M, int64_t N, float* out,
                 void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(B >= 0 && M >= 1 && M <= 128 && N >= 1 && N <= 128,
             "tsgemm_tn: needs M, N <= 128 (B=%lld M=%lld N=%lld)", (long long)B, (long long)M, (long long)N);
  PA_REQUIRE(out != nullptr, "tsgemm_tn: NULL output");
  hipStream_t s = pa::as_stream(stream);
  if (B == 0) {
    if (hipMemsetAsync(out, 0, (size_t)M * N * sizeof(float), s) != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "tsgemm_tn: memset failed");
    return PA_OK;
  }
  PA_REQUIRE(A && X && workspace, "tsgemm_tn: NULL pointer");
  PA_REQUIRE(workspace_bytes >= pa_tsgemm_tn_workspace(B, M, N), "tsgemm_tn: workspace too small");
  const int64_t Bp = (B + 31) / 32 * 32;
  const size_t pbytes = (size_t)3 * pa::BOW_HT * (Bp / 16) * 64 * 16;
  uint4* apl = (uint4*)workspace;
  uint4* xpl = (uint4*)((char*)workspace + pbytes);
  float* part = (float*)((char*)workspace + 2 * pbytes);
  const int64_t nchunks = (int64_t)pa::BOW_HT * (Bp / 16) * 64;
  const unsigned g = (unsigned)((nchunks + 255) / 256);
  hipLaunchKernelGGL(pa::bow_split_d_kernel, dim3(g), dim3(256), 0, s, A, (const float*)nullptr, B, Bp, (int)M,
                     apl, (float*)nullptr);
  hipLaunchKernelGGL(pa::bow_split_d_kernel, dim3(g), dim3(256), 0, s, X, (const float*)nullptr, B, Bp, (int)N,
                     xpl, (float*)nullptr);
  const int ks = pa_ts_ksplit(Bp);
  hipLaunchKernelGGL(pa::tsgemm_tn_kernel, dim3((unsigned)ks), dim3(256), 0, s, (const uint4*)apl,
                     (const uint4*)xpl, Bp, ks, part);
  hipLaunchKernelGGL(pa::tsgemm_reduce_kernel, dim3((unsigned)((M * N + 7) / 8)), dim3(256), 0, s, part,
                     ks, (int)M, (int)N, out);
  return pa::check_launch("tsgemm_tn");
}

int pa_bow_linear_fwd(const void* image_a, const float* W, const float* bias, int64_t B, int64_t V,
                      int64_t H, float* out, void* workspace, size_t workspace_bytes,
                      pa_stream_t stream) {
  return pa_bow_linear_fwd_act(image_a, W, bias, B, V, H, 0, out, workspace, workspace_bytes, stream);
}

int pa_bow_linear_fwd_act(const void* image_a, const float* W, const float* bias, int64_t B, int64_t V,
                          int64_t H, int sigmoid_out, float* out, void* workspace,
                          size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(B >= 0 && V >= 128 && V % 128 == 0 && H >= 1 && H <= 128,
             "bow_linear_fwd: needs V a multiple of 128 and H <= 128 (B=%lld V=%lld H=%lld)",
             (long long)B, (long long)V, (long long)H);
  if (B == 0) return PA_OK;
  PA_REQUIRE(image_a && W && out && workspace, "bow_linear_fwd: NULL pointer");
  PA_REQUIRE(workspace_bytes >= pa_bow_workspace(B, V, H), "bow_linear_fwd: workspace too small");
  hipStream_t s = pa::as_stream(stream);
  uint4* wpl = (uint4*)workspace;
  const int64_t nchunks = (V / 16) * pa::BOW_HT * 64;
  hipLaunchKernelGGL(pa::bow_split_w_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, W,
                     (int)H, (int)V, wpl);
  hipEvent_t ev0, ev1;
  const bool br = pa::take_bracket(PA_KERNEL_LDA, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  const int64_t nmt = (B + 31) / 32, nunits = (nmt + 7) / 8;
  const int64_t slots = (int64_t)pa::cu_count() * 2;
  const int64_t nwg = nunits < slots ? nunits : slots;
  (void)hipFuncSetAttribute((const void*)pa::bow_linear_fwd_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, pa::BOW_FWD_LDS);
  hipLaunchKernelGGL(pa::bow_linear_fwd_kernel, dim3((unsigned)nwg), dim3(256), pa::BOW_FWD_LDS, s,
                     (const uint4*)image_a, (const uint4*)wpl, bias, B, (int)V, (int)H, sigmoid_out, out);
  if (br) (void)hipEventRecord(ev1, s);
  return pa::check_launch("bow_linear_fwd_kernel");
}

int pa_bow_linear_bwd(const void* image_b, const float* d_out, int64_t B, int64_t V, int64_t H,
                      float* dW, void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  return pa_bow_linear_bwd_act(image_b, d_out, nullptr, B, V, H, dW, nullptr, workspace, workspace_bytes,
                               stream);
}

int pa_bow_linear_bwd_act(const void* image_b, const float* d_out, const float* y_mul, int64_t B, int64_t V,
                          int64_t H, float* dW, float* db_partial, void* workspace, size_t workspace_bytes,
                          pa_stream_t stream) {
  PA_REQUIRE(B >= 0 && V >= 128 && V % 128 == 0 && H >= 1 && H <= 128,
             "bow_linear_bwd: needs V a multiple of 128 and H <= 128 (B=%lld V=%lld H=%lld)",
             (long long)B, (long long)V, (long long)H);
  PA_REQUIRE(dW != nullptr, "bow_linear_bwd: NULL output");
  hipStream_t s = pa::as_stream(stream);
  if (B == 0) {
    if (hipMemsetAsync(dW, 0, (size_t)H * V * sizeof(float), s) != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "bow_linear_bwd: memset failed");
    return PA_OK;
  }
  PA_REQUIRE(image_b && d_out && workspace, "bow_linear_bwd: NULL pointer");
  PA_REQUIRE(workspace_bytes >= pa_bow_workspace(B, V, H), "bow_linear_bwd: workspace too small");
  const int64_t Bp = (B + 31) / 32 * 32;
  uint4* dpl = (uint4*)workspace;
  const size_t dbytes = (size_t)3 * pa::BOW_HT * (Bp / 16) * 64 * 16;
  float* part = (float*)((char*)workspace + dbytes);
  const int64_t nchunks = (int64_t)pa::BOW_HT * (Bp / 16) * 64;
  hipLaunchKernelGGL(pa::bow_split_d_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, d_out,
                     y_mul, B, Bp, (int)H, dpl, db_partial);
  const int ks = pa::bow_ksplit(Bp, (int)V);
  const int64_t ks8 = (ks + 7) / 8 * 8;           // (workgroups with s >= ks return at once)
  hipLaunchKernelGGL(pa::bow_linear_bwd_kernel, dim3((unsigned)((V / 128) * ks8)), dim3(256), 0, s,
                     (const uint4*)dpl, (const uint4*)image_b, Bp, (int)V, ks, part);
  hipLaunchKernelGGL(pa::bow_reduce_kernel, dim3((unsigned)((H * V + 63) / 64)), dim3(256), 0, s, part, ks,
                     (int)H, (int)V, dW);
  return pa::check_launch("bow_linear_bwd");
}

}  // extern "C"
