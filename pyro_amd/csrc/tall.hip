// tall.hip -- a Linear layer over a tall batch (B rows >> 128 features) without rocBLAS and without
// separate operand-split passes: the inner layers of an amortised guide (examples/lda.py:76-92,
// nn.Linear(100, 100) and nn.Linear(100, 8) over 1e5 documents per ELBO-gradient step).
//
// Reference path replaced per layer and step: F.linear forward (one rocBLAS product), and in the
// backward  dx = g W  (rocBLAS),  dW = g^T x  (rocBLAS: a 100 x 100 x 1e5 product, long dimension not
// split),  db = g.sum(0)  (a reduce kernel) -- torch/nn/functional.py linear + autograd's
// AddmmBackward / MmBackward.  Measured on the MI355X at B = 1e5: 75 us (forward, 100 -> 100), 71 us
// (dx) -- 27 TFLOP/s on 2 GFLOP -- and, on this library's round-3 route for dW, two split passes of
// 25 us each in front of a 42-us product plus an 11-23 us reduce for db.
//
// Here, two kernels, f32 operands split exactly into three bf16 pieces in registers (the six piece
// products of order >= 2^-16 on the matrix cores, f32 accumulation: f32-class, as glm_bf16.h):
//   pa_tall_linear      Y[B, C] = G[B, R] Wm[R, C] (+ bias[C]),  R, C <= 128.  Wm is addressed through two
//                       strides, so the forward (Wm = weight^T) and dx (Wm = weight) need no transposed
//                       copy.  A wave owns 32 rows: A operands straight from HBM (two 16-byte loads per
//                       lane and k-step, split in registers), B operands = the Wm planes, split once per
//                       workgroup into LDS in operand order.
//   pa_tall_wgrad       dW[R, K] = G^T X,  db[R] = sum_b G  for G[B, R], X[B, K]: the long dimension is the
//                       MFMA K index; lane (r, k-group) reads 8 consecutive rows of its column -- for a
//                       fixed row the 32 lanes of a group read 32 consecutive floats -- and splits them
//                       in registers; every wave accumulates its own [32, 128] tile over its share of
//                       the rows, partial tiles are reduced in a fixed order (bitwise reproducible).
// A Sigmoid behind the layer (examples/lda.py:84-87: every Linear of the predictor is followed by one) rides
// along instead of taking two more passes over [B, C] in each direction: pa_tall_linear_act stores
// sigmoid(G Wm + bias) from the accumulators, and in the backward the upstream gradient g of the ACTIVATION
// enters both kernels as g * y * (1 - y) with y read next to g (`y_mul`): torch's sigmoid_backward, fused
// into the operand loads of dx and dW / db.
#include "common.h"
#include "glm_bf16.h"

#include <cstdlib>

namespace pa {

constexpr int TALL_MAX = 128;
typedef float f32x16t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void tall_split8(const float (&v)[8], bf16x8& c1, bf16x8& c2, bf16x8& c3) {
  uint32_t p1[4], p2[4], p3[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], p1[j], p2[j], p3[j]);
  c1 = as_bf16x8(p1[0], p1[1], p1[2], p1[3]);
  c2 = as_bf16x8(p2[0], p2[1], p2[2], p2[3]);
  c3 = as_bf16x8(p3[0], p3[1], p3[2], p3[3]);
}

constexpr int TALL_TA[6] = {2, 1, 0, 1, 0, 0};      // piece products, smallest first
constexpr int TALL_TB[6] = {0, 1, 2, 0, 1, 0};

// ---- Y = G Wm + bias ---------------------------------------------------------------------------
// LDS: Wm planes [k-step][column tile][plane] blocks of 64 lanes x 16 B (lane = column (l & 31) of the
// tile, k group l >> 5: Wm[16 ks + 8 kg .. + 7][column]); NKS k-steps x NCT column tiles.
// NW waves per workgroup share the planes.  NW = 4 (round 3): one wave per SIMD, the next tile's loads travel
// under this tile's MFMAs -- nothing else hides a latency, and the phases of a tile add up (50 us for 8 us of
// matrix work at 100 -> 100 over 1e5 rows).  NW = 16 (round 6): four waves per SIMD, ONE tile each in a single
// round of the grid; a wave keeps three k-steps of loads in flight and the other three waves of its SIMD fill
// its waits (128 registers per wave: no second tile in registers).
template <int NKS, int NCT, int NW>
__global__ __launch_bounds__(64 * NW) void tall_linear_kernel(const float* __restrict__ G, int64_t B, int R,
                                                          const float* __restrict__ W, int64_t w_rs,
                                                          int64_t w_cs, int C,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ Ymul, int act,
                                                          float* __restrict__ Y, int stage) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tall_smem[];
  uint4* wsm = reinterpret_cast<uint4*>(tall_smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kg = lane >> 5;
  // the planes, once per workgroup: all of a wave's loads are requested before the first split
  if (stage) {                                               // (workgroup-uniform)
    // a contiguous weight (either orientation) travels ONCE, in 16-byte granules in memory order, into a staging
    // area behind the planes; the lanes then pick their operand-order values out of LDS.  (Lane = column of Wm:
    // for the forward's Wm = weight^T every lane of a load sits on its own cache line -- 9 us of the launch.)
    float* stg = reinterpret_cast<float*>(tall_smem + NKS * NCT * 3 * 1024);
    const int n4 = (R * C) >> 2;
    for (int i = (int)threadIdx.x; i < n4; i += 64 * NW)
      reinterpret_cast<float4*>(stg)[i] = reinterpret_cast<const float4*>(W)[i];
    __syncthreads();
    const int rs = (int)w_rs, cs = (int)w_cs;
    for (int blk = wave; blk < NKS * NCT; blk += NW) {
      const int ks = blk / NCT, ct = blk % NCT;
      const int c = 32 * ct + l31;
      float wv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 16 * ks + 8 * kg + q;
        const float x = stg[(c < C ? c : C - 1) * cs + (r < R ? r : R - 1) * rs];
        wv[q] = (r < R && c < C) ? x : 0.0f;
      }
      bf16x8 c1, c2, c3;
      tall_split8(wv, c1, c2, c3);
      wsm[(blk * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, c1);
      wsm[(blk * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, c2);
      wsm[(blk * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, c3);
    }
  } else {
    constexpr int NBLK = (NKS * NCT + NW - 1) / NW;
    float wv[NBLK][8];
    // (the forward's Wm = weight^T: a lane's 8 values are consecutive in memory -- two 16-byte loads instead of
    //  eight gathers that touch 64 cache lines each: 224 of those per workgroup were ~7 us of the CU's texture path)
    const bool wvec = w_rs == 1 && (w_cs & 3) == 0 && (R & 3) == 0 && ((uintptr_t)W & 15) == 0;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
      const int blk = wave + NW * i;
      const int ks = blk / NCT, ct = blk % NCT;
      const int c = 32 * ct + l31;
      const float* wp = W + (int64_t)(c < C ? c : C - 1) * w_cs;
      if (wvec) {                                            // (wave-uniform)
        const int r0 = 16 * ks + 8 * kg;                     // (R % 4 == 0: a granule is inside or outside)
        const bool in = blk < NKS * NCT && c < C;
        const bool ok0 = in && r0 + 4 <= R, ok1 = in && r0 + 8 <= R;
        const float4 a = *reinterpret_cast<const float4*>(wp + (r0 + 4 <= R ? r0 : 0));
        const float4 b = *reinterpret_cast<const float4*>(wp + (r0 + 8 <= R ? r0 + 4 : 0));
        wv[i][0] = ok0 ? a.x : 0.0f; wv[i][1] = ok0 ? a.y : 0.0f; wv[i][2] = ok0 ? a.z : 0.0f; wv[i][3] = ok0 ? a.w : 0.0f;
        wv[i][4] = ok1 ? b.x : 0.0f; wv[i][5] = ok1 ? b.y : 0.0f; wv[i][6] = ok1 ? b.z : 0.0f; wv[i][7] = ok1 ? b.w : 0.0f;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 16 * ks + 8 * kg + q;
          const float x = wp[(int64_t)(r < R ? r : R - 1) * w_rs];
          wv[i][q] = (blk < NKS * NCT && r < R && c < C) ? x : 0.0f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
      const int blk = wave + NW * i;
      if (blk >= NKS * NCT) break;
      bf16x8 c1, c2, c3;
      tall_split8(wv[i], c1, c2, c3);
      wsm[(blk * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, c1);
      wsm[(blk * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, c2);
      wsm[(blk * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, c3);
    }
  }
  __syncthreads();
  const int64_t ntiles = (B + 31) / 32;
  const bool vec4 = (R & 3) == 0;                       // 16-byte row loads
  // this lane's 8 floats of k-step ks of tile t (rows past the end: zeros)
  // NW > 4: buffer loads -- ONE 32-bit offset per lane and tile, the k-step in the instruction's immediate, rows
  // past the end of the batch answered with zeros by the bounds check of the descriptor.  (With 64-bit clamped
  // addresses per granule the compiler keeps 4 NKS pointers live across the tile loop: 58 spilled registers at 128.)
  const int buf_bytes = NW > 4 ? (int)(B * R * 4) : 0;
  __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, buf_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t y_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ymul != nullptr ? Ymul : G), 0, buf_bytes, 0x00020000);
  auto load_step = [&](int64_t t, int ks, float (&v)[8]) {
    auto dsig = [](float g, float y) { return g * (1.0f - y) * y; };      // (sigmoid_backward's expression)
    const int r0 = 16 * ks + 8 * kg;
    if constexpr (NW > 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const int row_off = ((int)(t * 32 + l31) * R + 8 * kg) * 4;         // (B R < 2^29: the launcher's condition)
      const bool ok0 = r0 + 4 <= R, ok1 = r0 + 8 <= R;                     // (R % 4 == 0: a granule is in or out)
      f4 a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, row_off + 64 * ks, 0, 0));
      f4 b = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, row_off + 64 * ks + 16, 0, 0));
      if (Ymul != nullptr) {                                 // (wave-uniform)
        const f4 ya = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, row_off + 64 * ks, 0, 0));
        const f4 yb = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(y_rsrc, row_off + 64 * ks + 16, 0, 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = dsig(a[q], ya[q]); b[q] = dsig(b[q], yb[q]); }
      }
      // (a granule past the end of the ROW belongs to the next row: zeroed here; past the end of the batch
      //  the loads returned zeros)
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] = ok0 ? a[q] : 0.0f; v[4 + q] = ok1 ? b[q] : 0.0f; }
      return;
    }
    // Every load is unconditional, from a clamped address, and a select zeroes what is out of range:
    // a predicated load compiles to a branch with its own s_waitcnt, and a tile has up to 56 of them
    // (measured: 110 waits per tile, 62 us for 11 us of MFMA work).
    const int64_t row = t * 32 + l31;
    const bool rok = row < B;
    const float* gr = G + (rok ? row : B - 1) * R;
    const float* yr = Ymul != nullptr ? Ymul + (rok ? row : B - 1) * R : nullptr;
    if (vec4) {                                              // (wave-uniform) R % 4 == 0: two 16-byte granules
      const bool ok0 = r0 + 4 <= R, ok1 = r0 + 8 <= R;
      float4 a = *reinterpret_cast<const float4*>(gr + (ok0 ? r0 : 0));
      float4 b = *reinterpret_cast<const float4*>(gr + (ok1 ? r0 + 4 : 0));
      if (yr != nullptr) {                                   // (wave-uniform)
        const float4 ya = *reinterpret_cast<const float4*>(yr + (ok0 ? r0 : 0));
        const float4 yb = *reinterpret_cast<const float4*>(yr + (ok1 ? r0 + 4 : 0));
        a.x = dsig(a.x, ya.x); a.y = dsig(a.y, ya.y); a.z = dsig(a.z, ya.z); a.w = dsig(a.w, ya.w);
        b.x = dsig(b.x, yb.x); b.y = dsig(b.y, yb.y); b.z = dsig(b.z, yb.z); b.w = dsig(b.w, yb.w);
      }
      const bool k0 = rok && ok0, k1 = rok && ok1;
      v[0] = k0 ? a.x : 0.0f; v[1] = k0 ? a.y : 0.0f; v[2] = k0 ? a.z : 0.0f; v[3] = k0 ? a.w : 0.0f;
      v[4] = k1 ? b.x : 0.0f; v[5] = k1 ? b.y : 0.0f; v[6] = k1 ? b.z : 0.0f; v[7] = k1 ? b.w : 0.0f;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = r0 + q < R ? r0 + q : 0;
        float x = gr[c];
        if (yr != nullptr) x = dsig(x, yr[c]);
        v[q] = (rok && r0 + q < R) ? x : 0.0f;
      }
    }
  };
  auto load_tile = [&](int64_t t, float (&v)[NKS][8]) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) load_step(t, ks, v[ks]);
  };
  const int64_t t_step = (int64_t)gridDim.x * NW;
  constexpr bool AHEAD = NW <= 4;                            // the next tile in registers (one wave per SIMD)
  constexpr int WIN = 2;                                     // otherwise: k-steps of loads in flight
  float vn[AHEAD ? NKS : 1][8];
  if constexpr (AHEAD) {
    const int64_t t0 = (int64_t)blockIdx.x * NW + wave;
    if (t0 < ntiles) load_tile(t0, vn);
  }
  for (int64_t t = (int64_t)blockIdx.x * NW + wave; t < ntiles; t += t_step) {
    float v[NKS][8];
    if constexpr (AHEAD) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[ks][q] = vn[ks][q];
      if (t + t_step < ntiles) load_tile(t + t_step, vn);    // the next tile travels under this one's MFMAs
    } else {
#pragma unroll
      for (int ks = 0; ks < (WIN < NKS ? WIN : NKS); ++ks) load_step(t, ks, v[ks]);
    }
    f32x16t acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if constexpr (!AHEAD) {
        if (ks + WIN < NKS) load_step(t, ks + WIN, v[ks + WIN]);
      }
      bf16x8 a[3];
      tall_split8(v[ks], a[0], a[1], a[2]);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        bf16x8 b[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          b[pl] = __builtin_bit_cast(bf16x8, wsm[((ks * NCT + ct) * 3 + pl) * 64 + lane]);
#pragma unroll
        for (int p = 0; p < 6; ++p)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TALL_TA[p]], b[TALL_TB[p]], acc[ct], 0, 0, 0);
      }
      // (keeps the scheduler from hoisting the B-operand reads of every later k-step above this
      // one's MFMAs: 84 ds_read_b128 in flight = 336 registers, measured as spills)
      __builtin_amdgcn_sched_barrier(0);
    }
    // C/D layout: lane = column, register r = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool full = t * 32 + 32 <= B;                      // (wave-uniform) no row test per store
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int c = 32 * ct + l31;
      if (c >= C) continue;
      const float bc = bias != nullptr ? bias[c] : 0.0f;
      float* yp = Y + (t * 32 + 4 * kg) * C + c;
      // (act: sigmoid on the hardware exp2 / rcp -- 1 ulp each; the library expf and the IEEE division cost a
      // wave that has nothing to overlap them with 19 us over the 1e5 x 100 layer, as much as the separate
      // operator they replace)
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float h = acc[ct][r] + bc;
          yp[((r & 3) + 8 * (r >> 2)) * C] = act ? fast_sigmoid(h) : h;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t orow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
          const float h = acc[ct][r] + bc;
          if (orow < B) yp[((r & 3) + 8 * (r >> 2)) * C] = act ? fast_sigmoid(h) : h;
        }
      }
    }
  }
}

// ---- dW = G^T X, db = sum G --------------------------------------------------------------------
// Wave `gw` (global index) owns row tile gw % nrt of dW and every (nwaves / nrt)-th k-step of 16 batch
// rows; partial[gw][32][128] and partial_db[gw][32].
// NW = 8 (round 6): two waves per SIMD -- one wave's loads / splits under the other's MFMAs -- and the waves w,
// w + 4 of a workgroup (same row tile) add their tiles through LDS, so the number of partial tiles stays 4 per CU.
template <int NCT, int NW>
__global__ __launch_bounds__(64 * NW) void tall_wgrad_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ X,
                                                         const float* __restrict__ Ymul, int64_t B, int R,
                                                         int K, int nrt, float* __restrict__ partial,
                                                         float* __restrict__ partial_db) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, kg = lane >> 5;
  const int64_t gw = (int64_t)blockIdx.x * NW + wave, nwaves = (int64_t)gridDim.x * NW;
  const int rt = (int)(gw % nrt);
  const int64_t share = gw / nrt, nshares = nwaves / nrt;            // (nwaves is a multiple of nrt)
  const int64_t nks = (B + 15) / 16;
  const int r = 32 * rt + l31;
  const bool r_ok = r < R;
  f32x16t acc[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ct][i] = 0.0f;
  float dbs = 0.0f;
  // The 8 + 8 NCT strided loads of a k-step feed 6 NCT MFMAs: requested in the iteration that uses
  // them the loop runs at the memory latency.  Three register sets, two k-steps ahead.
  struct Raw { float ga[8], xv[NCT][8]; };
  // Loads are unconditional from clamped columns (a predicated load is a branch with its own wait:
  // 40 per k-step); only a k-step that crosses the end of the batch takes the tested path.
  const int rc = r_ok ? r : R - 1;
  auto load = [&](int64_t ks, Raw& o) {
    const int64_t b0 = ks * 16 + 8 * kg;
    if (ks * 16 + 16 <= B) {                              // (wave-uniform)
      const float* gp = G + b0 * R + rc;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float x = gp[(int64_t)q * R];
        o.ga[q] = r_ok ? x : 0.0f;
      }
      if (Ymul != nullptr) {                              // (wave-uniform) g * ((1 - y) * y)
        const float* yp = Ymul + b0 * R + rc;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float y = yp[(int64_t)q * R];
          o.ga[q] = o.ga[q] * (1.0f - y) * y;
        }
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int k = 32 * ct + l31;
        const float* xp = X + b0 * K + (k < K ? k : K - 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float x = xp[(int64_t)q * K];
          o.xv[ct][q] = k < K ? x : 0.0f;
        }
      }
    } else {                                              // the last k-step (or one past the end: all zero)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        o.ga[q] = (r_ok && b0 + q < B) ? G[(b0 + q) * R + r] : 0.0f;
        if (Ymul != nullptr && r_ok && b0 + q < B) {
          const float y = Ymul[(b0 + q) * R + r];
          o.ga[q] = o.ga[q] * (1.0f - y) * y;
        }
      }
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int k = 32 * ct + l31;
#pragma unroll
        for (int q = 0; q < 8; ++q) o.xv[ct][q] = (k < K && b0 + q < B) ? X[(b0 + q) * K + k] : 0.0f;
      }
    }
  };
  auto mma = [&](const Raw& o) {
#pragma unroll
    for (int q = 0; q < 8; ++q) dbs += o.ga[q];
    bf16x8 a[3];
    tall_split8(o.ga, a[0], a[1], a[2]);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      bf16x8 b[3];
      tall_split8(o.xv[ct], b[0], b[1], b[2]);
#pragma unroll
      for (int p = 0; p < 6; ++p)
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TALL_TA[p]], b[TALL_TB[p]], acc[ct], 0, 0, 0);
    }
  };
  if (share < nks) {
    Raw o0, o1, o2;
    load(share, o0);
    load(share + nshares, o1);
    int64_t ks = share;
    for (; ks + 2 * nshares < nks; ks += 3 * nshares) {
      load(ks + 2 * nshares, o2);
      mma(o0);
      load(ks + 3 * nshares, o0);
      mma(o1);
      load(ks + 4 * nshares, o1);
      mma(o2);
    }
    if (ks < nks) mma(o0);
    if (ks + nshares < nks) mma(o1);
  }
  dbs += __shfl_xor(dbs, 32);
  if constexpr (NW == 8) {
    // waves 4..7 hand their tile to waves 0..3 (same row tile: 4 % nrt == 0), which add it to their own
    extern __shared__ __attribute__((aligned(16))) unsigned char tallw_smem[];
    float* xch = reinterpret_cast<float*>(tallw_smem) + (wave & 3) * (NCT * 16 + 1) * 64;
    if (wave >= 4) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) xch[(ct * 16 + i) * 64 + lane] = acc[ct][i];
      xch[NCT * 16 * 64 + lane] = dbs;
    }
    __syncthreads();
    if (wave >= 4) return;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ct][i] += xch[(ct * 16 + i) * 64 + lane];
    dbs += xch[NCT * 16 * 64 + lane];
  }
  const int64_t gp = (int64_t)blockIdx.x * 4 + wave;        // (gp % nrt == rt)
  float* dst = partial + gp * (32 * TALL_MAX);
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * kg;
      dst[m * TALL_MAX + 32 * ct + l31] = acc[ct][i];
    }
  if (kg == 0) partial_db[gp * 32 + l31] = dbs;
}

// dW[r][k] = the sum over the waves that own r's tile: JW outputs x 256 / JW groups per workgroup, thread (j, g)
// sums the parts g, g + NG, ... in fp64 and the NG group sums are added in order (fixed order, whole
// chip); the outputs past R K are db
template <int JW>
__global__ __launch_bounds__(256) void tall_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                const float* __restrict__ partial_db,
                                                                int64_t nwaves, int nrt, int R, int K,
                                                                float* __restrict__ dW,
                                                                float* __restrict__ db) {
  // 64 consecutive outputs per workgroup (a wave reads 256 contiguous bytes of every partial tile), four groups
  // of partial tiles side by side, eight loads in flight per thread: until round 6 eight outputs x 32 groups with
  // one load per iteration -- a chain of L2 round trips over 32-byte pieces (9-12 us for 16 MB of partial tiles)
  // (JW = 16 for few outputs -- an 8 x 100 gradient is 13 workgroups of 64: 16 x 16 keeps the chip busier)
  constexpr int NG = 256 / JW;
  __shared__ double sm[NG][JW];
  const int jj = threadIdx.x % JW, g = threadIdx.x / JW;
  const int idx = blockIdx.x * JW + jj;
  const int nout = R * K + (db != nullptr ? R : 0);
  const int64_t nparts = nwaves / nrt;
  double acc = 0.0;
  const float* src = nullptr;
  int64_t stride = 0;
  if (idx < R * K) {
    const int r = idx / K, k = idx - r * K;
    const int rt = r >> 5, m = r & 31;
    src = partial + (int64_t)rt * (32 * TALL_MAX) + m * TALL_MAX + k;
    stride = (int64_t)nrt * (32 * TALL_MAX);
  } else if (idx < nout) {
    const int r = idx - R * K;
    const int rt = r >> 5, m = r & 31;
    src = partial_db + (int64_t)rt * 32 + m;
    stride = (int64_t)nrt * 32;
  }
  if (src != nullptr) {
    int64_t p = g;
    for (; p + 7 * NG < nparts; p += 8 * NG) {
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = src[(p + NG * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)q[u];
    }
    for (; p < nparts; p += NG) acc += (double)src[p * stride];
  }
  sm[g][jj] = acc;
  __syncthreads();
  if (g == 0 && idx < nout) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < NG; ++q) t += sm[q][jj];
    if (idx < R * K) dW[idx] = (float)t;
    else db[idx - R * K] = (float)t;
  }
}

// waves per workgroup of tall_linear_kernel: 0 = by size; PA_TALL_WAVES=4 | 16 pins it (measurements)
static int tall_waves() {
  static const int v = [] {
    const char* e = getenv("PA_TALL_WAVES");
    const int n = e ? atoi(e) : 0;
    return (n == 4 || n == 16) ? n : 0;
  }();
  return v;
}

static int tall_wgrad_grid(int64_t B, int nw) {
  // one workgroup per CU (nw / 4 waves per SIMD; each keeps three k-steps in flight); never more waves than k-steps
  int64_t g = (int64_t)cu_count();
  const int64_t nks = (B + 15) / 16;
  if (g * nw > nks) g = (nks + nw - 1) / nw;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace pa

extern "C" {

int pa_tall_linear(const float* G, int64_t B, int64_t R, const float* W, int64_t w_row_stride,
                   int64_t w_col_stride, int64_t C, const float* bias, float* Y, pa_stream_t stream) {
  return pa_tall_linear_act(G, B, R, W, w_row_stride, w_col_stride, C, bias, nullptr, 0, Y, stream);
}

int pa_tall_linear_act(const float* G, int64_t B, int64_t R, const float* W, int64_t w_row_stride,
                       int64_t w_col_stride, int64_t C, const float* bias, const float* y_mul,
                       int sigmoid_out, float* Y, pa_stream_t stream) {
  PA_REQUIRE(B >= 0 && R >= 1 && R <= pa::TALL_MAX && C >= 1 && C <= pa::TALL_MAX,
             "tall_linear: needs 1 <= R, C <= 128 (B=%lld R=%lld C=%lld)", (long long)B, (long long)R,
             (long long)C);
  if (B == 0) return PA_OK;
  PA_REQUIRE(G && W && Y, "tall_linear: NULL pointer");
  const int nks = (int)((R + 15) / 16), nct = (int)((C + 31) / 32);
  const int64_t ntiles = (B + 31) / 32;
  // sixteen waves per workgroup (four per SIMD, one tile each) once there is a tile for every wave of a
  // quarter of the chip; below that the four-wave form (fewer, longer-lived waves: the planes' prologue is
  // per workgroup)
  int nw = pa::tall_waves() != 0 ? pa::tall_waves() : (ntiles >= (int64_t)pa::cu_count() * 4 ? 16 : 4);
  // (the sixteen-wave form reads rows in 16-byte granules through 32-bit buffer offsets)
  if ((R & 3) != 0 || B * R >= (1ll << 29) || ((uintptr_t)G & 15) != 0 || (y_mul && ((uintptr_t)y_mul & 15) != 0)) nw = 4;
  int64_t grid = (ntiles + nw - 1) / nw;
  const int64_t cap = (int64_t)pa::cu_count() * (nw == 16 || nks * nct > 16 ? 1 : 2);
  if (grid > cap) grid = cap;
  // a contiguous weight is staged through LDS in memory order (sixteen-wave form, when it fits behind the planes)
  const bool contiguous = (w_row_stride == 1 && w_col_stride == R) || (w_col_stride == 1 && w_row_stride == C);
  const size_t stage_bytes = ((size_t)R * C * 4 + 15) & ~(size_t)15;
  hipStream_t s = pa::as_stream(stream);
#define PA_TALL_LAUNCH(NKS_, NCT_, NW_)                                                                \
  {                                                                                                    \
    auto k = pa::tall_linear_kernel<NKS_, NCT_, NW_>;                                                  \
    size_t l = (size_t)NKS_ * NCT_ * 3 * 1024;                                                         \
    const int stage = NW_ == 16 && contiguous && ((R * C) & 3) == 0 && ((uintptr_t)W & 15) == 0 &&     \
                      l + stage_bytes <= (size_t)156 * 1024;                                           \
    if (stage) l += stage_bytes;                                                                       \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);     \
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64 * NW_), l, s, G, B, (int)R, W, w_row_stride,   \
                       w_col_stride, (int)C, bias, y_mul, sigmoid_out, Y, stage);                      \
    return pa::check_launch("tall_linear_kernel");                                                     \
  }
#define PA_TALL_CASE(NKS_, NCT_)                                                                       \
  if (nks <= NKS_ && nct <= NCT_) {                                                                    \
    if (nw == 16) PA_TALL_LAUNCH(NKS_, NCT_, 16)                                                       \
    PA_TALL_LAUNCH(NKS_, NCT_, 4)                                                                      \
  }
  PA_TALL_CASE(1, 4)
  PA_TALL_CASE(8, 1)
  PA_TALL_CASE(7, 4)
  PA_TALL_CASE(8, 4)
#undef PA_TALL_CASE
#undef PA_TALL_LAUNCH
  return pa::fail(PA_ERR_UNSUPPORTED, "tall_linear: no kernel for R=%lld C=%lld", (long long)R, (long long)C);
}

size_t pa_tall_wgrad_workspace(int64_t B, int64_t R, int64_t K) {
  if (B < 0 || R < 1 || R > pa::TALL_MAX || K < 1 || K > pa::TALL_MAX) return 0;
  const size_t nwaves = (size_t)pa::cu_count() * 2 * 4;
  return nwaves * (32 * pa::TALL_MAX + 32) * sizeof(float);
}

int pa_tall_wgrad(const float* G, const float* X, int64_t B, int64_t R, int64_t K, float* dW, float* db,
                  void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  return pa_tall_wgrad_act(G, X, nullptr, B, R, K, dW, db, workspace, workspace_bytes, stream);
}

int pa_tall_wgrad_act(const float* G, const float* X, const float* y_mul, int64_t B, int64_t R, int64_t K,
                      float* dW, float* db, void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(B >= 0 && R >= 1 && R <= pa::TALL_MAX && K >= 1 && K <= pa::TALL_MAX,
             "tall_wgrad: needs 1 <= R, K <= 128 (B=%lld R=%lld K=%lld)", (long long)B, (long long)R,
             (long long)K);
  PA_REQUIRE(dW != nullptr, "tall_wgrad: NULL output");
  hipStream_t s = pa::as_stream(stream);
  if (B == 0) {
    hipError_t e1 = hipMemsetAsync(dW, 0, (size_t)R * K * sizeof(float), s);
    hipError_t e2 = db ? hipMemsetAsync(db, 0, (size_t)R * sizeof(float), s) : hipSuccess;
    if (e1 != hipSuccess || e2 != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "tall_wgrad: memset failed");
    return PA_OK;
  }
  PA_REQUIRE(G && X && workspace, "tall_wgrad: NULL pointer");
  PA_REQUIRE(workspace_bytes >= pa_tall_wgrad_workspace(B, R, K), "tall_wgrad: workspace too small");
  int nrt = (int)((R + 31) / 32);
  if (nrt == 3) nrt = 4;                               // (a wave's tile index is gw % nrt, 4 waves per workgroup)
  // eight waves per workgroup (two per SIMD) once every wave has k-steps for its three register sets
  const int64_t nks16 = (B + 15) / 16;
  const int tw = pa::tall_waves();
  const int nw = tw == 4 ? 4 : tw == 16 ? 8 : (nks16 * nrt >= (int64_t)pa::cu_count() * 8 * 3 ? 8 : 4);
  const int grid = pa::tall_wgrad_grid(B, nw);
  const int64_t nwaves = (int64_t)grid * 4;            // partial tiles (the eight-wave form adds pairs in LDS)
  float* part = (float*)workspace;
  float* part_db = part + nwaves * (32 * pa::TALL_MAX);
  const int nct = (int)((K + 31) / 32);
#define PA_TALLW_CASE(NCT_)                                                                            \
  if (nct == NCT_) {                                                                                   \
    if (nw == 8) {                                                                                     \
      auto k = pa::tall_wgrad_kernel<NCT_, 8>;                                                         \
      const size_t l = (size_t)4 * (NCT_ * 16 + 1) * 64 * sizeof(float);                               \
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);   \
      hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), l, s, G, X, y_mul, B, (int)R, (int)K,     \
                         nrt, part, part_db);                                                          \
    } else {                                                                                           \
      hipLaunchKernelGGL((pa::tall_wgrad_kernel<NCT_, 4>), dim3((unsigned)grid), dim3(256), 0, s, G,   \
                         X, y_mul, B, (int)R, (int)K, nrt, part, part_db);                             \
    }                                                                                                  \
  }
  PA_TALLW_CASE(1)
  PA_TALLW_CASE(2)
  PA_TALLW_CASE(3)
  PA_TALLW_CASE(4)
#undef PA_TALLW_CASE
  const int64_t n = R * K + R;
  if (n >= 4096)
    hipLaunchKernelGGL(pa::tall_wgrad_reduce_kernel<64>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part,
                       part_db, nwaves, nrt, (int)R, (int)K, dW, db);
  else
    hipLaunchKernelGGL(pa::tall_wgrad_reduce_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, part,
                       part_db, nwaves, nrt, (int)R, (int)K, dW, db);
  return pa::check_launch("tall_wgrad");
}

}  // extern "C"
