// LDS-staged MLP tile primitives for one workgroup = one tile of IL_TILE_R (16) batch rows.
//
// Design (MI355X, wave64, fp32 parity => exact-fp32 MFMA 16x16x4):
//  * activations of the tile live in LDS ([16][ld] fp32, ld = K + 4 keeps 16-B alignment and spreads rows over banks);
//  * weights are streamed straight from L2 into MFMA B operands (each weight element is used exactly once per
//    workgroup, so staging it through LDS would only add traffic); rows of a torch [N][K] weight are read as
//    16-B lanes (64 contiguous bytes per 4-lane k-group);
//  * each wave owns 64 output columns (4 accumulator tiles) so one A operand (from LDS, ds_read_b128) feeds 16 MFMAs;
//  * the k-index of an MFMA step is permuted (lane group g covers k0+4g..k0+4g+3 over four steps) so that both
//    operands are single 16-byte loads. Summation order within a dot product changes, results stay exact-fp32 FMAs.
#pragma once
#include "il_common.hpp"

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// 4 consecutive floats p[k..k+3] with columns >= kvalid reading as 0. `vec` = row base and k are 16-B aligned.
__device__ __forceinline__ f32x4 load4_guard(const float* __restrict__ p, int k, int kvalid, bool vec) {
  if (vec && k + 3 < kvalid) return *reinterpret_cast<const f32x4*>(p + k);
  f32x4 r;
  r[0] = (k + 0 < kvalid) ? p[k + 0] : 0.f;
  r[1] = (k + 1 < kvalid) ? p[k + 1] : 0.f;
  r[2] = (k + 2 < kvalid) ? p[k + 2] : 0.f;
  r[3] = (k + 3 < kvalid) ? p[k + 3] : 0.f;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Y[16 x N] = Xs[16 x Kpad] . W^T      W: global row-major [N][ldw], columns >= Kw read as zero (Xs is zero-padded too)
// N % 64 == 0.  epi(c0, acc): acc[t][reg] = Y[row 4g+reg][col c0 + 16t + j],  j = lane&15, g = lane>>4.
// ---------------------------------------------------------------------------------------------
template <class Epi>
__device__ __forceinline__ void tile_fwd(const float* Xs, int ldx, int Kpad, const float* __restrict__ W, int ldw, int Kw, int N, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const bool vec = ((ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  for (int c0 = wave * 64; c0 < N; c0 += nw * 64) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float* wr0 = W + (size_t)(c0 + j) * ldw;
    const float* wr1 = wr0 + (size_t)16 * ldw;
    const float* wr2 = wr1 + (size_t)16 * ldw;
    const float* wr3 = wr2 + (size_t)16 * ldw;
    const float* xr = Xs + j * ldx + 4 * g;
    f32x4 b0 = load4_guard(wr0, 4 * g, Kw, vec), b1 = load4_guard(wr1, 4 * g, Kw, vec), b2 = load4_guard(wr2, 4 * g, Kw, vec),
          b3 = load4_guard(wr3, 4 * g, Kw, vec);
    for (int k0 = 0; k0 < Kpad; k0 += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0);
      const f32x4 c0v = b0, c1v = b1, c2v = b2, c3v = b3;
      const int kn = k0 + 16 + 4 * g;  // prefetch next k-block while the MFMAs below run
      if (k0 + 16 < Kpad) {
        b0 = load4_guard(wr0, kn, Kw, vec); b1 = load4_guard(wr1, kn, Kw, vec); b2 = load4_guard(wr2, kn, Kw, vec); b3 = load4_guard(wr3, kn, Kw, vec);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = mfma16(a[s], c0v[s], acc[0]);
        acc[1] = mfma16(a[s], c1v[s], acc[1]);
        acc[2] = mfma16(a[s], c2v[s], acc[2]);
        acc[3] = mfma16(a[s], c3v[s], acc[3]);
      }
    }
    epi(c0, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// dX[16 x K] = dYs[16 x Npad] . W      W: global row-major [Nvalid][ldw], K % 64 == 0, ldw % 4 == 0, rows >= Nvalid read as zero.
// epi(kb, acc): acc[i][reg] = dX[row 4g+reg][col kb + 4j + i]   (16 B per lane per row => 256 contiguous bytes per 16 lanes)
// ---------------------------------------------------------------------------------------------
template <class Epi>
__device__ __forceinline__ void tile_bwd_dx(const float* dYs, int ldy, int Npad, int Nvalid, const float* __restrict__ W, int ldw, int K, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  for (int kb = wave * 64; kb < K; kb += nw * 64) {
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float* wp = W + (size_t)(4 * g) * ldw + kb + 4 * j;
    const float* yr = dYs + j * ldy + 4 * g;
    f32x4 b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = (4 * g + s < Nvalid) ? *reinterpret_cast<const f32x4*>(wp + (size_t)s * ldw) : zero4();
    for (int n0 = 0; n0 < Npad; n0 += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(yr + n0);
      f32x4 c[4] = {b[0], b[1], b[2], b[3]};
      if (n0 + 16 < Npad) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          b[s] = (n0 + 16 + 4 * g + s < Nvalid) ? *reinterpret_cast<const f32x4*>(wp + (size_t)(n0 + 16 + s) * ldw) : zero4();
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = mfma16(a[s], c[s][0], acc[0]);
        acc[1] = mfma16(a[s], c[s][1], acc[1]);
        acc[2] = mfma16(a[s], c[s][2], acc[2]);
        acc[3] = mfma16(a[s], c[s][3], acc[3]);
      }
    }
    epi(kb, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Small output layer: Os[16][16] = Xs[16 x K] . W^T + b, W [N][ldw] with N <= 16 (actor head 2A, critic head 1).
// K (multiple of 64) is split across the waves; partial tiles are reduced through LDS (`part` >= nw*256 floats).
// Contains __syncthreads(); every thread of the block must call. Result valid after return.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_fwd_small(const float* Xs, int ldx, int K, const float* __restrict__ W, int ldw, int N, const float* __restrict__ bias,
                                               float* Os, float* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int kchunk = K / nw;  // K % (16*nw) == 0 for H % 64 == 0, nw = 4
  f32x4 acc = zero4();
  const float* wr = W + (size_t)j * ldw;
  for (int k0 = wave * kchunk; k0 < (wave + 1) * kchunk; k0 += 16) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(Xs + j * ldx + k0 + 4 * g);
    const f32x4 b = (j < N) ? *reinterpret_cast<const f32x4*>(wr + k0 + 4 * g) : zero4();
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a[s], b[s], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave * 256 + (4 * g + r) * 16 + j] = acc[r];
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += part[w * 256 + i];
    const int col = i & 15;
    Os[i] = (col < N) ? s + bias[col] : 0.f;
  }
  __syncthreads();
}

// Load a [16 x (K1+K2)] tile of concatenated fields (e.g. cat(state, action)) into LDS, zero-padded to Kpad columns.
__device__ __forceinline__ void load_rows_cat(float* Xs, int ldx, int Kpad, const float* __restrict__ f1, int ld1, int K1, const float* __restrict__ f2,
                                              int ld2, int K2, int row0, int nrows_valid) {
  for (int i = threadIdx.x; i < IL_TILE_R * Kpad; i += blockDim.x) {
    const int r = i / Kpad, k = i - r * Kpad;
    float v = 0.f;
    if (r < nrows_valid) {
      if (k < K1) v = f1[(size_t)(row0 + r) * ld1 + k];
      else if (k < K1 + K2) v = f2[(size_t)(row0 + r) * ld2 + (k - K1)];
    }
    Xs[r * ldx + k] = v;
  }
}

struct MlpView {  // flat torch-order parameter vector of a depth-2 MLP
  const float *W1, *b1, *W2, *b2, *W3, *b3;
};
__host__ __device__ inline int64_t mlp_numel(int in, int H, int out) { return (int64_t)H * in + H + (int64_t)H * H + H + (int64_t)out * H + out; }
__device__ __forceinline__ MlpView mlp_view(const float* p, int in, int H, int out) {
  MlpView v;
  v.W1 = p; v.b1 = v.W1 + (size_t)H * in; v.W2 = v.b1 + H; v.b2 = v.W2 + (size_t)H * H; v.W3 = v.b2 + H; v.b3 = v.W3 + (size_t)out * H;
  return v;
}
__host__ __device__ inline int round_up16(int x) { return (x + 15) & ~15; }
