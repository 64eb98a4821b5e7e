// The workgroup-level pieces of a small `_create_fcnn` network D -> H (-> H) -> 1 held in LDS (16 rows per workgroup, plain VALU loops): spectral-norm power iterations,
// forward, back-propagation of a per-row upstream gradient, the input-gradient pass dD/dx and the derivative of that pass (the gradient penalty's double backward, with
// the phi'' terms of tanh re-entering the forward graph; derivation in oracle/gail_deep.py). Shared by gail_deep.hip (the discriminator itself has this shape,
// reference models.py:152-162) and gail_shaped_deep.hip (the shaping potential h has it, models.py:157-160). Every function is called by all threads of the workgroup.
#pragma once
#include "il_common.hpp"

#define GD_R 16

struct GdLayout { int64_t oW[3], ob[3], P; int out[3], in[3]; int L; };   // layers 0 .. L-1 hidden, L = output (H -> 1); torch order: (b, W) with SN, (W, b) without
__host__ __device__ inline GdLayout gd_layout(int D, int H, int depth, int sn) {
  GdLayout l; l.L = depth; int64_t o = 0;
  for (int i = 0; i <= depth; ++i) {
    l.in[i] = i == 0 ? D : H; l.out[i] = i == depth ? 1 : H;
    const int64_t nw = (int64_t)l.out[i] * l.in[i];
    if (sn) { l.ob[i] = o; o += l.out[i]; l.oW[i] = o; o += nw; } else { l.oW[i] = o; o += nw; l.ob[i] = o; o += l.out[i]; }
  }
  for (int i = depth + 1; i < 3; ++i) { l.oW[i] = l.ob[i] = 0; l.out[i] = l.in[i] = 0; }
  l.P = o;
  return l;
}
// u / v buffers: per layer [u (out) | v (in)], layer order
__host__ __device__ inline int64_t gd_sn_numel(int D, int H, int depth) { int64_t n = 0; for (int i = 0; i <= depth; ++i) n += (i == depth ? 1 : H) + (i == 0 ? D : H); return n; }

struct GdLds {
  float *W[3], *b[3], *u[3], *v[3], *tmp, *sc;      // sc: [0..2] sigma per layer, [3] output bias
  float *X, *A[2], *Z[2], *U[2], *S1, *SB, *row, *red;
  int ldw[3], ldx, ldh;
};
__host__ __device__ inline size_t gd_lds_floats(int D, int H, int depth) {
  size_t n = 0;
  for (int i = 0; i <= depth; ++i) { const int out = i == depth ? 1 : H, in = i == 0 ? D : H; n += (size_t)out * (in + 1) + out + out + in; }
  n += (size_t)(D > H ? D : H) + 8;                                                        // tmp, sc
  n += (size_t)GD_R * (D + 1) + (size_t)(3 * depth + 1) * GD_R * (H + 1);                   // X; A, Z, U per hidden layer; S1
  n += (size_t)GD_R * ((D > H ? D : H) + 1) + 8 * GD_R + 64;                                // SB, row, red
  return n;
}
__device__ __forceinline__ GdLds gd_carve(float* p, int D, int H, int depth) {
  GdLds l; l.ldx = D + 1; l.ldh = H + 1;
  for (int i = 0; i < 3; ++i) {
    if (i <= depth) {
      const int out = i == depth ? 1 : H, in = i == 0 ? D : H;
      l.ldw[i] = in + 1; l.W[i] = p; p += out * (in + 1); l.b[i] = p; p += out; l.u[i] = p; p += out; l.v[i] = p; p += in;
    } else { l.W[i] = l.b[i] = l.u[i] = l.v[i] = nullptr; l.ldw[i] = 0; }
  }
  l.tmp = p; p += (D > H ? D : H); l.sc = p; p += 8;
  l.X = p; p += GD_R * (D + 1);
  for (int i = 0; i < 2; ++i) { l.A[i] = i < depth ? p : nullptr; if (i < depth) p += GD_R * (H + 1); }
  for (int i = 0; i < 2; ++i) { l.Z[i] = i < depth ? p : nullptr; if (i < depth) p += GD_R * (H + 1); }
  for (int i = 0; i < 2; ++i) { l.U[i] = i < depth ? p : nullptr; if (i < depth) p += GD_R * (H + 1); }
  l.S1 = p; p += GD_R * (H + 1);
  l.SB = p; p += GD_R * ((D > H ? D : H) + 1);
  l.row = p; p += 8 * GD_R; l.red = p;
  return l;
}

__device__ __forceinline__ float gd_phi(float z, int tanh_) { return tanh_ ? tanhf(z) : fmaxf(z, 0.f); }
__device__ __forceinline__ float gd_dphi(float a, int tanh_) { return tanh_ ? 1.f - a * a : (a > 0.f ? 1.f : 0.f); }
__device__ __forceinline__ float gd_d2phi(float a, int tanh_) { return tanh_ ? -2.f * a * (1.f - a * a) : 0.f; }

__device__ __forceinline__ float gd_dot(const float* a, const float* b, int n, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(a[i], b[i], s);
  return block_sum(s, red);
}
__device__ __forceinline__ void gd_normalize(float* v, int n, float* red) {   // torch F.normalize: v / max(||v||, 1e-12)
  const float inv = 1.f / fmaxf(sqrtf(gd_dot(v, v, n, red)), 1e-12f);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) v[i] *= inv;
  __syncthreads();
}
// y[j] = sum_k W[j][k] x[k]  (W in LDS [N][ld])
__device__ __forceinline__ void gd_mv(const float* W, int ld, int N, int K, const float* x, float* y) {
  for (int j = threadIdx.x; j < N; j += blockDim.x) { float s = 0.f; for (int k = 0; k < K; ++k) s = fmaf(W[j * ld + k], x[k], s); y[j] = s; }
  __syncthreads();
}
__device__ __forceinline__ void gd_mtv(const float* W, int ld, int N, int K, const float* u, float* y) {   // y[k] = sum_j W[j][k] u[j]
  for (int k = threadIdx.x; k < K; k += blockDim.x) { float s = 0.f; for (int j = 0; j < N; ++j) s = fmaf(W[j * ld + k], u[j], s); y[k] = s; }
  __syncthreads();
}

// parameters (layout `lay`, base `params`) and - with spectral norm - the u / v buffers (per layer [u (out) | v (in)], base `sn`; nullptr: none) into LDS
__device__ __forceinline__ void gd_stage_weights(const GdLds& l, const GdLayout& lay, const float* __restrict__ params) {
  for (int i = 0; i <= lay.L; ++i) {
    const int out = lay.out[i], in = lay.in[i];
    for (int e = threadIdx.x; e < out * in; e += blockDim.x) { const int n = e / in, k = e - n * in; l.W[i][n * l.ldw[i] + k] = params[lay.oW[i] + e]; }
    for (int e = threadIdx.x; e < out; e += blockDim.x) l.b[i][e] = params[lay.ob[i] + e];
  }
}
__device__ __forceinline__ void gd_stage(const GdLds& l, const GdLayout& lay, const float* __restrict__ params, const float* __restrict__ sn) {
  gd_stage_weights(l, lay, params);
  if (sn) {
    int64_t o = 0;
    for (int i = 0; i <= lay.L; ++i) {
      for (int e = threadIdx.x; e < lay.out[i]; e += blockDim.x) l.u[i][e] = sn[o + e];
      o += lay.out[i];
      for (int e = threadIdx.x; e < lay.in[i]; e += blockDim.x) l.v[i][e] = sn[o + e];
      o += lay.in[i];
    }
  }
  if (threadIdx.x == 0) { l.sc[0] = l.sc[1] = l.sc[2] = 1.f; }
  __syncthreads();
}
// `iters` power iterations of every layer (u = n(W v), v = n(W^T u)), then sigma_l = u . (W v) and W <- W / sigma in LDS (iters = 0: eval mode)
__device__ __forceinline__ void gd_spectral(const GdLds& l, const GdLayout& lay, int iters) {
  for (int i = 0; i <= lay.L; ++i) {
    const int N = lay.out[i], K = lay.in[i];
    for (int it = 0; it < iters; ++it) {
      gd_mv(l.W[i], l.ldw[i], N, K, l.v[i], l.u[i]);
      gd_normalize(l.u[i], N, l.red);
      gd_mtv(l.W[i], l.ldw[i], N, K, l.u[i], l.v[i]);
      gd_normalize(l.v[i], K, l.red);
    }
    gd_mv(l.W[i], l.ldw[i], N, K, l.v[i], l.tmp);
    const float s = gd_dot(l.u[i], l.tmp, N, l.red);
    __syncthreads();
    if (threadIdx.x == 0) l.sc[i] = s;
    const float inv = 1.f / s;
    for (int e = threadIdx.x; e < N * K; e += blockDim.x) { const int n = e / K, k = e - n * K; l.W[i][n * l.ldw[i] + k] *= inv; }
    __syncthreads();
  }
}
// A_l = phi(A_{l-1} W^_l^T + b_l) for the hidden layers, row[r] = logit
__device__ __forceinline__ void gd_forward(const GdLds& l, const GdLayout& lay, int H, int tanh_) {
  for (int i = 0; i < lay.L; ++i) {
    const int K = lay.in[i];
    const float* in = i == 0 ? l.X : l.A[i - 1]; const int ldi = i == 0 ? l.ldx : l.ldh;
    for (int e = threadIdx.x; e < GD_R * H; e += blockDim.x) {
      const int r = e / H, n = e - r * H;
      const float* w = l.W[i] + n * l.ldw[i]; const float* x = in + r * ldi;
      float s = 0.f;
      for (int k = 0; k < K; ++k) s = fmaf(x[k], w[k], s);
      l.A[i][r * l.ldh + n] = gd_phi(s + l.b[i][n], tanh_);
    }
    __syncthreads();
  }
  if (threadIdx.x < GD_R) {
    const float* a = l.A[lay.L - 1] + threadIdx.x * l.ldh;
    float s = 0.f;
    for (int j = 0; j < H; ++j) s = fmaf(a[j], l.W[lay.L][j], s);
    l.row[threadIdx.x] = s + l.b[lay.L][0];
  }
  __syncthreads();
}

// slab accumulation: the same thread owns the same elements in every pass (index e strided by the block), so `first` decides between = and +=
__device__ __forceinline__ void gd_acc(float* slab, int64_t o, float v, bool first) { if (first) slab[o] = v; else slab[o] += v; }
// GW_l[n][k] (+)= sum_r left[r][n] right[r][k];  gb_l[n] (+)= sum_r left[r][n]  (bias: only when with_bias)
__device__ __forceinline__ void gd_outer(float* slab, const GdLayout& lay, int i, const float* left, int ldl, const float* right, int ldr, bool first, bool with_bias) {
  const int N = lay.out[i], K = lay.in[i];
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int n = e / K, k = e - n * K;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < GD_R; ++r) s = fmaf(left[r * ldl + n], right[r * ldr + k], s);
    gd_acc(slab, lay.oW[i] + e, s, first);
  }
  if (with_bias)
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      float s = 0.f;
      for (int r = 0; r < GD_R; ++r) s += left[r * ldl + n];
      gd_acc(slab, lay.ob[i] + n, s, first);
    }
}

// The same as gd_spectral in separate steps, for a caller that needs the state between two uses of the weights (gail_shaped_deep.hip): power iterations only ...
__device__ __forceinline__ void gd_power(const GdLds& l, const GdLayout& lay, int iters) {
  for (int i = 0; i <= lay.L; ++i) {
    const int N = lay.out[i], K = lay.in[i];
    for (int it = 0; it < iters; ++it) {
      gd_mv(l.W[i], l.ldw[i], N, K, l.v[i], l.u[i]);
      gd_normalize(l.u[i], N, l.red);
      gd_mtv(l.W[i], l.ldw[i], N, K, l.u[i], l.v[i]);
      gd_normalize(l.v[i], K, l.red);
    }
  }
}
// ... sigma_l = u . (W v) of the UNSCALED weights into sc[l] ...
__device__ __forceinline__ void gd_sigma(const GdLds& l, const GdLayout& lay) {
  for (int i = 0; i <= lay.L; ++i) {
    const int N = lay.out[i], K = lay.in[i];
    gd_mv(l.W[i], l.ldw[i], N, K, l.v[i], l.tmp);
    const float s = gd_dot(l.u[i], l.tmp, N, l.red);
    __syncthreads();
    if (threadIdx.x == 0) l.sc[i] = s;
    __syncthreads();
  }
}
// ... and W <- W * (1 / sc[l]) (the arithmetic of gd_spectral)
__device__ __forceinline__ void gd_scale(const GdLds& l, const GdLayout& lay) {
  for (int i = 0; i <= lay.L; ++i) {
    const int N = lay.out[i], K = lay.in[i];
    const float inv = 1.f / l.sc[i];
    for (int e = threadIdx.x; e < N * K; e += blockDim.x) { const int n = e / K, k = e - n * K; l.W[i][n * l.ldw[i] + k] *= inv; }
  }
  __syncthreads();
}

// (DEPTH as a template parameter of everything below and of the kernels: with a run-time depth the per-layer tables of GdLds / GdLayout are indexed dynamically and
// live in scratch memory, 320 bytes per lane)

// Back-propagation of dL/dD = dzr[r] through the network whose activations gd_forward left in A: one slab of dL/dW^_l, dL/db_l (plain assignment: first pass of a slab)
template <int DEPTH>
__device__ __forceinline__ void gd_backprop(const GdLds& l, const GdLayout& lay, float* __restrict__ slab, const float* dzr, int H, int tanh_) {
  constexpr int depth = DEPTH;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const float* wo = l.W[depth];
  const float* AL = l.A[depth - 1];
  for (int j = tid; j < H; j += nthr) {   // output layer
    float s = 0.f;
    for (int r = 0; r < GD_R; ++r) s = fmaf(dzr[r], AL[r * l.ldh + j], s);
    slab[lay.oW[depth] + j] = s;
  }
  if (tid == 0) { float s = 0.f; for (int r = 0; r < GD_R; ++r) s += dzr[r]; slab[lay.ob[depth]] = s; }
  float* zb = l.Z[depth - 1];
  for (int e = tid; e < GD_R * H; e += nthr) { const int r = e / H, j = e - r * H; zb[r * l.ldh + j] = dzr[r] * wo[j] * gd_dphi(AL[r * l.ldh + j], tanh_); }
  __syncthreads();
#pragma unroll
  for (int i = depth - 1; i >= 0; --i) {
    gd_outer(slab, lay, i, l.Z[i], l.ldh, i == 0 ? l.X : l.A[i - 1], i == 0 ? l.ldx : l.ldh, true, true);
    if (i > 0) {
      for (int e = tid; e < GD_R * H; e += nthr) {
        const int r = e / H, k = e - r * H;
        float s = 0.f;
        for (int n = 0; n < H; ++n) s = fmaf(l.Z[i][r * l.ldh + n], l.W[i][n * l.ldw[i] + k], s);
        l.Z[i - 1][r * l.ldh + k] = s * gd_dphi(l.A[i - 1][r * l.ldh + k], tanh_);
      }
      __syncthreads();
    }
  }
}

// The input-gradient pass down to the first hidden layer: U[l] = dD/dz_{l+1} (and S1 = dD/da_1 for depth 2); the caller forms g = U[0] W^_0 = dD/dx
template <int DEPTH>
__device__ __forceinline__ void gd_input_grad_u(const GdLds& l, const GdLayout& lay, int H, int tanh_) {
  constexpr int depth = DEPTH;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const float* wo = l.W[depth];
  const float* AL = l.A[depth - 1];
  for (int e = tid; e < GD_R * H; e += nthr) { const int r = e / H, j = e - r * H; l.U[depth - 1][r * l.ldh + j] = gd_dphi(AL[r * l.ldh + j], tanh_) * wo[j]; }
  __syncthreads();
  if (depth == 2) {   // s_1 = u_1 W^_1 (= dD/da_1), u_0 = phi'(a_1) * s_1
    for (int e = tid; e < GD_R * H; e += nthr) {
      const int r = e / H, k = e - r * H;
      float s = 0.f;
      for (int n = 0; n < H; ++n) s = fmaf(l.U[1][r * l.ldh + n], l.W[1][n * l.ldw[1] + k], s);
      l.S1[r * l.ldh + k] = s;
      l.U[0][r * l.ldh + k] = gd_dphi(l.A[0][r * l.ldh + k], tanh_) * s;
    }
    __syncthreads();
  }
}

// Derivative of the input-gradient pass: given SB[r][k] = dL/dg (row stride ldx, D columns) after gd_input_grad_u, one slab of dL/dW^_l, dL/db_l of a loss that
// reaches the parameters through g = dD/dx only (oracle/gail_deep.py:_input_gradient_backward)
template <int DEPTH>
__device__ __forceinline__ void gd_input_grad_backward(const GdLds& l, const GdLayout& lay, float* __restrict__ slab, int D, int H, int tanh_) {
  constexpr int depth = DEPTH;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const float* wo = l.W[depth];
  // layer 0: GW_0 = u_0^T sbar_0 ; ubar_0 = sbar_0 W^_0^T
  gd_outer(slab, lay, 0, l.U[0], l.ldh, l.SB, l.ldx, true, false);
  for (int n = tid; n < H; n += nthr) slab[lay.ob[0] + n] = 0.f;
  float* ub = l.Z[0];
  for (int e = tid; e < GD_R * H; e += nthr) {
    const int r = e / H, n = e - r * H;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(l.SB[r * l.ldx + k], l.W[0][n * l.ldw[0] + k], s);
    ub[r * l.ldh + n] = s;
  }
  __syncthreads();
  float* z2top = nullptr;   // second-order term entering the forward graph at the top hidden layer
  if (depth == 1) {
    // u_0 = phi'(a_1) * w^_o :  GWo = sum_r ubar phi'(a_1) ;  zbar2_0 = ubar * w^_o * phi''(a_1)
    for (int j = tid; j < H; j += nthr) {
      float s = 0.f;
      for (int r = 0; r < GD_R; ++r) s = fmaf(ub[r * l.ldh + j], gd_dphi(l.A[0][r * l.ldh + j], tanh_), s);
      slab[lay.oW[1] + j] = s;
    }
    if (tid == 0) slab[lay.ob[1]] = 0.f;
    __syncthreads();
    for (int e = tid; e < GD_R * H; e += nthr) { const int r = e / H, j = e - r * H; ub[r * l.ldh + j] = ub[r * l.ldh + j] * wo[j] * gd_d2phi(l.A[0][r * l.ldh + j], tanh_); }
    z2top = ub;   // Z[0]
    __syncthreads();
  } else {
    // u_0 = phi'(a_1) * s_1 : zbar2_0 = ubar_0 * s_1 * phi''(a_1) (kept in Z[0]); sbar_1 = ubar_0 * phi'(a_1) (into SB, width H)
    for (int e = tid; e < GD_R * H; e += nthr) {
      const int r = e / H, k = e - r * H;
      const float u0 = ub[r * l.ldh + k], a1 = l.A[0][r * l.ldh + k];
      l.SB[r * l.ldh + k] = u0 * gd_dphi(a1, tanh_);
      ub[r * l.ldh + k] = u0 * l.S1[r * l.ldh + k] * gd_d2phi(a1, tanh_);
    }
    __syncthreads();
    // layer 1: GW_1 = u_1^T sbar_1 ; ubar_1 = sbar_1 W^_1^T
    gd_outer(slab, lay, 1, l.U[1], l.ldh, l.SB, l.ldh, true, false);
    for (int n = tid; n < H; n += nthr) slab[lay.ob[1] + n] = 0.f;
    float* ub1 = l.Z[1];
    for (int e = tid; e < GD_R * H; e += nthr) {
      const int r = e / H, n = e - r * H;
      float s = 0.f;
      for (int k = 0; k < H; ++k) s = fmaf(l.SB[r * l.ldh + k], l.W[1][n * l.ldw[1] + k], s);
      ub1[r * l.ldh + n] = s;
    }
    __syncthreads();
    for (int j = tid; j < H; j += nthr) {
      float s = 0.f;
      for (int r = 0; r < GD_R; ++r) s = fmaf(ub1[r * l.ldh + j], gd_dphi(l.A[1][r * l.ldh + j], tanh_), s);
      slab[lay.oW[2] + j] = s;
    }
    if (tid == 0) slab[lay.ob[2]] = 0.f;
    __syncthreads();
    for (int e = tid; e < GD_R * H; e += nthr) { const int r = e / H, j = e - r * H; ub1[r * l.ldh + j] = ub1[r * l.ldh + j] * wo[j] * gd_d2phi(l.A[1][r * l.ldh + j], tanh_); }
    z2top = ub1;  // Z[1]
    __syncthreads();
  }
  if (tanh_) {   // phi'' != 0: the second-order terms go back through the forward pass (ReLU: nothing, and no bias gradient)
#pragma unroll
    for (int i = depth - 1; i >= 0; --i) {
      const float* zb = i == depth - 1 ? z2top : l.Z[i];
      gd_outer(slab, lay, i, zb, l.ldh, i == 0 ? l.X : l.A[i - 1], i == 0 ? l.ldx : l.ldh, false, true);
      if (i > 0) {   // zbar_{i-1} = (zbar_i W^_i) phi'(a_i) + zbar2_{i-1}   (Z[i-1] holds zbar2_{i-1})
        for (int e = tid; e < GD_R * H; e += nthr) {
          const int r = e / H, k = e - r * H;
          float s = 0.f;
          for (int n = 0; n < H; ++n) s = fmaf(zb[r * l.ldh + n], l.W[i][n * l.ldw[i] + k], s);
          l.Z[i - 1][r * l.ldh + k] = s * gd_dphi(l.A[i - 1][r * l.ldh + k], tanh_) + l.Z[i - 1][r * l.ldh + k];
        }
        __syncthreads();
      }
    }
  }
}

// <G^_l, W_l> = sigma_l <G^_l, W^_l> of this tile's slab into ip_out[l] (the same threads re-read what they wrote; call after a __syncthreads)
template <int DEPTH>
__device__ __forceinline__ void gd_inner_products(const GdLds& l, const GdLayout& lay, const float* __restrict__ slab, float* __restrict__ ip_out) {
  const int tid = threadIdx.x, nthr = blockDim.x;
#pragma unroll
  for (int i = 0; i <= DEPTH; ++i) {
    const int N = lay.out[i], K = lay.in[i];
    float s = 0.f;
    for (int e = tid; e < N * K; e += nthr) { const int n = e / K, k = e - n * K; s = fmaf(slab[lay.oW[i] + e], l.W[i][n * l.ldw[i] + k], s); }
    s = block_sum(s, l.red);
    if (tid == 0) ip_out[i] = s * l.sc[i];
    __syncthreads();
  }
}
