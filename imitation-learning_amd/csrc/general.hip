// General actor / critic shapes (reference models.py:48-69 `_create_fcnn`: any depth, relu / tanh / sigmoid) for gfx950: sac_update (training.py:14-54), acting and
// log-probabilities (models.py:90-102) and behavioural cloning (training.py:57-64) for the networks the fused kernels of sac.hip do not cover (they are built for the
// shape every shipped configuration uses: depth 2, ReLU, hidden <= 256, 2A <= 16).
//
// Layer-at-a-time composition instead of one fused launch per phase: activations live in HBM FEATURE-MAJOR ([feature][Bp], Bp = batch rounded up to 16, padding rows zero),
// so that a 16-row tile of a layer's input is 64 contiguous bytes per feature, a lane's four rows of an output column are one 16-byte store, and the weight gradient
// dW = dZ^T X reads both operands as 16-byte lanes along the batch. Every product is an exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) tile:
//   k_g_linear   Y^T = act(W X + b)        one workgroup = 16 rows x 64 output features, the input tile staged in LDS, one wave per 16 x 16 tile
//   k_g_bwd      dZ_prev^T = (W^T dZ) * act'(H_prev)   same shape, dZ tile staged in LDS
//   k_g_dw       G_W = dZ^T X              one wave per 16 x 16 tile of the gradient, reduction over the batch; the bias sums (one wave per feature) ride in the same launch
// and the per-row pieces (tanh-Gaussian head, TD target, loss seeds, head backward, temperature step) are one thread per row. The optimiser steps and the target update
// are the library's elementwise kernels (il_adam_step, il_polyak). Same parameter layout as torch (`parameters()` order, twin critics at il_mlp_stride_general).
// Padding rows carry zeros in every dZ, so they contribute nothing to any gradient.
#include "il_common.hpp"
#include "mlp_tile.hpp"
#include "dw_block.hpp"

enum { G_ACT_NONE = -1, G_ACT_RELU = 0, G_ACT_TANH = 1, G_ACT_SIGMOID = 2 };

__device__ __forceinline__ float g_act(float z, int a) {
  if (a == G_ACT_RELU) return fmaxf(z, 0.f);
  if (a == G_ACT_TANH) return tanhf(z);
  if (a == G_ACT_SIGMOID) return 1.f / (1.f + expf(-z));
  return z;
}
__device__ __forceinline__ float g_act_grad(float h, int a) {   // d act / d z from the POST-activation value
  if (a == G_ACT_RELU) return h > 0.f ? 1.f : 0.f;
  if (a == G_ACT_TANH) return 1.f - h * h;
  if (a == G_ACT_SIGMOID) return h * (1.f - h);
  return 1.f;
}

struct GLayer { int64_t oW, ob; int K, N; };
__host__ __device__ static inline GLayer g_layer(int in, int H, int depth, int out, int l) {   // layer l of [0, depth]: W [N][K] at oW, b [N] at ob (torch parameters() order)
  GLayer r; int64_t o = 0; int K = in;
  for (int i = 0; i < l; ++i) { o += (int64_t)H * K + H; K = H; }
  r.K = K; r.N = (l == depth) ? out : H; r.oW = o; r.ob = o + (int64_t)r.N * K;
  return r;
}
__host__ __device__ static inline int64_t g_numel(int in, int H, int depth, int out) { const GLayer l = g_layer(in, H, depth, out, depth); return l.ob + out; }
__host__ __device__ static inline int64_t g_stride(int in, int H, int depth, int out) { return (g_numel(in, H, depth, out) + 3) & ~(int64_t)3; }
static inline int g_bp(int n) { return (n + 15) & ~15; }

// ---- 16 x 16 MFMA tiles (lane (j, g) = (lane & 15, lane >> 4) holds C(4g + reg, j)) ------------------------------------------------------------------------------
// Xs [16][ldx] (LDS, zero beyond K) . W[n0 .. n0 + 15][0 .. K)^T ; rows n >= N clamp their address (the caller drops those columns)
__device__ __forceinline__ f32x4 g_tile_fwd(const float* Xs, int ldx, int Kpad, const float* __restrict__ W, int K, int n0, int N) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const float* wr = W + (size_t)min(n0 + j, N - 1) * K;
  const float* xr = Xs + j * ldx + 4 * g;
  f32x4 acc0 = zero4(), acc1 = zero4();

  for (int k0 = 0; k0 < Kpad; k0 += 16) {
    float b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = gload(wr + min(k0 + 4 * g + s, K - 1));   // (k >= K: Xs is zero there)
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0);
    acc0 = mfma16(a[0], b[0], acc0); acc1 = mfma16(a[1], b[1], acc1); acc0 = mfma16(a[2], b[2], acc0); acc1 = mfma16(a[3], b[3], acc1);
  }
  return acc0 + acc1;
}
// dYs [16][ldy] (LDS, zero beyond N) . W[0 .. N)[k0 .. k0 + 15] ; columns k >= K clamp their address
__device__ __forceinline__ f32x4 g_tile_bwd(const float* dYs, int ldy, int Npad, const float* __restrict__ W, int K, int N, int k0) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const float* wp = W + min(k0 + j, K - 1);
  const float* yr = dYs + j * ldy + 4 * g;
  f32x4 acc0 = zero4(), acc1 = zero4();

  for (int n0 = 0; n0 < Npad; n0 += 16) {
    float b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = gload(wp + (size_t)min(n0 + 4 * g + s, N - 1) * K);
    const f32x4 a = *reinterpret_cast<const f32x4*>(yr + n0);
    acc0 = mfma16(a[0], b[0], acc0); acc1 = mfma16(a[1], b[1], acc1); acc0 = mfma16(a[2], b[2], acc0); acc1 = mfma16(a[3], b[3], acc1);
  }
  return acc0 + acc1;
}
// stage a 16-row tile of a feature-major matrix [F][Bp] into LDS as [16][ld], zero beyond F (up to Fpad)
__device__ __forceinline__ void g_stage(float* dst, int ld, int Fpad, const float* __restrict__ srcT, int F, int Bp, int row0) {
  for (int i = threadIdx.x; i < 16 * Fpad; i += blockDim.x) {
    const int r = i & 15, k = i >> 4;
    dst[r * ld + k] = k < F ? gload(srcT + (size_t)k * Bp + row0 + r) : 0.f;
  }
}

struct GLin { const float* XT; int64_t x_ns; const float* P; int64_t p_ns; int64_t oW, ob; int K, N; float* YT; int64_t y_ns; int Bp, act; };
__global__ __launch_bounds__(256) void k_g_linear(GLin a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int net = blockIdx.z, row0 = blockIdx.x * 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const float* P = a.P + net * a.p_ns;
  const int Kpad = round_up16(a.K), ldx = Kpad + 4;
  g_stage(smem, ldx, Kpad, a.XT + net * a.x_ns, a.K, a.Bp, row0);
  __syncthreads();
  const int n0 = (blockIdx.y * 4 + wave) * 16;
  if (n0 >= a.N) return;
  const f32x4 acc = g_tile_fwd(smem, ldx, Kpad, P + a.oW, a.K, n0, a.N);
  const int n = n0 + j;
  if (n < a.N) {
    const float bb = gload(P + a.ob + n);
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = g_act(acc[r] + bb, a.act);
    *reinterpret_cast<f32x4*>(a.YT + net * a.y_ns + (size_t)n * a.Bp + row0 + 4 * g) = v;
  }
}

// dXT[k][row] = (sum_n dZT[n][row] W[n][k]) * act'(HprevT[k][row])    (HprevT == NULL: the input layer, no activation behind it)
struct GBwd { const float* dZT; int64_t dz_ns; const float* P; int64_t p_ns; int64_t oW; int K, N; const float* HprevT; int64_t h_ns; float* dXT; int64_t dx_ns; int Bp, act; };
__global__ __launch_bounds__(256) void k_g_bwd(GBwd a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int net = blockIdx.z, row0 = blockIdx.x * 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int Npad = round_up16(a.N), ldy = Npad + 4;
  g_stage(smem, ldy, Npad, a.dZT + net * a.dz_ns, a.N, a.Bp, row0);
  __syncthreads();
  const int k0 = (blockIdx.y * 4 + wave) * 16;
  if (k0 >= a.K) return;
  const f32x4 acc = g_tile_bwd(smem, ldy, Npad, a.P + net * a.p_ns + a.oW, a.K, a.N, k0);
  const int k = k0 + j;
  if (k < a.K) {
    f32x4 v = acc;
    if (a.HprevT) {
      const f32x4 h = gload4(a.HprevT + net * a.h_ns + (size_t)k * a.Bp + row0 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] * g_act_grad(h[r], a.act);
    }
    *reinterpret_cast<f32x4*>(a.dXT + net * a.dx_ns + (size_t)k * a.Bp + row0 + 4 * g) = v;
  }
}

// G[oW + n K + k] = sum_row dZT[n][row] XT[k][row]: one wave per 16 x 16 tile; both operands as 16-byte lanes along the batch
struct GDw { const float* dZT; int64_t dz_ns; const float* XT; int64_t x_ns; float* G; int64_t g_ns; int64_t oW, ob; int N, K, Bp; };
__device__ __forceinline__ void g_dbias(const GDw& a, int net, int n) {   // one wave per output feature
  const int lane = threadIdx.x & 63;
  if (n >= a.N) return;
  const float* z = a.dZT + net * a.dz_ns + (size_t)n * a.Bp;
  float s = 0.f;
  for (int r = lane; r < a.Bp; r += 64) s += gload(z + r);
  s = wave_sum(s);
  if (lane == 0) a.G[net * a.g_ns + a.ob + n] = s;
}
// grid.x = ceil(tiles / 4) workgroups of weight-gradient tiles, then ceil(N / 4) workgroups of bias sums (one launch per layer instead of two)
__global__ __launch_bounds__(256) void k_g_dw(GDw a) {
  const int net = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int tk = (a.K + 15) >> 4, tn = (a.N + 15) >> 4, tile = blockIdx.x * 4 + wave, tile_wgs = (tk * tn + 3) >> 2;
  if ((int)blockIdx.x >= tile_wgs) { g_dbias(a, net, ((int)blockIdx.x - tile_wgs) * 4 + wave); return; }
  if (tile >= tk * tn) return;
  const int n0 = (tile / tk) * 16, k0 = (tile - (tile / tk) * tk) * 16;
  const float* zr = a.dZT + net * a.dz_ns + (size_t)min(n0 + j, a.N - 1) * a.Bp + 4 * g;
  const float* xr = a.XT + net * a.x_ns + (size_t)min(k0 + j, a.K - 1) * a.Bp + 4 * g;
  f32x4 acc0 = zero4(), acc1 = zero4();

  for (int r0 = 0; r0 < a.Bp; r0 += 16) {
    const f32x4 z = gload4(zr + r0), x = gload4(xr + r0);
    acc0 = mfma16(z[0], x[0], acc0); acc1 = mfma16(z[1], x[1], acc1); acc0 = mfma16(z[2], x[2], acc0); acc1 = mfma16(z[3], x[3], acc1);
  }
  const f32x4 acc = acc0 + acc1;
  float* G = a.G + net * a.g_ns + a.oW;
  const int k = k0 + j;
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int n = n0 + 4 * g + r; if (n < a.N && k < a.K) G[(size_t)n * a.K + k] = acc[r]; }
}
// X0T[k][row] = cat(f1, f2)[row][k] for row < n, zero padding rows (f2 == NULL: K2 columns are left to the head kernel that produces them)
__global__ __launch_bounds__(256) void k_g_pack(const float* __restrict__ f1, int ld1, int K1, const float* __restrict__ f2, int ld2, int K2, int n, int Bp, float* __restrict__ XT) {
  const int total = (K1 + (f2 ? K2 : 0)) * Bp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i / Bp, row = i - k * Bp;
    float v = 0.f;
    if (row < n) v = k < K1 ? f1[(size_t)row * ld1 + k] : f2[(size_t)row * ld2 + (k - K1)];
    XT[i] = v;
  }
}

// the five inputs of one SAC update in one launch: blockIdx.y selects (s'), (s), (s', .), (s, a), (s, .)
struct GPackSac { il_batch b; int S, A, Bp; float* xa2; float* xa; float* xt; float* xc; float* xp; };
__global__ __launch_bounds__(256) void k_g_pack_sac(GPackSac a) {
  const int which = blockIdx.y, S = a.S, A = a.A, Bp = a.Bp;
  const bool next = which == 0 || which == 2, with_a = which == 3;
  float* XT = which == 0 ? a.xa2 : (which == 1 ? a.xa : (which == 2 ? a.xt : (which == 3 ? a.xc : a.xp)));
  const float* f1 = next ? a.b.next_states : a.b.states; const int ld1 = next ? a.b.ld_next_states : a.b.ld_states;
  const int total = (S + (with_a ? A : 0)) * Bp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i / Bp, row = i - k * Bp;
    float v = 0.f;
    if (row < a.b.n) v = k < S ? f1[(size_t)row * ld1 + k] : a.b.actions[(size_t)row * a.b.ld_actions + (k - S)];
    XT[i] = v;
  }
}

// ---- per-row pieces ---------------------------------------------------------------------------------------------------------------------------------------------
// models.py:90-94 + torch.distributions: the tanh-Gaussian head of one row (op order of oracle/nets.py tanh_gaussian_logp)
__device__ __forceinline__ void g_head(float mean, float ls_raw, float eps, float& x, float& a, float& nlp, float& ladj) {
  const float sd = expf(fminf(fmaxf(ls_raw, -20.f), 2.f));
  x = __fadd_rn(__fmul_rn(eps, sd), mean);
  a = tanhf(x);
  const float d = __fsub_rn(x, mean);
  nlp = -(d * d) / (2.f * (sd * sd)) - logf(sd) - LOG_SQRT_2PI;
  ladj = 2.f * (LOG_2 - x - softplus_f(-2.f * x));
}
struct GSample {
  const float* outT; int Bp, n, A;                       // actor output [2A][Bp]
  const float* eps; uint64_t seed; const uint32_t* ctr_ptr; uint32_t ctr; int stream_id;   // eps [n][A] or Philox(seed, ctr_ptr ? *ctr_ptr : ctr, stream_id)
  const float* absorbing; int ld_abs;                    // non-NULL: the action is multiplied by (1 - absorbing) (training.py:21 on s')
  float* aT; float* a_rows; int ld_a;                    // destinations: feature-major rows [A][Bp] and / or row-major [n][ld_a]
  float* xT; float* epsT; float* logp; int greedy;
};
__global__ __launch_bounds__(256) void k_g_sample(GSample a) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= a.Bp) return;
  if (row >= a.n) {   // padding rows: zero inputs for the next network
    if (a.aT) for (int c = 0; c < a.A; ++c) a.aT[(size_t)c * a.Bp + row] = 0.f;
    return;
  }
  const uint32_t ctr = a.ctr_ptr ? *a.ctr_ptr : a.ctr;
  const float m = a.absorbing ? 1.f - a.absorbing[(size_t)row * a.ld_abs] : 1.f;
  float sn = 0.f, sl = 0.f;
  for (int c = 0; c < a.A; ++c) {
    const float mean = a.outT[(size_t)c * a.Bp + row], lsr = a.outT[(size_t)(a.A + c) * a.Bp + row];
    float x, act, nlp, ladj, e = 0.f;
    if (a.greedy) { act = tanhf(mean); x = mean; nlp = 0.f; ladj = 0.f; }
    else {
      e = a.eps ? a.eps[(size_t)row * a.A + c] : philox_normal(a.seed, ctr, a.stream_id, (uint32_t)(row * a.A + c));
      g_head(mean, lsr, e, x, act, nlp, ladj);
    }
    sn += nlp; sl += ladj;
    if (a.aT) a.aT[(size_t)c * a.Bp + row] = m * act;
    if (a.a_rows) a.a_rows[(size_t)row * a.ld_a + c] = m * act;
    if (a.xT) a.xT[(size_t)c * a.Bp + row] = x;
    if (a.epsT) a.epsT[(size_t)c * a.Bp + row] = e;
  }
  if (a.logp) a.logp[row] = (0.f - sl) + sn;
}
// models.py:97-99 log pi(a | s) of GIVEN actions (clamped, atanh); with `weights` also the behavioural-cloning seed (training.py:57-64): d(-mean(w logp)) / d(head outputs)
__global__ __launch_bounds__(256) void k_g_logp(const float* __restrict__ outT, int Bp, int n, int A, const float* __restrict__ actions, int ld_a, float* __restrict__ logp,
                                                const float* __restrict__ weights, int ld_w, float* __restrict__ doutT, float* __restrict__ loss_rows) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Bp) return;
  if (row >= n) { if (doutT) for (int c = 0; c < 2 * A; ++c) doutT[(size_t)c * Bp + row] = 0.f; return; }
  const float up = weights ? -weights[(size_t)row * ld_w] / (float)n : 0.f;
  float sn = 0.f, sl = 0.f;
  for (int c = 0; c < A; ++c) {
    const float mean = outT[(size_t)c * Bp + row], lsr = outT[(size_t)(A + c) * Bp + row];
    const float sd = expf(fminf(fmaxf(lsr, -20.f), 2.f));
    const float act = fminf(fmaxf(actions[(size_t)row * ld_a + c], -1.f + 1e-6f), 1.f - 1e-6f);
    const float x = atanhf(act), df = x - mean, var = sd * sd;
    sn += -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;
    sl += 2.f * (LOG_2 - x - softplus_f(-2.f * x));
    if (doutT) {
      const float dsd = up * (df * df / (var * sd) - 1.f / sd);
      doutT[(size_t)c * Bp + row] = up * df / var;
      doutT[(size_t)(A + c) * Bp + row] = (lsr >= -20.f && lsr <= 2.f) ? dsd * sd : 0.f;
    }
  }
  const float lp = (0.f - sl) + sn;
  if (logp) logp[row] = lp;
  if (loss_rows) loss_rows[row] = weights ? -weights[(size_t)row * ld_w] * lp : 0.f;
}
// training.py:22-31: y = r + (1 - d) gamma (min Q'(s', a') - (1 - absorbing) alpha log pi(a'|s')); dL/dQ_k = w 2 (Q_k - y) / B; Q_values = min(Q_1, Q_2) (:54)
__global__ __launch_bounds__(256) void k_g_critic_seed(il_batch b, const float* __restrict__ qtT, const float* __restrict__ qT, const float* __restrict__ logp2, const float* __restrict__ log_alpha,
                                                       float discount, int Bp, float* __restrict__ dqT, float* __restrict__ out_q) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Bp) return;
  if (row >= b.n) { dqT[row] = 0.f; dqT[Bp + row] = 0.f; return; }
  const float alpha = expf(log_alpha[0]);
  const float m = 1.f - b.absorbing[(size_t)row * b.ld_absorbing];
  const float tv = fminf(qtT[row], qtT[Bp + row]) - m * alpha * logp2[row];
  const float y = b.rewards[(size_t)row * b.ld_rewards] + (1.f - b.terminals[(size_t)row * b.ld_terminals]) * discount * tv;
  const float w = b.weights[(size_t)row * b.ld_weights], q1 = qT[row], q2 = qT[Bp + row];
  dqT[row] = (w * (2.f * (q1 - y))) / (float)b.n;
  dqT[Bp + row] = (w * (2.f * (q2 - y))) / (float)b.n;
  if (out_q) out_q[row] = fminf(q1, q2);
}
// training.py:37-38: d(-mean(min(Q_1, Q_2))) / dQ_k = -[k is the smaller] / B (a tie splits, like torch.min's backward)
__global__ __launch_bounds__(256) void k_g_policy_seed(const float* __restrict__ qnT, int n, int Bp, float* __restrict__ dqT) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Bp) return;
  if (row >= n) { dqT[row] = 0.f; dqT[Bp + row] = 0.f; return; }
  const float q1 = qnT[row], q2 = qnT[Bp + row];
  const float sel = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
  dqT[row] = -(sel) / (float)n;
  dqT[Bp + row] = -(1.f - sel) / (float)n;
}
// training.py:35-42 back through the head: dL/d(mean, log_std_raw) from dQ/da (both critics' input gradients) and the entropy term; alpha_rows = w m (log pi + target entropy)
struct GHeadBwd { il_batch b; const float* outT; const float* xT; const float* epsT; const float* logp; const float* dx0T; int64_t dx_ns; const float* log_alpha; float entropy_target;
                  int S, A, Bp; float* doutT; float* alpha_rows; float* out_logp; };
__global__ __launch_bounds__(256) void k_g_head_bwd(GHeadBwd a) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x, A = a.A, Bp = a.Bp;
  if (row >= Bp) return;
  if (row >= a.b.n) { for (int c = 0; c < 2 * A; ++c) a.doutT[(size_t)c * Bp + row] = 0.f; a.alpha_rows[row] = 0.f; return; }
  const float alpha = expf(a.log_alpha[0]);
  const float w = a.b.weights[(size_t)row * a.b.ld_weights], m = 1.f - a.b.absorbing[(size_t)row * a.b.ld_absorbing];
  const float cc = (w * m * alpha) / (float)a.b.n;
  for (int c = 0; c < A; ++c) {
    const float x = a.xT[(size_t)c * Bp + row], e = a.epsT[(size_t)c * Bp + row], lsr = a.outT[(size_t)(A + c) * Bp + row];
    const float sd = expf(fminf(fmaxf(lsr, -20.f), 2.f)), an = tanhf(x);
    const float da = a.dx0T[(size_t)(a.S + c) * Bp + row] + a.dx0T[a.dx_ns + (size_t)(a.S + c) * Bp + row];
    const float dx = cc * (2.f * an) + da * (1.f - an * an);
    const float dsd = dx * e - cc / sd;
    a.doutT[(size_t)c * Bp + row] = dx;
    a.doutT[(size_t)(A + c) * Bp + row] = (lsr >= -20.f && lsr <= 2.f) ? dsd * sd : 0.f;
  }
  const float lp = a.logp[row];
  a.alpha_rows[row] = w * m * (lp + a.entropy_target);
  if (a.out_logp) a.out_logp[row] = lp;
}
// training.py:45-49: Adam (not AdamW) on log_alpha; the update's Philox counter advances here, as in the fused path's tail
__global__ void k_g_alpha(const float* __restrict__ alpha_rows, int n, float* __restrict__ log_alpha, il_adam opt, float* __restrict__ alpha_grad, int grads_only, uint32_t* noise_counter) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += alpha_rows[i];
  const float gr = -(expf(log_alpha[0])) * (s / (float)n);
  if (alpha_grad) alpha_grad[0] = gr;
  if (!grads_only) {
    adam_tick(opt);
    const adam_consts ac = load_adam_consts(opt);
    float pp = log_alpha[0], mm = opt.m[0], vv = opt.v[0];
    adam_update(pp, gr, mm, vv, ac);
    log_alpha[0] = pp; opt.m[0] = mm; opt.v[0] = vv;
  }
  if (noise_counter) noise_counter[0] += 1;
}
__global__ void k_g_sum_rows(const float* __restrict__ rows, int n, float* __restrict__ out) {   // loss = sum / n (one thread: deterministic order)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += rows[i];
  out[0] = s / (float)n;
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------------------------------------
struct GNet { int in, H, depth, out, act; };
static int g_check_shape(const GNet& s, const char* who) {
  IL_CHECK_ARG(s.depth >= 1 && s.depth <= 8 && s.H >= 1 && s.H <= 2048 && s.in >= 1 && s.in <= 2048 && s.out >= 1 && s.out <= 2048 && s.act >= 0 && s.act <= 2,
               "%s: general shapes cover depth 1-8, widths <= 2048, activation 0 relu / 1 tanh / 2 sigmoid (got in=%d hidden=%d depth=%d out=%d activation=%d)", who, s.in, s.H, s.depth, s.out, s.act);
  return IL_OK;
}
static size_t g_lds(int F) { return (size_t)16 * (round_up16(F) + 4) * sizeof(float); }
static int g_lds_ok(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}
// hidden activations of `nets` networks: [net][layer 1 .. depth][H][Bp]; output [net][out][Bp]
static inline int64_t g_hidden_floats(const GNet& s, int Bp) { return (int64_t)s.depth * s.H * Bp; }

static int g_forward(hipStream_t st, const GNet& s, const float* P, int64_t p_ns, int nets, const float* X0T, int64_t x_ns, float* HT, float* OT, int Bp) {
  const int64_t h_ns = g_hidden_floats(s, Bp);
  for (int l = 0; l <= s.depth; ++l) {
    const GLayer L = g_layer(s.in, s.H, s.depth, s.out, l);
    GLin a;
    a.XT = l == 0 ? X0T : HT + (int64_t)(l - 1) * s.H * Bp; a.x_ns = l == 0 ? x_ns : h_ns;
    a.P = P; a.p_ns = p_ns; a.oW = L.oW; a.ob = L.ob; a.K = L.K; a.N = L.N;
    a.YT = l == s.depth ? OT : HT + (int64_t)l * s.H * Bp; a.y_ns = l == s.depth ? (int64_t)s.out * Bp : h_ns;
    a.Bp = Bp; a.act = l == s.depth ? G_ACT_NONE : s.act;
    const size_t lds = g_lds(L.K);
    if (int rc = g_lds_ok((const void*)k_g_linear, lds)) return rc;
    { IL_TRACE("k_g_linear", st); k_g_linear<<<dim3(Bp / 16, (L.N + 63) / 64, nets), 256, lds, st>>>(a); }
  }
  return IL_OK;
}
// dOT [net][out][Bp] -> gradients G (flat, NULL: none) and dX0T [net][in][Bp] (NULL: none); dZ scratch [net][depth][H][Bp]
static int g_backward(hipStream_t st, const GNet& s, const float* P, int64_t p_ns, int nets, const float* X0T, int64_t x_ns, const float* HT, const float* dOT, float* dZT, float* dX0T,
                      float* G, int64_t g_ns, int Bp) {
  const int64_t h_ns = g_hidden_floats(s, Bp);
  for (int l = s.depth; l >= 0; --l) {
    const GLayer L = g_layer(s.in, s.H, s.depth, s.out, l);
    const float* dz = l == s.depth ? dOT : dZT + (int64_t)l * s.H * Bp; const int64_t dz_ns = l == s.depth ? (int64_t)s.out * Bp : h_ns;
    const float* xin = l == 0 ? X0T : HT + (int64_t)(l - 1) * s.H * Bp; const int64_t xin_ns = l == 0 ? x_ns : h_ns;
    if (G) {
      GDw w; w.dZT = dz; w.dz_ns = dz_ns; w.XT = xin; w.x_ns = xin_ns; w.G = G; w.g_ns = g_ns; w.oW = L.oW; w.ob = L.ob; w.N = L.N; w.K = L.K; w.Bp = Bp;
      const int tiles = ((L.N + 15) / 16) * ((L.K + 15) / 16);
      { IL_TRACE("k_g_dw", st); k_g_dw<<<dim3((tiles + 3) / 4 + (L.N + 3) / 4, 1, nets), 256, 0, st>>>(w); }
    }
    if (l > 0 || dX0T) {
      GBwd a; a.dZT = dz; a.dz_ns = dz_ns; a.P = P; a.p_ns = p_ns; a.oW = L.oW; a.K = L.K; a.N = L.N;
      a.HprevT = l > 0 ? HT + (int64_t)(l - 1) * s.H * Bp : nullptr; a.h_ns = h_ns;
      a.dXT = l > 0 ? dZT + (int64_t)(l - 1) * s.H * Bp : dX0T; a.dx_ns = l > 0 ? h_ns : (int64_t)s.in * Bp;
      a.Bp = Bp; a.act = s.act;
      const size_t lds = g_lds(L.N);
      if (int rc = g_lds_ok((const void*)k_g_bwd, lds)) return rc;
      { IL_TRACE("k_g_bwd", st); k_g_bwd<<<dim3(Bp / 16, (L.K + 63) / 64, nets), 256, lds, st>>>(a); }
    }
  }
  return IL_OK;
}
static void g_pack(hipStream_t st, const float* f1, int ld1, int K1, const float* f2, int ld2, int K2, int n, int Bp, float* XT) {
  const int total = (K1 + (f2 ? K2 : 0)) * Bp;
  IL_TRACE("k_g_pack", st);
  k_g_pack<<<(total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024, 256, 0, st>>>(f1, ld1, K1, f2, ld2, K2, n, Bp, XT);
}


// =================================================================================================================================================================
// Tile engine (round 6): the same networks as 16-row tiles that stay in LDS across ALL layers of a pass, built from the fused path's primitives (mlp_tile.hpp: tile_fwd,
// tile_fwd_packed / tile_bwd_packed on lane-ordered copies of the H x H layers, tile_bwd_dx, tile_fwd_small). A depth-3 update is 9 launches instead of 49:
//   k_gt_repack   lane-ordered PF / PB copies of every H x H layer (actor, critics, targets)
//   k_gt_fwd      up to four PASSES per launch (blockIdx.y): actor(s') and actor(s) with the tanh-Gaussian head in the epilogue, critic_1,2(s, a); then target_1,2(s', a');
//                 later critic_1,2(s, a~). A workgroup = (tile, pass): input rows -> every hidden layer -> output, hidden activations written feature-major for the backward
//   k_gt_bwd      (tile, net): the loss seed of the tile's rows in the prologue (critic loss / policy / actor head / behavioural cloning - the per-row kernels' arithmetic),
//                 then dZ of every layer back to front, written feature-major for k_gt_dw; the policy pass ends with the action columns of dL/d(input)
//   k_gt_dw       every layer's weight and bias gradient of up to two networks in one launch (k_g_dw's tiles), AdamW in the epilogue (and the lane-ordered copies kept in step);
//                 the actor's launch carries the temperature step, the target update and the Philox counter as tail workgroups
// Applies when both networks have hidden widths that are multiples of 16 (<= 512), inputs <= 512 wide and 2A <= 16; anything else keeps the layer-at-a-time launches above
// (IL_GENERAL_TILES=0 forces them: the A/B and the bit-for-bit reference of the hidden layers). Per element the hidden layers run k_g_linear / k_g_bwd's MFMA order; the output
// layer's dot product is split over the waves (tile_fwd_small: partial tiles summed in k-block order), so results equal the layer-at-a-time path to rounding, not bit for bit.
// =================================================================================================================================================================
struct GtNet { const float* P; int in, H, depth, out, act; const float* PF; const float* PB; };   // PF / PB: [depth - 1][H * H] lane-ordered copies (NULL: read W directly)
__host__ __device__ static inline int gt_threads(int H) { return H * 4 < 256 ? 256 : (H * 4 > 512 ? 512 : H * 4); }   // (8 waves: the 256-VGPR budget keeps a whole 16 KB weight panel in registers without a spill; 16 waves spilled 18 - 30 VGPRs)

struct GtFwdPass {
  GtNet net;
  const float* f1; int ld1, K1; const float* f2; int ld2, K2;   // input rows cat(f1, f2), row-major; rows >= n are zero
  float* X0T; float* HT; float* OT;                              // feature-major outputs (NULL: not kept): input copy [in][Bp], hidden activations [depth][H][Bp], raw output [out][Bp]
  int head;                                                      // 1: tanh-Gaussian sample of an actor pass (GSample's fields below)
  const float* eps; uint64_t seed; const uint32_t* ctr_ptr; uint32_t ctr; int stream_id; const float* absorbing; int ld_abs; int greedy;
  float* a_rows; int ld_a; float* xT; float* epsT; float* logp;
};
struct GtFwd { GtFwdPass p[4]; int n, Bp, npass; };

__global__ __launch_bounds__(512) void k_gt_fwd(GtFwd a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // 1-D grid of (tile, pass) pairs, decoded so that the tiles of ONE pass share 8 / passes XCDs (xcd_tile_net): an XCD's private L2 then pulls one or two networks through
  // the fabric per layer instead of all of them (the first launch of an update, 4 passes x 16 tiles: 31 -> us with every XCD pulling all four networks)
  int tile_, pass_;
  xcd_tile_net((int)blockIdx.x, a.Bp / 16, a.npass, tile_, pass_);
  const GtFwdPass& q = a.p[pass_];
  const GtNet& nn = q.net;
  const int H = nn.H, in = nn.in, depth = nn.depth, out = nn.out, act = nn.act, Bp = a.Bp;
  const int row0 = tile_ * 16, nrows = min(16, a.n - row0);
  const int inp = round_up16(in), ldx = inp + 4, ldh = H + 4;
  float* Xs = smem; float* A0 = Xs + 16 * ldx; float* A1 = A0 + 16 * ldh; float* part = A1 + 16 * ldh; float* Os = part + (H >> 4) * 256;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, tid = threadIdx.x;
  const bool stamp = tile_ == 0 && pass_ == 1;
  IL_STAMP(stamp, 0);
  load_rows_cat(Xs, ldx, inp, q.f1, q.ld1, q.K1, q.f2, q.ld2, q.K2, row0, nrows);
  __syncthreads();
  IL_STAMP(stamp, 1);
  if (q.X0T)
    for (int i = tid; i < 16 * in; i += blockDim.x) { const int c = i >> 4, r = i & 15; q.X0T[(size_t)c * Bp + row0 + r] = Xs[r * ldx + c]; }
  const float* P = nn.P;
  float* cur = A0; float* nxt = A1;
  const int wcol = min((int)(threadIdx.x >> 6) * 16 + j, H - 1);   // this lane's column of the wave's FIRST output tile: its bias is requested ahead of each layer's MFMAs
  {
    const GLayer L = g_layer(in, H, depth, out, 0);
    const float* bias = P + L.ob;
    const float pb = gload(bias + wcol);
    tile_fwd(Xs, ldx, inp, P + L.oW, in, in, H, [&](int c0, f32x4 acc) {
      const int col = c0 + j; const float bb = col == wcol ? pb : gload(bias + col);
      f32x4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { hv[r] = g_act(acc[r] + bb, act); cur[(4 * g + r) * ldh + col] = hv[r]; }
      if (q.HT) *reinterpret_cast<f32x4*>(q.HT + (size_t)col * Bp + row0 + 4 * g) = hv;
    });
  }
  __syncthreads();
  IL_STAMP(stamp, 2);
  for (int l = 1; l < depth; ++l) {
    const GLayer L = g_layer(in, H, depth, out, l);
    const float* bias = P + L.ob;
    float* ht = q.HT ? q.HT + (size_t)l * H * Bp : nullptr;
    const float pb = gload(bias + wcol);
    auto epi = [&](int c0, f32x4 acc) {
      const int col = c0 + j; const float bb = col == wcol ? pb : gload(bias + col);
      f32x4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { hv[r] = g_act(acc[r] + bb, act); nxt[(4 * g + r) * ldh + col] = hv[r]; }
      if (ht) *reinterpret_cast<f32x4*>(ht + (size_t)col * Bp + row0 + 4 * g) = hv;
    };
    if (nn.PF) tile_packed_x2(cur, ldh, H, nn.PF + (size_t)(l - 1) * H * H, epi);
    else tile_fwd(cur, ldh, H, P + L.oW, H, H, H, epi);
    __syncthreads();
    IL_STAMP(stamp, 2 + l);
    float* t = cur; cur = nxt; nxt = t;
  }
  {
    const GLayer L = g_layer(in, H, depth, out, depth);
    tile_fwd_small(cur, ldh, H, P + L.oW, H, out, P + L.ob, Os, part);   // (barriers inside)
  }
  IL_STAMP(stamp, 10);
  if (q.OT)
    for (int i = tid; i < 16 * out; i += blockDim.x) { const int c = i >> 4, r = i & 15; q.OT[(size_t)c * Bp + row0 + r] = Os[r * 16 + c]; }
  if (q.head) {   // k_g_sample's arithmetic: one thread per (row, component) (a thread per row walked A Philox draws and heads one after the other: ~5 us of the launch), then
    const int A = out >> 1;                                                       // the per-row sums in component order, as k_g_sample adds them
    float* nl = part; float* la = part + 256;
    if (tid < 16 * A) {
      const int r = tid / A, c = tid - r * A, row = row0 + r;
      if (row < a.n) {
        const uint32_t ctr = q.ctr_ptr ? *q.ctr_ptr : q.ctr;
        const float m = q.absorbing ? 1.f - q.absorbing[(size_t)row * q.ld_abs] : 1.f;
        const float mean = Os[r * 16 + c], lsr = Os[r * 16 + A + c];
        float x, av, nlp, ladj, e = 0.f;
        if (q.greedy) { av = tanhf(mean); x = mean; nlp = 0.f; ladj = 0.f; }
        else {
          e = q.eps ? q.eps[(size_t)row * A + c] : philox_normal(q.seed, ctr, q.stream_id, (uint32_t)(row * A + c));
          g_head(mean, lsr, e, x, av, nlp, ladj);
        }
        nl[r * 16 + c] = nlp; la[r * 16 + c] = ladj;
        if (q.a_rows) q.a_rows[(size_t)row * q.ld_a + c] = m * av;
        if (q.xT) q.xT[(size_t)c * Bp + row] = x;
        if (q.epsT) q.epsT[(size_t)c * Bp + row] = e;
      }
    }
    __syncthreads();
    if (tid < 16 && row0 + tid < a.n && q.logp) {
      float sn = 0.f, sl = 0.f;
      for (int c = 0; c < A; ++c) { sn += nl[tid * 16 + c]; sl += la[tid * 16 + c]; }
      q.logp[row0 + tid] = (0.f - sl) + sn;
    }
  }
  IL_STAMP(stamp, 11);
}

// seeds of k_gt_bwd (the per-row kernels' arithmetic: k_g_critic_seed, k_g_policy_seed, k_g_head_bwd, k_g_logp)
enum { GT_SEED_CRITIC = 1, GT_SEED_POLICY = 2, GT_SEED_HEAD = 3, GT_SEED_BC = 4 };
struct GtBwd {
  GtNet net; int64_t p_ns, pk_ns;            // twin critics: parameters / lane-ordered copies of net k at + k * stride
  const float* HT; int64_t h_ns; float* dZT; int64_t dz_ns; float* dOT; int64_t do_ns;   // [net][depth][H][Bp] activations in, dZ out; [net][out][Bp] output-layer dZ out
  float* dX0T; int64_t dx_ns; int dx_c0, dx_c1;   // columns [c0, c1) of dL/d(input) -> dX0T[net][in][Bp] (NULL: none)
  int seed, n, Bp, nets;
  il_batch b;
  const float* qtT; const float* qT; const float* logp2; const float* log_alpha; float discount; float* out_q;   // critic loss (qT also: the policy pass's Q(s, a~))
  const float* outT; const float* xT; const float* epsT; const float* logp; const float* dx0T; int64_t dx0_ns; float entropy_target; int S; float* alpha_rows; float* out_logp;   // actor head
  float* loss_rows;                                                                                           // behavioural cloning
  il_adam tick;                                                                                              // ticked by (tile 0, net 0): consumed by the k_gt_dw that follows (step == NULL: none)
};
__global__ __launch_bounds__(512) void k_gt_bwd(GtBwd a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const GtNet& nn = a.net;
  int tile_, net;
  xcd_tile_net((int)blockIdx.x, a.Bp / 16, a.nets, tile_, net);   // (a network's tiles on 8 / nets XCDs: see k_gt_fwd)
  const int H = nn.H, in = nn.in, depth = nn.depth, out = nn.out, act = nn.act, Bp = a.Bp;
  const int row0 = tile_ * 16, ldh = H + 4, ldy = 20;
  float* dYs = smem; float* Z0 = dYs + 16 * ldy; float* Z1 = Z0 + 16 * ldh;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, tid = threadIdx.x;
  const float* P = nn.P + net * a.p_ns;
  const float* HT = a.HT + net * a.h_ns;
  float* dZT = a.dZT + net * a.dz_ns;
  const bool stamp = tile_ == 0 && net == 0;
  IL_STAMP(stamp, 16);
  for (int i = tid; i < 16 * ldy; i += blockDim.x) dYs[i] = 0.f;
  __syncthreads();
  if (tid < 16) {
    const int row = row0 + tid, n = a.n;
    const bool live = row < n;
    if (a.seed == GT_SEED_CRITIC) {
      float dq = 0.f;
      if (live) {
        const float alpha = expf(a.log_alpha[0]);
        const float m = 1.f - a.b.absorbing[(size_t)row * a.b.ld_absorbing];
        const float tv = fminf(a.qtT[row], a.qtT[Bp + row]) - m * alpha * a.logp2[row];
        const float y = a.b.rewards[(size_t)row * a.b.ld_rewards] + (1.f - a.b.terminals[(size_t)row * a.b.ld_terminals]) * a.discount * tv;
        const float w = a.b.weights[(size_t)row * a.b.ld_weights], q1 = a.qT[row], q2 = a.qT[Bp + row];
        dq = (w * (2.f * ((net == 0 ? q1 : q2) - y))) / (float)n;
        if (net == 0 && a.out_q) a.out_q[row] = fminf(q1, q2);
      }
      dYs[tid * ldy] = dq;
    } else if (a.seed == GT_SEED_POLICY) {
      float dq = 0.f;
      if (live) {
        const float q1 = a.qT[row], q2 = a.qT[Bp + row];
        const float sel = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
        dq = net == 0 ? -(sel) / (float)n : -(1.f - sel) / (float)n;
      }
      dYs[tid * ldy] = dq;
    } else if (a.seed == GT_SEED_HEAD) {   // (the per-row outputs; the head gradient itself: one thread per (row, component) below)
      if (live) {
        const float w = a.b.weights[(size_t)row * a.b.ld_weights], m = 1.f - a.b.absorbing[(size_t)row * a.b.ld_absorbing];
        const float lp = a.logp[row];
        a.alpha_rows[row] = w * m * (lp + a.entropy_target);
        if (a.out_logp) a.out_logp[row] = lp;
      } else if (row < Bp) a.alpha_rows[row] = 0.f;
    } else {   // GT_SEED_BC: d(-mean(w log pi(a | s))) / d(head outputs)
      const int A = out >> 1;
      if (live) {
        const float wt = a.b.weights[(size_t)row * a.b.ld_weights], up = -wt / (float)n;
        float sn = 0.f, sl = 0.f;
        for (int c = 0; c < A; ++c) {
          const float mean = a.outT[(size_t)c * Bp + row], lsr = a.outT[(size_t)(A + c) * Bp + row];
          const float sd = expf(fminf(fmaxf(lsr, -20.f), 2.f));
          const float av = fminf(fmaxf(a.b.actions[(size_t)row * a.b.ld_actions + c], -1.f + 1e-6f), 1.f - 1e-6f);
          const float x = atanhf(av), df = x - mean, var = sd * sd;
          sn += -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;
          sl += 2.f * (LOG_2 - x - softplus_f(-2.f * x));
          const float dsd = up * (df * df / (var * sd) - 1.f / sd);
          dYs[tid * ldy + c] = up * df / var;
          dYs[tid * ldy + A + c] = (lsr >= -20.f && lsr <= 2.f) ? dsd * sd : 0.f;
        }
        if (a.loss_rows) a.loss_rows[row] = -wt * ((0.f - sl) + sn);
      } else if (row < Bp && a.loss_rows) a.loss_rows[row] = 0.f;
    }
  }
  if (a.seed == GT_SEED_HEAD) {   // k_g_head_bwd's arithmetic with every operand of the tile requested at once (a thread per row walked A dependent round trips: 5.5 us)
    const int A = out >> 1, t2 = tid - 64;   // (waves 1, 2: wave 0 holds the per-row threads above)
    if (t2 >= 0 && t2 < 16 * A) {
      const int r = t2 / A, c = t2 - r * A, row = row0 + r;
      if (row < a.n) {
        const float alpha = expf(a.log_alpha[0]);
        const float w = a.b.weights[(size_t)row * a.b.ld_weights], m = 1.f - a.b.absorbing[(size_t)row * a.b.ld_absorbing];
        const float cc = (w * m * alpha) / (float)a.n;
        const float x = a.xT[(size_t)c * Bp + row], e = a.epsT[(size_t)c * Bp + row], lsr = a.outT[(size_t)(A + c) * Bp + row];
        const float da = a.dx0T[(size_t)(a.S + c) * Bp + row] + a.dx0T[a.dx0_ns + (size_t)(a.S + c) * Bp + row];
        const float sd = expf(fminf(fmaxf(lsr, -20.f), 2.f)), an = tanhf(x);
        const float dx = cc * (2.f * an) + da * (1.f - an * an);
        const float dsd = dx * e - cc / sd;
        dYs[r * ldy + c] = dx;
        dYs[r * ldy + A + c] = (lsr >= -20.f && lsr <= 2.f) ? dsd * sd : 0.f;
      }
    }
  }
  if (tile_ == 0 && net == 0 && tid == 192 && a.tick.step) adam_tick(a.tick);
  __syncthreads();
  IL_STAMP(stamp, 17);
  if (a.dOT)
    for (int i = tid; i < 16 * out; i += blockDim.x) { const int c = i >> 4, r = i & 15; a.dOT[net * a.do_ns + (size_t)c * Bp + row0 + r] = dYs[r * ldy + c]; }
  float* cur = Z0; float* nxt = Z1;
  const int wk = min((int)(threadIdx.x >> 6) * 16 + j, H - 1);   // this lane's feature of the wave's FIRST tile: its activation lane is requested ahead of each layer's MFMAs
  {   // through the output layer: dZ of the last hidden layer
    const GLayer L = g_layer(in, H, depth, out, depth);
    const float* hp = HT + (size_t)(depth - 1) * H * Bp;
    float* dz = dZT + (size_t)(depth - 1) * H * Bp;
    const f32x4 hpre = gload4(hp + (size_t)wk * Bp + row0 + 4 * g);
    tile_bwd_dx(dYs, ldy, 16, out, P + L.oW, H, H, [&](int kb, f32x4 acc) {
      const size_t off = (size_t)(kb + j) * Bp + row0 + 4 * g;
      const f32x4 h = kb + j == wk ? hpre : gload4(hp + off);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[r] * g_act_grad(h[r], act); cur[(4 * g + r) * ldh + kb + j] = v[r]; }
      *reinterpret_cast<f32x4*>(dz + off) = v;
    });
  }
  __syncthreads();
  IL_STAMP(stamp, 18);
  for (int l = depth - 1; l >= 1; --l) {   // dZ of hidden layer l (in `cur`) -> dZ of hidden layer l - 1
    const GLayer L = g_layer(in, H, depth, out, l);
    const float* hp = HT + (size_t)(l - 1) * H * Bp;
    float* dz = dZT + (size_t)(l - 1) * H * Bp;
    const f32x4 hpre = gload4(hp + (size_t)wk * Bp + row0 + 4 * g);
    auto epi = [&](int kb, f32x4 acc) {
      const size_t off = (size_t)(kb + j) * Bp + row0 + 4 * g;
      const f32x4 h = kb + j == wk ? hpre : gload4(hp + off);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[r] * g_act_grad(h[r], act); nxt[(4 * g + r) * ldh + kb + j] = v[r]; }
      *reinterpret_cast<f32x4*>(dz + off) = v;
    };
    if (nn.PB) tile_packed_x2(cur, ldh, H, nn.PB + net * a.pk_ns + (size_t)(l - 1) * H * H, epi);
    else tile_bwd_dx(cur, ldh, H, H, P + L.oW, H, H, epi);
    __syncthreads();
    IL_STAMP(stamp, 18 + (depth - l));
    float* t = cur; cur = nxt; nxt = t;
  }
  if (a.dX0T) {   // dL/d(input), columns [dx_c0, dx_c1): no activation behind the input
    const GLayer L = g_layer(in, H, depth, out, 0);
    float* dx = a.dX0T + net * a.dx_ns;
    // a few columns of a K = in wide product: the N-reduction split over the waves (tile_bwd_dx would keep one or two waves busy with H / 4 dependent MFMAs each)
    tile_bwd_dx_cols(cur, ldh, H, P + L.oW, in, in, a.dx_c0, a.dx_c1, nxt, [&](int col, int row, float v) {
      if (col >= a.dx_c0 && col < a.dx_c1) dx[(size_t)col * Bp + row0 + row] = v;
    });
  }
  IL_STAMP(stamp, 27);
}

// every layer's dW and db of `nets` networks with the optimiser in the epilogue
struct GtDwLayer { const float* dZT; int64_t dz_ns; const float* XT; int64_t x_ns; int64_t oW, ob; int N, K; int tile0, bias0, blk0; float* PF; float* PB; };   // blk0: first 32 x 32 block job of the layer (k_gt_dw32)   // tile0 / bias0: first wave-job of the layer's tiles / bias features; PF / PB: this layer's lane-ordered copies (NULL: none)
struct GtDw {
  GtDwLayer L[9]; int n_layers, jobs_per_net, blocks_per_net, nets; int64_t p_ns, pk_ns;
  float* P; float* G; int64_t g_ns; il_adam opt; int grads_only, Bp;
  int n_job_wgs;   // workgroups of tile / bias jobs; behind them the tail workgroups
  // tail (the actor's launch): temperature step (k_g_alpha), target update (k_polyak), Philox counter
  const float* alpha_rows; int n_rows; float* log_alpha; il_adam alpha_opt; float* alpha_grad; uint32_t* noise_counter;
  float* target; const float* polyak_src; int64_t polyak_n; double tau;
};
__device__ __forceinline__ void gt_dw_tail(const GtDw& a) {   // the workgroups behind the jobs: temperature step + Philox counter (workgroup 0), target update (all)
    const int tb = (int)blockIdx.x - a.n_job_wgs, ntb = (int)gridDim.x - a.n_job_wgs;
    __shared__ float red[32];
    float srow = 0.f;
    if (tb == 0 && a.log_alpha) {   // the rows' terms: thread-strided partial sums, then the block reduction (fixed order; a single thread adding n_rows dependent loads took ~9 us)
      for (int i = threadIdx.x; i < a.n_rows; i += blockDim.x) srow += gload(a.alpha_rows + i);
      srow = block_sum(srow, red);
    }
    if (tb == 0 && threadIdx.x == 0 && a.log_alpha) {
      const float s = srow;
      const float gr = -(expf(a.log_alpha[0])) * (s / (float)a.n_rows);
      if (a.alpha_grad) a.alpha_grad[0] = gr;
      if (!a.grads_only) {
        adam_tick(a.alpha_opt);
        const adam_consts ac = load_adam_consts(a.alpha_opt);
        float pp = a.log_alpha[0], mm = a.alpha_opt.m[0], vv = a.alpha_opt.v[0];
        adam_update(pp, gr, mm, vv, ac);
        a.log_alpha[0] = pp; a.alpha_opt.m[0] = mm; a.alpha_opt.v[0] = vv;
      }
      if (a.noise_counter) a.noise_counter[0] += 1;
    }
    if (a.target && !a.grads_only) {
      const float omt = (float)(1.0 - a.tau), tau = (float)a.tau;
      // 16-byte lanes, four per thread and trip with all eight loads requested first (the elementwise loop was a dependent HBM round trip per element: 8 us of this launch)
      const int64_t n4 = ((a.polyak_n & 3) == 0 && (reinterpret_cast<uintptr_t>(a.target) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.polyak_src) & 15) == 0) ? a.polyak_n >> 2 : 0, stride = (int64_t)ntb * blockDim.x;
      for (int64_t i = (int64_t)tb * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
        f32x4 t[4], p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t q = i + u * stride < n4 ? i + u * stride : i; t[u] = gload4(a.target + 4 * q); p[u] = gload4(a.polyak_src + 4 * q); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int c = 0; c < 4; ++c) t[u][c] = __fadd_rn(__fmul_rn(t[u][c], tau), __fmul_rn(omt, p[u][c]));
          if (i + u * stride < n4) *reinterpret_cast<f32x4*>(a.target + 4 * (i + u * stride)) = t[u];
        }
      }
      for (int64_t i = 4 * n4 + (int64_t)tb * blockDim.x + threadIdx.x; i < a.polyak_n; i += stride)
        a.target[i] = __fadd_rn(__fmul_rn(a.target[i], tau), __fmul_rn(omt, a.polyak_src[i]));
    }
}
__global__ __launch_bounds__(256) void k_gt_dw(GtDw a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  if ((int)blockIdx.x >= a.n_job_wgs) { gt_dw_tail(a); return; }
  int job = blockIdx.x * 4 + wave;
  if (job >= a.jobs_per_net * a.nets) return;
  const int net = job / a.jobs_per_net; job -= net * a.jobs_per_net;
  int li = 0;
  for (int i = 1; i < a.n_layers; ++i) if (job >= a.L[i].tile0) li = i;
  const GtDwLayer& L = a.L[li];
  float* P = a.P + net * a.p_ns;
  il_adam opt = a.opt; opt.m += net * a.p_ns; opt.v += net * a.p_ns;
  adam_consts ac = {};
  if (!a.grads_only) ac = load_adam_consts(a.opt);
  const int tile = job - L.tile0, tk = (L.K + 15) >> 4;
  const int n0 = (tile / tk) * 16, k0 = (tile - (tile / tk) * tk) * 16;
  const float* zr = L.dZT + net * L.dz_ns + (size_t)min(n0 + j, L.N - 1) * a.Bp + 4 * g;
  const float* xr = L.XT + net * L.x_ns + (size_t)min(k0 + j, L.K - 1) * a.Bp + 4 * g;
  f32x4 acc0 = zero4(), acc1 = zero4();
  float bs = 0.f;   // the layer's bias gradient rides in the tiles of its first k-column: this lane's rows of dZ[n0 + j], summed in row order
  int r0 = 0;
  for (; r0 + 64 <= a.Bp; r0 += 64) {   // four row groups per trip, their eight lanes requested together (k_g_dw's MFMA order)
    f32x4 z[4], x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { z[u] = gload4(zr + r0 + 16 * u); x[u] = gload4(xr + r0 + 16 * u); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = mfma16(z[u][0], x[u][0], acc0); acc1 = mfma16(z[u][1], x[u][1], acc1); acc0 = mfma16(z[u][2], x[u][2], acc0); acc1 = mfma16(z[u][3], x[u][3], acc1);
      bs += (z[u][0] + z[u][1]) + (z[u][2] + z[u][3]);
    }
  }
  for (; r0 < a.Bp; r0 += 16) {
    const f32x4 z = gload4(zr + r0), x = gload4(xr + r0);
    acc0 = mfma16(z[0], x[0], acc0); acc1 = mfma16(z[1], x[1], acc1); acc0 = mfma16(z[2], x[2], acc0); acc1 = mfma16(z[3], x[3], acc1);
    bs += (z[0] + z[1]) + (z[2] + z[3]);
  }
  if (k0 == 0) {   // (wave-uniform) the four row groups g of feature n0 + j meet in lane j
    bs += __shfl_xor(bs, 16, 64);
    bs += __shfl_xor(bs, 32, 64);
    if (g == 0 && n0 + j < L.N) {
      const int64_t o = L.ob + n0 + j;
      if (a.G) a.G[net * a.g_ns + o] = bs;   // (the gradient arena is part of the entry points' contract: behavioural cloning's callers read it back)
      if (!a.grads_only) { float pp = P[o], mm = opt.m[o], vv = opt.v[o]; adam_update(pp, bs, mm, vv, ac); P[o] = pp; opt.m[o] = mm; opt.v[o] = vv; }
    }
  }
  const f32x4 acc = acc0 + acc1;
  const int k = k0 + j;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + 4 * g + r;
    if (n >= L.N || k >= L.K) continue;
    const int64_t o = L.oW + (int64_t)n * L.K + k;
    if (a.G) a.G[net * a.g_ns + o] = acc[r];
    if (a.grads_only) continue;
    float pp = P[o], mm = opt.m[o], vv = opt.v[o];
    adam_update(pp, acc[r], mm, vv, ac);
    P[o] = pp; opt.m[o] = mm; opt.v[o] = vv;
    if (L.PF) { L.PF[net * a.pk_ns + packed_fwd_index(n, k, L.K)] = pp; L.PB[net * a.pk_ns + packed_bwd_index(n, k, L.K)] = pp; }
  }
}

// the same launch as 32 x 32 BLOCK jobs (dw_block.hpp dw_block32: operands staged through LDS with whole-line loads, the layer's bias in the blocks of its first k-column, AdamW and
// the lane-ordered copies in the epilogue) - the form the single learner's k_dw_adam runs; needs Bp % 128 == 0. k_gt_dw's wave-per-tile gathers are texture-address bound (17.6 us
// per launch at depth 3 / 256 against ~6 here).
__global__ __launch_bounds__(256) void k_gt_dw32(GtDw a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * DWS * DWS_LD];
  if ((int)blockIdx.x >= a.n_job_wgs) { gt_dw_tail(a); return; }
  int job = blockIdx.x;
  const int net = job / a.blocks_per_net; job -= net * a.blocks_per_net;
  int li = 0;
  for (int i = 1; i < a.n_layers; ++i) if (job >= a.L[i].blk0) li = i;
  const GtDwLayer& L = a.L[li];
  const int b = job - L.blk0, nbk = (L.K + DWS - 1) / DWS;
  DwArgs d = {};
  d.params = a.P; d.grads = a.G; d.opt = a.opt; d.grads_only = a.grads_only; d.batch = a.Bp;
  const int64_t base = net * a.p_ns;
  dw_block32<false>(d, L.dZT + net * L.dz_ns, L.N, L.XT + net * L.x_ns, L.K, (b / nbk) * DWS, (b % nbk) * DWS, base + L.oW, base + L.ob, L.PF ? L.PF + net * a.pk_ns : nullptr,
                    L.PB ? L.PB + net * a.pk_ns : nullptr, smem);
}
// lane-ordered copies of the H x H layers: blockIdx.y = slot; slot s copies W (src[s]) into PF[s] / PB[s] (PB NULL: forward copy only)
struct GtRepack { const float* W[16]; float* PF[16]; float* PB[16]; int H[16]; };
__global__ __launch_bounds__(256) void k_gt_repack(GtRepack a) {
  const int s = blockIdx.y, H = a.H[s], kq = H / 4;
  const float* W2 = a.W[s]; float* pf = a.PF[s]; float* pb = a.PB[s];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kq * kq; i += gridDim.x * blockDim.x) {   // one thread = a 4 x 4 block (k_repack)
    const int n = (i / kq) * 4, k = (i - (i / kq) * kq) * 4;
    f32x4 w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(W2 + (size_t)(n + r) * H + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(pf + packed_fwd_index(n + r, k, H)) = w[r];
    if (pb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { f32x4 c; c[0] = w[0][q]; c[1] = w[1][q]; c[2] = w[2][q]; c[3] = w[3][q]; *reinterpret_cast<f32x4*>(pb + packed_bwd_index(n, k + q, H)) = c; }
    }
  }
}

static bool gt_env() { static const int on = [] { const char* e = getenv("IL_GENERAL_TILES"); return e && e[0] == '0' ? 0 : 1; }(); return on != 0; }
static bool gt_shape_ok(const GNet& s) {
  const auto al = [](int64_t v) { return (v & 3) == 0; };
  (void)al;
  return s.H % 16 == 0 && s.H >= 16 && s.H <= 512 && s.in <= 512 && s.out <= 16 && s.depth >= 1 && s.depth <= 8;
}
static size_t gt_fwd_lds(const GNet& s) { return sizeof(float) * ((size_t)16 * (round_up16(s.in) + 4) + 2 * (size_t)16 * (s.H + 4) + (size_t)(s.H >> 4) * 256 + 256); }
static size_t gt_bwd_lds(const GNet& s) { return sizeof(float) * ((size_t)16 * 20 + 2 * (size_t)16 * (s.H + 4)); }
// packed copies need 16-byte aligned H x H layers (their offset in the flat vector is a multiple of 4 floats) - otherwise the passes read W directly
static bool gt_packable(const GNet& s, const float* P) {
  if (s.depth < 2 || s.H % 64 != 0) return false;   // (tile_packed walks a panel four 16-wide k-blocks at a time)
  for (int l = 1; l < s.depth; ++l) { const GLayer L = g_layer(s.in, s.H, s.depth, s.out, l); if ((L.oW & 3) != 0 || (reinterpret_cast<uintptr_t>(P) & 15) != 0) return false; }
  return true;
}
// fills the layer table of a k_gt_dw launch; returns the wave-jobs per network
static int gt_dw_layers(GtDw& w, const GNet& s, const float* X0T, int64_t x_ns, const float* HT, int64_t h_ns, const float* dZT, int64_t dz_ns, const float* dOT, int64_t do_ns, float* PF, float* PB, int Bp) {
  int job = 0, blk = 0;
  w.n_layers = s.depth + 1;
  for (int l = 0; l <= s.depth; ++l) {
    const GLayer L = g_layer(s.in, s.H, s.depth, s.out, l);
    GtDwLayer& d = w.L[l];
    d.dZT = l == s.depth ? dOT : dZT + (int64_t)l * s.H * Bp; d.dz_ns = l == s.depth ? do_ns : dz_ns;
    d.XT = l == 0 ? X0T : HT + (int64_t)(l - 1) * s.H * Bp; d.x_ns = l == 0 ? x_ns : h_ns;
    d.oW = L.oW; d.ob = L.ob; d.N = L.N; d.K = L.K;
    d.tile0 = job; job += ((L.N + 15) / 16) * ((L.K + 15) / 16);
    d.bias0 = job;   // (no bias jobs: the tiles of a layer's first k-column carry its bias gradient)
    d.blk0 = blk; blk += ((L.N + DWS - 1) / DWS) * ((L.K + DWS - 1) / DWS);
    const bool hh = l >= 1 && l < s.depth && PF;
    d.PF = hh ? PF + (int64_t)(l - 1) * s.H * s.H : nullptr; d.PB = hh ? PB + (int64_t)(l - 1) * s.H * s.H : nullptr;
  }
  w.blocks_per_net = blk;
  return job;
}

extern "C" int64_t il_mlp_numel_general(int32_t in_dim, int32_t hidden, int32_t depth, int32_t out_dim) { return g_numel(in_dim, hidden, depth, out_dim); }
extern "C" int64_t il_mlp_stride_general(int32_t in_dim, int32_t hidden, int32_t depth, int32_t out_dim) { return g_stride(in_dim, hidden, depth, out_dim); }

// workspace of il_sac_update_general (floats)
struct GSacWs { int64_t xa2, xa, xt, xc, xp, ha2, oa2, ha, oa, ht, qt, hc, qc, hp, qp, dz, dq, dout, dx0, logp2, logp, xpre, epsu, arows, ga, gc, a2r, anr, pk_af, pk_ab, pk_cf, pk_cb, pk_tf, total; };   // a2r / anr: a' and a~ row-major [B][A]; pk_*: lane-ordered copies of the H x H layers (tile engine)
static GSacWs g_sac_ws(int S, int A, int Ha, int da, int Hc, int dc, int B) {   // actor: da hidden layers of Ha units; critics: dc of Hc
  const int Bp = g_bp(B), IN = S + A;
  const int64_t hid = (int64_t)da * Ha * Bp, hidc = (int64_t)dc * Hc * Bp, hidm = hid > 2 * hidc ? hid : 2 * hidc;
  GSacWs w; int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) & ~(int64_t)3; return at; };
  w.xa2 = take((int64_t)S * Bp); w.xa = take((int64_t)S * Bp); w.xt = take((int64_t)IN * Bp); w.xc = take((int64_t)IN * Bp); w.xp = take((int64_t)IN * Bp);
  w.ha2 = take(hid); w.oa2 = take((int64_t)2 * A * Bp); w.ha = take(hid); w.oa = take((int64_t)2 * A * Bp);
  w.ht = take(2 * hidc); w.qt = take(2 * Bp); w.hc = take(2 * hidc); w.qc = take(2 * Bp); w.hp = take(2 * hidc); w.qp = take(2 * Bp);
  w.dz = take(hidm); w.dq = take(2 * Bp); w.dout = take((int64_t)2 * A * Bp); w.dx0 = take((int64_t)2 * IN * Bp);
  w.logp2 = take(Bp); w.logp = take(Bp); w.xpre = take((int64_t)A * Bp); w.epsu = take((int64_t)A * Bp); w.arows = take(Bp);
  w.ga = take(g_numel(S, Ha, da, 2 * A)); w.gc = take(2 * g_stride(IN, Hc, dc, 1));
  w.a2r = take((int64_t)B * A); w.anr = take((int64_t)B * A);
  const int64_t pa = (int64_t)(da > 1 ? da - 1 : 0) * Ha * Ha, pc = (int64_t)(dc > 1 ? dc - 1 : 0) * Hc * Hc;
  w.pk_af = take(pa); w.pk_ab = take(pa); w.pk_cf = take(2 * pc); w.pk_cb = take(2 * pc); w.pk_tf = take(2 * pc);
  w.total = o;
  return w;
}
extern "C" int64_t il_sac_workspace_floats_general(int32_t S, int32_t A, int32_t Ha, int32_t da, int32_t Hc, int32_t dc, int32_t B) { return g_sac_ws(S, A, Ha, da, Hc, dc, B).total; }

// training.py:14-54 through the tile engine (9 launches; see the engine's header). Same arguments, workspace and results as the layer-at-a-time sequence below.
static int g_sac_update_tiles(const il_sac* d, const il_batch* b, const GNet& an, const GNet& cn, const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, bool grads_only,
                              hipStream_t st) {
  const int S = d->state_dim, A = d->action_dim, B = d->batch, IN = S + A, Bp = g_bp(B), nt = Bp / 16;
  const GSacWs ws = g_sac_ws(S, A, an.H, an.depth, cn.H, cn.depth, B);
  float* W = d->workspace;
  const int64_t Pa = g_numel(S, an.H, an.depth, 2 * A), Pc = g_numel(IN, cn.H, cn.depth, 1), Ps = g_stride(IN, cn.H, cn.depth, 1);
  const int64_t hid_a = g_hidden_floats(an, Bp), hid_c = g_hidden_floats(cn, Bp), pkc = (int64_t)(cn.depth > 1 ? cn.depth - 1 : 0) * cn.H * cn.H;
  const bool pack_a = gt_packable(an, d->actor), pack_c = gt_packable(cn, d->critic) && gt_packable(cn, d->target) && (Ps & 3) == 0;
  float* ga = grads_only ? d->actor_grad : W + ws.ga; float* gc = grads_only ? d->critic_grad : W + ws.gc;
  const size_t lds_f = gt_fwd_lds(an) > gt_fwd_lds(cn) ? gt_fwd_lds(an) : gt_fwd_lds(cn), lds_b = gt_bwd_lds(an) > gt_bwd_lds(cn) ? gt_bwd_lds(an) : gt_bwd_lds(cn);
  if (int rc = g_lds_ok((const void*)k_gt_fwd, lds_f)) return rc;
  if (int rc = g_lds_ok((const void*)k_gt_bwd, lds_b)) return rc;
  const int th_a = gt_threads(an.H), th_c = gt_threads(cn.H), th_m = th_a > th_c ? th_a : th_c;
  // 0. lane-ordered copies of every H x H layer (the critics' are kept in step by their optimiser launch below: the policy pass reads the stepped critics)
  {
    GtRepack r = {}; int n = 0;
    auto add = [&](const float* P, const GNet& s, float* pf, float* pb) {
      for (int l = 1; l < s.depth; ++l) { const GLayer L = g_layer(s.in, s.H, s.depth, s.out, l); r.W[n] = P + L.oW; r.PF[n] = pf + (int64_t)(l - 1) * s.H * s.H; r.PB[n] = pb ? pb + (int64_t)(l - 1) * s.H * s.H : nullptr; r.H[n] = s.H; ++n; }
    };
    if (pack_a && an.depth - 1 <= 3) add(d->actor, an, W + ws.pk_af, W + ws.pk_ab);
    if (pack_c && 4 * (cn.depth - 1) + n <= 16) for (int k = 0; k < 2; ++k) { add(d->critic + k * Ps, cn, W + ws.pk_cf + k * pkc, W + ws.pk_cb + k * pkc); add(d->target + k * Ps, cn, W + ws.pk_tf + k * pkc, nullptr); }
    if (n > 0) { int hm = 0; for (int i = 0; i < n; ++i) hm = r.H[i] > hm ? r.H[i] : hm; IL_TRACE("k_gt_repack", st); k_gt_repack<<<dim3((hm * hm / 16 + 255) / 256, n), 256, 0, st>>>(r); }
  }
  const bool pa = pack_a && an.depth - 1 <= 3, pc = pack_c && 4 * (cn.depth - 1) + (pa ? an.depth - 1 : 0) <= 16;
  const GtNet actor = {d->actor, S, an.H, an.depth, 2 * A, an.act, pa ? W + ws.pk_af : nullptr, pa ? W + ws.pk_ab : nullptr};
  auto critic = [&](int k) { GtNet c = {d->critic + k * Ps, IN, cn.H, cn.depth, 1, cn.act, pc ? W + ws.pk_cf + k * pkc : nullptr, pc ? W + ws.pk_cb + k * pkc : nullptr}; return c; };
  auto target = [&](int k) { GtNet c = {d->target + k * Ps, IN, cn.H, cn.depth, 1, cn.act, pc ? W + ws.pk_tf + k * pkc : nullptr, nullptr}; return c; };
  // A. actor(s') -> a', log pi(a'|s'); actor(s) -> a~, log pi, pre-tanh sample, noise; critic_1,2(s, a)
  {
    GtFwd f = {}; f.n = B; f.Bp = Bp;
    GtFwdPass& p0 = f.p[0]; p0.net = actor; p0.f1 = b->next_states; p0.ld1 = b->ld_next_states; p0.K1 = S; p0.head = 1; p0.eps = eps_next; p0.seed = d->noise_seed; p0.ctr_ptr = d->noise_counter;
    p0.stream_id = IL_STREAM_EPS_NEXT; p0.absorbing = b->absorbing; p0.ld_abs = b->ld_absorbing; p0.a_rows = W + ws.a2r; p0.ld_a = A; p0.logp = W + ws.logp2;
    GtFwdPass& p1 = f.p[1]; p1.net = actor; p1.f1 = b->states; p1.ld1 = b->ld_states; p1.K1 = S; p1.X0T = W + ws.xa; p1.HT = W + ws.ha; p1.OT = W + ws.oa; p1.head = 1; p1.eps = eps_cur; p1.seed = d->noise_seed;
    p1.ctr_ptr = d->noise_counter; p1.stream_id = IL_STREAM_EPS_CUR; p1.a_rows = W + ws.anr; p1.ld_a = A; p1.xT = W + ws.xpre; p1.epsT = W + ws.epsu; p1.logp = W + ws.logp;
    for (int k = 0; k < 2; ++k) {
      GtFwdPass& pk = f.p[2 + k]; pk.net = critic(k); pk.f1 = b->states; pk.ld1 = b->ld_states; pk.K1 = S; pk.f2 = b->actions; pk.ld2 = b->ld_actions; pk.K2 = A;
      pk.X0T = k == 0 ? W + ws.xc : nullptr; pk.HT = W + ws.hc + k * hid_c; pk.OT = W + ws.qc + (int64_t)k * Bp;
    }
    f.npass = 4; IL_TRACE("k_gt_fwd", st); k_gt_fwd<<<nt * 4, th_m, lds_f, st>>>(f);
  }
  // B. target_1,2(s', a')
  {
    GtFwd f = {}; f.n = B; f.Bp = Bp;
    for (int k = 0; k < 2; ++k) {
      GtFwdPass& pk = f.p[k]; pk.net = target(k); pk.f1 = b->next_states; pk.ld1 = b->ld_next_states; pk.K1 = S; pk.f2 = W + ws.a2r; pk.ld2 = A; pk.K2 = A; pk.OT = W + ws.qt + (int64_t)k * Bp;
    }
    f.npass = 2; IL_TRACE("k_gt_fwd", st); k_gt_fwd<<<nt * 2, th_c, lds_f, st>>>(f);
  }
  // C. critic loss seed + backward (training.py:22-31), D. every layer's dW + AdamW
  {
    GtBwd g = {}; g.net = critic(0); g.p_ns = Ps; g.pk_ns = pkc; g.HT = W + ws.hc; g.h_ns = hid_c; g.dZT = W + ws.dz; g.dz_ns = hid_c; g.dOT = W + ws.dq; g.do_ns = Bp; g.seed = GT_SEED_CRITIC; g.n = B; g.Bp = Bp;
    g.b = *b; g.qtT = W + ws.qt; g.qT = W + ws.qc; g.logp2 = W + ws.logp2; g.log_alpha = d->log_alpha; g.discount = d->discount; g.out_q = out_q ? out_q : d->out_q;
    if (!grads_only) g.tick = d->critic_opt;
    g.nets = 2; IL_TRACE("k_gt_bwd", st); k_gt_bwd<<<nt * 2, th_c, lds_b, st>>>(g);
  }
  {
    GtDw w = {}; w.nets = 2; w.p_ns = Ps; w.pk_ns = pkc; w.P = d->critic; w.G = gc; w.g_ns = Ps; w.opt = d->critic_opt; w.grads_only = grads_only ? 1 : 0; w.Bp = Bp;
    w.jobs_per_net = gt_dw_layers(w, cn, W + ws.xc, 0, W + ws.hc, hid_c, W + ws.dz, hid_c, W + ws.dq, Bp, pc ? W + ws.pk_cf : nullptr, pc ? W + ws.pk_cb : nullptr, Bp);
    if (Bp % DWS_ROWS == 0) { w.n_job_wgs = w.blocks_per_net * 2; IL_TRACE("k_gt_dw32", st); k_gt_dw32<<<w.n_job_wgs, 256, 0, st>>>(w); }
    else { w.n_job_wgs = (w.jobs_per_net * 2 + 3) / 4; IL_TRACE("k_gt_dw", st); k_gt_dw<<<w.n_job_wgs, 256, 0, st>>>(w); }
  }
  // E. the stepped critics on (s, a~), F. policy seed + backward down to dQ/da~ (training.py:34-38)
  {
    GtFwd f = {}; f.n = B; f.Bp = Bp;
    for (int k = 0; k < 2; ++k) {
      GtFwdPass& pk = f.p[k]; pk.net = critic(k); pk.f1 = b->states; pk.ld1 = b->ld_states; pk.K1 = S; pk.f2 = W + ws.anr; pk.ld2 = A; pk.K2 = A; pk.HT = W + ws.hp + k * hid_c; pk.OT = W + ws.qp + (int64_t)k * Bp;
    }
    f.npass = 2; IL_TRACE("k_gt_fwd", st); k_gt_fwd<<<nt * 2, th_c, lds_f, st>>>(f);
  }
  {
    GtBwd g = {}; g.net = critic(0); g.p_ns = Ps; g.pk_ns = pkc; g.HT = W + ws.hp; g.h_ns = hid_c; g.dZT = W + ws.dz; g.dz_ns = hid_c; g.seed = GT_SEED_POLICY; g.n = B; g.Bp = Bp; g.qT = W + ws.qp;
    g.dX0T = W + ws.dx0; g.dx_ns = (int64_t)IN * Bp; g.dx_c0 = S; g.dx_c1 = IN;
    g.nets = 2; IL_TRACE("k_gt_bwd", st); k_gt_bwd<<<nt * 2, th_c, lds_b, st>>>(g);
  }
  // G. back through the tanh-Gaussian head and the actor (training.py:35-42), H. the actor's dW + AdamW, temperature step, target update (training.py:45-52)
  {
    GtBwd g = {}; g.net = actor; g.HT = W + ws.ha; g.dZT = W + ws.dz; g.dOT = W + ws.dout; g.seed = GT_SEED_HEAD; g.n = B; g.Bp = Bp; g.b = *b;
    g.outT = W + ws.oa; g.xT = W + ws.xpre; g.epsT = W + ws.epsu; g.logp = W + ws.logp; g.dx0T = W + ws.dx0; g.dx0_ns = (int64_t)IN * Bp; g.log_alpha = d->log_alpha; g.entropy_target = d->entropy_target; g.S = S;
    g.alpha_rows = W + ws.arows; g.out_logp = out_logp ? out_logp : d->out_logp;
    if (!grads_only) g.tick = d->actor_opt;
    g.nets = 1; IL_TRACE("k_gt_bwd", st); k_gt_bwd<<<nt, th_a, lds_b, st>>>(g);
  }
  {
    GtDw w = {}; w.nets = 1; w.P = d->actor; w.G = ga; w.opt = d->actor_opt; w.grads_only = grads_only ? 1 : 0; w.Bp = Bp;
    w.jobs_per_net = gt_dw_layers(w, an, W + ws.xa, 0, W + ws.ha, hid_a, W + ws.dz, hid_a, W + ws.dout, 0, nullptr, nullptr, Bp);   // (the actor's copies are re-derived at the next update's start)
    const bool blocks = Bp % DWS_ROWS == 0;
    w.n_job_wgs = blocks ? w.blocks_per_net : (w.jobs_per_net + 3) / 4;
    w.alpha_rows = W + ws.arows; w.n_rows = B; w.log_alpha = d->log_alpha; w.alpha_opt = d->alpha_opt; w.alpha_grad = d->alpha_grad; w.noise_counter = d->noise_counter;
    w.target = d->target; w.polyak_src = d->critic; w.polyak_n = Ps + Pc; w.tau = d->polyak;
    const int tail = grads_only ? 1 : 1 + (int)(((Ps + Pc) / 256 + 7) / 8 < 64 ? ((Ps + Pc) / 256 + 7) / 8 : 64);
    if (blocks) { IL_TRACE("k_gt_dw32", st); k_gt_dw32<<<w.n_job_wgs + tail, 256, 0, st>>>(w); }
    else { IL_TRACE("k_gt_dw", st); k_gt_dw<<<w.n_job_wgs + tail, 256, 0, st>>>(w); }
  }
  (void)Pa;
  IL_CHECK_LAUNCH("il_sac_update_general (tile engine)");
  return IL_OK;
}

// training.py:14-54 for general shapes. The descriptor is the fused path's (`hidden` = the ACTOR's hidden width; parameter arenas in torch order, twin critics at
// il_mlp_stride_general); the actor and the critics have their own (hidden, depth, activation), as reinforcement.actor / reinforcement.critic do in the reference's configuration.
// workspace >= il_sac_workspace_floats_general. eps_next / eps_cur [B][A] or NULL (Philox, the fused path's streams and counter).
extern "C" int il_sac_update_general(const il_sac* d, const il_batch* b, int32_t actor_depth, int32_t actor_activation, int32_t critic_hidden, int32_t critic_depth, int32_t critic_activation,
                                     const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream_) {
  IL_CHECK_ARG(d && b && d->actor && d->critic && d->target && d->log_alpha && d->workspace, "il_sac_update_general: null descriptor field");
  IL_NO_GATHER(b, "il_sac_update_general");
  IL_CHECK_ARG(!(flags & ~(uint32_t)IL_FLAG_GRADS_ONLY), "il_sac_update_general: only IL_FLAG_GRADS_ONLY is understood");
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, IN = S + A, Bp = g_bp(B);
  IL_CHECK_ARG(b->n == B && B >= 1, "il_sac_update_general: batch has %d rows, descriptor %d", b->n, B);
  const GNet an = {S, H, actor_depth, 2 * A, actor_activation}, cn = {IN, critic_hidden, critic_depth, 1, critic_activation};
  if (int rc = g_check_shape(an, "il_sac_update_general (actor)")) return rc;
  if (int rc = g_check_shape(cn, "il_sac_update_general (critic)")) return rc;
  const GSacWs ws = g_sac_ws(S, A, H, actor_depth, critic_hidden, critic_depth, B);
  if (d->workspace_floats < ws.total) return il_set_error(IL_ERR_WORKSPACE, "il_sac_update_general: workspace has %lld floats, needs %lld", (long long)d->workspace_floats, (long long)ws.total);
  const bool grads_only = (flags & IL_FLAG_GRADS_ONLY) != 0;
  IL_CHECK_ARG(!grads_only || (d->actor_grad && d->critic_grad && d->alpha_grad), "il_sac_update_general: IL_FLAG_GRADS_ONLY needs the gradient arenas");
  hipStream_t st = (hipStream_t)stream_;
  if (gt_env() && gt_shape_ok(an) && gt_shape_ok(cn)) return g_sac_update_tiles(d, b, an, cn, eps_next, eps_cur, out_logp, out_q, grads_only, st);
  float* W = d->workspace;
  const int64_t Pa = g_numel(S, H, actor_depth, 2 * A), Pc = g_numel(IN, critic_hidden, critic_depth, 1), Ps = g_stride(IN, critic_hidden, critic_depth, 1);
  float* ga = grads_only ? d->actor_grad : W + ws.ga; float* gc = grads_only ? d->critic_grad : W + ws.gc;
  const int rb = (Bp + 255) / 256;
  // inputs, feature-major
  {   // (the action rows of xt / xp are written by the head kernels below: a', a~)
    GPackSac pk = {*b, S, A, Bp, W + ws.xa2, W + ws.xa, W + ws.xt, W + ws.xc, W + ws.xp};
    const int total = IN * Bp;
    IL_TRACE("k_g_pack", st); k_g_pack_sac<<<dim3((total + 255) / 256 < 256 ? (total + 255) / 256 : 256, 5), 256, 0, st>>>(pk);
  }
  // target values (training.py:19-25)
  if (int rc = g_forward(st, an, d->actor, 0, 1, W + ws.xa2, 0, W + ws.ha2, W + ws.oa2, Bp)) return rc;
  {
    GSample h = {}; h.outT = W + ws.oa2; h.Bp = Bp; h.n = B; h.A = A; h.eps = eps_next; h.seed = d->noise_seed; h.ctr_ptr = d->noise_counter; h.stream_id = IL_STREAM_EPS_NEXT;
    h.absorbing = b->absorbing; h.ld_abs = b->ld_absorbing; h.aT = W + ws.xt + (int64_t)S * Bp; h.logp = W + ws.logp2;
    IL_TRACE("k_g_sample", st); k_g_sample<<<rb, 256, 0, st>>>(h);
  }
  if (int rc = g_forward(st, cn, d->target, Ps, 2, W + ws.xt, 0, W + ws.ht, W + ws.qt, Bp)) return rc;
  // critic loss, backward, AdamW (training.py:26-31)
  if (int rc = g_forward(st, cn, d->critic, Ps, 2, W + ws.xc, 0, W + ws.hc, W + ws.qc, Bp)) return rc;
  { IL_TRACE("k_g_critic_seed", st); k_g_critic_seed<<<rb, 256, 0, st>>>(*b, W + ws.qt, W + ws.qc, W + ws.logp2, d->log_alpha, d->discount, Bp, W + ws.dq, out_q ? out_q : d->out_q); }
  if (int rc = g_backward(st, cn, d->critic, Ps, 2, W + ws.xc, 0, W + ws.hc, W + ws.dq, W + ws.dz, nullptr, gc, Ps, Bp)) return rc;
  if (!grads_only) {
    if (int rc = il_adam_step(d->critic, gc, &d->critic_opt, Pc, IL_FLAG_TICK, stream_)) return rc;
    il_adam o2 = d->critic_opt; o2.m += Ps; o2.v += Ps;   // the second critic's slice of the arena (one tick per step: above)
    if (int rc = il_adam_step(d->critic + Ps, gc + Ps, &o2, Pc, 0, stream_)) return rc;
  }
  // policy loss through the updated critic (training.py:34-42)
  if (int rc = g_forward(st, an, d->actor, 0, 1, W + ws.xa, 0, W + ws.ha, W + ws.oa, Bp)) return rc;
  {
    GSample h = {}; h.outT = W + ws.oa; h.Bp = Bp; h.n = B; h.A = A; h.eps = eps_cur; h.seed = d->noise_seed; h.ctr_ptr = d->noise_counter; h.stream_id = IL_STREAM_EPS_CUR;
    h.aT = W + ws.xp + (int64_t)S * Bp; h.xT = W + ws.xpre; h.epsT = W + ws.epsu; h.logp = W + ws.logp;
    IL_TRACE("k_g_sample", st); k_g_sample<<<rb, 256, 0, st>>>(h);
  }
  if (int rc = g_forward(st, cn, d->critic, Ps, 2, W + ws.xp, 0, W + ws.hp, W + ws.qp, Bp)) return rc;
  { IL_TRACE("k_g_policy_seed", st); k_g_policy_seed<<<rb, 256, 0, st>>>(W + ws.qp, B, Bp, W + ws.dq); }
  if (int rc = g_backward(st, cn, d->critic, Ps, 2, W + ws.xp, 0, W + ws.hp, W + ws.dq, W + ws.dz, W + ws.dx0, nullptr, 0, Bp)) return rc;
  {
    GHeadBwd h = {}; h.b = *b; h.outT = W + ws.oa; h.xT = W + ws.xpre; h.epsT = W + ws.epsu; h.logp = W + ws.logp; h.dx0T = W + ws.dx0; h.dx_ns = (int64_t)IN * Bp; h.log_alpha = d->log_alpha;
    h.entropy_target = d->entropy_target; h.S = S; h.A = A; h.Bp = Bp; h.doutT = W + ws.dout; h.alpha_rows = W + ws.arows; h.out_logp = out_logp ? out_logp : d->out_logp;
    IL_TRACE("k_g_head_bwd", st); k_g_head_bwd<<<rb, 256, 0, st>>>(h);
  }
  if (int rc = g_backward(st, an, d->actor, 0, 1, W + ws.xa, 0, W + ws.ha, W + ws.dout, W + ws.dz, nullptr, ga, 0, Bp)) return rc;
  if (!grads_only) { if (int rc = il_adam_step(d->actor, ga, &d->actor_opt, Pa, IL_FLAG_TICK, stream_)) return rc; }
  // temperature, target network (training.py:45-52)
  { IL_TRACE("k_g_alpha", st); k_g_alpha<<<1, 64, 0, st>>>(W + ws.arows, B, d->log_alpha, d->alpha_opt, d->alpha_grad, grads_only ? 1 : 0, d->noise_counter); }
  if (!grads_only) {
    if (int rc = il_polyak(d->target, d->critic, Ps + Pc, d->polyak, stream_)) return rc;   // both networks in one launch (the <= 3 pad floats between them are never parameters)
  }
  IL_CHECK_LAUNCH("il_sac_update_general");
  return IL_OK;
}

// workspace of the actor-only entry points below: input, hidden activations, head outputs, and (behavioural cloning) dZ, head gradient, loss rows, gradient arena
struct GActWs { int64_t x, h, o, dz, dout, rows, g, total; };
static GActWs g_act_ws(int S, int A, int H, int depth, int Bp) {
  GActWs w; int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) & ~(int64_t)3; return at; };
  w.x = take((int64_t)S * Bp); w.h = take((int64_t)depth * H * Bp); w.o = take((int64_t)2 * A * Bp); w.dz = take((int64_t)depth * H * Bp); w.dout = take((int64_t)2 * A * Bp); w.rows = take(Bp);
  w.g = take(g_numel(S, H, depth, 2 * A));
  w.total = o;
  return w;
}
extern "C" int64_t il_actor_workspace_floats_general(int32_t S, int32_t A, int32_t H, int32_t depth, int32_t n) {   // the layout's own total: a layout change cannot outgrow the size check
  return g_act_ws(S, A, H, depth, g_bp(n)).total;
}
// train.py:152 `actor(state).sample()` / models.py:101-102 get_greedy_action for general shapes (arguments of il_actor_act + depth, activation, workspace)
extern "C" int il_actor_act_general(const float* actor, int32_t S, int32_t A, int32_t H, int32_t depth, int32_t activation, const float* states, int32_t ld_states, int32_t n, const float* eps,
                                    uint64_t noise_seed, uint32_t noise_offset, int32_t greedy, float* out_action, float* out_logp, float* workspace, int64_t workspace_floats, il_stream_t stream_) {
  IL_CHECK_ARG(actor && states && out_action && workspace && n > 0, "il_actor_act_general: null argument");
  const GNet an = {S, H, depth, 2 * A, activation};
  if (int rc = g_check_shape(an, "il_actor_act_general")) return rc;
  IL_CHECK_ARG(workspace_floats >= il_actor_workspace_floats_general(S, A, H, depth, n), "il_actor_act_general: workspace too small");
  hipStream_t st = (hipStream_t)stream_;
  const int Bp = g_bp(n);
  const GActWs ws = g_act_ws(S, A, H, depth, Bp);
  if (gt_env() && gt_shape_ok(an)) {   // tile engine: one launch (rows -> every layer -> head)
    const size_t lds = gt_fwd_lds(an);
    if (int rc = g_lds_ok((const void*)k_gt_fwd, lds)) return rc;
    GtFwd f = {}; f.n = n; f.Bp = Bp;
    GtFwdPass& p0 = f.p[0]; p0.net = GtNet{actor, S, H, depth, 2 * A, activation, nullptr, nullptr}; p0.f1 = states; p0.ld1 = ld_states; p0.K1 = S; p0.head = 1; p0.eps = eps; p0.seed = noise_seed; p0.ctr = noise_offset;
    p0.stream_id = IL_STREAM_ACT; p0.greedy = greedy; p0.a_rows = out_action; p0.ld_a = A; p0.logp = out_logp;
    f.npass = 1; { IL_TRACE("k_gt_fwd", st); k_gt_fwd<<<Bp / 16, gt_threads(H), lds, st>>>(f); }
    IL_CHECK_LAUNCH("il_actor_act_general (tile engine)");
    return IL_OK;
  }
  g_pack(st, states, ld_states, S, nullptr, 0, 0, n, Bp, workspace + ws.x);
  if (int rc = g_forward(st, an, actor, 0, 1, workspace + ws.x, 0, workspace + ws.h, workspace + ws.o, Bp)) return rc;
  GSample h = {}; h.outT = workspace + ws.o; h.Bp = Bp; h.n = n; h.A = A; h.eps = eps; h.seed = noise_seed; h.ctr = noise_offset; h.stream_id = IL_STREAM_ACT; h.a_rows = out_action; h.ld_a = A;
  h.logp = out_logp; h.greedy = greedy;
  { IL_TRACE("k_g_sample", st); k_g_sample<<<(Bp + 255) / 256, 256, 0, st>>>(h); }
  IL_CHECK_LAUNCH("il_actor_act_general");
  return IL_OK;
}
// models.py:97-99 SoftActor.log_prob for general shapes
extern "C" int il_actor_log_prob_general(const float* actor, int32_t S, int32_t A, int32_t H, int32_t depth, int32_t activation, const float* states, int32_t ld_states, const float* actions,
                                         int32_t ld_actions, int32_t n, float* out_logp, float* workspace, int64_t workspace_floats, il_stream_t stream_) {
  IL_CHECK_ARG(actor && states && actions && out_logp && workspace && n > 0, "il_actor_log_prob_general: null argument");
  const GNet an = {S, H, depth, 2 * A, activation};
  if (int rc = g_check_shape(an, "il_actor_log_prob_general")) return rc;
  IL_CHECK_ARG(workspace_floats >= il_actor_workspace_floats_general(S, A, H, depth, n), "il_actor_log_prob_general: workspace too small");
  hipStream_t st = (hipStream_t)stream_;
  const int Bp = g_bp(n);
  const GActWs ws = g_act_ws(S, A, H, depth, Bp);
  g_pack(st, states, ld_states, S, nullptr, 0, 0, n, Bp, workspace + ws.x);
  if (int rc = g_forward(st, an, actor, 0, 1, workspace + ws.x, 0, workspace + ws.h, workspace + ws.o, Bp)) return rc;
  { IL_TRACE("k_g_logp", st); k_g_logp<<<(Bp + 255) / 256, 256, 0, st>>>(workspace + ws.o, Bp, n, A, actions, ld_actions, out_logp, nullptr, 0, nullptr, nullptr); }
  IL_CHECK_LAUNCH("il_actor_log_prob_general");
  return IL_OK;
}
// training.py:57-64 behavioural_cloning_update for general shapes (arguments of il_bc_step + depth, activation); out_loss [1] = mean(w * -log pi) or NULL
extern "C" int il_bc_step_general(float* actor, float* actor_grad, const il_adam* opt, int32_t S, int32_t A, int32_t H, int32_t depth, int32_t activation, const il_batch* b, float* workspace,
                                  int64_t workspace_floats, float* out_loss, uint32_t flags, il_stream_t stream_) {
  IL_CHECK_ARG(actor && b && workspace && b->n > 0 && (opt || (flags & IL_FLAG_GRADS_ONLY)), "il_bc_step_general: null argument");
  IL_NO_GATHER(b, "il_bc_step_general");
  const GNet an = {S, H, depth, 2 * A, activation};
  if (int rc = g_check_shape(an, "il_bc_step_general")) return rc;
  const int n = b->n, Bp = g_bp(n);
  IL_CHECK_ARG(workspace_floats >= il_actor_workspace_floats_general(S, A, H, depth, n), "il_bc_step_general: workspace too small");
  const bool grads_only = (flags & IL_FLAG_GRADS_ONLY) != 0;
  IL_CHECK_ARG(!grads_only || actor_grad, "il_bc_step_general: IL_FLAG_GRADS_ONLY needs actor_grad");
  hipStream_t st = (hipStream_t)stream_;
  const GActWs ws = g_act_ws(S, A, H, depth, Bp);
  float* G = actor_grad ? actor_grad : workspace + ws.g;
  if (gt_env() && gt_shape_ok(an)) {   // tile engine: forward, BC seed + backward, every layer's dW + AdamW: three launches
    const size_t lds_f = gt_fwd_lds(an), lds_b = gt_bwd_lds(an);
    if (int rc = g_lds_ok((const void*)k_gt_fwd, lds_f)) return rc;
    if (int rc = g_lds_ok((const void*)k_gt_bwd, lds_b)) return rc;
    const GtNet net = {actor, S, H, depth, 2 * A, activation, nullptr, nullptr};
    GtFwd f = {}; f.n = n; f.Bp = Bp;
    GtFwdPass& p0 = f.p[0]; p0.net = net; p0.f1 = b->states; p0.ld1 = b->ld_states; p0.K1 = S; p0.X0T = workspace + ws.x; p0.HT = workspace + ws.h; p0.OT = workspace + ws.o;
    f.npass = 1; { IL_TRACE("k_gt_fwd", st); k_gt_fwd<<<Bp / 16, gt_threads(H), lds_f, st>>>(f); }
    GtBwd g = {}; g.net = net; g.HT = workspace + ws.h; g.dZT = workspace + ws.dz; g.dOT = workspace + ws.dout; g.seed = GT_SEED_BC; g.n = n; g.Bp = Bp; g.b = *b; g.outT = workspace + ws.o; g.loss_rows = workspace + ws.rows;
    if (!grads_only) g.tick = *opt;
    g.nets = 1; { IL_TRACE("k_gt_bwd", st); k_gt_bwd<<<Bp / 16, gt_threads(H), lds_b, st>>>(g); }
    if (out_loss) { IL_TRACE("k_g_sum_rows", st); k_g_sum_rows<<<1, 64, 0, st>>>(workspace + ws.rows, n, out_loss); }
    GtDw w = {}; w.nets = 1; w.P = actor; w.G = G; if (opt) w.opt = *opt; w.grads_only = grads_only ? 1 : 0; w.Bp = Bp;
    w.jobs_per_net = gt_dw_layers(w, an, workspace + ws.x, 0, workspace + ws.h, 0, workspace + ws.dz, 0, workspace + ws.dout, 0, nullptr, nullptr, Bp);
    w.n_job_wgs = (w.jobs_per_net + 3) / 4;
    { IL_TRACE("k_gt_dw", st); k_gt_dw<<<w.n_job_wgs, 256, 0, st>>>(w); }
    IL_CHECK_LAUNCH("il_bc_step_general (tile engine)");
    return IL_OK;
  }
  g_pack(st, b->states, b->ld_states, S, nullptr, 0, 0, n, Bp, workspace + ws.x);
  if (int rc = g_forward(st, an, actor, 0, 1, workspace + ws.x, 0, workspace + ws.h, workspace + ws.o, Bp)) return rc;
  { IL_TRACE("k_g_logp", st); k_g_logp<<<(Bp + 255) / 256, 256, 0, st>>>(workspace + ws.o, Bp, n, A, b->actions, b->ld_actions, nullptr, b->weights, b->ld_weights, workspace + ws.dout, workspace + ws.rows); }
  if (out_loss) { IL_TRACE("k_g_sum_rows", st); k_g_sum_rows<<<1, 64, 0, st>>>(workspace + ws.rows, n, out_loss); }
  if (int rc = g_backward(st, an, actor, 0, 1, workspace + ws.x, 0, workspace + ws.h, workspace + ws.dout, workspace + ws.dz, nullptr, G, 0, Bp)) return rc;
  if (!grads_only) { if (int rc = il_adam_step(actor, G, opt, g_numel(S, H, depth, 2 * A), IL_FLAG_TICK, stream_)) return rc; }
  IL_CHECK_LAUNCH("il_bc_step_general");
  return IL_OK;
}
IL_STAMP_READER(il_debug_stamps_general)
