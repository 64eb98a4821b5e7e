// GAIL discriminator with reward shaping (reference models.py:152-180 with reward_shaping=True; training.py:85-134) for gfx950.
//
//   f(s, a, s', t) = g(x) + (1 - t) (discount h(s') - h(s)),  g = Linear(Dg, 1),  h = Linear(S, H) -> ReLU -> Linear(H, 1),  optional spectral norm.
// torch's _SpectralNorm runs one power iteration per ACCESS of a parametrised weight in train mode and `forward` evaluates g(x), h(s'), h(s) in that
// order: one discriminator call advances g's (u, v) once and h's twice, and h(s') / h(s) are normalised by different sigmas. The three calls of an
// update (policy, expert, gradient-penalty mix) are independent workgroups: call c replays c + 1 iterations of g and 2c + 1 / 2c + 2 of h from the
// stored buffers (they only depend on the parameters). Chain rule per use: dW = G^/sigma - <G^, W>/sigma^2 u v^T, linear in G^, so every tile applies
// it to its own partial sums. Gradient penalty: dD/ds = Wg_s^ - (1 - t)(w2^ [pre > 0]) W1^ (second use of h), dD/da = Wg_a^, ReLU mask constant.
//
// This is an optional configuration (the default has reward_shaping = false) with a few-KB network: like RED / DRIL it is written as plain VALU loops
// over LDS-resident tiles (32 rows per workgroup), one gradient slab per (call, tile), a reduce + AdamW kernel, and an eval kernel for the rewards.
#include "il_common.hpp"

#define GS_R 32

struct GsLayout { int64_t oWg, obg, oW1, ob1, oW2, ob2, P; };
__host__ __device__ inline GsLayout gs_layout(int S, int Dg, int H, int sn) {
  GsLayout l; int64_t o = 0;
  if (sn) { l.obg = o; o += 1; l.oWg = o; o += Dg; l.ob1 = o; o += H; l.oW1 = o; o += (int64_t)H * S; l.ob2 = o; o += 1; l.oW2 = o; o += H; }
  else { l.oWg = o; o += Dg; l.obg = o; o += 1; l.oW1 = o; o += (int64_t)H * S; l.ob1 = o; o += H; l.oW2 = o; o += H; l.ob2 = o; o += 1; }
  l.P = o;
  return l;
}
__host__ __device__ inline int gs_calls(const il_disc_shaped& d) { return (d.loss_function == IL_LOSS_MIXUP ? 1 : 2) + (d.grad_penalty > 0.f ? 1 : 0); }   // Mixup: ONE call on the convex combinations (training.py:104-113)
// what a call runs on: 0 policy, 1 expert, 2 gradient-penalty mix (training.py:116-126), 3 Mixup mix
__host__ __device__ inline int gs_kind(const il_disc_shaped& d, int call) { return d.loss_function == IL_LOSS_MIXUP ? (call == 0 ? 3 : 2) : call; }
struct GsWs { int64_t slabs, sn_new, pu, total; };   // pu: [2][nt] per-tile sums of w softplus(z) of the policy / expert call (PUGAIL with a finite nonnegative_margin)
__host__ __device__ inline GsWs gs_ws(int S, int Dg, int H, int B) {
  GsWs w; const int64_t P = gs_layout(S, Dg, H, 1).P, nt = (B + GS_R - 1) / GS_R;
  w.slabs = 0; w.sn_new = (3 * nt * P + 3) & ~(int64_t)3; w.pu = w.sn_new + ((2 + Dg + 2 * H + S + 3) & ~3); w.total = w.pu + ((2 * nt + 3) & ~(int64_t)3);
  return w;
}
extern "C" int64_t il_disc_shaped_numel(int32_t S, int32_t A, int32_t H, int32_t state_only) { return gs_layout(S, state_only ? S : S + A, H, 1).P; }
extern "C" int64_t il_disc_shaped_workspace_floats(int32_t S, int32_t A, int32_t H, int32_t B, int32_t state_only) { return gs_ws(S, state_only ? S : S + A, H, B).total; }

struct GsLds {
  float *Wg, *W1, *b1, *W2, *vg, *u1, *v1, *v2, *u1n, *v1n, *tH, *tS, *sc, *X, *Xn, *pn, *ps, *gin, *row, *red;
  int ldw, ldx, ldn, ldh;
};
// scalars in sc[]: 0 bg, 1 b2, 2 ug, 3 u2, 4 sg, 5 s1n, 6 s2n, 7 s1s, 8 s2s, 9 u2n (u2 at the h(s') use)
__host__ __device__ inline size_t gs_lds_floats(int S, int Dg, int H) {
  return (size_t)Dg + (size_t)H * (S + 1) + 2 * H + Dg + 2 * H + 2 * S + 2 * H + H + S + 16 + (size_t)GS_R * (Dg + 1) + (size_t)GS_R * (S + 1) + 2 * (size_t)GS_R * (H + 1) +
         (size_t)GS_R * (Dg + 1) + 8 * GS_R + 64;
}
__device__ __forceinline__ GsLds gs_carve(float* p, int S, int Dg, int H) {
  GsLds l; l.ldw = S + 1; l.ldx = Dg + 1; l.ldn = S + 1; l.ldh = H + 1;
  l.Wg = p; p += Dg; l.W1 = p; p += H * (S + 1); l.b1 = p; p += H; l.W2 = p; p += H; l.vg = p; p += Dg; l.u1 = p; p += H; l.v2 = p; p += H;
  l.v1 = p; p += S; l.v1n = p; p += S; l.u1n = p; p += H; l.tH = p; p += H; l.tS = p; p += S; l.sc = p; p += 16;
  l.X = p; p += GS_R * (Dg + 1); l.Xn = p; p += GS_R * (S + 1); l.pn = p; p += GS_R * (H + 1); l.ps = p; p += GS_R * (H + 1);
  l.gin = p; p += GS_R * (Dg + 1); l.row = p; p += 8 * GS_R; l.red = p;
  return l;
}

__device__ __forceinline__ float gs_dot(const float* a, const float* b, int n, float* red) {   // block-wide dot product of two LDS vectors
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(a[i], b[i], s);
  return block_sum(s, red);
}
__device__ __forceinline__ void gs_normalize(float* v, int n, float* red) {   // v /= max(||v||, 1e-12) (torch F.normalize), block-wide, in place
  const float nrm = sqrtf(gs_dot(v, v, n, red));
  const float inv = 1.f / fmaxf(nrm, 1e-12f);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) v[i] *= inv;
  __syncthreads();
}
// one power iteration of a [1 x n] weight: u = normalize(W v) (a scalar), v = normalize(W^T u); returns nothing, state in (u_s, v)
__device__ __forceinline__ void gs_iter_row(const float* W, int n, float* u_s, float* v, float* red) {
  const float wv = gs_dot(W, v, n, red);
  const float u = wv / fmaxf(fabsf(wv), 1e-12f);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) v[i] = W[i] * u;
  if (threadIdx.x == 0) *u_s = u;
  __syncthreads();
  gs_normalize(v, n, red);
}
// one power iteration of W1 [H x S] (LDS, row stride ldw): u = normalize(W v), v = normalize(W^T u)
__device__ __forceinline__ void gs_iter_mat(const GsLds& l, int S, int H, float* u, float* v) {
  for (int j = threadIdx.x; j < H; j += blockDim.x) { float s = 0.f; for (int k = 0; k < S; ++k) s = fmaf(l.W1[j * l.ldw + k], v[k], s); u[j] = s; }
  __syncthreads();
  gs_normalize(u, H, l.red);
  for (int k = threadIdx.x; k < S; k += blockDim.x) { float s = 0.f; for (int j = 0; j < H; ++j) s = fmaf(l.W1[j * l.ldw + k], u[j], s); v[k] = s; }
  __syncthreads();
  gs_normalize(v, S, l.red);
}
__device__ __forceinline__ float gs_sigma_mat(const GsLds& l, int S, int H, const float* u, const float* v) {   // u^T W1 v
  float s = 0.f;
  for (int j = threadIdx.x; j < H; j += blockDim.x) { float t = 0.f; for (int k = 0; k < S; ++k) t = fmaf(l.W1[j * l.ldw + k], v[k], t); s = fmaf(u[j], t, s); }
  return block_sum(s, l.red);
}

__device__ __forceinline__ void gs_stage_params(const GsLds& l, const il_disc_shaped& d, const GsLayout& lay, int S, int Dg, int H) {
  const float* P = d.params;
  for (int i = threadIdx.x; i < Dg; i += blockDim.x) { l.Wg[i] = P[lay.oWg + i]; l.vg[i] = d.spectral_norm ? d.vg[i] : 0.f; }
  for (int i = threadIdx.x; i < H * S; i += blockDim.x) { const int j = i / S, k = i - j * S; l.W1[j * l.ldw + k] = P[lay.oW1 + i]; }
  for (int i = threadIdx.x; i < H; i += blockDim.x) { l.b1[i] = P[lay.ob1 + i]; l.W2[i] = P[lay.oW2 + i]; l.u1[i] = d.spectral_norm ? d.u1[i] : 0.f; l.v2[i] = d.spectral_norm ? d.v2[i] : 0.f; }
  for (int i = threadIdx.x; i < S; i += blockDim.x) l.v1[i] = d.spectral_norm ? d.v1[i] : 0.f;
  if (threadIdx.x == 0) {
    l.sc[0] = P[lay.obg]; l.sc[1] = P[lay.ob2];
    l.sc[2] = d.spectral_norm ? d.ug[0] : 0.f; l.sc[3] = d.spectral_norm ? d.u2[0] : 0.f;
    l.sc[4] = l.sc[5] = l.sc[6] = l.sc[7] = l.sc[8] = 1.f; l.sc[9] = 0.f;
  }
  __syncthreads();
}

// Spectral norm of one discriminator call: `iters_g` iterations of g, then h iterated to the state of its first use (kept in u1n, v1n, sc[5], sc[6], sc[9]) and one
// more to the second use (u1, v1, v2, sc[7], sc[8], sc[3]). iters_g = 0 and iters_h1 = 0: eval mode (sigmas from the stored buffers, both uses equal).
__device__ __forceinline__ void gs_spectral(const GsLds& l, int S, int Dg, int H, int iters_g, int iters_h1) {
  for (int it = 0; it < iters_g; ++it) gs_iter_row(l.Wg, Dg, &l.sc[2], l.vg, l.red);
  { const float s = l.sc[2] * gs_dot(l.Wg, l.vg, Dg, l.red); if (threadIdx.x == 0) l.sc[4] = s; }
  for (int it = 0; it < iters_h1; ++it) { gs_iter_mat(l, S, H, l.u1, l.v1); gs_iter_row(l.W2, H, &l.sc[3], l.v2, l.red); }
  for (int i = threadIdx.x; i < H; i += blockDim.x) l.u1n[i] = l.u1[i];
  for (int i = threadIdx.x; i < S; i += blockDim.x) l.v1n[i] = l.v1[i];
  { const float s1 = gs_sigma_mat(l, S, H, l.u1, l.v1); const float s2 = l.sc[3] * gs_dot(l.W2, l.v2, H, l.red);
    __syncthreads();
    if (threadIdx.x == 0) { l.sc[5] = s1; l.sc[6] = s2; l.sc[9] = l.sc[3]; } }
  // the first use's v2 is not kept: for a [1 x H] weight v = normalize(W u) is the same vector after every iteration >= 1 and is only needed together with
  // u2 (sign) in the chain rule, which uses u2n * v2 (see gs_chain_row)
  if (iters_h1 > 0) { gs_iter_mat(l, S, H, l.u1, l.v1); gs_iter_row(l.W2, H, &l.sc[3], l.v2, l.red); }
  { const float s1 = gs_sigma_mat(l, S, H, l.u1, l.v1); const float s2 = l.sc[3] * gs_dot(l.W2, l.v2, H, l.red);
    __syncthreads();
    if (threadIdx.x == 0) { l.sc[7] = s1; l.sc[8] = s2; } }
  __syncthreads();
}

// rows of one call into LDS: x = cat(s, a) (or s), s' ; mixing for the gradient-penalty call. Also t, w per row -> row[0..R), row[R..2R)
__device__ __forceinline__ void gs_stage_rows(const GsLds& l, const il_disc_shaped& d, const il_batch& pol, const il_batch& exp, int kind, const float* eps_given, uint32_t ctr, int row0,
                                              int S, int Dg) {
  const int call = kind == 3 ? 2 : kind;   // both mixes stage their rows the same way
  const int B = pol.n;
  float* epsr = l.row + 6 * GS_R;   // the draw of each row of a mix call (training.py:118 U(0,1); :106 Beta(alpha, alpha), on the chip only for alpha = 1), once
  if (call == 2) {
    if (threadIdx.x < GS_R) { const int row = row0 + threadIdx.x; epsr[threadIdx.x] = row < B ? (eps_given ? eps_given[row] : philox_uniform(d.noise_seed, ctr, kind == 3 ? IL_STREAM_MIX : IL_STREAM_GP, (uint32_t)row)) : 0.f; }
    __syncthreads();
  }
  auto pick = [&](const il_batch& b, int row, int k, bool next) {
    if (next) return b.next_states[(size_t)row * b.ld_next_states + k];
    return k < S ? b.states[(size_t)row * b.ld_states + k] : b.actions[(size_t)row * b.ld_actions + k - S];
  };
  for (int i = threadIdx.x; i < GS_R * (Dg + S); i += blockDim.x) {
    const int r = i / (Dg + S), c = i - r * (Dg + S), row = row0 + r;
    const bool next = c >= Dg; const int k = next ? c - Dg : c;
    float v = 0.f;
    if (row < B) {
      if (call == 0) v = pick(pol, row, k, next);
      else if (call == 1) v = pick(exp, row, k, next);
      else { const float e = epsr[r]; v = e * pick(exp, row, k, next) + (1.f - e) * pick(pol, row, k, next); }
    }
    if (next) l.Xn[r * l.ldn + k] = v; else l.X[r * l.ldx + k] = v;
  }
  if (threadIdx.x < GS_R) {
    const int r = threadIdx.x, row = row0 + r;
    float t = 0.f, w = 0.f;
    if (row < B) {
      const float tp = pol.terminals[(size_t)row * pol.ld_terminals], te = exp.terminals[(size_t)row * exp.ld_terminals];
      const float wp = pol.weights[(size_t)row * pol.ld_weights], we = exp.weights[(size_t)row * exp.ld_weights];
      if (call == 0) { t = tp; w = wp; } else if (call == 1) { t = te; w = we; }
      else { const float e = epsr[r]; t = e * te + (1.f - e) * tp; w = e * we + (1.f - e) * wp; }
    }
    l.row[r] = t; l.row[GS_R + r] = w;
  }
  __syncthreads();
}

// pre-activations of h for s' (pn, sigma s1n) and s (ps, sigma s1s), then f per row -> row[2R..3R) ; g(x) uses sigma sg
__device__ __forceinline__ void gs_forward(const GsLds& l, int S, int Dg, int H, float discount) {
  const float s1n = l.sc[5], s2n = l.sc[6], s1s = l.sc[7], s2s = l.sc[8], sg = l.sc[4];
  for (int i = threadIdx.x; i < 2 * GS_R * H; i += blockDim.x) {
    const int use = i >= GS_R * H, ii = i - use * GS_R * H, r = ii / H, j = ii - r * H;
    const float* x = use ? l.X + r * l.ldx : l.Xn + r * l.ldn;   // use 0: s' ; use 1: s (the state part of x)
    float s = 0.f;
    for (int k = 0; k < S; ++k) s = fmaf(x[k], l.W1[j * l.ldw + k], s);
    (use ? l.ps : l.pn)[r * l.ldh + j] = s / (use ? s1s : s1n) + l.b1[j];
  }
  __syncthreads();
  if (threadIdx.x < GS_R) {
    const int r = threadIdx.x;
    float gx = 0.f, hn = 0.f, hs = 0.f;
    for (int k = 0; k < Dg; ++k) gx = fmaf(l.X[r * l.ldx + k], l.Wg[k], gx);
    for (int j = 0; j < H; ++j) { hn = fmaf(fmaxf(l.pn[r * l.ldh + j], 0.f), l.W2[j], hn); hs = fmaf(fmaxf(l.ps[r * l.ldh + j], 0.f), l.W2[j], hs); }
    gx = gx / sg + l.sc[0]; hn = hn / s2n + l.sc[1]; hs = hs / s2s + l.sc[1];
    l.row[2 * GS_R + r] = gx + (1.f - l.row[r]) * (discount * hn - hs);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_gs_grad(il_disc_shaped d, il_batch pol, il_batch exp, const float* __restrict__ eps_gp, il_gail_extra x, int pu_value_pass) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = d.state_dim, Dg = d.state_only ? S : S + d.action_dim, H = d.hidden, B = d.batch, tid = threadIdx.x;
  const int tile = blockIdx.x, call = blockIdx.y, ncalls = gridDim.y, nt = gridDim.x, row0 = tile * GS_R;
  const GsLayout lay = gs_layout(S, Dg, H, d.spectral_norm);
  const GsWs wsl = gs_ws(S, Dg, H, B);
  const GsLds l = gs_carve(smem, S, Dg, H);
  float* slab = d.workspace + wsl.slabs + ((size_t)call * nt + tile) * lay.P;
  if (tid == 0 && tile == 0 && call == 0 && !pu_value_pass) adam_tick(d.opt);
  gs_stage_params(l, d, lay, S, Dg, H);
  if (d.spectral_norm) gs_spectral(l, S, Dg, H, call + 1, 2 * call + 1);
  const uint32_t ctr = d.noise_counter ? *d.noise_counter : 0u;
  const int kind = gs_kind(d, call);
  gs_stage_rows(l, d, pol, exp, kind, kind == 3 ? x.eps_mix : eps_gp, ctr, row0, S, Dg);
  gs_forward(l, S, Dg, H, d.discount);
  const float s1n = l.sc[5], s2n = l.sc[6], s1s = l.sc[7], s2s = l.sc[8], sg = l.sc[4], fB = (float)B;
  float* coef_n = l.row + 3 * GS_R; float* coef_s = l.row + 4 * GS_R; float* dzr = l.row + 5 * GS_R;
  const bool is_gp = kind == 2;
  if (!is_gp) {
    if (tid < GS_R) {
      const int r = tid, row = row0 + r;
      float dz = 0.f;
      const bool pu_gate = d.loss_function == IL_LOSS_PUGAIL && d.pu_clamped;
      if (pu_value_pass) {   // training.py:100-102 with a finite margin: this launch (the same power iterations as the real one) only leaves the per-tile sums of
        // w softplus(z) = w bce(z, 0) of the policy / expert call; the gradient launch reads them all and decides, every workgroup the same way (gail.hip does the same)
        float ws_ = 0.f;
        if (row < B) {
          const float* off = kind == 0 ? x.logit_offset_policy : x.logit_offset_expert;
          const float f = l.row[2 * GS_R + r];
          ws_ = l.row[GS_R + r] * softplus_f(off ? f - off[row] : f);
        }
        float part = 0.f;
        for (int o = 0; o < GS_R; ++o) part += __shfl(ws_, o, GS_R);
        if (tid == 0) d.workspace[wsl.pu + (size_t)call * nt + tile] = part;
      }
      float pu_on = 1.f;   // 1: the clamp passes the gradient (always, with nonnegative_margin = inf)
      if (pu_gate && !pu_value_pass) {
        float se = 0.f, sp = 0.f;
        for (int t = 0; t < nt; ++t) { sp += d.workspace[wsl.pu + t]; se += d.workspace[wsl.pu + nt + t]; }
        pu_on = d.pos_class_prior * (se / fB) - sp / fB >= -d.nonnegative_margin ? 1.f : 0.f;   // torch.clamp(min = -margin): gradient where the input is not below the bound
      }
      if (row < B) {
        const float* off = kind == 0 ? x.logit_offset_policy : (kind == 1 ? x.logit_offset_expert : x.logit_offset_mix);
        const float f = l.row[2 * GS_R + r], z = off ? f - off[row] : f, w = l.row[GS_R + r];
        const bool pu = d.loss_function == IL_LOSS_PUGAIL;
        // d loss / d z = w (c_sig sigmoid(z) - c_lab) / B: BCE {1, label}; PUGAIL policy {-1, 0}, expert {2 prior, prior} (clamped away: {0, 0}, {prior, prior}); Mixup {1, eps}
        const float c_sig = pu ? (kind == 1 ? (1.f + pu_on) * d.pos_class_prior : -pu_on) : 1.f;
        const float c_lab = kind == 3 ? l.row[6 * GS_R + r] : (kind == 1 ? (pu ? d.pos_class_prior : 1.f) : 0.f);
        const float p = sigmoid_f(z);
        dz = w * (c_sig * p - c_lab) / fB;
        if (d.entropy_bonus > 0.f) dz += d.entropy_bonus * w * z * p * (1.f - p) / fB;
      }
      dzr[r] = dz; coef_n[r] = dz * (1.f - l.row[r]) * d.discount; coef_s[r] = -dz * (1.f - l.row[r]);
    }
    if (pu_value_pass) return;   // uniform: every thread of the workgroup leaves here
    __syncthreads();
    // ---- g: G^g[k] = sum_r dz x ; bias
    {
      float ipl = 0.f;
      for (int k = tid; k < Dg; k += blockDim.x) { float s = 0.f; for (int r = 0; r < GS_R; ++r) s = fmaf(dzr[r], l.X[r * l.ldx + k], s); l.gin[k] = s; ipl = fmaf(s, l.Wg[k], ipl); }
      const float ip = block_sum(ipl, l.red);
      for (int k = tid; k < Dg; k += blockDim.x) slab[lay.oWg + k] = l.gin[k] / sg - (d.spectral_norm ? (ip / (sg * sg)) * l.sc[2] * l.vg[k] : 0.f);
      if (tid == 0) { float s = 0.f; for (int r = 0; r < GS_R; ++r) s += dzr[r]; slab[lay.obg] = s; }
    }
    __syncthreads();
    // ---- h, two uses. dpre = coef (w2^ [pre > 0]) in place; G^2[j] = sum_r coef relu(pre)
    float g2n = 0.f, g2s = 0.f;   // for j = tid (H <= 256)
    if (tid < H) for (int r = 0; r < GS_R; ++r) { g2n = fmaf(coef_n[r], fmaxf(l.pn[r * l.ldh + tid], 0.f), g2n); g2s = fmaf(coef_s[r], fmaxf(l.ps[r * l.ldh + tid], 0.f), g2s); }
    __syncthreads();
    for (int i = tid; i < 2 * GS_R * H; i += blockDim.x) {
      const int use = i >= GS_R * H, ii = i - use * GS_R * H, r = ii / H, j = ii - r * H;
      float* p = (use ? l.ps : l.pn) + r * l.ldh + j;
      *p = *p > 0.f ? (use ? coef_s[r] : coef_n[r]) * (l.W2[j] / (use ? s2s : s2n)) : 0.f;
    }
    __syncthreads();
    const float ip2n = block_sum(tid < H ? g2n * l.W2[tid] : 0.f, l.red), ip2s = block_sum(tid < H ? g2s * l.W2[tid] : 0.f, l.red);
    if (tid < H) {
      float gb = 0.f;
      for (int r = 0; r < GS_R; ++r) gb += l.pn[r * l.ldh + tid] + l.ps[r * l.ldh + tid];
      slab[lay.ob1 + tid] = gb;
      // (u2, v2) of a [1 x H] weight is a fixed point after its first power iteration, so both uses of this call share it
      slab[lay.oW2 + tid] = g2n / s2n + g2s / s2s - (d.spectral_norm ? (ip2n / (s2n * s2n) + ip2s / (s2s * s2s)) * l.sc[3] * l.v2[tid] : 0.f);
    }
    if (tid == 0) { float s = 0.f; for (int r = 0; r < GS_R; ++r) s += coef_n[r] + coef_s[r]; slab[lay.ob2] = s; }
    // G^1 per use, chain rule per use
    float ipn = 0.f, ips = 0.f;
    for (int i = tid; i < H * S; i += blockDim.x) {
      const int j = i / S, k = i - j * S;
      float an = 0.f, as = 0.f;
      for (int r = 0; r < GS_R; ++r) { an = fmaf(l.pn[r * l.ldh + j], l.Xn[r * l.ldn + k], an); as = fmaf(l.ps[r * l.ldh + j], l.X[r * l.ldx + k], as); }
      const float wjk = l.W1[j * l.ldw + k];
      ipn = fmaf(an, wjk, ipn); ips = fmaf(as, wjk, ips);
      slab[lay.oW1 + i] = an / s1n + as / s1s;
    }
    ipn = block_sum(ipn, l.red); ips = block_sum(ips, l.red);
    if (d.spectral_norm)
      for (int i = tid; i < H * S; i += blockDim.x) {
        const int j = i / S, k = i - j * S;
        slab[lay.oW1 + i] -= (ipn / (s1n * s1n)) * l.u1n[j] * l.v1n[k] + (ips / (s1s * s1s)) * l.u1[j] * l.v1[k];
      }
  } else {
    // ---- gradient penalty (training.py:117-127). q = w2^ [pre_s > 0] -> ps in place ; gin = dD/dx ; c = 2 gp w / B
    for (int i = tid; i < GS_R * H; i += blockDim.x) { const int r = i / H, j = i - r * H; float* p = l.ps + r * l.ldh + j; const bool on = *p > 0.f; l.pn[r * l.ldh + j] = on ? 1.f : 0.f; *p = on ? l.W2[j] / s2s : 0.f; }   // mask -> pn (h(s') plays no part here)
    __syncthreads();
    for (int i = tid; i < GS_R * Dg; i += blockDim.x) {
      const int r = i / Dg, k = i - r * Dg;
      float v = l.Wg[k] / sg;
      if (k < S) { float s = 0.f; for (int j = 0; j < H; ++j) s = fmaf(l.ps[r * l.ldh + j], l.W1[j * l.ldw + k], s); v += -(1.f - l.row[r]) * (s / s1s); }
      const float c = (row0 + r < B) ? 2.f * d.grad_penalty * l.row[GS_R + r] / fB : 0.f;
      l.gin[r * l.ldx + k] = c * v;                       // d penalty / d(dD/dx)
    }
    __syncthreads();
    {  // g
      float ipl = 0.f;
      for (int k = tid; k < Dg; k += blockDim.x) { float s = 0.f; for (int r = 0; r < GS_R; ++r) s += l.gin[r * l.ldx + k]; l.X[k] = s; ipl = fmaf(s, l.Wg[k], ipl); }   // X (no longer needed) row 0 holds G^g
      const float ip = block_sum(ipl, l.red);
      for (int k = tid; k < Dg; k += blockDim.x) slab[lay.oWg + k] = l.X[k] / sg - (d.spectral_norm ? (ip / (sg * sg)) * l.sc[2] * l.vg[k] : 0.f);
      if (tid == 0) { slab[lay.obg] = 0.f; slab[lay.ob2] = 0.f; }
    }
    __syncthreads();
    // cs_in = gin[:, :S] * k_r (k_r = -(1 - t)) -> Xn in place ; G^1 = q^T cs_in ; G^2[j] = sum_r (cs_in . W1^[j]) [pre_s > 0]
    for (int i = tid; i < GS_R * S; i += blockDim.x) { const int r = i / S, k = i - r * S; l.Xn[r * l.ldn + k] = l.gin[r * l.ldx + k] * -(1.f - l.row[r]); }
    __syncthreads();
    float g2 = 0.f;
    if (tid < H) {
      for (int r = 0; r < GS_R; ++r) {
        if (l.pn[r * l.ldh + tid] != 0.f) {   // [pre_s > 0]
          float s = 0.f;
          for (int k = 0; k < S; ++k) s = fmaf(l.Xn[r * l.ldn + k], l.W1[tid * l.ldw + k], s);
          g2 += s / s1s;
        }
      }
      slab[lay.ob1 + tid] = 0.f;
    }
    const float ip2 = block_sum(tid < H ? g2 * l.W2[tid] : 0.f, l.red);
    if (tid < H) slab[lay.oW2 + tid] = g2 / s2s - (d.spectral_norm ? (ip2 / (s2s * s2s)) * l.sc[3] * l.v2[tid] : 0.f);
    float ips = 0.f;
    for (int i = tid; i < H * S; i += blockDim.x) {
      const int j = i / S, k = i - j * S;
      float a = 0.f;
      for (int r = 0; r < GS_R; ++r) a = fmaf(l.ps[r * l.ldh + j], l.Xn[r * l.ldn + k], a);
      ips = fmaf(a, l.W1[j * l.ldw + k], ips);
      slab[lay.oW1 + i] = a / s1s;
    }
    ips = block_sum(ips, l.red);
    if (d.spectral_norm)
      for (int i = tid; i < H * S; i += blockDim.x) { const int j = i / S, k = i - j * S; slab[lay.oW1 + i] -= (ips / (s1s * s1s)) * l.u1[j] * l.v1[k]; }
  }
  if (tile == 0 && call == ncalls - 1 && d.spectral_norm) {   // buffers after this update: the last call's second-use state
    float* o = d.workspace + wsl.sn_new;
    if (tid == 0) { o[0] = l.sc[2]; o[1] = l.sc[3]; }
    for (int i = tid; i < Dg; i += blockDim.x) o[2 + i] = l.vg[i];
    for (int i = tid; i < H; i += blockDim.x) { o[2 + Dg + i] = l.u1[i]; o[2 + Dg + H + i] = l.v2[i]; }
    for (int i = tid; i < S; i += blockDim.x) o[2 + Dg + 2 * H + i] = l.v1[i];
  }
}

__global__ __launch_bounds__(256) void k_gs_reduce(il_disc_shaped d, int apply) {
  const int S = d.state_dim, Dg = d.state_only ? S : S + d.action_dim, H = d.hidden;
  const GsLayout lay = gs_layout(S, Dg, H, d.spectral_norm);
  const GsWs wsl = gs_ws(S, Dg, H, d.batch);
  const int nslabs = ((d.batch + GS_R - 1) / GS_R) * gs_calls(d);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < lay.P) {
    float pp = d.params[e], mm = 0.f, vv = 0.f;
    if (apply) { mm = d.opt.m[e]; vv = d.opt.v[e]; }
    float g = 0.f;
    for (int t = 0; t < nslabs; ++t) g += d.workspace[wsl.slabs + (size_t)t * lay.P + e];
    d.grad[e] = g;
    if (apply) {
      const adam_consts ac = load_adam_consts(d.opt);
      adam_update(pp, g, mm, vv, ac);
      d.params[e] = pp; d.opt.m[e] = mm; d.opt.v[e] = vv;
    }
  }
  if (blockIdx.x == 0 && d.spectral_norm) {
    const float* o = d.workspace + wsl.sn_new;
    if (threadIdx.x == 0) { d.ug[0] = o[0]; d.u2[0] = o[1]; }
    for (int i = threadIdx.x; i < Dg; i += blockDim.x) d.vg[i] = o[2 + i];
    for (int i = threadIdx.x; i < H; i += blockDim.x) { d.u1[i] = o[2 + Dg + i]; d.v2[i] = o[2 + Dg + H + i]; }
    for (int i = threadIdx.x; i < S; i += blockDim.x) d.v1[i] = o[2 + Dg + 2 * H + i];
  }
}

// eval mode (no power iteration): reward head of models.py:177-180 on f (minus the optional log-policy offset)
__global__ __launch_bounds__(256) void k_gs_reward(il_disc_shaped d, il_batch b, float* __restrict__ out_r, float* __restrict__ out_logit, const float* __restrict__ logit_offset) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = d.state_dim, Dg = d.state_only ? S : S + d.action_dim, H = d.hidden, row0 = blockIdx.x * GS_R;
  const GsLayout lay = gs_layout(S, Dg, H, d.spectral_norm);
  const GsLds l = gs_carve(smem, S, Dg, H);
  gs_stage_params(l, d, lay, S, Dg, H);
  if (d.spectral_norm) gs_spectral(l, S, Dg, H, 0, 0);
  gs_stage_rows(l, d, b, b, 0, nullptr, 0u, row0, S, Dg);
  gs_forward(l, S, Dg, H, d.discount);
  if (threadIdx.x < GS_R && row0 + threadIdx.x < b.n) {
    const int row = row0 + threadIdx.x;
    const float f = l.row[2 * GS_R + threadIdx.x], z = logit_offset ? f - logit_offset[row] : f, Dp = sigmoid_f(z);
    float h = d.reward_function == 1 ? -log1pf(-Dp + 1e-6f) : logf(Dp + 1e-6f) - log1pf(-Dp + 1e-6f);
    if (d.reward_function == 2) h = expf(h) * -h;
    out_r[row] = h;
    if (out_logit) out_logit[row] = z;
  }
}

static int check_gs(const il_disc_shaped* d) {
  IL_CHECK_ARG(d && d->params && d->workspace, "il_disc_shaped: null descriptor field");
  const int S = d->state_dim, Dg = d->state_only ? S : S + d->action_dim;
  IL_CHECK_ARG(S >= 1 && Dg <= 512 && d->hidden >= 1 && d->hidden <= 256, "il_disc_shaped: dims out of range (state=%d, input=%d, hidden=%d; hidden <= 256)", S, Dg, d->hidden);
  IL_CHECK_ARG(gs_lds_floats(S, Dg, d->hidden) * sizeof(float) <= 160 * 1024, "il_disc_shaped: state=%d hidden=%d needs more than 160 KiB of LDS", S, d->hidden);
  IL_CHECK_ARG(d->reward_function >= 0 && d->reward_function <= 2 && d->loss_function >= IL_LOSS_BCE && d->loss_function <= IL_LOSS_MIXUP,
               "il_disc_shaped: reward_function in {0,1,2}, loss_function BCE, PUGAIL or Mixup");
  if (d->spectral_norm) IL_CHECK_ARG(d->ug && d->vg && d->u1 && d->v1 && d->u2 && d->v2, "il_disc_shaped: spectral-norm buffers missing");
  if (d->workspace_floats < gs_ws(S, Dg, d->hidden, d->batch).total) return il_set_error(IL_ERR_WORKSPACE, "il_disc_shaped: workspace too small");
  return IL_OK;
}
static int gs_ensure_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}

extern "C" int il_gail_shaped_step(const il_disc_shaped* d, const il_batch* pol, const il_batch* exp, const float* eps_gp, const il_gail_extra* extra, uint32_t flags, il_stream_t stream_) {
  IL_NO_GATHER(pol, "il_gail_shaped_step"); IL_NO_GATHER(exp, "il_gail_shaped_step");
  if (int rc = check_gs(d)) return rc;
  IL_CHECK_ARG(pol && exp && pol->n == d->batch && exp->n == d->batch && d->grad && d->opt.m && d->opt.v && d->opt.step, "il_gail_shaped_step: bad batches / optimiser state");
  IL_CHECK_ARG(pol->next_states && pol->terminals && exp->next_states && exp->terminals && pol->weights && exp->weights, "il_gail_shaped_step: the shaping term needs next_states, terminals and weights");
  il_gail_extra x = {};
  if (extra) x = *extra;
  IL_CHECK_ARG(d->loss_function != IL_LOSS_MIXUP || (!x.logit_offset_policy && !x.logit_offset_expert), "il_gail_shaped_step: with Mixup the log-policy offset belongs to the mixed batch (logit_offset_mix)");
  const int S = d->state_dim, Dg = d->state_only ? S : S + d->action_dim;
  const size_t lds = gs_lds_floats(S, Dg, d->hidden) * sizeof(float);
  if (int rc = gs_ensure_lds((const void*)k_gs_grad, lds)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  if (d->loss_function == IL_LOSS_PUGAIL && d->pu_clamped) {   // finite nonnegative_margin: a value pass (logits only) ahead of the gradient pass, which reads the clamp decision
    IL_CHECK_ARG(d->nonnegative_margin >= 0.f, "il_gail_shaped_step: nonnegative_margin must be >= 0");
    { IL_TRACE("k_gs_grad", st); k_gs_grad<<<dim3(ceil_div(d->batch, GS_R), 2), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, 1); }
  }
  { IL_TRACE("k_gs_grad", st); k_gs_grad<<<dim3(ceil_div(d->batch, GS_R), gs_calls(*d)), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, 0); }
  const int64_t P = gs_layout(S, Dg, d->hidden, d->spectral_norm).P;
  { IL_TRACE("k_gs_reduce", st); k_gs_reduce<<<(int)((P + 255) / 256), 256, 0, st>>>(*d, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1); }
  IL_CHECK_LAUNCH("il_gail_shaped_step");
  return IL_OK;
}

extern "C" int il_gail_shaped_reward(const il_disc_shaped* d, const il_batch* b, float* out_rewards, float* out_logits, const float* logit_offset, il_stream_t stream_) {
  IL_NO_GATHER(b, "il_gail_shaped_reward");
  if (int rc = check_gs(d)) return rc;
  IL_CHECK_ARG(b && out_rewards && b->n > 0 && b->next_states && b->terminals, "il_gail_shaped_reward: bad arguments (next_states and terminals are inputs of the shaping term)");
  const int S = d->state_dim, Dg = d->state_only ? S : S + d->action_dim;
  const size_t lds = gs_lds_floats(S, Dg, d->hidden) * sizeof(float);
  if (int rc = gs_ensure_lds((const void*)k_gs_reward, lds)) return rc;
  { IL_TRACE("k_gs_reward", (hipStream_t)stream_); k_gs_reward<<<ceil_div(b->n, GS_R), 256, lds, (hipStream_t)stream_>>>(*d, *b, out_rewards, out_logits, logit_offset); }
  IL_CHECK_LAUNCH("il_gail_shaped_reward");
  return IL_OK;
}
