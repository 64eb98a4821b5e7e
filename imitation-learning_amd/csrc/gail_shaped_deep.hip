// GAIL discriminator with reward shaping and a shaping potential of any `_create_fcnn` shape (reference models.py:152-180 with reward_shaping = true and
// discriminator.depth in {1, 2}, activation in {relu, tanh}: conf/hyperparameter_search_space/GAIL.yaml; training.py:85-134) for gfx950.
//
//   f(s, a, s', t) = g(x) + (1 - t) (discount h(s') - h(s)),   g = SN(Linear(Dg, 1)),   h = [SN(Linear) - act] x depth - SN(Linear(H, 1)) on the state.
//
// gail_shaped.hip is the depth-1 ReLU potential of the default `discriminator` block written out by hand (hidden <= 256); this file is the general one and is built
// from the workgroup-level pieces gail_deep.hip uses for a discriminator of the same shape (gail_deep_tile.hpp). What shaping adds (oracle/gail_shaped_deep.py):
//   * torch's _SpectralNorm runs one power iteration per ACCESS of a weight and `forward` evaluates g(x), h(s'), h(s) in that order: call c uses g after c + 1
//     iterations, h(s') after 2c + 1 and h(s) after 2c + 2 iterations of every layer of h, and the chain rule dW = G^/sigma - <G^, W>/sigma^2 u v^T holds per USE.
//   * the gradient penalty reaches h through dh/ds of the SECOND use only: dD/ds = Wg_s^ + k dh/ds, k = -(1 - t); dL/d(dh/ds) = 2 c k dD/ds, c = lambda w / B.
//   k_gsd_grad    grid (tiles, calls, 2 uses): one workgroup = 16 rows of one USE of h in one discriminator call. It needs the logit f of its rows, i.e. both uses'
//                 forward passes: it replays the power iterations, runs the OTHER use first (logits only), then its own (activations kept), and back-propagates its
//                 own coefficient dz (1 - t) discount / -dz (1 - t). The second-use workgroup also owns g (chain rule applied per tile: it is linear in G^) and,
//                 in the gradient-penalty call, the whole penalty (the first-use workgroups of that call have nothing to do and leave at once).
//   k_gsd_reduce  slab sums in (call, use, tile) order, chain rule per (call, use) for h, AdamW, the last call's second-use (u, v) become the buffers.
//   k_gsd_reward  eval-mode forward (no power iteration: both uses see the same weights) + AIRL / GAIL / FAIRL head.
#include "il_common.hpp"
#include "gail_deep_tile.hpp"

struct GsdLayout { int64_t oWg, obg, oh, P; };   // g, then h's layers in gd_layout order from `oh`
__host__ __device__ inline GsdLayout gsd_layout(int S, int Dg, int H, int depth, int sn) {
  GsdLayout l;
  if (sn) { l.obg = 0; l.oWg = 1; } else { l.oWg = 0; l.obg = Dg; }
  l.oh = Dg + 1; l.P = l.oh + gd_layout(S, H, depth, sn).P;
  return l;
}
__host__ __device__ inline int gsd_depth(const il_disc_shaped_deep& d) { return d.depth == 2 ? 2 : 1; }
__host__ __device__ inline int gsd_dg(const il_disc_shaped_deep& d) { return d.state_only ? d.state_dim : d.state_dim + d.action_dim; }
__host__ __device__ inline int gsd_calls(const il_disc_shaped_deep& d) { return (d.loss_function == IL_LOSS_MIXUP ? 1 : 2) + (d.grad_penalty > 0.f ? 1 : 0); }
// what a call runs on: 0 policy, 1 expert, 2 gradient-penalty mix (training.py:116-126), 3 Mixup mix (training.py:104-113)
__host__ __device__ inline int gsd_kind(const il_disc_shaped_deep& d, int call) { return d.loss_function == IL_LOSS_MIXUP ? (call == 0 ? 3 : 2) : call; }
__host__ __device__ inline int64_t gsd_sn_numel(int S, int Dg, int H, int depth) { return 1 + Dg + gd_sn_numel(S, H, depth); }   // ug | vg | h: per layer u | v
// workspace: slabs [3 calls][2 uses][tiles][P + 4: the parameters, then <G^, W> of h's layers] | context [3][2][4 sigmas of h + h's u | v] |
//            g's (u, v) after the update [1 + Dg] | pu [2][tiles]: per-tile sums of w softplus(z) of the policy / expert call (PUGAIL with a finite nonnegative_margin)
struct GsdWs { int64_t slabs, slab_stride, ctx, ctx_stride, sn_g, pu, total; };
__host__ __device__ inline GsdWs gsd_ws(int S, int Dg, int H, int depth, int B) {
  GsdWs w; const int64_t P = gsd_layout(S, Dg, H, depth, 1).P, nt = (B + GD_R - 1) / GD_R;
  w.slab_stride = (P + 4 + 3) & ~(int64_t)3; w.slabs = 0;
  w.ctx = 6 * nt * w.slab_stride; w.ctx_stride = (4 + gd_sn_numel(S, H, depth) + 3) & ~(int64_t)3;
  w.sn_g = w.ctx + 6 * w.ctx_stride;
  w.pu = w.sn_g + ((1 + Dg + 3) & ~3);
  w.total = w.pu + ((2 * nt + 3) & ~(int64_t)3);
  return w;
}
// LDS: the potential's tile (gd_carve with input width S), then g and the per-row scalars
struct GsdLds { float *Wg, *vg, *gsc, *XA, *GQ, *rw; int lda; };   // gsc: 0 bg, 1 ug, 2 sigma_g, 4..6 the first use's sigmas of h
#define GSD_ROWS 12   // per-row arrays in rw: 0 t, 1 w, 2 h(s'), 3 h(s), 4 g(x), 5 f, 6 dz, 7 this use's coefficient, 8 the mix draw, 9 2 c, 10 k = -(1 - t)
__host__ __device__ inline size_t gsd_lds_floats(int S, int Dg, int H, int depth) {
  return gd_lds_floats(S, H, depth) + 2 * (size_t)Dg + 8 + (size_t)GD_R * (Dg - S + 1) + (size_t)GD_R * (S + 1) + Dg + GSD_ROWS * GD_R;   // GQ: a [16][S + 1] tile or a [Dg] vector
}
__device__ __forceinline__ GsdLds gsd_carve(float* p, int S, int Dg) {
  GsdLds g; g.lda = Dg - S + 1;
  g.Wg = p; p += Dg; g.vg = p; p += Dg; g.gsc = p; p += 8; g.XA = p; p += GD_R * g.lda; g.GQ = p; p += GD_R * (S + 1) + Dg; g.rw = p;
  return g;
}
extern "C" int64_t il_disc_shaped_deep_numel(int32_t S, int32_t A, int32_t H, int32_t depth, int32_t state_only) { return gsd_layout(S, state_only ? S : S + A, H, depth == 2 ? 2 : 1, 1).P; }
extern "C" int64_t il_disc_shaped_deep_sn_numel(int32_t S, int32_t A, int32_t H, int32_t depth, int32_t state_only) { return gsd_sn_numel(S, state_only ? S : S + A, H, depth == 2 ? 2 : 1); }
extern "C" int64_t il_disc_shaped_deep_workspace_floats(int32_t S, int32_t A, int32_t H, int32_t depth, int32_t B, int32_t state_only) {
  return gsd_ws(S, state_only ? S : S + A, H, depth == 2 ? 2 : 1, B).total;
}
extern "C" int64_t il_disc_shaped_deep_lds_bytes(int32_t S, int32_t A, int32_t H, int32_t depth, int32_t state_only) {
  return (int64_t)(gsd_lds_floats(S, state_only ? S : S + A, H, depth == 2 ? 2 : 1) * sizeof(float));
}

// g's parameters and buffers into LDS
__device__ __forceinline__ void gsd_stage_g(const GsdLds& g, const il_disc_shaped_deep& d, const GsdLayout& lay, int Dg) {
  for (int i = threadIdx.x; i < Dg; i += blockDim.x) { g.Wg[i] = d.params[lay.oWg + i]; g.vg[i] = d.spectral_norm ? d.sn[1 + i] : 0.f; }
  if (threadIdx.x == 0) { g.gsc[0] = d.params[lay.obg]; g.gsc[1] = d.spectral_norm ? d.sn[0] : 0.f; g.gsc[2] = 1.f; }
  __syncthreads();
}
// `iters` power iterations of the [1 x Dg] weight (u = normalize(W v): a sign; v = normalize(W^T u)), then sigma_g = u (W . v) into gsc[2]
__device__ __forceinline__ void gsd_spectral_g(const GsdLds& g, int Dg, int iters, float* red) {
  for (int it = 0; it < iters; ++it) {
    const float wv = gd_dot(g.Wg, g.vg, Dg, red);
    const float u = wv / fmaxf(fabsf(wv), 1e-12f);
    __syncthreads();
    for (int i = threadIdx.x; i < Dg; i += blockDim.x) g.vg[i] = g.Wg[i] * u;
    if (threadIdx.x == 0) g.gsc[1] = u;
    __syncthreads();
    gd_normalize(g.vg, Dg, red);
  }
  const float s = g.gsc[1] * gd_dot(g.Wg, g.vg, Dg, red);
  __syncthreads();
  if (threadIdx.x == 0) g.gsc[2] = s;
  __syncthreads();
}
// the mix draw, terminal and weight of the tile's rows (kinds 2 / 3: convex combinations, training.py:107,120) -> rw[8], rw[0], rw[1]
__device__ __forceinline__ void gsd_stage_scalars(const GsdLds& g, const il_disc_shaped_deep& d, const il_batch& pol, const il_batch& exp, int kind, const float* eps_given, uint32_t ctr,
                                                  int row0, int nrows) {
  if (threadIdx.x < GD_R) {
    const int r = threadIdx.x, row = row0 + r;
    float e = 0.f, t = 0.f, w = 0.f;
    if (r < nrows) {
      if (kind >= 2) e = eps_given ? eps_given[row] : philox_uniform(d.noise_seed, ctr, kind == 3 ? IL_STREAM_MIX : IL_STREAM_GP, (uint32_t)row);
      const float tp = kind != 1 ? pol.terminals[(size_t)row * pol.ld_terminals] : 0.f, te = kind != 0 ? exp.terminals[(size_t)row * exp.ld_terminals] : 0.f;
      const float wp = kind != 1 ? pol.weights[(size_t)row * pol.ld_weights] : 0.f, we = kind != 0 ? exp.weights[(size_t)row * exp.ld_weights] : 0.f;
      if (kind >= 2) { t = e * te + (1.f - e) * tp; w = e * we + (1.f - e) * wp; } else { t = kind == 0 ? tp : te; w = kind == 0 ? wp : we; }
    }
    g.rw[8 * GD_R + r] = e; g.rw[r] = t; g.rw[GD_R + r] = w;
  }
  __syncthreads();
}
// the tile's states (next = 0; and its actions into XA) or next states (next = 1) into the potential's input tile X
__device__ __forceinline__ void gsd_stage_rows(const GdLds& l, const GsdLds& g, const il_batch& pol, const il_batch& exp, int kind, int next, int row0, int nrows, int S, int A) {
  const float* eps = g.rw + 8 * GD_R;
  for (int i = threadIdx.x; i < GD_R * S; i += blockDim.x) {
    const int r = i / S, k = i - r * S; float v = 0.f;
    if (r < nrows) {
      const size_t row = (size_t)(row0 + r);
      float xp = 0.f, xe = 0.f;
      if (kind != 1) xp = next ? pol.next_states[row * pol.ld_next_states + k] : pol.states[row * pol.ld_states + k];
      if (kind != 0) xe = next ? exp.next_states[row * exp.ld_next_states + k] : exp.states[row * exp.ld_states + k];
      if (kind >= 2) { const float e = eps[r]; v = e * xe + (1.f - e) * xp; } else v = kind == 0 ? xp : xe;
    }
    l.X[r * l.ldx + k] = v;
  }
  if (!next)
    for (int i = threadIdx.x; i < GD_R * A; i += blockDim.x) {
      const int r = i / A, k = i - r * A; float v = 0.f;
      if (r < nrows) {
        const size_t row = (size_t)(row0 + r);
        float xp = 0.f, xe = 0.f;
        if (kind != 1) xp = pol.actions[row * pol.ld_actions + k];
        if (kind != 0) xe = exp.actions[row * exp.ld_actions + k];
        if (kind >= 2) { const float e = eps[r]; v = e * xe + (1.f - e) * xp; } else v = kind == 0 ? xp : xe;
      }
      g.XA[r * g.lda + k] = v;
    }
  __syncthreads();
}
// g(x) of the staged rows (X = states, XA = actions) -> rw[4]
__device__ __forceinline__ void gsd_forward_g(const GdLds& l, const GsdLds& g, int S, int Dg) {
  if (threadIdx.x < GD_R) {
    const int r = threadIdx.x;
    float gx = 0.f;
    for (int k = 0; k < S; ++k) gx = fmaf(l.X[r * l.ldx + k], g.Wg[k], gx);
    for (int k = S; k < Dg; ++k) gx = fmaf(g.XA[r * g.lda + k - S], g.Wg[k], gx);
    g.rw[4 * GD_R + r] = gx / g.gsc[2] + g.gsc[0];
  }
  __syncthreads();
}
// Both uses of h for this tile, the workgroup's own use (q: 0 = h(s'), 1 = h(s)) LAST so that its activations and scaled weights are what stays in LDS; h(s'), h(s), g(x)
// and f per row into rw. iters0 = power iterations before the first use (train: 2 call + 1; -1: eval mode, no iteration at all). `ctx`: where to leave this use's
// (sigmas, u | v) for the reduce kernel, or nullptr.
template <int DEPTH>
__device__ __forceinline__ void gsd_forward(const GdLds& l, const GsdLds& g, const GdLayout& layh, const il_disc_shaped_deep& d, const GsdLayout& lay, const il_batch& pol,
                                            const il_batch& exp, int kind, int q, int iters0, float* __restrict__ ctx, int row0, int nrows, int S, int A, int Dg, int H, int tanh_) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const bool sn = d.spectral_norm != 0, train = iters0 >= 0;
  auto leave_ctx = [&]() {
    if (!ctx || !sn) return;
    if (tid < 4) ctx[tid] = tid <= DEPTH ? l.sc[tid] : 1.f;
    int64_t o = 4;
#pragma unroll
    for (int i = 0; i <= DEPTH; ++i) {
      for (int e = tid; e < layh.out[i]; e += nthr) ctx[o + e] = l.u[i][e];
      o += layh.out[i];
      for (int e = tid; e < layh.in[i]; e += nthr) ctx[o + e] = l.v[i][e];
      o += layh.in[i];
    }
  };
  auto use = [&](int next) {   // forward of the staged weights on s' (next) or s; logits into rw[2] / rw[3]; with s also g(x)
    gsd_stage_rows(l, g, pol, exp, kind, next, row0, nrows, S, A);
    gd_forward(l, layh, H, tanh_);
    if (tid < GD_R) g.rw[(next ? 2 : 3) * GD_R + tid] = l.row[tid];
    if (!next) gsd_forward_g(l, g, S, Dg); else __syncthreads();
  };
  if (!sn || !train) {            // one set of weights for both uses
    if (sn) { gd_sigma(l, layh); gd_scale(l, layh); }
    if (q == 1) leave_ctx();
    use(q == 1 ? 1 : 0); use(q == 1 ? 0 : 1);
    if (q == 0) leave_ctx();
  } else if (q == 1) {
    gd_power(l, layh, iters0); gd_sigma(l, layh); gd_scale(l, layh);
    use(1);
    gd_stage_weights(l, layh, d.params + lay.oh); __syncthreads();
    gd_power(l, layh, 1); gd_sigma(l, layh);
    leave_ctx();
    gd_scale(l, layh);
    use(0);
  } else {
    gd_power(l, layh, iters0); gd_sigma(l, layh);
    leave_ctx();
    if (tid < 3) g.gsc[4 + tid] = l.sc[tid];
    __syncthreads();
    gd_power(l, layh, 1); gd_sigma(l, layh); gd_scale(l, layh);
    use(0);
    gd_stage_weights(l, layh, d.params + lay.oh);
    if (tid < 3) l.sc[tid] = g.gsc[4 + tid];
    __syncthreads();
    gd_scale(l, layh);
    use(1);
  }
  if (tid < GD_R) {
    const float t = g.rw[tid];
    g.rw[5 * GD_R + tid] = g.rw[4 * GD_R + tid] + (1.f - t) * (d.discount * g.rw[2 * GD_R + tid] - g.rw[3 * GD_R + tid]);
  }
  __syncthreads();
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_gsd_grad(il_disc_shaped_deep d, il_batch pol, il_batch exp, const float* __restrict__ eps_gp, il_gail_extra x, int pu_value_pass) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = d.state_dim, Dg = gsd_dg(d), A = Dg - S, H = d.hidden, B = d.batch, tanh_ = d.activation == 1;
  const int tile = blockIdx.x, call = blockIdx.y, nt = gridDim.x, row0 = tile * GD_R, tid = threadIdx.x, nthr = blockDim.x;
  const int q = pu_value_pass ? 1 : (int)blockIdx.z;
  const int nrows = min(GD_R, B - row0);
  const int kind = gsd_kind(d, call);
  if (kind == 2 && q == 0) return;   // the penalty goes through the second use only (k_gsd_reduce skips this slab)
  const GsdLayout lay = gsd_layout(S, Dg, H, DEPTH, d.spectral_norm);
  const GdLayout layh = gd_layout(S, H, DEPTH, d.spectral_norm);
  const GsdWs ws = gsd_ws(S, Dg, H, DEPTH, B);
  const GdLds l = gd_carve(smem, S, H, DEPTH);
  const GsdLds g = gsd_carve(smem + gd_lds_floats(S, H, DEPTH), S, Dg);
  float* slab = d.workspace + ws.slabs + (((size_t)call * 2 + q) * nt + tile) * ws.slab_stride;
  float* slab_h = slab + lay.oh;
  if (tile == 0 && call == 0 && q == 0 && tid == 0 && !pu_value_pass) adam_tick(d.opt);
  gsd_stage_g(g, d, lay, Dg);
  gd_stage(l, layh, d.params + lay.oh, d.spectral_norm ? d.sn + 1 + Dg : nullptr);
  if (d.spectral_norm) gsd_spectral_g(g, Dg, call + 1, l.red);
  const uint32_t ctr = d.noise_counter ? *d.noise_counter : 0u;
  gsd_stage_scalars(g, d, pol, exp, kind, kind == 3 ? x.eps_mix : eps_gp, ctr, row0, nrows);
  float* ctx = (tile == 0 && !pu_value_pass) ? d.workspace + ws.ctx + ((size_t)call * 2 + q) * ws.ctx_stride : nullptr;
  gsd_forward<DEPTH>(l, g, layh, d, lay, pol, exp, kind, q, 2 * call + 1, ctx, row0, nrows, S, A, Dg, H, tanh_);
  const float sg = g.gsc[2], fB = (float)B;
  float* dzr = g.rw + 6 * GD_R; float* coef = g.rw + 7 * GD_R; float* c2 = g.rw + 9 * GD_R; float* kr = g.rw + 10 * GD_R;
  if (kind != 2) {
    // ---- first-order call: dL/dz = w (c_sig sigmoid(z) - c_lab) / B (+ entropy bonus); this use's share of it goes back through h
    if (tid < GD_R) {
      const int row = row0 + min(tid, nrows - 1);
      const float* off = kind == 0 ? x.logit_offset_policy : (kind == 1 ? x.logit_offset_expert : x.logit_offset_mix);
      const float f = g.rw[5 * GD_R + tid], z = off ? f - off[row] : f, w = g.rw[GD_R + tid], t = g.rw[tid];
      const bool pu = d.loss_function == IL_LOSS_PUGAIL;
      if (pu_value_pass) {   // training.py:100-102 with a finite margin: this launch (policy and expert call, the same power iterations as the real one) only leaves the
        // per-tile sums of w softplus(z) = w bce(z, 0); the gradient launch reads them all and decides, every workgroup the same way (gail.hip does the same)
        const float ws_ = tid < nrows ? w * softplus_f(z) : 0.f;
        float part = 0.f;
        for (int o = 0; o < GD_R; ++o) part += __shfl(ws_, o, GD_R);
        if (tid == 0) d.workspace[ws.pu + (size_t)kind * nt + tile] = part;
      }
      float pu_on = 1.f;   // 1: the clamp passes the gradient (always, with nonnegative_margin = inf)
      if (pu && d.pu_clamped && !pu_value_pass) {
        float se = 0.f, sp = 0.f;
        for (int tt = 0; tt < nt; ++tt) { sp += d.workspace[ws.pu + tt]; se += d.workspace[ws.pu + nt + tt]; }
        pu_on = d.pos_class_prior * (se / fB) - sp / fB >= -d.nonnegative_margin ? 1.f : 0.f;   // torch.clamp(min = -margin): gradient where the input is not below the bound
      }
      // BCE {1, label}; PUGAIL policy {-1, 0}, expert {2 prior, prior} (clamped away: {0, 0}, {prior, prior}); Mixup {1, eps}
      const float c_sig = pu ? (kind == 1 ? (1.f + pu_on) * d.pos_class_prior : -pu_on) : 1.f;
      const float c_lab = kind == 3 ? g.rw[8 * GD_R + tid] : (kind == 1 ? (pu ? d.pos_class_prior : 1.f) : 0.f);
      const float p = sigmoid_f(z);
      float dz = tid < nrows ? w * (c_sig * p - c_lab) / fB : 0.f;
      if (d.entropy_bonus > 0.f && tid < nrows) dz += d.entropy_bonus * w * z * p * (1.f - p) / fB;
      dzr[tid] = dz;
      coef[tid] = q == 0 ? dz * (1.f - t) * d.discount : -dz * (1.f - t);
    }
    if (pu_value_pass) return;   // uniform: every thread of the workgroup leaves here
    __syncthreads();
    gd_backprop<DEPTH>(l, layh, slab_h, coef, H, tanh_);
    // ---- g (second-use workgroup; X / XA hold its states and actions): G^g[k] = sum_r dz x, chain rule applied here
    if (q == 1) {
      float ipl = 0.f;
      for (int k = tid; k < Dg; k += nthr) {
        float s = 0.f;
        for (int r = 0; r < GD_R; ++r) s = fmaf(dzr[r], k < S ? l.X[r * l.ldx + k] : g.XA[r * g.lda + k - S], s);
        g.GQ[k] = s; ipl = fmaf(s, g.Wg[k], ipl);
      }
      const float ip = block_sum(ipl, l.red);
      for (int k = tid; k < Dg; k += nthr) slab[lay.oWg + k] = g.GQ[k] / sg - (d.spectral_norm ? (ip / (sg * sg)) * g.gsc[1] * g.vg[k] : 0.f);
      if (tid == 0) { float s = 0.f; for (int r = 0; r < GD_R; ++r) s += dzr[r]; slab[lay.obg] = s; }
    } else {
      for (int k = tid; k < Dg; k += nthr) slab[lay.oWg + k] = 0.f;
      if (tid == 0) slab[lay.obg] = 0.f;
    }
  } else {
    // ---- gradient penalty (training.py:117-127), second use: gin = dD/dx = Wg^ + k dh/ds (state columns), L = sum_r c ||gin||^2
    if (tid < GD_R) { c2[tid] = tid < nrows ? 2.f * d.grad_penalty * g.rw[GD_R + tid] / fB : 0.f; kr[tid] = -(1.f - g.rw[tid]); }
    __syncthreads();
    gd_input_grad_u<DEPTH>(l, layh, H, tanh_);
    for (int e = tid; e < GD_R * S; e += nthr) {
      const int r = e / S, k = e - r * S;
      float s = 0.f;
      for (int n = 0; n < H; ++n) s = fmaf(l.U[0][r * l.ldh + n], l.W[0][n * l.ldw[0] + k], s);   // dh/ds
      const float cg = c2[r] * (g.Wg[k] / sg + kr[r] * s);   // d penalty / d(dD/ds)
      g.GQ[r * l.ldx + k] = cg;
      l.SB[r * l.ldx + k] = cg * kr[r];                      // d penalty / d(dh/ds)
    }
    __syncthreads();
    {  // g: G^g = sum_r c2 gin (action columns: gin = Wg^)
      float c2s = 0.f;
      for (int r = 0; r < GD_R; ++r) c2s += c2[r];
      float ipl = 0.f;
      for (int k = tid; k < Dg; k += nthr) {
        float s = 0.f;
        if (k < S) { for (int r = 0; r < GD_R; ++r) s += g.GQ[r * l.ldx + k]; } else s = c2s * (g.Wg[k] / sg);
        ipl = fmaf(s, g.Wg[k], ipl);
        slab[lay.oWg + k] = s;           // G^g for now; chain rule below
      }
      const float ip = block_sum(ipl, l.red);
      for (int k = tid; k < Dg; k += nthr) slab[lay.oWg + k] = slab[lay.oWg + k] / sg - (d.spectral_norm ? (ip / (sg * sg)) * g.gsc[1] * g.vg[k] : 0.f);
      if (tid == 0) slab[lay.obg] = 0.f;
    }
    gd_input_grad_backward<DEPTH>(l, layh, slab_h, S, H, tanh_);
  }
  __syncthreads();
  gd_inner_products<DEPTH>(l, layh, slab_h, slab + lay.P);
  if (tile == 0 && call == (int)gridDim.y - 1 && q == 1 && d.spectral_norm) {   // g's buffers after this update: the last call's iteration
    float* o = d.workspace + ws.sn_g;
    if (tid == 0) o[0] = g.gsc[1];
    for (int i = tid; i < Dg; i += nthr) o[1 + i] = g.vg[i];
  }
}

// grid = ceil(P / 256): one parameter per thread; slabs summed in (call, use, tile) order (deterministic)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_gsd_reduce(il_disc_shaped_deep d, int apply) {
  constexpr int depth = DEPTH;
  const int S = d.state_dim, Dg = gsd_dg(d), H = d.hidden, B = d.batch;
  const GsdLayout lay = gsd_layout(S, Dg, H, depth, d.spectral_norm);
  const GdLayout layh = gd_layout(S, H, depth, d.spectral_norm);
  const GsdWs ws = gsd_ws(S, Dg, H, depth, B);
  const int nt = (B + GD_R - 1) / GD_R, calls = gsd_calls(d);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < lay.P) {
    int layer = -1, n = 0, k = 0, out_l = 0;
    int64_t o_l = 4, o_run = 4;   // where the layer's u | v sit in a (call, use) context (compile-time indices: a table looked up by `layer` would live in scratch memory)
    const int64_t eh = e - lay.oh;
#pragma unroll
    for (int i = 0; i <= depth; ++i) {
      if (eh >= layh.oW[i] && eh < layh.oW[i] + (int64_t)layh.out[i] * layh.in[i]) { layer = i; n = (int)((eh - layh.oW[i]) / layh.in[i]); k = (int)((eh - layh.oW[i]) % layh.in[i]); o_l = o_run; out_l = layh.out[i]; }
      o_run += layh.out[i] + layh.in[i];
    }
    float gsum = 0.f;
    for (int c = 0; c < calls; ++c)
      for (int q = 0; q < 2; ++q) {
        if (gsd_kind(d, c) == 2 && q == 0) continue;   // nothing was written there
        const float* sl = d.workspace + ws.slabs + ((size_t)c * 2 + q) * nt * ws.slab_stride;
        float gc = 0.f;
        for (int t = 0; t < nt; ++t) gc += sl[(size_t)t * ws.slab_stride + e];
        if (layer >= 0 && d.spectral_norm) {
          const float* ctx = d.workspace + ws.ctx + ((size_t)c * 2 + q) * ws.ctx_stride;
          float ip = 0.f;
          for (int t = 0; t < nt; ++t) ip += sl[(size_t)t * ws.slab_stride + lay.P + layer];
          const float sg = ctx[layer], u = ctx[o_l + n], v = ctx[o_l + out_l + k];
          gc = gc / sg - (ip / (sg * sg)) * (u * v);
        }
        gsum += gc;
      }
    d.grad[e] = gsum;
    if (apply) {
      const adam_consts ac = load_adam_consts(d.opt);
      float pp = d.params[e], mm = d.opt.m[e], vv = d.opt.v[e];
      adam_update(pp, gsum, mm, vv, ac);
      d.params[e] = pp; d.opt.m[e] = mm; d.opt.v[e] = vv;
    }
  }
  if (blockIdx.x == 0 && d.spectral_norm) {   // the buffers after this update: g's from the last call, h's from the last call's second use
    const float* og = d.workspace + ws.sn_g;
    for (int i = threadIdx.x; i < 1 + Dg; i += blockDim.x) d.sn[i] = og[i];
    const float* ctx = d.workspace + ws.ctx + ((size_t)(calls - 1) * 2 + 1) * ws.ctx_stride + 4;
    for (int64_t i = threadIdx.x; i < gd_sn_numel(S, H, depth); i += blockDim.x) d.sn[1 + Dg + i] = ctx[i];
  }
}

// eval mode (no power iteration): reward head of models.py:177-180 on f (minus the optional log-policy offset)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_gsd_reward(il_disc_shaped_deep d, il_batch b, float* __restrict__ out_r, float* __restrict__ out_logit, const float* __restrict__ logit_offset) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = d.state_dim, Dg = gsd_dg(d), A = Dg - S, H = d.hidden, tanh_ = d.activation == 1;
  const int row0 = blockIdx.x * GD_R, tid = threadIdx.x, nrows = min(GD_R, b.n - row0);
  const GsdLayout lay = gsd_layout(S, Dg, H, DEPTH, d.spectral_norm);
  const GdLayout layh = gd_layout(S, H, DEPTH, d.spectral_norm);
  const GdLds l = gd_carve(smem, S, H, DEPTH);
  const GsdLds g = gsd_carve(smem + gd_lds_floats(S, H, DEPTH), S, Dg);
  gsd_stage_g(g, d, lay, Dg);
  gd_stage(l, layh, d.params + lay.oh, d.spectral_norm ? d.sn + 1 + Dg : nullptr);
  if (d.spectral_norm) gsd_spectral_g(g, Dg, 0, l.red);
  if (tid < GD_R) {   // kind 0 on (b, b): the rows as they are
    g.rw[8 * GD_R + tid] = 0.f; g.rw[tid] = tid < nrows ? b.terminals[(size_t)(row0 + tid) * b.ld_terminals] : 0.f; g.rw[GD_R + tid] = 0.f;
  }
  __syncthreads();
  gsd_forward<DEPTH>(l, g, layh, d, lay, b, b, 0, 1, -1, nullptr, row0, nrows, S, A, Dg, H, tanh_);
  if (tid < nrows) {
    const int row = row0 + tid;
    const float f = g.rw[5 * GD_R + tid], z = logit_offset ? f - logit_offset[row] : f, Dp = sigmoid_f(z);
    float h = d.reward_function == 1 ? -log1pf(-Dp + 1e-6f) : logf(Dp + 1e-6f) - log1pf(-Dp + 1e-6f);
    if (d.reward_function == 2) h = expf(h) * -h;
    out_r[row] = h;
    if (out_logit) out_logit[row] = z;
  }
}

static int check_gsd(const il_disc_shaped_deep* d) {
  IL_CHECK_ARG(d && d->params && d->workspace, "il_disc_shaped_deep: null descriptor field");
  const int S = d->state_dim, Dg = d->state_only ? S : S + d->action_dim;
  IL_CHECK_ARG(S >= 1 && S <= 128 && Dg <= 256 && d->hidden >= 2 && d->hidden <= 128, "il_disc_shaped_deep: unsupported dims (state=%d <= 128, input=%d <= 256, hidden=%d <= 128)", S, Dg, d->hidden);
  IL_CHECK_ARG(d->depth >= 0 && d->depth <= 2 && (d->activation == 0 || d->activation == 1), "il_disc_shaped_deep: depth must be 1 or 2 (0 = 1) and activation 0 (relu) or 1 (tanh)");
  IL_CHECK_ARG(d->reward_function >= 0 && d->reward_function <= 2 && d->loss_function >= IL_LOSS_BCE && d->loss_function <= IL_LOSS_MIXUP,
               "il_disc_shaped_deep: reward_function in {0,1,2}, loss_function BCE, PUGAIL or Mixup");
  IL_CHECK_ARG(!d->spectral_norm || d->sn, "il_disc_shaped_deep: spectral-norm buffers missing");
  if (d->workspace_floats < gsd_ws(S, Dg, d->hidden, gsd_depth(*d), d->batch).total) return il_set_error(IL_ERR_WORKSPACE, "il_disc_shaped_deep: workspace too small");
  return IL_OK;
}
static int gsd_ensure_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  if (bytes > 160 * 1024) return il_set_error(IL_ERR_UNSUPPORTED, "il_disc_shaped_deep: this shape needs %zu bytes of LDS (> 160 KiB per CU)", bytes);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}

extern "C" int il_gail_shaped_deep_step(const il_disc_shaped_deep* d, const il_batch* pol, const il_batch* exp, const float* eps_gp, const il_gail_extra* extra, uint32_t flags,
                                        il_stream_t stream_) {
  IL_NO_GATHER(pol, "il_gail_shaped_deep_step"); IL_NO_GATHER(exp, "il_gail_shaped_deep_step");
  if (int rc = check_gsd(d)) return rc;
  IL_CHECK_ARG(pol && exp && pol->n == d->batch && exp->n == d->batch && d->grad && d->opt.m && d->opt.v && d->opt.step, "il_gail_shaped_deep_step: bad batches / optimiser state");
  IL_CHECK_ARG(pol->next_states && pol->terminals && exp->next_states && exp->terminals && pol->weights && exp->weights,
               "il_gail_shaped_deep_step: the shaping term needs next_states, terminals and weights");
  il_gail_extra x = {};
  if (extra) x = *extra;
  IL_CHECK_ARG(d->loss_function != IL_LOSS_MIXUP || (!x.logit_offset_policy && !x.logit_offset_expert),
               "il_gail_shaped_deep_step: with Mixup the log-policy offset belongs to the mixed batch (logit_offset_mix)");
  const int S = d->state_dim, Dg = d->state_only ? S : S + d->action_dim, depth = gsd_depth(*d), nt = ceil_div(d->batch, GD_R);
  const size_t lds = gsd_lds_floats(S, Dg, d->hidden, depth) * sizeof(float);
  const auto grad = depth == 2 ? k_gsd_grad<2> : k_gsd_grad<1>;
  const auto reduce = depth == 2 ? k_gsd_reduce<2> : k_gsd_reduce<1>;
  if (int rc = gsd_ensure_lds((const void*)grad, lds)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  if (d->loss_function == IL_LOSS_PUGAIL && d->pu_clamped) {   // finite nonnegative_margin: a value pass (logits only) ahead of the gradient pass, which reads the clamp decision
    IL_CHECK_ARG(d->nonnegative_margin >= 0.f, "il_gail_shaped_deep_step: nonnegative_margin must be >= 0");
    { IL_TRACE("k_gsd_grad", st); grad<<<dim3(nt, 2, 1), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, 1); }
  }
  { IL_TRACE("k_gsd_grad", st); grad<<<dim3(nt, gsd_calls(*d), 2), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, 0); }
  const int64_t P = gsd_layout(S, Dg, d->hidden, depth, d->spectral_norm).P;
  { IL_TRACE("k_gsd_reduce", st); reduce<<<(int)((P + 255) / 256), 256, 0, st>>>(*d, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1); }
  IL_CHECK_LAUNCH("il_gail_shaped_deep_step");
  return IL_OK;
}

extern "C" int il_gail_shaped_deep_reward(const il_disc_shaped_deep* d, const il_batch* b, float* out_rewards, float* out_logits, const float* logit_offset, il_stream_t stream_) {
  IL_NO_GATHER(b, "il_gail_shaped_deep_reward");
  if (int rc = check_gsd(d)) return rc;
  IL_CHECK_ARG(b && out_rewards && b->n > 0 && b->next_states && b->terminals, "il_gail_shaped_deep_reward: bad arguments (next_states and terminals are inputs of the shaping term)");
  const int S = d->state_dim, Dg = d->state_only ? S : S + d->action_dim, depth = gsd_depth(*d);
  const size_t lds = gsd_lds_floats(S, Dg, d->hidden, depth) * sizeof(float);
  const auto reward = depth == 2 ? k_gsd_reward<2> : k_gsd_reward<1>;
  if (int rc = gsd_ensure_lds((const void*)reward, lds)) return rc;
  { IL_TRACE("k_gsd_reward", (hipStream_t)stream_); reward<<<ceil_div(b->n, GD_R), 256, lds, (hipStream_t)stream_>>>(*d, *b, out_rewards, out_logits, logit_offset); }
  IL_CHECK_LAUNCH("il_gail_shaped_deep_reward");
  return IL_OK;
}
