// GAIL discriminator of any `_create_fcnn` shape without reward shaping (reference models.py:152-180, training.py:85-134) for gfx950:
//   g = [SN(Linear) - act] x depth - SN(Linear(H, 1)) on x = cat(s, a),  depth in {1, 2},  act in {relu, tanh}   (conf/hyperparameter_search_space/GAIL.yaml).
// gail.hip is the fast path of the depth-1 ReLU discriminator every shipped configuration uses; this file is the general one. Like RED / DRIL / the shaped
// discriminator it is a few-KB network written as plain VALU loops over LDS-resident tiles (16 rows per workgroup):
//   k_gd_grad    grid (tiles, calls): one workgroup = 16 rows of ONE discriminator call (policy / expert or the Mixup combination, then the gradient-penalty
//                mix). Replays the spectral-norm power iterations up to its call (torch runs one per weight access: call c has seen c + 1), scales the
//                weights in LDS, forward, closed-form backward - for the penalty the derivative of the input-gradient pass, with the phi'' terms of tanh
//                re-entering the forward graph (oracle/gail_deep.py has the derivation) - and writes one slab of dL/dW^ plus <dL/dW^, W> per layer.
//   k_gd_reduce  slab sums, spectral-norm chain rule per call  dW = G^/sigma - <G^, W>/sigma^2 u v^T,  AdamW, u / v of the last call become the buffers.
//   k_gd_reward  eval-mode forward + AIRL / GAIL / FAIRL head.
#include "il_common.hpp"
#include "gail_deep_tile.hpp"

__host__ __device__ inline int gd_depth(const il_disc_deep& d) { return d.depth == 2 ? 2 : 1; }
__host__ __device__ inline int gd_calls(const il_disc_deep& d) { return (d.loss_function == IL_LOSS_MIXUP ? 1 : 2) + (d.grad_penalty > 0.f ? 1 : 0); }
// workspace: slabs [calls][tiles][P + 4] | call context [3][4 sigmas + sn_numel] | pu [2][tiles]: per-tile sums of w softplus(z) of the policy / expert call
// (PUGAIL with a finite nonnegative_margin)
struct GdWs { int64_t slabs, ctx, ctx_stride, slab_stride, pu, total; };
__host__ __device__ inline GdWs gd_ws(int D, int H, int depth, int B) {
  GdWs w; const int64_t P = gd_layout(D, H, depth, 1).P, nt = (B + GD_R - 1) / GD_R;
  w.slab_stride = (P + 4 + 3) & ~(int64_t)3; w.slabs = 0; w.ctx = 3 * nt * w.slab_stride; w.ctx_stride = (4 + gd_sn_numel(D, H, depth) + 3) & ~(int64_t)3;
  w.pu = w.ctx + 3 * w.ctx_stride;
  w.total = w.pu + ((2 * nt + 3) & ~(int64_t)3);
  return w;
}
extern "C" int64_t il_disc_deep_numel(int32_t D, int32_t H, int32_t depth) { return gd_layout(D, H, depth == 2 ? 2 : 1, 1).P; }
extern "C" int64_t il_disc_deep_sn_numel(int32_t D, int32_t H, int32_t depth) { return gd_sn_numel(D, H, depth == 2 ? 2 : 1); }
extern "C" int64_t il_disc_deep_workspace_floats(int32_t D, int32_t H, int32_t depth, int32_t B) { return gd_ws(D, H, depth == 2 ? 2 : 1, B).total; }

extern "C" int64_t il_disc_deep_lds_bytes(int32_t D, int32_t H, int32_t depth) { return (int64_t)(gd_lds_floats(D, H, depth == 2 ? 2 : 1) * sizeof(float)); }

// (DEPTH as a template parameter: with a run-time depth the per-layer tables of GdLds / GdLayout are indexed dynamically and live in scratch memory, 320 bytes per lane)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_gd_grad(il_disc_deep d, il_batch pol, il_batch exp, const float* __restrict__ eps_gp, il_gail_extra x, int pu_value_pass) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int depth = DEPTH;
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, B = d.batch, tanh_ = d.activation == 1;
  const int tile = blockIdx.x, call = blockIdx.y, nt = gridDim.x, row0 = tile * GD_R, tid = threadIdx.x, nthr = blockDim.x;
  const int nrows = min(GD_R, B - row0);
  const bool mixup = d.loss_function == IL_LOSS_MIXUP;
  const int kind = mixup ? (call == 0 ? 3 : 2) : call;   // 0 policy, 1 expert, 2 gradient-penalty mix, 3 mixup mix
  const GdLayout lay = gd_layout(D, H, depth, d.spectral_norm);
  const GdWs ws = gd_ws(D, H, depth, B);
  const GdLds l = gd_carve(smem, D, H, depth);
  float* slab = d.workspace + ws.slabs + ((size_t)call * nt + tile) * ws.slab_stride;
  if (tile == 0 && call == 0 && tid == 0 && !pu_value_pass) adam_tick(d.opt);
  gd_stage(l, lay, d.params, d.spectral_norm ? d.sn : nullptr);
  if (d.spectral_norm) gd_spectral(l, lay, call + 1);
  if (tile == 0 && !pu_value_pass) {   // this call's (sigma, u, v) for the reduce kernel
    float* c = d.workspace + ws.ctx + (size_t)call * ws.ctx_stride;
    if (tid < 4) c[tid] = tid <= depth ? l.sc[tid] : 1.f;
    int64_t o = 4;
    for (int i = 0; i <= depth; ++i) {
      for (int e = tid; e < lay.out[i]; e += nthr) c[o + e] = l.u[i][e];
      o += lay.out[i];
      for (int e = tid; e < lay.in[i]; e += nthr) c[o + e] = l.v[i][e];
      o += lay.in[i];
    }
  }
  // ---- rows (and per-row weights) of this call
  const uint32_t ctr = d.noise_counter ? *d.noise_counter : 0u;
  auto mix_eps = [&](int row) -> float {
    const float* given = kind == 2 ? eps_gp : x.eps_mix;
    return given ? given[row] : philox_uniform(d.noise_seed, ctr, kind == 2 ? IL_STREAM_GP : IL_STREAM_MIX, (uint32_t)row);
  };
  for (int i = tid; i < GD_R * D; i += nthr) {
    const int r = i / D, k = i - r * D; float xv = 0.f;
    if (r < nrows) {
      const int row = row0 + r;
      float xp = 0.f, xe = 0.f;
      if (kind != 1) xp = k < S ? pol.states[(size_t)row * pol.ld_states + k] : pol.actions[(size_t)row * pol.ld_actions + k - S];
      if (kind != 0) xe = k < S ? exp.states[(size_t)row * exp.ld_states + k] : exp.actions[(size_t)row * exp.ld_actions + k - S];
      if (kind >= 2) { const float e = mix_eps(row); xv = e * xe + (1.f - e) * xp; } else xv = kind == 0 ? xp : xe;
    }
    l.X[r * l.ldx + k] = xv;
  }
  float* wt = l.row + GD_R; float* dzr = l.row + 2 * GD_R; float* crow = l.row + 3 * GD_R;
  if (tid < GD_R) {
    const int row = row0 + tid; float w = 0.f;
    if (tid < nrows) {
      const float wp = kind != 1 ? pol.weights[(size_t)row * pol.ld_weights] : 0.f, we = kind != 0 ? exp.weights[(size_t)row * exp.ld_weights] : 0.f;
      if (kind >= 2) { const float e = mix_eps(row); w = e * we + (1.f - e) * wp; } else w = kind == 0 ? wp : we;
    }
    wt[tid] = w;
  }
  __syncthreads();
  gd_forward(l, lay, H, tanh_);
  const float fB = (float)B;
  if (kind != 2) {
    // ---- first-order call: dL/dz = w (c_sig sigmoid(z) - c_lab) / B (+ entropy bonus), then plain back-propagation
    if (tid < GD_R) {
      const int row = row0 + min(tid, nrows - 1);
      const float* off = kind == 0 ? x.logit_offset_policy : (kind == 1 ? x.logit_offset_expert : (kind == 3 ? x.logit_offset_mix : nullptr));
      const float f = l.row[tid], z = off ? f - off[row] : f;
      const bool pu = d.loss_function == IL_LOSS_PUGAIL;
      if (pu_value_pass) {   // training.py:100-102 with a finite margin: this launch (policy and expert call, the same power iterations as the real one) only leaves the
        // per-tile sums of w softplus(z) = w bce(z, 0); the gradient launch reads them all and decides, every workgroup the same way (gail.hip does the same)
        const float ws_ = tid < nrows ? wt[tid] * softplus_f(z) : 0.f;
        float part = 0.f;
        for (int o = 0; o < GD_R; ++o) part += __shfl(ws_, o, GD_R);
        if (tid == 0) d.workspace[ws.pu + (size_t)kind * nt + tile] = part;
      }
      float pu_on = 1.f;   // 1: the clamp passes the gradient (always, with nonnegative_margin = inf)
      if (pu && d.pu_clamped && !pu_value_pass) {
        float se = 0.f, sp = 0.f;
        for (int t = 0; t < nt; ++t) { sp += d.workspace[ws.pu + t]; se += d.workspace[ws.pu + nt + t]; }
        pu_on = d.pos_class_prior * (se / fB) - sp / fB >= -d.nonnegative_margin ? 1.f : 0.f;   // torch.clamp(min = -margin): gradient where the input is not below the bound
      }
      const float c_sig = pu ? (kind == 1 ? (1.f + pu_on) * d.pos_class_prior : -pu_on) : 1.f;
      const float c_lab = kind == 3 ? mix_eps(row) : (kind == 1 ? (pu ? d.pos_class_prior : 1.f) : 0.f);
      const float p = sigmoid_f(z), w = wt[tid];
      float dz = tid < nrows ? w * (c_sig * p - c_lab) / fB : 0.f;
      if (d.entropy_bonus > 0.f && tid < nrows) dz += d.entropy_bonus * w * z * p * (1.f - p) / fB;
      dzr[tid] = dz;
    }
    if (pu_value_pass) return;   // uniform: every thread of the workgroup leaves here
    __syncthreads();
    gd_backprop<DEPTH>(l, lay, slab, dzr, H, tanh_);
  } else {
    // ---- gradient penalty: L = sum_r c_r ||g_r||^2, c_r = lambda w_r / B, g = dD/dx (see oracle/gail_deep.py)
    if (tid < GD_R) crow[tid] = tid < nrows ? d.grad_penalty * wt[tid] / fB : 0.f;
    __syncthreads();
    gd_input_grad_u<DEPTH>(l, lay, H, tanh_);
    // sbar_0 = 2 c g,  g = u_0 W^_0
    for (int e = tid; e < GD_R * D; e += nthr) {
      const int r = e / D, k = e - r * D;
      float s = 0.f;
      for (int n = 0; n < H; ++n) s = fmaf(l.U[0][r * l.ldh + n], l.W[0][n * l.ldw[0] + k], s);
      l.SB[r * l.ldx + k] = 2.f * crow[r] * s;
    }
    __syncthreads();
    gd_input_grad_backward<DEPTH>(l, lay, slab, D, H, tanh_);
  }
  __syncthreads();
  gd_inner_products<DEPTH>(l, lay, slab, slab + lay.P);   // <G^_l, W_l> of this tile, behind the slab
}

// grid = ceil(P / 256): one parameter per thread; slabs summed in (call, tile) order (deterministic)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_gd_reduce(il_disc_deep d, int apply) {
  constexpr int depth = DEPTH;
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, B = d.batch;
  const GdLayout lay = gd_layout(D, H, depth, d.spectral_norm);
  const GdWs ws = gd_ws(D, H, depth, B);
  const int nt = (B + GD_R - 1) / GD_R, calls = gd_calls(d);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < lay.P) {
    int layer = -1, n = 0, k = 0, out_l = 0;
    int64_t o_l = 4, o_run = 4;   // where the layer's u | v sit in a call's context (found with compile-time indices: a table looked up by `layer` would live in scratch memory)
#pragma unroll
    for (int i = 0; i <= depth; ++i) {
      if (e >= lay.oW[i] && e < lay.oW[i] + (int64_t)lay.out[i] * lay.in[i]) { layer = i; n = (int)((e - lay.oW[i]) / lay.in[i]); k = (int)((e - lay.oW[i]) % lay.in[i]); o_l = o_run; out_l = lay.out[i]; }
      o_run += lay.out[i] + lay.in[i];
    }
    float g = 0.f;
    for (int c = 0; c < calls; ++c) {
      const float* sl = d.workspace + ws.slabs + (size_t)c * nt * ws.slab_stride;
      float gc = 0.f;
      for (int t = 0; t < nt; ++t) gc += sl[(size_t)t * ws.slab_stride + e];
      if (layer >= 0 && d.spectral_norm) {
        const float* ctx = d.workspace + ws.ctx + (size_t)c * ws.ctx_stride;
        float ip = 0.f;
        for (int t = 0; t < nt; ++t) ip += sl[(size_t)t * ws.slab_stride + lay.P + layer];
        const float sg = ctx[layer], u = ctx[o_l + n], v = ctx[o_l + out_l + k];
        gc = gc / sg - (ip / (sg * sg)) * (u * v);
      }
      g += gc;
    }
    d.grad[e] = g;
    if (apply) {
      const adam_consts ac = load_adam_consts(d.opt);
      float pp = d.params[e], mm = d.opt.m[e], vv = d.opt.v[e];
      adam_update(pp, g, mm, vv, ac);
      d.params[e] = pp; d.opt.m[e] = mm; d.opt.v[e] = vv;
    }
  }
  if (blockIdx.x == 0 && d.spectral_norm) {   // the buffers after this update = the last call's iteration
    const float* ctx = d.workspace + ws.ctx + (size_t)(calls - 1) * ws.ctx_stride + 4;
    for (int64_t i = threadIdx.x; i < gd_sn_numel(D, H, depth); i += blockDim.x) d.sn[i] = ctx[i];
  }
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_gd_reward(il_disc_deep d, il_batch b, float* __restrict__ out_r, float* __restrict__ out_logit, const float* __restrict__ logit_offset) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int depth = DEPTH;
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, tanh_ = d.activation == 1;
  const int row0 = blockIdx.x * GD_R, tid = threadIdx.x, nrows = min(GD_R, b.n - row0);
  const GdLayout lay = gd_layout(D, H, depth, d.spectral_norm);
  const GdLds l = gd_carve(smem, D, H, depth);
  gd_stage(l, lay, d.params, d.spectral_norm ? d.sn : nullptr);
  if (d.spectral_norm) gd_spectral(l, lay, 0);
  for (int i = tid; i < GD_R * D; i += blockDim.x) {
    const int r = i / D, k = i - r * D;
    l.X[r * l.ldx + k] = r < nrows ? (k < S ? b.states[(size_t)(row0 + r) * b.ld_states + k] : b.actions[(size_t)(row0 + r) * b.ld_actions + k - S]) : 0.f;
  }
  __syncthreads();
  gd_forward(l, lay, H, tanh_);
  if (tid < nrows) {
    const float f = l.row[tid], z = logit_offset ? f - logit_offset[row0 + tid] : f, Dp = sigmoid_f(z);
    float h = d.reward_function == 1 ? -log1pf(-Dp + 1e-6f) : logf(Dp + 1e-6f) - log1pf(-Dp + 1e-6f);
    if (d.reward_function == 2) h = expf(h) * -h;
    out_r[row0 + tid] = h;
    if (out_logit) out_logit[row0 + tid] = z;
  }
}

static int check_gd(const il_disc_deep* d) {
  IL_CHECK_ARG(d, "il_disc_deep: null descriptor");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  IL_CHECK_ARG(D >= 1 && D <= 128 && d->hidden >= 2 && d->hidden <= 128, "il_disc_deep: unsupported dims (input=%d <= 128, hidden=%d <= 128)", D, d->hidden);
  IL_CHECK_ARG(d->depth >= 0 && d->depth <= 2 && (d->activation == 0 || d->activation == 1), "il_disc_deep: depth must be 1 or 2 (0 = 1) and activation 0 (relu) or 1 (tanh)");
  IL_CHECK_ARG(d->params && (!d->spectral_norm || d->sn), "il_disc_deep: null parameter / spectral-norm arena");
  IL_CHECK_ARG(d->reward_function >= 0 && d->reward_function <= 2, "il_disc_deep: reward_function must be 0 (AIRL), 1 (GAIL) or 2 (FAIRL)");
  return IL_OK;
}
static int gd_ensure_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  if (bytes > 160 * 1024) return il_set_error(IL_ERR_UNSUPPORTED, "il_disc_deep: this shape needs %zu bytes of LDS (> 160 KiB per CU)", bytes);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}

extern "C" int il_gail_deep_step(const il_disc_deep* d, const il_batch* pol, const il_batch* exp, const float* eps_gp, const il_gail_extra* extra, uint32_t flags,
                                 il_stream_t stream_) {
  IL_NO_GATHER(pol, "il_gail_deep_step"); IL_NO_GATHER(exp, "il_gail_deep_step");
  if (int rc = check_gd(d)) return rc;
  IL_CHECK_ARG(pol && exp && pol->n == d->batch && exp->n == d->batch, "il_gail_deep_step: policy/expert batches must both have %d rows", d->batch);
  IL_CHECK_ARG(d->grad && d->workspace && d->opt.m && d->opt.v && d->opt.step && pol->weights && exp->weights, "il_gail_deep_step: null optimiser / workspace / weights");
  il_gail_extra x = {};
  if (extra) x = *extra;
  IL_CHECK_ARG(d->loss_function != IL_LOSS_MIXUP || (!x.logit_offset_policy && !x.logit_offset_expert), "il_gail_deep_step: with Mixup the log-policy offset belongs to the mixed batch (logit_offset_mix)");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim), depth = gd_depth(*d), nt = ceil_div(d->batch, GD_R);
  const size_t lds = gd_lds_floats(D, d->hidden, depth) * sizeof(float);
  const auto grad = depth == 2 ? k_gd_grad<2> : k_gd_grad<1>;
  const auto reduce = depth == 2 ? k_gd_reduce<2> : k_gd_reduce<1>;
  if (int rc = gd_ensure_lds((const void*)grad, lds)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  if (d->loss_function == IL_LOSS_PUGAIL && d->pu_clamped) {   // finite nonnegative_margin: a value pass (logits only) ahead of the gradient pass, which reads the clamp decision
    IL_CHECK_ARG(d->nonnegative_margin >= 0.f, "il_gail_deep_step: nonnegative_margin must be >= 0");
    { IL_TRACE("k_gd_grad", st); grad<<<dim3(nt, 2), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, 1); }
  }
  { IL_TRACE("k_gd_grad", st); grad<<<dim3(nt, gd_calls(*d)), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, 0); }
  const int64_t P = gd_layout(D, d->hidden, depth, d->spectral_norm).P;
  { IL_TRACE("k_gd_reduce", st); reduce<<<(int)((P + 255) / 256), 256, 0, st>>>(*d, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1); }
  IL_CHECK_LAUNCH("il_gail_deep_step");
  return IL_OK;
}

extern "C" int il_gail_deep_reward(const il_disc_deep* d, const il_batch* b, float* out_rewards, float* out_logits, const float* logit_offset, il_stream_t stream_) {
  IL_NO_GATHER(b, "il_gail_deep_reward");
  if (int rc = check_gd(d)) return rc;
  IL_CHECK_ARG(b && out_rewards && b->n > 0, "il_gail_deep_reward: bad arguments");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const size_t lds = gd_lds_floats(D, d->hidden, gd_depth(*d)) * sizeof(float);
  const auto reward = gd_depth(*d) == 2 ? k_gd_reward<2> : k_gd_reward<1>;
  if (int rc = gd_ensure_lds((const void*)reward, lds)) return rc;
  { IL_TRACE("k_gd_reward", stream_); reward<<<ceil_div(b->n, GD_R), 256, lds, (hipStream_t)stream_>>>(*d, *b, out_rewards, out_logits, logit_offset); }
  IL_CHECK_LAUNCH("il_gail_deep_reward");
  return IL_OK;
}
