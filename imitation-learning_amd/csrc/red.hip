// Random Expert Distillation (reference models.py:252-284 REDDiscriminator, training.py:68-75 target_estimation_update) for gfx950.
//
// predictor / frozen target: Linear(D,H) -> ReLU -> Linear(H,D) on x = cat(state, action) (D <= 128, H <= 256; 32 by default).
// The networks are a few KB, so the update is launch-latency bound; it is two launches:
//   k_red_grad   one workgroup per 32-row tile: both forwards, loss, predictor backward; per-tile gradient slab (no atomics,
//                deterministic), plain VALU dot products over LDS-resident activations (the matrices are far too small for MFMA
//                tiles to pay: 32x32x24 at the default sizes);
//   k_red_apply  slab sum -> grad (+ AdamW unless IL_FLAG_GRADS_ONLY).
//   k_red_eval   eval forward: reward = exp(-sigma_1 * mean_c (pred - target)^2) and / or the raw embeddings (for set_sigma).
#include "il_common.hpp"

#define RT 32  // rows per tile

struct RedLayout { int64_t oW1, ob1, oW2, ob2, P; };
__host__ __device__ inline RedLayout red_layout(int D, int H) {
  RedLayout l; l.oW1 = 0; l.ob1 = (int64_t)H * D; l.oW2 = l.ob1 + H; l.ob2 = l.oW2 + (int64_t)D * H; l.P = l.ob2 + D;
  return l;
}
extern "C" int64_t il_red_numel(int32_t D, int32_t H) { return red_layout(D, H).P; }
extern "C" int64_t il_red_workspace_floats(int32_t D, int32_t H, int32_t B) {
  const int64_t nt = (B + RT - 1) / RT;
  return nt * red_layout(D, H).P + nt + 4;
}

struct RedLds { float* X; float* Hp; float* Ht; float* E; int ldx, ldh; };
__host__ __device__ inline size_t red_lds_floats(int D, int H) { return (size_t)2 * RT * (D + 1) + (size_t)2 * RT * (H + 1) + 64; }
__device__ __forceinline__ RedLds red_carve(float* smem, int D, int H) {
  RedLds l; l.ldx = D + 1; l.ldh = H + 1;
  l.X = smem; l.E = l.X + RT * l.ldx; l.Hp = l.E + RT * l.ldx; l.Ht = l.Hp + RT * l.ldh;
  return l;
}

__device__ __forceinline__ float red_in(const il_batch& b, int S, int r, int k) {
  return k < S ? b.states[(size_t)r * b.ld_states + k] : b.actions[(size_t)r * b.ld_actions + (k - S)];
}

// X tile, hidden activations of both networks, E = pred - target (all in LDS). Rows >= n are zero inputs.
__device__ __forceinline__ void red_forward_tile(const RedLds& l, const il_red& d, const il_batch& b, int row0, int D, int H) {
  const RedLayout lay = red_layout(D, H);
  const int S = d.state_dim, tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < RT * D; i += nthr) {
    const int r = i / D, k = i - r * D;
    l.X[r * l.ldx + k] = (row0 + r < b.n) ? red_in(b, S, row0 + r, k) : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < 2 * RT * H; i += nthr) {  // hidden units of predictor (first RT*H items) and target
    const int net = i >= RT * H, ii = i - net * RT * H, r = ii / H, j = ii - r * H;
    const float* P = net ? d.target : d.predictor;
    const float* w = P + lay.oW1 + (size_t)j * D; const float* x = l.X + r * l.ldx;
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 2 <= D; k += 2) { s0 = fmaf(x[k], w[k], s0); s1 = fmaf(x[k + 1], w[k + 1], s1); }
    if (k < D) s0 = fmaf(x[k], w[k], s0);
    (net ? l.Ht : l.Hp)[r * l.ldh + j] = fmaxf((s0 + s1) + P[lay.ob1 + j], 0.f);
  }
  __syncthreads();
  for (int i = tid; i < RT * D; i += nthr) {
    const int r = i / D, c = i - r * D;
    float o[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      const float* P = net ? d.target : d.predictor;
      const float* w = P + lay.oW2 + (size_t)c * H; const float* h = (net ? l.Ht : l.Hp) + r * l.ldh;
      float s0 = 0.f, s1 = 0.f;
      for (int j = 0; j + 2 <= H; j += 2) { s0 = fmaf(h[j], w[j], s0); s1 = fmaf(h[j + 1], w[j + 1], s1); }
      o[net] = (s0 + s1) + P[lay.ob2 + c];
    }
    l.E[r * l.ldx + c] = o[0] - o[1];
    if (d.out_pred && row0 + r < b.n) { d.out_pred[(size_t)(row0 + r) * D + c] = o[0]; d.out_target[(size_t)(row0 + r) * D + c] = o[1]; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_red_grad(il_red d, il_batch b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = d.state_dim + (d.state_only ? 0 : d.action_dim), H = d.hidden, B = b.n;
  const RedLayout lay = red_layout(D, H);
  const RedLds l = red_carve(smem, D, H);
  float* rowsum = l.Ht + RT * l.ldh;  // [RT] + [RT] scratch
  const int tile = blockIdx.x, row0 = tile * RT, tid = threadIdx.x, nthr = blockDim.x;
  if (tile == 0 && tid == 0) adam_tick(d.opt);
  d.out_pred = nullptr;
  red_forward_tile(l, d, b, row0, D, H);
  // loss partial and G = dLoss/dpred = 2 w_r E / (B D)   (training.py:72: (w * err^2.mean(1)).mean())
  if (tid < RT) {
    const int r = tid;
    const float w = (row0 + r < B) ? b.weights[(size_t)(row0 + r) * b.ld_weights] : 0.f;
    float s = 0.f;
    for (int c = 0; c < D; ++c) { const float e = l.E[r * l.ldx + c]; s = fmaf(e, e, s); }
    rowsum[r] = w * (s / (float)D);
    rowsum[RT + r] = (2.f * w) / ((float)B * (float)D);
  }
  __syncthreads();
  for (int i = tid; i < RT * D; i += nthr) { const int r = i / D, c = i - r * D; l.E[r * l.ldx + c] *= rowsum[RT + r]; }
  __syncthreads();
  float* slab = d.workspace + (size_t)tile * lay.P;
  if (tid == 0) {
    float s = 0.f;
    for (int r = 0; r < RT; ++r) s += rowsum[r];
    d.workspace[(size_t)gridDim.x * lay.P + tile] = s;
  }
  // dW2[c][j] = sum_r G[r][c] Hp[r][j];  db2[c] = sum_r G[r][c]
  for (int i = tid; i < D * H; i += nthr) {
    const int c = i / H, j = i - c * H;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RT; ++r) s = fmaf(l.E[r * l.ldx + c], l.Hp[r * l.ldh + j], s);
    slab[lay.oW2 + i] = s;
  }
  for (int c = tid; c < D; c += nthr) {
    float s = 0.f;
    for (int r = 0; r < RT; ++r) s += l.E[r * l.ldx + c];
    slab[lay.ob2 + c] = s;
  }
  // dHid[r][j] = [Hp > 0] sum_c G[r][c] W2[c][j]   (into Ht: the target's hidden activations are no longer needed)
  for (int i = tid; i < RT * H; i += nthr) {
    const int r = i / H, j = i - r * H;
    const float* w = d.predictor + lay.oW2 + j;
    float s = 0.f;
    for (int c = 0; c < D; ++c) s = fmaf(l.E[r * l.ldx + c], w[(size_t)c * H], s);
    l.Ht[r * l.ldh + j] = l.Hp[r * l.ldh + j] > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < H * D; i += nthr) {
    const int j = i / D, k = i - j * D;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RT; ++r) s = fmaf(l.Ht[r * l.ldh + j], l.X[r * l.ldx + k], s);
    slab[lay.oW1 + i] = s;
  }
  for (int j = tid; j < H; j += nthr) {
    float s = 0.f;
    for (int r = 0; r < RT; ++r) s += l.Ht[r * l.ldh + j];
    slab[lay.ob1 + j] = s;
  }
}

__global__ __launch_bounds__(256) void k_red_apply(il_red d, int nt, int apply, float* __restrict__ out_loss) {
  const int D = d.state_dim + (d.state_only ? 0 : d.action_dim);
  const int64_t P = red_layout(D, d.hidden).P;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < P) {
    float g = 0.f;
    for (int t = 0; t < nt; ++t) g += d.workspace[(size_t)t * P + e];
    d.grad[e] = g;
    if (apply) {
      const adam_consts ac = load_adam_consts(d.opt);
      float pp = d.predictor[e], mm = d.opt.m[e], vv = d.opt.v[e];
      adam_update(pp, g, mm, vv, ac);
      d.predictor[e] = pp; d.opt.m[e] = mm; d.opt.v[e] = vv;
    }
  }
  if (e == 0 && out_loss) {
    float s = 0.f;
    for (int t = 0; t < nt; ++t) s += d.workspace[(size_t)nt * P + t];
    out_loss[0] = s / (float)d.batch;
  }
}

__global__ __launch_bounds__(256) void k_red_eval(il_red d, il_batch b, float* __restrict__ out_reward) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = d.state_dim + (d.state_only ? 0 : d.action_dim), H = d.hidden;
  const RedLds l = red_carve(smem, D, H);
  const int row0 = blockIdx.x * RT;
  red_forward_tile(l, d, b, row0, D, H);
  if (out_reward && threadIdx.x < RT && row0 + threadIdx.x < b.n) {
    const int r = threadIdx.x;
    float s = 0.f;
    for (int c = 0; c < D; ++c) { const float e = l.E[r * l.ldx + c]; s = fmaf(e, e, s); }
    out_reward[row0 + r] = expf(-d.sigma_1 * (s / (float)D));   // models.py:280
  }
}

static int check_red(const il_red* d, const il_batch* b) {
  IL_CHECK_ARG(d && b, "il_red: null descriptor");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  IL_CHECK_ARG(D >= 1 && D <= 128 && d->hidden >= 2 && d->hidden <= 256 && d->hidden % 2 == 0, "il_red: unsupported dims (input=%d, hidden=%d)", D, d->hidden);
  IL_CHECK_ARG(d->predictor && d->target, "il_red: null parameter arena");
  IL_CHECK_ARG(b->n > 0 && b->states && (d->state_only || b->actions), "il_red: bad batch");
  return IL_OK;
}

static int red_ensure_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  if (bytes > 160 * 1024) return il_set_error(IL_ERR_UNSUPPORTED, "kernel needs %zu bytes of LDS (> 160 KiB per CU)", bytes);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}

extern "C" int il_red_step(const il_red* d, const il_batch* expert, float* out_loss, uint32_t flags, il_stream_t stream_) {
  IL_NO_GATHER(expert, "il_red_step");
  if (int rc = check_red(d, expert)) return rc;
  IL_CHECK_ARG(d->grad && d->workspace && d->opt.m && d->opt.v && d->opt.step && expert->weights, "il_red_step: null optimiser / workspace / weights");
  IL_CHECK_ARG(d->batch == expert->n, "il_red_step: descriptor batch %d != batch rows %d", d->batch, expert->n);
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim), nt = ceil_div(expert->n, RT);
  const size_t lds = red_lds_floats(D, d->hidden) * sizeof(float);
  if (int rc = red_ensure_lds((const void*)k_red_grad, lds)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  const int64_t P = red_layout(D, d->hidden).P;
  { IL_TRACE("k_red_grad", st); k_red_grad<<<nt, 256, lds, st>>>(*d, *expert); }
  { IL_TRACE("k_red_apply", st); k_red_apply<<<(int)((P + 255) / 256), 256, 0, st>>>(*d, nt, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1, out_loss); }
  IL_CHECK_LAUNCH("il_red_step");
  return IL_OK;
}

extern "C" int il_red_forward(const il_red* d, const il_batch* batch, float* out_reward, float* out_pred, float* out_target, il_stream_t stream_) {
  IL_NO_GATHER(batch, "il_red_forward");
  if (int rc = check_red(d, batch)) return rc;
  IL_CHECK_ARG(out_reward || (out_pred && out_target), "il_red_forward: nothing to write");
  IL_CHECK_ARG((out_pred == nullptr) == (out_target == nullptr), "il_red_forward: out_pred and out_target go together");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const size_t lds = red_lds_floats(D, d->hidden) * sizeof(float);
  if (int rc = red_ensure_lds((const void*)k_red_eval, lds)) return rc;
  il_red dd = *d; dd.out_pred = out_pred; dd.out_target = out_target;
  { IL_TRACE("k_red_eval", (hipStream_t)stream_); k_red_eval<<<ceil_div(batch->n, RT), 256, lds, (hipStream_t)stream_>>>(dd, *batch, out_reward); }
  IL_CHECK_LAUNCH("il_red_forward");
  return IL_OK;
}
