// Random Expert Distillation (reference models.py:252-284 REDDiscriminator, training.py:68-75 target_estimation_update) for gfx950.
//
// predictor / frozen target = `_create_fcnn` (models.py:49-70) on x = cat(state, action):  [Dropout(p_in)] -> Linear(D,H) -> [Dropout(p)] -> act
// (-> Linear(H,H) -> [Dropout(p)] -> act when depth = 2) -> Linear(H,D), act in {ReLU, Tanh}. Only the predictor has the dropout layers (models.py:256-257),
// and only in train mode: target_estimation_update and set_sigma run before train.py:147 `discriminator.eval()`, predict_reward after it.
// conf/algorithm/RED.yaml is depth 1 / relu / no dropout; conf/optimised_hyperparameters/RED_*.yaml use depth 1-2, relu / tanh, both dropouts.
// The networks are a few KB, so the update is launch-latency bound; it is two launches:
//   k_red_grad   one workgroup per 32-row tile: both forwards, loss, predictor backward; per-tile gradient slab (no atomics,
//                deterministic), plain VALU dot products over LDS-resident activations (the matrices are far too small for MFMA
//                tiles to pay: 32x32x24 at the default sizes);
//   k_red_apply  slab sum -> grad (+ AdamW unless IL_FLAG_GRADS_ONLY).
//   k_red_eval   forward (train or eval mode): reward = exp(-sigma_1 * mean_c (pred - target)^2) and / or the raw embeddings (for set_sigma).
// Dropout keep-masks: supplied by the caller (parity tests feed the masks the reference drew) or drawn on chip from the Philox stream.
#include "il_common.hpp"

#define RT 32  // rows per tile
enum { RED_STREAM_IN = 8, RED_STREAM_H1 = 9, RED_STREAM_H2 = 10 };   // Philox stream ids of the three dropout layers (il_common.hpp lists the others)

struct RedLayout { int64_t oW1, ob1, oW2, ob2, oWo, obo, P; };   // torch order: W1[H,D] b1 (W2[H,H] b2) Wo[D,H] bo
__host__ __device__ inline RedLayout red_layout(int D, int H, int depth) {
  RedLayout l; l.oW1 = 0; l.ob1 = (int64_t)H * D; l.oW2 = l.ob1 + H; l.ob2 = l.oW2 + (depth == 2 ? (int64_t)H * H : 0);
  l.oWo = l.ob2 + (depth == 2 ? H : 0); l.obo = l.oWo + (int64_t)D * H; l.P = l.obo + D;
  return l;
}
__host__ __device__ inline int red_depth(const il_red& d) { return d.depth == 2 ? 2 : 1; }
extern "C" int64_t il_red_numel(int32_t D, int32_t H, int32_t depth) { return red_layout(D, H, depth == 2 ? 2 : 1).P; }
extern "C" int64_t il_red_workspace_floats(int32_t D, int32_t H, int32_t B, int32_t depth) {
  const int64_t nt = (B + RT - 1) / RT;
  return nt * red_layout(D, H, depth == 2 ? 2 : 1).P + nt + 4;
}

// LDS: X raw rows, Xm = rows after the predictor's input dropout, E = pred - target (later dLoss/dpred); per hidden layer l: Hp[l] / Ht[l] the two
// networks' activations and M[l] the predictor's keep-scale (all [RT][H+1]); the backward reuses Ht[l] for dz_l.
struct RedLds { float *X, *Xm, *E, *Hp[2], *Ht[2], *M[2], *scratch; int ldx, ldh; };
__host__ __device__ inline size_t red_lds_floats(int D, int H, int depth) { return (size_t)3 * RT * (D + 1) + (size_t)3 * depth * RT * (H + 1) + 2 * RT + 64; }
__device__ __forceinline__ RedLds red_carve(float* smem, int D, int H, int depth) {
  RedLds l; l.ldx = D + 1; l.ldh = H + 1;
  float* p = smem;
  l.X = p; p += RT * l.ldx; l.Xm = p; p += RT * l.ldx; l.E = p; p += RT * l.ldx;
  for (int i = 0; i < 2; ++i) {
    if (i < depth) { l.Hp[i] = p; p += RT * l.ldh; l.Ht[i] = p; p += RT * l.ldh; l.M[i] = p; p += RT * l.ldh; }
    else { l.Hp[i] = l.Ht[i] = l.M[i] = nullptr; }
  }
  l.scratch = p;
  return l;
}

__device__ __forceinline__ float red_in(const il_batch& b, int S, int r, int k) {
  return k < S ? b.states[(size_t)r * b.ld_states + k] : b.actions[(size_t)r * b.ld_actions + (k - S)];
}
__device__ __forceinline__ float red_act(float z, int tanh_) { return tanh_ ? tanhf(z) : fmaxf(z, 0.f); }
__device__ __forceinline__ float red_act_grad(float h, int tanh_) { return tanh_ ? 1.f - h * h : (h > 0.f ? 1.f : 0.f); }   // in terms of the activation's output
// keep / (1 - p): ATen multiplies by noise = bernoulli(1 - p) / (1 - p)
__device__ __forceinline__ float red_keep(const float* mask, size_t idx, float p, uint64_t seed, uint32_t ctr, uint32_t stream) {
  if (p <= 0.f) return 1.f;
  const float keep = mask ? mask[idx] : (philox_uniform(seed, ctr, stream, (uint32_t)idx) >= p ? 1.f : 0.f);
  return keep * (1.f / (1.f - p));
}
// s = in[r] . w + bias: two partial sums over even / odd elements (fixed order)
__device__ __forceinline__ float red_dot(const float* x, const float* __restrict__ w, int K, float bias) {
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 2 <= K; k += 2) { s0 = fmaf(x[k], w[k], s0); s1 = fmaf(x[k + 1], w[k + 1], s1); }
  if (k < K) s0 = fmaf(x[k], w[k], s0);
  return (s0 + s1) + bias;
}

struct RedMasks { const float* in; const float* h[2]; uint32_t ctr; int training; };

// Both networks' forward on one tile. Leaves X, Xm, Hp / Ht / M per layer and E = pred - target in LDS. Rows >= n are zero inputs with zero keep-scales.
// (DEPTH as a template parameter of the kernels: with a run-time depth the per-layer tables of RedLds / RedLayout / RedMasks are indexed dynamically and live in scratch memory)
template <int DEPTH>
__device__ __forceinline__ void red_forward_tile(const RedLds& l, const il_red& d, const il_batch& b, const RedMasks& mk, int row0, int D, int H) {
  constexpr int depth = DEPTH;
  const int tanh_ = d.activation == 1;
  const RedLayout lay = red_layout(D, H, depth);
  const int S = d.state_dim, tid = threadIdx.x, nthr = blockDim.x;
  const bool drop_in = mk.training && d.p_in > 0.f, drop_h = mk.training && d.p > 0.f;
  for (int i = tid; i < RT * D; i += nthr) {
    const int r = i / D, k = i - r * D;
    const bool valid = row0 + r < b.n;
    const float x = valid ? red_in(b, S, row0 + r, k) : 0.f;
    l.X[r * l.ldx + k] = x;
    l.Xm[r * l.ldx + k] = (drop_in && valid) ? x * red_keep(mk.in, (size_t)(row0 + r) * D + k, d.p_in, d.noise_seed, mk.ctr, RED_STREAM_IN) : x;
  }
  __syncthreads();
  for (int layer = 0; layer < depth; ++layer) {
    const int K = layer == 0 ? D : H;
    const int64_t oW = layer == 0 ? lay.oW1 : lay.oW2, ob = layer == 0 ? lay.ob1 : lay.ob2;
    for (int i = tid; i < 2 * RT * H; i += nthr) {  // hidden units of predictor (first RT*H items) and target
      const int net = i >= RT * H, ii = i - net * RT * H, r = ii / H, j = ii - r * H;
      const float* P = net ? d.target : d.predictor;
      const float* in = layer == 0 ? ((net ? l.X : l.Xm) + r * l.ldx) : ((net ? l.Ht[0] : l.Hp[0]) + r * l.ldh);
      float z = red_dot(in, P + oW + (size_t)j * K, K, P[ob + j]);
      if (!net) {
        float m = 1.f;
        if (drop_h) m = (row0 + r < b.n) ? red_keep(mk.h[layer], (size_t)(row0 + r) * H + j, d.p, d.noise_seed, mk.ctr, layer == 0 ? RED_STREAM_H1 : RED_STREAM_H2) : 0.f;
        l.M[layer][r * l.ldh + j] = m;
        if (drop_h) z *= m;
      }
      (net ? l.Ht[layer] : l.Hp[layer])[r * l.ldh + j] = red_act(z, tanh_);
    }
    __syncthreads();
  }
  for (int i = tid; i < RT * D; i += nthr) {
    const int r = i / D, c = i - r * D;
    float o[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      const float* P = net ? d.target : d.predictor;
      const float* h = (net ? l.Ht[depth - 1] : l.Hp[depth - 1]) + r * l.ldh;
      const float* w = P + lay.oWo + (size_t)c * H;
      float s0 = 0.f, s1 = 0.f;
      for (int j = 0; j + 2 <= H; j += 2) { s0 = fmaf(h[j], w[j], s0); s1 = fmaf(h[j + 1], w[j + 1], s1); }
      o[net] = (s0 + s1) + P[lay.obo + c];
    }
    l.E[r * l.ldx + c] = o[0] - o[1];
    if (d.out_pred && row0 + r < b.n) { d.out_pred[(size_t)(row0 + r) * D + c] = o[0]; d.out_target[(size_t)(row0 + r) * D + c] = o[1]; }
  }
  __syncthreads();
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_red_grad(il_red d, il_batch b, RedMasks mk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = d.state_dim + (d.state_only ? 0 : d.action_dim), H = d.hidden, B = b.n;
  constexpr int depth = DEPTH;
  const int tanh_ = d.activation == 1;
  const RedLayout lay = red_layout(D, H, depth);
  const RedLds l = red_carve(smem, D, H, depth);
  float* rowsum = l.scratch;  // [RT] + [RT]
  const int tile = blockIdx.x, row0 = tile * RT, tid = threadIdx.x, nthr = blockDim.x;
  if (tile == 0 && tid == 0) adam_tick(d.opt);
  d.out_pred = nullptr;
  red_forward_tile<DEPTH>(l, d, b, mk, row0, D, H);
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
  // output layer: dWo[c][j] = sum_r G[r][c] h_last[r][j];  dbo[c] = sum_r G[r][c]
  const float* hl = l.Hp[depth - 1];
  for (int i = tid; i < D * H; i += nthr) {
    const int c = i / H, j = i - c * H;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RT; ++r) s = fmaf(l.E[r * l.ldx + c], hl[r * l.ldh + j], s);
    slab[lay.oWo + i] = s;
  }
  for (int c = tid; c < D; c += nthr) {
    float s = 0.f;
    for (int r = 0; r < RT; ++r) s += l.E[r * l.ldx + c];
    slab[lay.obo + c] = s;
  }
  // dz_last[r][j] = (sum_c G[r][c] Wo[c][j]) act'(h_last) m_last   (into Ht[last]: the target's activations are no longer needed)
  float* dzl = l.Ht[depth - 1];
  for (int i = tid; i < RT * H; i += nthr) {
    const int r = i / H, j = i - r * H;
    const float* w = d.predictor + lay.oWo + j;
    float s = 0.f;
    for (int c = 0; c < D; ++c) s = fmaf(l.E[r * l.ldx + c], w[(size_t)c * H], s);
    const float h = hl[r * l.ldh + j];
    dzl[r * l.ldh + j] = tanh_ ? (s * (1.f - h * h)) * l.M[depth - 1][r * l.ldh + j] : (h > 0.f ? s * l.M[depth - 1][r * l.ldh + j] : 0.f);
  }
  __syncthreads();
  if (depth == 2) {
    // second hidden layer: dW2[j][i] = sum_r dz2[r][j] h1[r][i]; db2; dz1[r][i] = (sum_j dz2[r][j] W2[j][i]) act'(h1) m1   (into Ht[0])
    for (int i = tid; i < H * H; i += nthr) {
      const int j = i / H, k = i - j * H;
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < RT; ++r) s = fmaf(dzl[r * l.ldh + j], l.Hp[0][r * l.ldh + k], s);
      slab[lay.oW2 + i] = s;
    }
    for (int j = tid; j < H; j += nthr) {
      float s = 0.f;
      for (int r = 0; r < RT; ++r) s += dzl[r * l.ldh + j];
      slab[lay.ob2 + j] = s;
    }
    for (int i = tid; i < RT * H; i += nthr) {
      const int r = i / H, k = i - r * H;
      const float* w = d.predictor + lay.oW2 + k;
      float s = 0.f;
      for (int j = 0; j < H; ++j) s = fmaf(dzl[r * l.ldh + j], w[(size_t)j * H], s);
      const float h = l.Hp[0][r * l.ldh + k];
      l.Ht[0][r * l.ldh + k] = tanh_ ? (s * (1.f - h * h)) * l.M[0][r * l.ldh + k] : (h > 0.f ? s * l.M[0][r * l.ldh + k] : 0.f);
    }
    __syncthreads();
  }
  const float* dz1 = l.Ht[0];
  for (int i = tid; i < H * D; i += nthr) {
    const int j = i / D, k = i - j * D;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RT; ++r) s = fmaf(dz1[r * l.ldh + j], l.Xm[r * l.ldx + k], s);
    slab[lay.oW1 + i] = s;
  }
  for (int j = tid; j < H; j += nthr) {
    float s = 0.f;
    for (int r = 0; r < RT; ++r) s += dz1[r * l.ldh + j];
    slab[lay.ob1 + j] = s;
  }
}

__global__ __launch_bounds__(256) void k_red_apply(il_red d, int nt, int apply, float* __restrict__ out_loss) {
  const int D = d.state_dim + (d.state_only ? 0 : d.action_dim);
  const int64_t P = red_layout(D, d.hidden, red_depth(d)).P;
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

template <int DEPTH>
__global__ __launch_bounds__(256) void k_red_eval(il_red d, il_batch b, RedMasks mk, float* __restrict__ out_reward) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = d.state_dim + (d.state_only ? 0 : d.action_dim), H = d.hidden;
  const RedLds l = red_carve(smem, D, H, DEPTH);
  const int row0 = blockIdx.x * RT;
  red_forward_tile<DEPTH>(l, d, b, mk, row0, D, H);
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
  IL_CHECK_ARG(d->depth >= 0 && d->depth <= 2 && (d->activation == 0 || d->activation == 1), "il_red: depth must be 1 or 2 (0 = 1) and activation 0 (relu) or 1 (tanh)");
  IL_CHECK_ARG(d->p_in >= 0.f && d->p_in < 1.f && d->p >= 0.f && d->p < 1.f, "il_red: dropout probabilities must be in [0,1)");
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

extern "C" int il_red_step(const il_red* d, const il_batch* expert, const float* mask_in, const float* mask_h1, const float* mask_h2, uint32_t noise_offset, float* out_loss,
                           uint32_t flags, il_stream_t stream_) {
  IL_NO_GATHER(expert, "il_red_step");
  if (int rc = check_red(d, expert)) return rc;
  IL_CHECK_ARG(d->grad && d->workspace && d->opt.m && d->opt.v && d->opt.step && expert->weights, "il_red_step: null optimiser / workspace / weights");
  IL_CHECK_ARG(d->batch == expert->n, "il_red_step: descriptor batch %d != batch rows %d", d->batch, expert->n);
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim), nt = ceil_div(expert->n, RT), depth = red_depth(*d);
  const size_t lds = red_lds_floats(D, d->hidden, depth) * sizeof(float);
  const auto grad = depth == 2 ? k_red_grad<2> : k_red_grad<1>;
  if (int rc = red_ensure_lds((const void*)grad, lds)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  const int64_t P = red_layout(D, d->hidden, depth).P;
  const RedMasks mk = {mask_in, {mask_h1, mask_h2}, noise_offset, 1};   // target_estimation_update runs in train mode (train.py:115-123 precede :147)
  { IL_TRACE("k_red_grad", st); grad<<<nt, 256, lds, st>>>(*d, *expert, mk); }
  { IL_TRACE("k_red_apply", st); k_red_apply<<<(int)((P + 255) / 256), 256, 0, st>>>(*d, nt, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1, out_loss); }
  IL_CHECK_LAUNCH("il_red_step");
  return IL_OK;
}

extern "C" int il_red_forward(const il_red* d, const il_batch* batch, int32_t training, const float* mask_in, const float* mask_h1, const float* mask_h2, uint32_t noise_offset,
                              float* out_reward, float* out_pred, float* out_target, il_stream_t stream_) {
  IL_NO_GATHER(batch, "il_red_forward");
  if (int rc = check_red(d, batch)) return rc;
  IL_CHECK_ARG(out_reward || (out_pred && out_target), "il_red_forward: nothing to write");
  IL_CHECK_ARG((out_pred == nullptr) == (out_target == nullptr), "il_red_forward: out_pred and out_target go together");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const size_t lds = red_lds_floats(D, d->hidden, red_depth(*d)) * sizeof(float);
  const auto eval = red_depth(*d) == 2 ? k_red_eval<2> : k_red_eval<1>;
  if (int rc = red_ensure_lds((const void*)eval, lds)) return rc;
  il_red dd = *d; dd.out_pred = out_pred; dd.out_target = out_target;
  const RedMasks mk = {mask_in, {mask_h1, mask_h2}, noise_offset, training ? 1 : 0};
  { IL_TRACE("k_red_eval", (hipStream_t)stream_); eval<<<ceil_div(batch->n, RT), 256, lds, (hipStream_t)stream_>>>(dd, *batch, mk, out_reward); }
  IL_CHECK_LAUNCH("il_red_forward");
  return IL_OK;
}
