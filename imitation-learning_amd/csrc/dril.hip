// DRIL dropout policy ensemble (reference models.py:84-120 SoftActor with conf/algorithm/DRIL.yaml's discriminator config, trained by
// training.py:57-64 behavioural_cloning_update in train mode) for gfx950.
//
// network (`_create_fcnn`, models.py:49-70): Dropout(p_in) -> Linear(S,H) -> Dropout(p) -> act (-> Linear(H,H) -> Dropout(p) -> act when depth = 2) -> Linear(H,2A),
// act in {Tanh, ReLU}: conf/algorithm/DRIL.yaml is depth 1 / tanh, conf/optimised_hyperparameters/DRIL_{10,25}_trajectories.yaml depth 2 / relu.
// flat arena in torch order [W1 (H,S) | b1 | (Wh (H,H) | bh) | W2 (2A,H) | b2].
// Like RED this is a few-KB network, launch-latency bound, VALU dot products over LDS-resident tiles:
//   k_dril_grad    one workgroup per 32 rows: masked forward, tanh-Gaussian log-prob of the (clamped) expert action, backward, gradient slab
//   k_dril_apply   slab sum -> grad (+ AdamW)
//   k_dril_unc     8 rows x 5 ensemble members per workgroup: exp(log_prob) under 5 independent masks, unbiased variance, +-1 reward
// Dropout keep-masks are either supplied (parity tests feed the masks the reference drew) or drawn on chip (Philox, one uniform per element).
#include "il_common.hpp"

#define DT 32        // rows per tile (BC step)
#define DU 8         // rows per tile (uncertainty), x DRIL_ENSEMBLE virtual rows (40 x (S + 2H) floats of LDS: 103 KB at the largest supported dims)
#define DRIL_ENSEMBLE 5
enum { IL_STREAM_DROP_IN = 5, IL_STREAM_DROP_HID = 6, IL_STREAM_DROP_HID2 = 11 };

struct DrilLayout { int64_t oW1, ob1, oWh, obh, oW2, ob2, P; };   // (Wh, bh): the second hidden layer when depth = 2; (W2, b2): the output layer
__host__ __device__ inline DrilLayout dril_layout(int S, int A, int H, int depth) {
  DrilLayout l; l.oW1 = 0; l.ob1 = (int64_t)H * S; l.oWh = l.ob1 + H; l.obh = l.oWh + (depth == 2 ? (int64_t)H * H : 0);
  l.oW2 = l.obh + (depth == 2 ? H : 0); l.ob2 = l.oW2 + (int64_t)2 * A * H; l.P = l.ob2 + 2 * A;
  return l;
}
__host__ __device__ inline int dril_depth(const il_dril& d) { return d.depth == 2 ? 2 : 1; }
extern "C" int64_t il_dril_numel(int32_t S, int32_t A, int32_t H, int32_t depth) { return dril_layout(S, A, H, depth == 2 ? 2 : 1).P; }
extern "C" int64_t il_dril_workspace_floats(int32_t S, int32_t A, int32_t H, int32_t B, int32_t depth) {
  const int64_t nt = (B + DT - 1) / DT;
  return nt * dril_layout(S, A, H, depth == 2 ? 2 : 1).P + nt + 4;
}

// rows x (S+1) inputs | per hidden layer: rows x (H+1) activations (+ rows x (H+1) keep-scales when the backward needs them) | rows x 17 head outputs / their gradients
__host__ __device__ inline size_t dril_lds_floats(int rows, int S, int H, int depth, int keep_scales) {
  return (size_t)rows * (S + 1) + (size_t)depth * (1 + keep_scales) * rows * (H + 1) + (size_t)rows * 17 + 2 * rows;
}
struct DrilLds { float *X, *Hh[2], *Ms[2], *O, *tail; };
__device__ __forceinline__ DrilLds dril_carve(float* p, int rows, int S, int H, int depth, int keep_scales) {
  DrilLds l; l.X = p; p += rows * (S + 1);
  for (int i = 0; i < 2; ++i) {
    l.Hh[i] = i < depth ? p : nullptr; if (i < depth) p += rows * (H + 1);
    l.Ms[i] = (i < depth && keep_scales) ? p : nullptr; if (i < depth && keep_scales) p += rows * (H + 1);
  }
  l.O = p; p += rows * 17; l.tail = p;
  return l;
}

__device__ __forceinline__ float keep_scale(const float* mask, size_t idx, float p, uint64_t seed, uint32_t ctr, uint32_t stream) {
  if (p <= 0.f) return 1.f;
  const float keep = mask ? mask[idx] : (philox_uniform(seed, ctr, stream, (uint32_t)idx) >= p ? 1.f : 0.f);
  return keep * (1.f / (1.f - p));   // ATen: noise.bernoulli_(1 - p).div_(1 - p), then input * noise
}
__device__ __forceinline__ float dril_act(float z, int relu) { return relu ? fmaxf(z, 0.f) : tanhf(z); }

struct DrilMasks { const float* in; const float* h[2]; uint32_t ctr; };

// Masked forward of `rows` virtual rows (virtual row v reads batch row row0 + v / rep). Leaves x~ in X, the hidden activations in Hh[l], m_l/(1-p) in Ms[l]
// (when carved), the head in O.
// (DEPTH as a template parameter of the kernels: with a run-time depth the per-layer tables of DrilLds / DrilLayout / DrilMasks are indexed dynamically and live in scratch memory)
template <int DEPTH>
__device__ __forceinline__ void dril_forward(const il_dril& d, const il_batch& b, const DrilMasks& mk, int row0, int rows, int rep, const DrilLds& L) {
  constexpr int depth = DEPTH;
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, ldx = S + 1, ldh = H + 1, relu = d.activation == 1;
  const DrilLayout lay = dril_layout(S, A, H, depth);
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < rows * S; i += nthr) {
    const int v = i / S, k = i - v * S, r = row0 + v / rep;
    const size_t gv = (size_t)row0 * rep + v;   // global virtual row: index into the masks
    L.X[v * ldx + k] = (r < b.n) ? b.states[(size_t)r * b.ld_states + k] * keep_scale(mk.in, gv * S + k, d.p_in, d.noise_seed, mk.ctr, IL_STREAM_DROP_IN) : 0.f;
  }
  __syncthreads();
  for (int layer = 0; layer < depth; ++layer) {
    const int K = layer == 0 ? S : H;
    const int64_t oW = layer == 0 ? lay.oW1 : lay.oWh, ob = layer == 0 ? lay.ob1 : lay.obh;
    for (int i = tid; i < rows * H; i += nthr) {
      const int v = i / H, j = i - v * H;
      const float* w = d.params + oW + (size_t)j * K; const float* x = layer == 0 ? L.X + v * ldx : L.Hh[0] + v * ldh;
      float s0 = 0.f, s1 = 0.f;
      int k = 0;
      for (; k + 2 <= K; k += 2) { s0 = fmaf(x[k], w[k], s0); s1 = fmaf(x[k + 1], w[k + 1], s1); }
      if (k < K) s0 = fmaf(x[k], w[k], s0);
      const float ms = (row0 + v / rep < b.n) ? keep_scale(mk.h[layer], ((size_t)row0 * rep + v) * H + j, d.p, d.noise_seed, mk.ctr, layer == 0 ? IL_STREAM_DROP_HID : IL_STREAM_DROP_HID2)
                                              : 0.f;   // rows past the batch: no mask entry exists
      if (L.Ms[layer]) L.Ms[layer][v * ldh + j] = ms;
      L.Hh[layer][v * ldh + j] = dril_act(((s0 + s1) + d.params[ob + j]) * ms, relu);
    }
    __syncthreads();
  }
  const float* Hl = L.Hh[depth - 1];
  for (int i = tid; i < rows * 2 * A; i += nthr) {
    const int v = i / (2 * A), c = i - v * 2 * A;
    const float* w = d.params + lay.oW2 + (size_t)c * H; const float* h = Hl + v * ldh;
    float s0 = 0.f, s1 = 0.f;
    for (int j = 0; j + 2 <= H; j += 2) { s0 = fmaf(h[j], w[j], s0); s1 = fmaf(h[j + 1], w[j + 1], s1); }
    L.O[v * 17 + c] = (s0 + s1) + d.params[lay.ob2 + c];
  }
  __syncthreads();
}

// tanh-Gaussian log-density of the clamped action for virtual row v (one thread), optionally the head gradients (scaled by up = -w/B) into O.
__device__ __forceinline__ float dril_logp_row(const il_dril& d, const il_batch& b, int r, float* Orow, bool backward, float up) {
  const int A = d.action_dim;
  float sn = 0.f, sl = 0.f;
  for (int c = 0; c < A; ++c) {
    const float mean = Orow[c], lsr = Orow[A + c];
    const float ls = fminf(fmaxf(lsr, -20.f), 2.f), sd = expf(ls);
    const float a = fminf(fmaxf(b.actions[(size_t)r * b.ld_actions + c], -1.f + 1e-6f), 1.f - 1e-6f);   // models.py:98
    const float x = atanhf(a), dx = x - mean;
    sn += -(dx * dx) / (2.f * (sd * sd)) - logf(sd) - LOG_SQRT_2PI;
    sl += 2.f * (LOG_2 - x - softplus_f(-2.f * x));
    if (backward) {
      const float dmean = up * dx / (sd * sd);
      const float dstd = up * (dx * dx / (sd * sd * sd) - 1.f / sd);
      Orow[c] = dmean;
      Orow[A + c] = (lsr >= -20.f && lsr <= 2.f) ? dstd * sd : 0.f;
    }
  }
  return (0.f - sl) + sn;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_dril_grad(il_dril d, il_batch b, DrilMasks mk) {
  if (d.noise_counter) mk.ctr += *d.noise_counter;   // captured plans: the per-update part of the Philox counter lives on the device
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int depth = DEPTH;
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = b.n, ldx = S + 1, ldh = H + 1, relu = d.activation == 1;
  const DrilLayout lay = dril_layout(S, A, H, depth);
  const DrilLds L = dril_carve(smem, DT, S, H, depth, 1);
  float* X = L.X; float* O = L.O; float* lossr = L.tail;
  const int tile = blockIdx.x, row0 = tile * DT, tid = threadIdx.x, nthr = blockDim.x;
  if (tile == 0 && tid == 0) adam_tick(d.opt);
  dril_forward<DEPTH>(d, b, mk, row0, DT, 1, L);
  if (tid < DT) {
    const int r = row0 + tid;
    float l = 0.f;
    if (r < B) {
      const float w = b.weights[(size_t)r * b.ld_weights];
      const float logp = dril_logp_row(d, b, r, O + tid * 17, true, -w / (float)B);
      l = w * -logp;
    } else {
      for (int c = 0; c < 2 * A; ++c) O[tid * 17 + c] = 0.f;
    }
    lossr[tid] = l;
  }
  __syncthreads();
  float* slab = d.workspace + (size_t)tile * lay.P;
  if (tid == 0) {
    float s = 0.f;
    for (int r = 0; r < DT; ++r) s += lossr[r];
    d.workspace[(size_t)gridDim.x * lay.P + tile] = s;
  }
  const float* Hl = L.Hh[depth - 1];
  for (int i = tid; i < 2 * A * H; i += nthr) {   // dW2[c][j] = sum_r dO[r][c] h_last[r][j]
    const int c = i / H, j = i - c * H;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < DT; ++r) s = fmaf(O[r * 17 + c], Hl[r * ldh + j], s);
    slab[lay.oW2 + i] = s;
  }
  for (int c = tid; c < 2 * A; c += nthr) {
    float s = 0.f;
    for (int r = 0; r < DT; ++r) s += O[r * 17 + c];
    slab[lay.ob2 + c] = s;
  }
  __syncthreads();
  // dz_last[r][j] = (sum_c dO[r][c] W2[c][j]) act'(h_last) m_last/(1-p)   (overwrites Ms[last])
  float* dzl = L.Ms[depth - 1];
  for (int i = tid; i < DT * H; i += nthr) {
    const int r = i / H, j = i - r * H;
    const float* w = d.params + lay.oW2 + j;
    float s = 0.f;
    for (int c = 0; c < 2 * A; ++c) s = fmaf(O[r * 17 + c], w[(size_t)c * H], s);
    const float h = Hl[r * ldh + j];
    dzl[r * ldh + j] = relu ? (h > 0.f ? s * dzl[r * ldh + j] : 0.f) : s * (1.f - h * h) * dzl[r * ldh + j];
  }
  __syncthreads();
  if (depth == 2) {
    const float* H1 = L.Hh[0];
    for (int i = tid; i < H * H; i += nthr) {     // dWh[j][k] = sum_r dz2[r][j] h1[r][k]
      const int j = i / H, k = i - j * H;
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < DT; ++r) s = fmaf(dzl[r * ldh + j], H1[r * ldh + k], s);
      slab[lay.oWh + i] = s;
    }
    for (int j = tid; j < H; j += nthr) {
      float s = 0.f;
      for (int r = 0; r < DT; ++r) s += dzl[r * ldh + j];
      slab[lay.obh + j] = s;
    }
    float* dz1 = L.Ms[0];                          // dz1[r][k] = (sum_j dz2[r][j] Wh[j][k]) act'(h1) m1/(1-p)   (overwrites Ms[0])
    for (int i = tid; i < DT * H; i += nthr) {
      const int r = i / H, k = i - r * H;
      const float* w = d.params + lay.oWh + k;
      float s = 0.f;
      for (int j = 0; j < H; ++j) s = fmaf(dzl[r * ldh + j], w[(size_t)j * H], s);
      const float h = H1[r * ldh + k];
      dz1[r * ldh + k] = relu ? (h > 0.f ? s * dz1[r * ldh + k] : 0.f) : s * (1.f - h * h) * dz1[r * ldh + k];
    }
    __syncthreads();
  }
  const float* dz1 = L.Ms[0];
  for (int i = tid; i < H * S; i += nthr) {       // dW1[j][k] = sum_r dz1[r][j] x~[r][k]
    const int j = i / S, k = i - j * S;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < DT; ++r) s = fmaf(dz1[r * ldh + j], X[r * ldx + k], s);
    slab[lay.oW1 + i] = s;
  }
  for (int j = tid; j < H; j += nthr) {
    float s = 0.f;
    for (int r = 0; r < DT; ++r) s += dz1[r * ldh + j];
    slab[lay.ob1 + j] = s;
  }
}

__global__ __launch_bounds__(256) void k_dril_apply(il_dril d, int nt, int apply, float* __restrict__ out_loss) {
  const int64_t P = dril_layout(d.state_dim, d.action_dim, d.hidden, dril_depth(d)).P;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < P) {
    float g = 0.f;
    for (int t = 0; t < nt; ++t) g += d.workspace[(size_t)t * P + e];
    d.grad[e] = g;
    if (apply) {
      const adam_consts ac = load_adam_consts(d.opt);
      float pp = d.params[e], mm = d.opt.m[e], vv = d.opt.v[e];
      adam_update(pp, g, mm, vv, ac);
      d.params[e] = pp; d.opt.m[e] = mm; d.opt.v[e] = vv;
    }
  }
  if (e == 0 && out_loss) {
    float s = 0.f;
    for (int t = 0; t < nt; ++t) s += d.workspace[(size_t)nt * P + t];
    out_loss[0] = s / (float)d.batch;
  }
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_dril_unc(il_dril d, il_batch b, DrilMasks mk, float* __restrict__ out_unc, float* __restrict__ out_reward) {
  if (d.noise_counter) mk.ctr += *d.noise_counter;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int V = DU * DRIL_ENSEMBLE;
  const DrilLds L = dril_carve(smem, V, d.state_dim, d.hidden, DEPTH, 0);   // forward only: the keep-scales are not kept
  float* O = L.O; float* prob = L.tail;
  const int row0 = blockIdx.x * DU, tid = threadIdx.x;
  dril_forward<DEPTH>(d, b, mk, row0, V, DRIL_ENSEMBLE, L);
  if (tid < V) {
    const int r = row0 + tid / DRIL_ENSEMBLE;
    prob[tid] = (r < b.n) ? expf(dril_logp_row(d, b, r, O + tid * 17, false, 0.f)) : 0.f;   // models.py:106
  }
  __syncthreads();
  if (tid < DU && row0 + tid < b.n) {
    const float* p = prob + tid * DRIL_ENSEMBLE;
    float mean = 0.f;
    for (int k = 0; k < DRIL_ENSEMBLE; ++k) mean += p[k];
    mean /= (float)DRIL_ENSEMBLE;
    float var = 0.f;
    for (int k = 0; k < DRIL_ENSEMBLE; ++k) { const float e = p[k] - mean; var = fmaf(e, e, var); }
    var /= (float)(DRIL_ENSEMBLE - 1);                                                     // torch.var: unbiased
    if (out_unc) out_unc[row0 + tid] = var;
    if (out_reward) out_reward[row0 + tid] = (var <= d.q) ? 1.f : -1.f;                     // models.py:114-120
  }
}

static int check_dril(const il_dril* d, const il_batch* b) {
  IL_CHECK_ARG(d && b, "il_dril: null descriptor");
  IL_CHECK_ARG(d->state_dim >= 1 && d->state_dim <= 128 && d->action_dim >= 1 && 2 * d->action_dim <= 16 && d->hidden >= 2 && d->hidden <= 256 && d->hidden % 2 == 0,
               "il_dril: unsupported dims (state=%d, action=%d, hidden=%d)", d->state_dim, d->action_dim, d->hidden);
  IL_CHECK_ARG(d->p_in >= 0.f && d->p_in < 1.f && d->p >= 0.f && d->p < 1.f, "il_dril: dropout probabilities must be in [0,1)");
  IL_CHECK_ARG(d->depth >= 0 && d->depth <= 2 && (d->activation == 0 || d->activation == 1), "il_dril: depth must be 1 or 2 (0 = 1) and activation 0 (tanh) or 1 (relu)");
  IL_CHECK_ARG(d->params && b->n > 0 && b->states && b->actions, "il_dril: null parameters / batch");
  return IL_OK;
}

static int dril_ensure_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  if (bytes > 160 * 1024) return il_set_error(IL_ERR_UNSUPPORTED, "kernel needs %zu bytes of LDS (> 160 KiB per CU)", bytes);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}

extern "C" int il_dril_bc_step(const il_dril* d, const il_batch* expert, const float* mask_in, const float* mask_hidden, const float* mask_hidden2, uint32_t noise_offset,
                               float* out_loss, uint32_t flags, il_stream_t stream_) {
  IL_NO_GATHER(expert, "il_dril_bc_step");
  if (int rc = check_dril(d, expert)) return rc;
  IL_CHECK_ARG(d->grad && d->workspace && d->opt.m && d->opt.v && d->opt.step && expert->weights, "il_dril_bc_step: null optimiser / workspace / weights");
  IL_CHECK_ARG(d->batch == expert->n, "il_dril_bc_step: descriptor batch %d != batch rows %d", d->batch, expert->n);
  const int nt = ceil_div(expert->n, DT), depth = dril_depth(*d);
  const size_t lds = dril_lds_floats(DT, d->state_dim, d->hidden, depth, 1) * sizeof(float);
  const auto grad = depth == 2 ? k_dril_grad<2> : k_dril_grad<1>;
  if (int rc = dril_ensure_lds((const void*)grad, lds)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  const int64_t P = dril_layout(d->state_dim, d->action_dim, d->hidden, depth).P;
  const DrilMasks mk = {mask_in, {mask_hidden, mask_hidden2}, noise_offset};
  { IL_TRACE("k_dril_grad", st); grad<<<nt, 256, lds, st>>>(*d, *expert, mk); }
  { IL_TRACE("k_dril_apply", st); k_dril_apply<<<(int)((P + 255) / 256), 256, 0, st>>>(*d, nt, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1, out_loss); }
  IL_CHECK_LAUNCH("il_dril_bc_step");
  return IL_OK;
}

extern "C" int il_dril_uncertainty(const il_dril* d, const il_batch* batch, const float* mask_in, const float* mask_hidden, const float* mask_hidden2, uint32_t noise_offset,
                                   float* out_uncertainty, float* out_reward, il_stream_t stream_) {
  IL_NO_GATHER(batch, "il_dril_uncertainty");
  if (int rc = check_dril(d, batch)) return rc;
  IL_CHECK_ARG(out_uncertainty || out_reward, "il_dril_uncertainty: nothing to write");
  const size_t lds = dril_lds_floats(DU * DRIL_ENSEMBLE, d->state_dim, d->hidden, dril_depth(*d), 0) * sizeof(float);
  const auto unc = dril_depth(*d) == 2 ? k_dril_unc<2> : k_dril_unc<1>;
  if (int rc = dril_ensure_lds((const void*)unc, lds)) return rc;
  const DrilMasks mk = {mask_in, {mask_hidden, mask_hidden2}, noise_offset};
  { IL_TRACE("k_dril_unc", (hipStream_t)stream_);
    unc<<<ceil_div(batch->n, DU), 256, lds, (hipStream_t)stream_>>>(*d, *batch, mk, out_uncertainty, out_reward); }
  IL_CHECK_LAUNCH("il_dril_uncertainty");
  return IL_OK;
}
