// Replay ring kernels (reference memory.py:12-68) + bit-exact MT19937 index draws (host and device).
//
// HBM layout: [capacity][row] fp32, row = roundup4(2S + A + 5) floats so every row is a whole number of 16-B lanes:
//   states @0 | actions @S | next_states @S+A | rewards @2S+A | terminals | timeouts | weights | step | pad
// A gather moves row-granular 16-B lanes: consecutive lanes of a wave read consecutive 16-B pieces of ONE ring row
// (188-192 B contiguous per row at HalfCheetah dims), the only coalescing a random-row gather admits.
#include "il_common.hpp"
#include "mt_device.hpp"

extern "C" int32_t il_ring_row_floats(int32_t S, int32_t A) { return (2 * S + A + 5 + 3) & ~3; }

__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ ring, int64_t capacity, int row4, const int32_t* __restrict__ idx, int n, float* __restrict__ out) {
  const f32x4* src = reinterpret_cast<const f32x4*>(ring);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  const int64_t total = (int64_t)n * row4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / row4), c = (int)(i - (int64_t)r * row4);
    int64_t s = idx[r];
    s = s < 0 ? 0 : (s >= capacity ? capacity - 1 : s);  // never fault on a bad index; the host layer validates ranges
    dst[i] = src[s * row4 + c];
  }
}

extern "C" int il_replay_gather(const float* ring, int64_t capacity, int32_t row_floats, const int32_t* idx, int32_t n, float* out_rows, il_stream_t stream) {
  IL_CHECK_ARG(ring && idx && out_rows && n > 0 && capacity > 0, "il_replay_gather: bad arguments");
  IL_CHECK_ARG(row_floats % 4 == 0, "il_replay_gather: row_floats=%d must be a multiple of 4 (use il_ring_row_floats)", row_floats);
  const int row4 = row_floats / 4;
  const int64_t total = (int64_t)n * row4;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  { IL_TRACE("k_gather", (hipStream_t)stream); k_gather<<<blocks, 256, 0, (hipStream_t)stream>>>(ring, capacity, row4, idx, n, out_rows); }
  IL_CHECK_LAUNCH("il_replay_gather");
  return IL_OK;
}

__global__ __launch_bounds__(256) void k_write_rows(float* __restrict__ ring, int64_t capacity, int row, int64_t cursor, const float* __restrict__ src, int n_rows) {
  const int64_t total = (int64_t)n_rows * row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / row), c = (int)(i - (int64_t)r * row);
    ring[((cursor + r) % capacity) * row + c] = src[i];
  }
}

extern "C" int il_replay_write_rows(float* ring, int64_t capacity, int32_t row_floats, int64_t cursor, const float* rows_src, int32_t n_rows, il_stream_t stream) {
  IL_CHECK_ARG(ring && rows_src && n_rows > 0 && capacity > 0 && cursor >= 0 && cursor < capacity, "il_replay_write_rows: bad arguments");
  const int64_t total = (int64_t)n_rows * row_floats;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  { IL_TRACE("k_write_rows", (hipStream_t)stream); k_write_rows<<<blocks, 256, 0, (hipStream_t)stream>>>(ring, capacity, row_floats, cursor, rows_src, n_rows); }
  IL_CHECK_LAUNCH("il_replay_write_rows");
  return IL_OK;
}

__global__ void k_wrap_absorbing(float* __restrict__ ring, int row, int S, int A, int64_t last, int64_t cursor) {
  float* lr = ring + last * row; float* nr = ring + cursor * row;
  const int o_next = S + A, o_rew = 2 * S + A;
  for (int c = threadIdx.x; c < row; c += blockDim.x) {
    // new row: absorbing state -> absorbing state, zero action, reward 0, terminal 0, timeout 0, weight 1, step copied (memory.py:68)
    float v = 0.f;
    if (c < S) v = (c == S - 1) ? 1.f : 0.f;
    else if (c >= o_next && c < o_next + S) v = (c == o_next + S - 1) ? 1.f : 0.f;
    else if (c == o_rew + 3) v = 1.f;
    else if (c == o_rew + 4) v = lr[o_rew + 4];
    nr[c] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < S; c += blockDim.x) lr[o_next + c] = (c == S - 1) ? 1.f : 0.f;  // memory.py:67
  if (threadIdx.x == 0) lr[o_rew + 1] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// Batch mixing + constant-reward relabelling on packed rows (models.py:287-318: mix_expert_agent_transitions, RewardRelabeller).
// rows [n][row] <- expert rows for r < n_expert (every field: the reference overwrites every key); label: 0 none, 1 SQIL, 2 AdRIL.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mix_relabel(float* __restrict__ rows, const float* __restrict__ expert, int n, int row, int o_rew, int n_expert, int label,
                                                    float update_freq, float round_num, float reward_expert, float policy_trajectories, const long long* __restrict__ dyn) {
  if (dyn) {   // il_batch_mix_relabel_dyn: per-update quantities from device memory (same conversions as the host entry point)
    n_expert = (int)(dyn[0] < 0 ? 0 : (dyn[0] > n ? n : dyn[0])); round_num = (float)dyn[1]; policy_trajectories = (float)(dyn[2] > 1 ? dyn[2] : 1);
  }
  const int total = n * row;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / row, c = i - r * row;
    const bool from_expert = r < n_expert;
    float v = from_expert ? expert[i] : rows[i];
    if (label && c == o_rew) {
      if (from_expert) v = (label == 1) ? 1.f : reward_expert;
      else if (label == 1) v = 0.f;
      else {
        const float stamped = ceilf(__fdiv_rn(rows[(size_t)r * row + o_rew + 4], update_freq));  // round in which the row was collected
        v = __fdiv_rn(-1.f * ((round_num > stamped) ? 1.f : 0.f), policy_trajectories);          // -0.0 for the current round, like torch
      }
    }
    if (from_expert || (label && c == o_rew)) rows[i] = v;
  }
}

extern "C" int il_batch_mix_relabel(float* rows, const float* expert_rows, int32_t n, int32_t S, int32_t A, int32_t n_expert, int32_t label, int32_t update_freq,
                                    int64_t round_num, float reward_expert, int64_t policy_trajectories, il_stream_t stream) {
  IL_CHECK_ARG(rows && n > 0 && n_expert >= 0 && n_expert <= n && (n_expert == 0 || expert_rows), "il_batch_mix_relabel: bad arguments");
  IL_CHECK_ARG(label >= 0 && label <= 2 && (label != 2 || update_freq > 0), "il_batch_mix_relabel: label=%d update_freq=%d", label, update_freq);
  const int row = il_ring_row_floats(S, A);
  const int blocks = (n * row + 255) / 256 < 512 ? (n * row + 255) / 256 : 512;
  { IL_TRACE("k_mix_relabel", (hipStream_t)stream);
    k_mix_relabel<<<blocks, 256, 0, (hipStream_t)stream>>>(rows, expert_rows, n, row, 2 * S + A, n_expert, label, (float)update_freq, (float)round_num, reward_expert,
                                                           (float)(policy_trajectories > 1 ? policy_trajectories : 1), nullptr); }
  IL_CHECK_LAUNCH("il_batch_mix_relabel");
  return IL_OK;
}

extern "C" int il_batch_mix_relabel_dyn(float* rows, const float* expert_rows, int32_t n, int32_t S, int32_t A, int32_t label, int32_t update_freq, float reward_expert, const int64_t* dyn,
                                        il_stream_t stream) {
  IL_CHECK_ARG(rows && expert_rows && dyn && n > 0, "il_batch_mix_relabel_dyn: bad arguments");
  IL_CHECK_ARG(label >= 0 && label <= 2 && (label != 2 || update_freq > 0), "il_batch_mix_relabel_dyn: label=%d update_freq=%d", label, update_freq);
  const int row = il_ring_row_floats(S, A);
  const int blocks = (n * row + 255) / 256 < 512 ? (n * row + 255) / 256 : 512;
  { IL_TRACE("k_mix_relabel", (hipStream_t)stream);
    k_mix_relabel<<<blocks, 256, 0, (hipStream_t)stream>>>(rows, expert_rows, n, row, 2 * S + A, 0, label, (float)update_freq, 0.f, reward_expert, 1.f, (const long long*)dyn); }
  IL_CHECK_LAUNCH("il_batch_mix_relabel_dyn");
  return IL_OK;
}

extern "C" int il_replay_wrap_absorbing(float* ring, int64_t capacity, int32_t S, int32_t A, int64_t last, int64_t cursor, il_stream_t stream) {
  IL_CHECK_ARG(ring && last >= 0 && last < capacity && cursor >= 0 && cursor < capacity && last != cursor, "il_replay_wrap_absorbing: bad arguments");
  { IL_TRACE("k_wrap_absorbing", (hipStream_t)stream); k_wrap_absorbing<<<1, 64, 0, (hipStream_t)stream>>>(ring, il_ring_row_floats(S, A), S, A, last, cursor); }
  IL_CHECK_LAUNCH("il_replay_wrap_absorbing");
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// MT19937 (Matsumoto & Nishimura) + numpy legacy masked-rejection randint, as consumed by memory.py:51-56.
// state[0..623] = mt words, state[624] = position.
// ---------------------------------------------------------------------------------------------
static void mt_twist_host(uint32_t* mt) {
  for (int k = 0; k < MT_N; ++k) mt[k] = mt_mix(mt[k], mt[(k + 1) % MT_N], mt[(k + MT_M) % MT_N]);
}
extern "C" int il_mt19937_seed(uint32_t* s, uint32_t seed) {
  IL_CHECK_ARG(s, "il_mt19937_seed: null state");
  s[0] = seed;
  for (uint32_t i = 1; i < MT_N; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + i;
  s[MT_N] = MT_N;
  return IL_OK;
}
static inline uint32_t mt_next_host(uint32_t* s) {
  if (s[MT_N] >= MT_N) { mt_twist_host(s); s[MT_N] = 0; }
  return mt_temper(s[s[MT_N]++]);
}
static inline uint32_t mask_for(uint32_t rng) {
  uint32_t m = rng; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
  return m;
}
extern "C" int il_mt19937_sample_indices(uint32_t* s, int32_t n, int64_t size, int64_t idx, int32_t full, int32_t* out) {
  IL_CHECK_ARG(s && out && n >= 0 && size > 0, "il_mt19937_sample_indices: bad arguments");
  const int64_t high = full ? size : idx - 1;  // np.random.randint(0, high)
  IL_CHECK_ARG(high >= 1 && high <= 0x7FFFFFFFll, "il_mt19937_sample_indices: empty or oversized range (high=%lld)", (long long)high);
  const int64_t excl = ((idx - 1) % size + size) % size;
  IL_CHECK_ARG(!(high == 1 && excl == 0), "il_mt19937_sample_indices: the only candidate slot is the excluded one");
  const uint32_t rng = (uint32_t)(high - 1), mask = mask_for(rng);
  for (int32_t i = 0; i < n; ++i) {
    uint32_t v;
    do {
      if (rng == 0) v = 0;
      else do { v = mt_next_host(s) & mask; } while (v > rng);
    } while ((int64_t)v == excl);
    out[i] = (int32_t)v;
  }
  return IL_OK;
}

// plain np.random.randint(0, high) draws from the same stream (environments.py:113 `np.random.choice(subsample)` consumes it too)
extern "C" int il_mt19937_randint(uint32_t* s, int64_t high, int32_t n, int32_t* out) {
  IL_CHECK_ARG(s && out && n >= 0 && high >= 1 && high <= 0x7FFFFFFFll, "il_mt19937_randint: bad arguments");
  const uint32_t rng = (uint32_t)(high - 1), mask = mask_for(rng);
  for (int32_t i = 0; i < n; ++i) {
    uint32_t v = 0;
    if (rng != 0) do { v = mt_next_host(s) & mask; } while (v > rng);
    out[i] = (int32_t)v;
  }
  return IL_OK;
}

__device__ __forceinline__ void gather_rows(const float* __restrict__ ring, int64_t capacity, int row4, const int32_t* __restrict__ idx, int n, float* __restrict__ out) {
  const f32x4* src = reinterpret_cast<const f32x4*>(ring);
  f32x4* dst = reinterpret_cast<f32x4*>(out);
  for (int i = threadIdx.x; i < n * row4; i += blockDim.x) {
    const int r = i / row4, c = i - r * row4;
    int64_t s = idx[r];
    s = s < 0 ? 0 : (s >= capacity ? capacity - 1 : s);
    dst[i] = src[s * row4 + c];
  }
}

// memory.sample(B) for the agent ring THEN the expert ring (the order train.py:173 consumes the stream) + both gathers: one launch
__global__ __launch_bounds__(256) void k_sample2(uint32_t* __restrict__ state, int n, const int64_t* __restrict__ rs_a, const float* __restrict__ ring_a, int64_t cap_a, int row4_a,
                                                  int32_t* __restrict__ idx_a, float* __restrict__ rows_a, const int64_t* __restrict__ rs_b, const float* __restrict__ ring_b,
                                                  int64_t cap_b, int row4_b, int32_t* __restrict__ idx_b, float* __restrict__ rows_b, long long* __restrict__ sync, int resident) {
  __shared__ MtShared sh;
  mt_sample_update(sh, state, n, rs_a, idx_a, rs_b, idx_b, sync, resident);
  if (rows_a) gather_rows(ring_a, cap_a, row4_a, idx_a, n, rows_a);
  if (ring_b && rows_b) gather_rows(ring_b, cap_b, row4_b, idx_b, n, rows_b);
}

extern "C" int il_replay_draw_resident(uint32_t* state_dev, int32_t n, const int64_t* ring_state_a, int32_t* idx_a, const int64_t* ring_state_b, int32_t* idx_b, int64_t* sync,
                                       il_stream_t stream) {
  IL_CHECK_ARG(state_dev && ring_state_a && idx_a && n > 0 && sync && (!ring_state_b == !idx_b), "il_replay_draw_resident: bad arguments (the il_sync counters are required)");
  { IL_TRACE("k_sample2", stream); k_sample2<<<1, 256, 0, (hipStream_t)stream>>>(state_dev, n, ring_state_a, nullptr, 0, 0, idx_a, nullptr, ring_state_b, nullptr, 0, 0,
                                                                            idx_b, nullptr, (long long*)sync, 1); }
  IL_CHECK_LAUNCH("il_replay_draw_resident");
  return IL_OK;
}

// both gathers of an update in one launch, one 16-byte lane per thread (the index draw above is a single-workgroup kernel)
__global__ __launch_bounds__(256) void k_gather2(const float* __restrict__ ring_a, int64_t cap_a, int row4_a, const int32_t* __restrict__ idx_a, float* __restrict__ rows_a,
                                                 const float* __restrict__ ring_b, int64_t cap_b, int row4_b, const int32_t* __restrict__ idx_b, float* __restrict__ rows_b, int n, long long* __restrict__ sync) {
  const int na = n * row4_a, nb = ring_b ? n * row4_b : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += gridDim.x * blockDim.x) {
    const bool isb = i >= na;
    const int ii = isb ? i - na : i, row4 = isb ? row4_b : row4_a;
    const int r = ii / row4, c = ii - r * row4;
    int64_t s = (isb ? idx_b : idx_a)[r];
    const int64_t cap = isb ? cap_b : cap_a;
    s = s < 0 ? 0 : (s >= cap ? cap - 1 : s);
    const f32x4 v = reinterpret_cast<const f32x4*>(isb ? ring_b : ring_a)[s * row4 + c];
    if (!sync) reinterpret_cast<f32x4*>(isb ? rows_b : rows_a)[ii] = v;
    else if (isb) wstore4<true>(rows_b, 4 * (int64_t)ii, v);   // device hand-off ([IL_SYNC_ROWS]): the consumer is a resident launch of another stream - written through (round 6,
    else wstore4<true>(rows_a, 4 * (int64_t)ii, v);            // profiles/r06_soak_under_load.md), in memory once sync_signal's drain has passed
  }
  if (sync) sync_signal(sync + IL_SYNC_ROWS);   // rows of this workgroup are in place
}

// population axis: one workgroup per learner draws (own MT19937 state), then one launch gathers every learner's rows
__global__ __launch_bounds__(256) void k_sample2_pop(const il_sample_args* __restrict__ aL, int n) {
  il_sample_args a = aL[blockIdx.x];
  globalize(a);
  __shared__ MtShared sh;
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += 256) sh.mt[i] = a.state[i];
  if (tid == 0) { sh.pos = (int)a.state[MT_N]; sh.have_next = 0; }
  __syncthreads();
  mt_draw(sh, a.ring_state_a, n, a.idx_a);
  if (a.ring_b) { __syncthreads(); mt_draw(sh, a.ring_state_b, n, a.idx_b); }
  __syncthreads();
  for (int i = tid; i < MT_N; i += 256) a.state[i] = sh.mt[i];
  if (tid == 0) a.state[MT_N] = (uint32_t)sh.pos;
}
__global__ __launch_bounds__(256) void k_gather2_pop(const il_sample_args* __restrict__ aL, int n) {
  il_sample_args a = aL[blockIdx.y];
  globalize(a);
  const int row4_a = a.row_floats_a / 4, row4_b = a.row_floats_b / 4;
  const int na = n * row4_a, nb = a.ring_b ? n * row4_b : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += gridDim.x * blockDim.x) {
    const bool isb = i >= na;
    const int ii = isb ? i - na : i, row4 = isb ? row4_b : row4_a;
    const int r = ii / row4, c = ii - r * row4;
    int64_t s = (isb ? a.idx_b : a.idx_a)[r];
    const int64_t cap = isb ? a.capacity_b : a.capacity_a;
    s = s < 0 ? 0 : (s >= cap ? cap - 1 : s);
    reinterpret_cast<f32x4*>(isb ? a.rows_b : a.rows_a)[ii] = reinterpret_cast<const f32x4*>(isb ? a.ring_b : a.ring_a)[s * row4 + c];
  }
}
extern "C" int il_replay_sample_population(const il_sample_args* args_dev, int32_t n_learners, int32_t n, int32_t max_row_floats, il_stream_t stream) {
  IL_CHECK_ARG(args_dev && n_learners >= 1 && n > 0 && max_row_floats > 0, "il_replay_sample_population: bad arguments");
  { IL_TRACE("k_sample2", stream); k_sample2_pop<<<n_learners, 256, 0, (hipStream_t)stream>>>(args_dev, n); }
  const int lanes = 2 * n * (max_row_floats / 4);
  { IL_TRACE("k_gather2", stream); k_gather2_pop<<<dim3((lanes + 255) / 256, n_learners), 256, 0, (hipStream_t)stream>>>(args_dev, n); }
  IL_CHECK_LAUNCH("il_replay_sample_population");
  return IL_OK;
}

extern "C" int il_mt19937_sample_indices_device(uint32_t* state_dev, const int64_t* ring_state_dev, int32_t n, int32_t* out_dev, il_stream_t stream) {
  IL_CHECK_ARG(state_dev && ring_state_dev && out_dev && n > 0, "il_mt19937_sample_indices_device: bad arguments");
  { IL_TRACE("k_sample2", stream); k_sample2<<<1, 256, 0, (hipStream_t)stream>>>(state_dev, n, ring_state_dev, nullptr, 0, 0, out_dev, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0); }
  IL_CHECK_LAUNCH("il_mt19937_sample_indices_device");
  return IL_OK;
}

extern "C" int32_t il_replay_gather_workgroups(int32_t n, int32_t row_floats_a, int32_t row_floats_b) { return (n * (row_floats_a / 4) + n * (row_floats_b / 4) + 255) / 256; }

extern "C" int il_replay_sample_device(uint32_t* state_dev, int32_t n, const int64_t* ring_state_a, const float* ring_a, int64_t capacity_a, int32_t row_floats_a, int32_t* idx_a,
                                       float* rows_a, const int64_t* ring_state_b, const float* ring_b, int64_t capacity_b, int32_t row_floats_b, int32_t* idx_b, float* rows_b,
                                       int64_t* sync, il_stream_t stream) {
  IL_CHECK_ARG(state_dev && ring_state_a && ring_a && idx_a && n > 0, "il_replay_sample_device: bad arguments for ring A");
  IL_CHECK_ARG(row_floats_a % 4 == 0 && (!ring_b || (row_floats_b % 4 == 0 && ring_state_b && idx_b && (!rows_a == !rows_b))), "il_replay_sample_device: bad arguments for ring B");
  { IL_TRACE("k_sample2", stream); k_sample2<<<1, 256, 0, (hipStream_t)stream>>>(state_dev, n, ring_state_a, ring_a, capacity_a, row_floats_a / 4, idx_a, nullptr, ring_state_b, ring_b, capacity_b,
                                                                            row_floats_b / 4, idx_b, nullptr, (long long*)sync, 0); }
  if (!rows_a) { IL_CHECK_LAUNCH("il_replay_sample_device"); return IL_OK; }   // draw only
  const int lanes = n * (row_floats_a / 4) + (ring_b ? n * (row_floats_b / 4) : 0);   // il_replay_gather_workgroups() = ceil(lanes / 256)
  { IL_TRACE("k_gather2", stream); k_gather2<<<(lanes + 255) / 256, 256, 0, (hipStream_t)stream>>>(ring_a, capacity_a, row_floats_a / 4, idx_a, rows_a, ring_b, capacity_b, row_floats_b / 4, idx_b, rows_b, n, (long long*)sync); }
  IL_CHECK_LAUNCH("il_replay_sample_device");
  return IL_OK;
}
