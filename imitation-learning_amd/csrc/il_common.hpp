// Shared device helpers for libil_hip.so (gfx950 only: wave = 64, MFMA f32 16x16x4, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/il_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define IL_WAVE 64
#define IL_TILE_R 16  // batch rows per workgroup tile (= MFMA M)
#ifndef IL_CTR_STRIDE
#define IL_CTR_STRIDE 32  // 32-bit words between two per-tile arrival counters (one 128-byte line each: agent-scope atomics and polls on one line serialise at the memory side)
#endif

int il_set_error(int code, const char* fmt, ...);
#define IL_CHECK_ARG(cond, ...)                         \
  do {                                                  \
    if (!(cond)) return il_set_error(IL_ERR_ARG, __VA_ARGS__); \
  } while (0)
#define IL_CHECK_LAUNCH(name)                                                                  \
  do {                                                                                         \
    hipError_t e__ = hipGetLastError();                                                        \
    if (e__ != hipSuccess) return il_set_error(IL_ERR_HIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// MFMA v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16], exact fp32 (bitwise an fmaf chain).
//   lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
//   lane l holds   D[row = 4*(l>>4) + reg][col = l&15], reg = 0..3.
// ---------------------------------------------------------------------------------------------
// Loads through a pointer the compiler cannot prove to be global memory (every pointer that is a FIELD of a by-value descriptor struct is generic)
// become FLAT instructions. FLAT loads count against both vmcnt and lgkmcnt and may return out of order with LDS traffic, so hipcc puts
// `s_waitcnt vmcnt(0) lgkmcnt(0)` before the first use of ANY of them: a wave that requested its 16-load weight panel waited for all of it before the
// first MFMA, and LDS reads could not overlap it. An explicit global address space makes them global_load: precise vmcnt(N) waits, MFMAs start as the
// operands arrive.
template <class T>
__device__ __forceinline__ T gload(const T* p) { return *(const __attribute__((address_space(1))) T*)p; }
__device__ __forceinline__ f32x4 gload4(const float* p) { return *(const __attribute__((address_space(1))) f32x4*)p; }
// Write-through stores for data that the NEXT launch reads on other XCDs (activations / dZ in the workspace, the optimiser's p / m / v and lane-ordered copies). A plain
// store leaves a dirty line in this XCD's L2; the end-of-kernel release writes all of them back, and that write-back sits between this launch and the next one on the
// update's critical path (round 3: `sc0 sc1` stores in the dW epilogue alone: 14.76k -> 14.97k updates/s; `nt` stores: no change). `base` must be wave-uniform (it
// becomes the buffer resource), `off` is in floats (< 2^29). IL_WT_STORES=0: plain stores (A/B builds).
#ifndef IL_WT_STORES
#define IL_WT_STORES 1
#endif
typedef unsigned il_u32x4 __attribute__((ext_vector_type(4)));
template <bool WT = true>
__device__ __forceinline__ void wstore4(float* base, int64_t off, const f32x4& v) {
#if IL_WT_STORES
  if (!WT) { *reinterpret_cast<f32x4*>(base + off) = v; return; }   // (the 80-VGPR population builds keep plain stores: their chip is oversubscribed, the flush is not exposed)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);   // raw buffer, byte offsets
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(il_u32x4, v), rs, (int)(off * 4), 0, 17);       // sc0 | sc1
#else
  *reinterpret_cast<f32x4*>(base + off) = v;
#endif
}

// One float written through / read below this CU's L1 (the fence-free hand-offs of the pair-mode kernels: a few scalars per lane next to a relaxed arrival counter)
__device__ __forceinline__ void wstore1(float* base, int64_t off, float v) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)(off * 4), 0, 17);   // sc0 | sc1
}
__device__ __forceinline__ float sload1(const float* base, int64_t off) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7ffffff0, 0x00020000);
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4), 0, 17));
}

// Whole-descriptor version of the same: a descriptor fetched from memory (population axis: d = dL[blockIdx.y]) carries generic pointers, and
// ONE pending flat access (a flat_store of an activation slab is enough) makes the compiler turn every later wait into vmcnt(0) lgkmcnt(0).
// Round-tripping the fields through address space 1 lets address-space inference type every derived access as global.
// entry points whose kernels index batch rows directly
#define IL_NO_GATHER(b, who) IL_CHECK_ARG(!(b) || !(b)->gather, who ": il_batch.gather is only honoured by il_gail_disc_step, il_gail_reward and il_sac_update_gather")
// batch row -> source row of an il_batch: identity, or through the batch's index array (il_batch.gather, clamped like il_replay_gather)
__device__ __forceinline__ size_t brow(const il_batch& b, int row) {
  if (!b.gather) return (size_t)row;
  const int64_t s = b.gather[row];
  return (size_t)(s < 0 ? 0 : (s >= b.gather_capacity ? b.gather_capacity - 1 : s));
}
template <class T>
__device__ __forceinline__ T* as_global(T* p) { auto g = (__attribute__((address_space(1))) T*)p; asm volatile("" : "+s"(g)); return (T*)g; }
__device__ __forceinline__ void globalize(il_adam& o) { o.m = as_global(o.m); o.v = as_global(o.v); o.step = as_global(o.step); }
__device__ __forceinline__ void globalize(il_batch& b) {
  b.states = as_global(b.states); b.actions = as_global(b.actions); b.rewards = as_global(b.rewards); b.next_states = as_global(b.next_states);
  b.terminals = as_global(b.terminals); b.weights = as_global(b.weights); b.absorbing = as_global(b.absorbing);
  b.gather = as_global(b.gather);
}
__device__ __forceinline__ void globalize(il_sac& d) {
  d.actor = as_global(d.actor); d.critic = as_global(d.critic); d.target = as_global(d.target); d.log_alpha = as_global(d.log_alpha);
  d.actor_grad = as_global(d.actor_grad); d.critic_grad = as_global(d.critic_grad); d.alpha_grad = as_global(d.alpha_grad);
  globalize(d.actor_opt); globalize(d.critic_opt); globalize(d.alpha_opt);
  d.workspace = as_global(d.workspace); d.noise_counter = as_global(d.noise_counter); d.out_logp = as_global(d.out_logp); d.out_q = as_global(d.out_q);
  d.sync = as_global(d.sync); d.debug_masks = as_global(d.debug_masks);
}

__device__ __forceinline__ void globalize(il_disc& d) {
  d.params = as_global(d.params); d.u1 = as_global(d.u1); d.v1 = as_global(d.v1); d.u2 = as_global(d.u2); d.v2 = as_global(d.v2); d.grad = as_global(d.grad);
  globalize(d.opt); d.workspace = as_global(d.workspace); d.noise_counter = as_global(d.noise_counter); d.sync = as_global(d.sync);
}
__device__ __forceinline__ void globalize(il_gail_extra& x) {
  x.eps_mix = as_global(x.eps_mix); x.logit_offset_policy = as_global(x.logit_offset_policy); x.logit_offset_expert = as_global(x.logit_offset_expert); x.logit_offset_mix = as_global(x.logit_offset_mix);
}

__device__ __forceinline__ void globalize(il_sample_args& a) {
  a.state = as_global(a.state);
  a.ring_state_a = as_global(a.ring_state_a); a.ring_a = as_global(a.ring_a); a.idx_a = as_global(a.idx_a); a.rows_a = as_global(a.rows_a);
  a.ring_state_b = as_global(a.ring_state_b); a.ring_b = as_global(a.ring_b); a.idx_b = as_global(a.idx_b); a.rows_b = as_global(a.rows_b);
}

// x / d for 0 <= x < 2^16, 1 <= d <= 2^10 as one v_mul_hi (a runtime integer division is ~40 VALU instructions; the staging loops of the pair-mode tiles and of the
// discriminator's reward tile did twenty of them per thread: measured as 2 us of prologue). m = fastdiv_magic(d) is wave-uniform.
__device__ __forceinline__ unsigned fastdiv_magic(int d) { return 0xFFFFFFFFu / (unsigned)d + 1u; }
__device__ __forceinline__ int fastdiv(int x, unsigned m) { return m ? (int)(((unsigned long long)(unsigned)x * m) >> 32) : x; }   // (m = 0: d = 1, whose magic does not fit 32 bits)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Cross-lane sums on DPP (data-parallel primitives: the operand of a VALU op is fetched from another lane of the same 16-lane
// row, ~1 issue slot) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 -- an LDS round trip of ~100+ cycles per step.
// row_ror:n rotates within each row of 16 lanes, so four rotate-and-add steps leave the row total in every lane of the row.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes that share (lane >> 4); every lane of the group gets the total
__device__ __forceinline__ float group16_sum(float v) {
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  v = dpp_add<0x121>(v);  // row_ror:1
  return v;
}
// sum over the 64 lanes of the wave, broadcast to every lane (all lanes must be active)
__device__ __forceinline__ float wave_sum(float v) {
  v = group16_sum(v);
  const int iv = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), b = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), d = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return (a + b) + (c + d);
}

// block-wide sum, result broadcast to every thread; `red` = LDS scratch of >= 32 floats. All threads must call.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (Salmon et al. 2011) + Box-Muller; used only when the caller passes eps == NULL.
// ---------------------------------------------------------------------------------------------
struct philox_out { uint32_t v[4]; };
__host__ __device__ inline philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  philox_out o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}
__device__ __forceinline__ float u32_to_unit_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
// standard normal #idx of stream `stream_id` at update `ctr`
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t ctr, uint32_t stream_id, uint32_t idx) {
  const philox_out o = philox4x32_10(idx, ctr, stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float u1 = u32_to_unit_open(o.v[0]), u2 = u32_to_unit_open(o.v[1]);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t ctr, uint32_t stream_id, uint32_t idx) {
  const philox_out o = philox4x32_10(idx, ctr, stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (float)(o.v[0] >> 8) * (1.0f / 16777216.0f);  // [0,1) like torch.rand
}
// Beta(alpha, alpha) #idx of stream `stream_id` at update `ctr` (Mixup's coefficients, training.py:105-107, for alpha != 1; Beta(1, 1) is philox_uniform): X / (X + Y) with
// X, Y ~ Gamma(alpha) by Marsaglia & Tsang's squeeze-free form (alpha < 1: Gamma(alpha + 1) U^(1/alpha)). Attempt t of variate `which` keys the Philox counter's fourth word
// (1 + 2 t + which; word 0 is philox_uniform's), so a draw is a pure function of (seed, ctr, stream, idx, alpha): every workgroup that needs row idx's coefficient computes
// the same bits. The reference draws with torch's CPU Beta sampler: equal in distribution, not in bits.
__device__ __forceinline__ float philox_gamma(uint64_t seed, uint32_t ctr, uint32_t stream_id, uint32_t idx, float alpha, uint32_t which, float* log_out = nullptr) {
  const float a = alpha < 1.f ? alpha + 1.f : alpha, d = a - 1.f / 3.f, c = 1.f / sqrtf(9.f * d);
  float out = d;   // (never left as is: 64 attempts at a > 95 % acceptance rate)
  if (log_out) *log_out = logf(d);
  for (uint32_t t = 0; t < 64u; ++t) {
    const philox_out o = philox4x32_10(idx, ctr, stream_id, 1u + 2u * t + which, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float x = sqrtf(-2.0f * logf(u32_to_unit_open(o.v[0]))) * cosf(6.28318530717958647692f * u32_to_unit_open(o.v[1]));
    const float v0 = 1.f + c * x;
    if (v0 <= 0.f) continue;
    const float v = v0 * v0 * v0, u = u32_to_unit_open(o.v[2]);
    if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) {
      out = d * v;
      if (log_out) *log_out = logf(out) + (alpha < 1.f ? logf(u32_to_unit_open(o.v[3])) / alpha : 0.f);   // the variate's logarithm: finite where U^(1/alpha) underflows
      if (alpha < 1.f) out *= powf(u32_to_unit_open(o.v[3]), 1.f / alpha);
      break;
    }
  }
  return out;
}
__device__ __forceinline__ float philox_beta(uint64_t seed, uint32_t ctr, uint32_t stream_id, uint32_t idx, float alpha) {
  float lx, ly;
  const float x = philox_gamma(seed, ctr, stream_id, idx, alpha, 0u, &lx), y = philox_gamma(seed, ctr, stream_id, idx, alpha, 1u, &ly);
  // small alpha: both Gamma(alpha + 1) U^(1/alpha) variates can underflow to 0 in fp32 (alpha = 0.05: U^20) and 0 / 0 would put a NaN into the discriminator step; the ratio
  // of their logarithms is still exact there (the reference's sampler is guarded the same way). Any other draw keeps the plain quotient's bits.
  if (!(x + y > 0.f)) return 1.f / (1.f + expf(ly - lx));
  return x / (x + y);
}
enum { IL_STREAM_EPS_NEXT = 1, IL_STREAM_EPS_CUR = 2, IL_STREAM_GP = 3, IL_STREAM_ACT = 4, IL_STREAM_MIX = 7 };   // 5, 6: dropout masks (dril.hip)

// ---------------------------------------------------------------------------------------------
// AdamW single-tensor step, op order of torch._single_tensor_adam (fp32 tensors, python-double scalars).
// ---------------------------------------------------------------------------------------------
struct adam_consts {
  float decay;      // 1 - lr*wd (1 => no decay)
  float one_m_b1;   // 1 - beta1
  float beta2, one_m_b2;
  float step_size;  // lr / (1 - beta1^t)
  float bc2_sqrt;   // sqrt(1 - beta2^t)
  float eps;
  int has_decay;
};
__device__ __forceinline__ adam_consts make_adam_consts(double lr, double b1, double b2, double eps, double wd, int t) {
  adam_consts c;
  c.has_decay = wd != 0.0;
  c.decay = (float)(1.0 - lr * wd);
  c.one_m_b1 = (float)(1.0 - b1);
  c.beta2 = (float)b2;
  c.one_m_b2 = (float)(1.0 - b2);
  c.step_size = (float)(lr / (1.0 - pow(b1, (double)t)));
  c.bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)t));
  c.eps = (float)eps;
  return c;
}
// The kernel that produces a gradient ticks the optimiser: step[0] += 1 and the bias-correction constants of that step are
// stored next to the counter (floats at int32 offsets 4..11 of the `step` buffer, which holds >= 16 int32), so the hundreds of
// waves of the following Adam kernel load 8 floats instead of each evaluating two double-precision pow().
__device__ __forceinline__ void adam_tick(const il_adam& o) {
  const int t = o.step[0] + 1;
  o.step[0] = t;
  // beta^t: carried as running products in double (int32 slots 12..15, valid for step o.step[1]) so that the one thread that ticks does two
  // multiplications instead of two double-precision pow() (~1.5 us of a single lane, on the critical path of the workgroup that ticks).
  // A step counter that was set from outside (load_state_dict) does not match o.step[1]: recompute once with pow().
  double* pw = reinterpret_cast<double*>(o.step + 12);
  double b1t, b2t;
  if (o.step[1] == t - 1 && t > 1) { b1t = pw[0] * o.beta1; b2t = pw[1] * o.beta2; }
  else { b1t = pow(o.beta1, (double)t); b2t = pow(o.beta2, (double)t); }
  pw[0] = b1t; pw[1] = b2t; o.step[1] = t;
  float* f = reinterpret_cast<float*>(o.step) + 4;
  f[0] = (float)(1.0 - o.lr * o.weight_decay); f[1] = (float)(1.0 - o.beta1); f[2] = (float)o.beta2; f[3] = (float)(1.0 - o.beta2);
  f[4] = (float)(o.lr / (1.0 - b1t)); f[5] = (float)sqrt(1.0 - b2t); f[6] = (float)o.eps; f[7] = o.weight_decay != 0.0 ? 1.f : 0.f;
}
__device__ __forceinline__ adam_consts load_adam_consts(const il_adam& o) {
  const float* f = reinterpret_cast<const float*>(o.step) + 4;
  adam_consts c;
  c.decay = f[0]; c.one_m_b1 = f[1]; c.beta2 = f[2]; c.one_m_b2 = f[3]; c.step_size = f[4]; c.bc2_sqrt = f[5]; c.eps = f[6]; c.has_decay = f[7] != 0.f;
  return c;
}
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const adam_consts& c) {
  if (c.has_decay) p = __fmul_rn(p, c.decay);
  m = __fadd_rn(m, __fmul_rn(c.one_m_b1, __fsub_rn(g, m)));                       // lerp_
  v = __fadd_rn(__fmul_rn(v, c.beta2), __fmul_rn(__fmul_rn(c.one_m_b2, g), g));   // mul_ + addcmul_ ((value*g)*g)
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), c.bc2_sqrt), c.eps);
  p = __fsub_rn(p, __fmul_rn(c.step_size, __fdiv_rn(m, denom)));                  // addcdiv_
}

// ---------------------------------------------------------------------------------------------
// Device-side hand-off between kernels of ONE update that run on different streams (il_sync, include/il_hip.h). A cross-stream edge
// in a hipGraph costs 7-16 us of queue-to-queue signalling (rocprofv3 timeline); here the consumer kernel is already
// resident and its thread 0 polls a monotonic counter the producer's workgroups bump with one agent-scope release each.
// Bounded: after IL_SYNC_SPIN_LIMIT polls the waiter gives up, counts a timeout (the host checks it) and proceeds, so a runtime that
// serialises the two queues (a counter-collecting profiler) can never hang the GPU.
// ---------------------------------------------------------------------------------------------
#define IL_SYNC_SPIN_LIMIT (1 << 20)   // ~1 s of polling: the first replay of a freshly instantiated graph can reach the device >10 ms after the other branch's
// Every wave, in front of the barrier that precedes a workgroup's release (round 6): its own stores have been acknowledged by the L2. The release that follows is ONE thread's
// (an L2 write-back + its own vmcnt); the workgroup barrier in between does not wait for the other waves' stores in flight (workgroup scope needs no vmcnt on this target), so a
// store that reached the L2 behind the write-back stayed there, dirty, until the kernel's end - and a consumer on another XCD read the previous contents from memory.
__device__ __forceinline__ void sync_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void sync_signal(long long* ctr) {   // all threads of the workgroup, after their stores
  sync_drain_stores();
  __syncthreads();
#ifdef IL_SYNC_UNSAFE
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void sync_timed_out(long long* sync) {   // one thread: count the expired wait, and raise the host's flag if it gave us one ([IL_SYNC_HOST_FLAG])
  const long long n = __hip_atomic_fetch_add(sync + IL_SYNC_TIMEOUTS, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  // poison the learner (round 6): the optimiser epilogues of this and every later launch skip their stores until the host has seen it (il_sync_clear_poison) - an update
  // whose hand-off expired must not reach the weights
  __hip_atomic_store(sync + IL_SYNC_POISON, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const long long host = sync[IL_SYNC_HOST_FLAG];
  if (host) __hip_atomic_store((__attribute__((address_space(1))) long long*)host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // typed global: a FLAT store would make every later wait of the kernel conservative
}
// wave-uniform: has a bounded wait of this learner expired (now or in an earlier launch)? Read by the optimiser epilogues right before their stores.
__device__ __forceinline__ bool sync_poisoned(const long long* sync) {
  return sync && __hip_atomic_load(sync + IL_SYNC_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// The acquire behind a workgroup's poll: EVERY wave executes it, after the barrier that follows the poll (round 6). The agent-scope acquire is a cache invalidate
// (buffer_inv sc1) that is ordered only against the later loads of the wave that issued it and is not counted by vmcnt, so the polling wave cannot wait for its completion
// before it releases the barrier; two XCDs' L2s are not coherent inside a launch (profiles/tools/l2_stale_probe.hip: a plain re-load of a line another XCD rewrote was stale
// 20 000 times of 20 000), e.g. the discriminator's parameters, which the concurrently running k_gail_grad of the same update reads (pre-step) into the L2 the critic-loss
// launch uses. Hardening found while chasing profiles/r06_soak_under_load.md (whose main cause was on the release side: sync_drain_stores). IL_SYNC_LEADER_ACQUIRE: the
// round-5 form, for A/B builds.
__device__ __forceinline__ void sync_acquire_all() {
#ifndef IL_SYNC_UNSAFE
#ifdef IL_SYNC_LEADER_ACQUIRE
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
#endif
}
// a counter read as a wait TARGET (the epoch of this launch): an agent-scope load like the polls, never a cached one - a poll of another resident kernel that was in flight
// when this launch's start invalidated the L2 can install the line's previous value behind the invalidation, and a plain load would then hit it (IL_SYNC_PLAIN_EPOCH: the
// round-5 plain loads, for the soak A/B under profiles/)
__device__ __forceinline__ long long sync_read(const long long* sync, int which) {
#ifdef IL_SYNC_PLAIN_EPOCH
  return sync[which];
#else
  return __hip_atomic_load(sync + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// LEADER_ACQUIRE (sync_wait_leader): one invalidate, by the polling thread, in front of the barrier - ONLY where no cache of this XCD can hold a line of the guarded data that
// was fetched after this launch started and before its producer wrote it (each use says why); everywhere else every wave acquires (sync_acquire_all).
// NO_ACQUIRE (sync_wait_only): the poll and the barrier alone - for the first of two waits in a row, whose guarded data is only read behind the second one's acquire.
template <bool LEADER_ACQUIRE = false, bool NO_ACQUIRE = false>
__device__ __forceinline__ void sync_wait(long long* sync, int which, long long target, int limit = 0) {   // all threads of the workgroup, before their loads
  if (threadIdx.x == 0) {   // limit 0: the learner's own bound [IL_SYNC_SPIN] (0 there = IL_SYNC_SPIN_LIMIT), read only once a poll has failed: nothing on the fast path
    int spins = 0;
    while (__hip_atomic_load(sync + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (limit == 0) { const long long own = sync[IL_SYNC_SPIN]; limit = own > 0 ? (int)(own > 0x7fffffffLL ? 0x7fffffffLL : own) : IL_SYNC_SPIN_LIMIT; }
      if (++spins > limit) { sync_timed_out(sync); break; }
    }
#ifndef IL_SYNC_UNSAFE
    if (LEADER_ACQUIRE && !NO_ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
  if (!LEADER_ACQUIRE && !NO_ACQUIRE) sync_acquire_all();
}
__device__ __forceinline__ void sync_wait_only(long long* sync, int which, long long target) { sync_wait<false, true>(sync, which, target); }
__device__ __forceinline__ void sync_wait_leader(long long* sync, int which, long long target) { sync_wait<true>(sync, which, target); }

// ---------------------------------------------------------------------------------------------
// Stage hand-offs of the overlapped SAC branch (il_sac_update_gather_overlap; include/il_hip.h [IL_SYNC_OV_EPOCH] / [IL_SYNC_OV_TICKET]). The four launches of an update alternate
// over two streams; a launch is dispatched while its predecessor (the other stream's head) still runs and waits for it behind its independent prologue.
//   producer (ov_done, all threads, at the workgroup's end): every wave drains its stores, barrier, one agent-scope release (L2 write-back) and a fire-and-forget add on the
//            stage's ticket line - no returned value, the workgroup retires without a round trip to memory.
//   consumer (ov_wait, all threads): workgroup 0 of the waiting launch is its LEADER, the only poller of the ticket line (160 workgroups polling one line were measured to
//            slow every kernel on the chip down and to delay the tickets themselves: 11.4k against 17.4k updates/s): ticket >= the producer's grid (passed by the host with
//            the launch) or epoch >= target; it acquires, CLOSES the producer's stage (epoch = target first, ticket = 0 second) and stores `target` into the flag line of
//            every workgroup of its own launch; the others poll their own line. By the time the stage is closed every producer workgroup has arrived, and the producer
//            stage's next launch cannot have started (it follows, in its own stream, a launch that waits for this one).
//   ov_own:  a stage's epoch = the number of updates it has completed, as seen by its OWN launch: closed by its consumer before its next launch starts, so every workgroup of
//            a launch reads the same value. il_sac_overlap_enter() sets all four to [IL_SYNC_MAIN_EPOCH].
// ---------------------------------------------------------------------------------------------
#ifndef IL_OV_POLL_SLEEP
#define IL_OV_POLL_SLEEP 2
#endif
__device__ __forceinline__ long long ov_own(long long* sync, int stage) {
  return __hip_atomic_load(sync + IL_SYNC_OV_EPOCH + stage * IL_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits for launch number `target` (1-based) of `stage`, whose grid has `grid` workgroups. Workgroup 0 of the waiting launch is the LEADER: the only poller of the ticket
// line; it closes the stage and tells every other workgroup of its launch through that workgroup's own flag line.
__device__ __forceinline__ void ov_wait(long long* sync, int stage, long long target, int grid) {
  const bool leader = blockIdx.x == 0;
  long long* flags = sync + IL_SYNC_OV_FLAGS + (long long)stage * IL_OV_MAX_GRID * IL_SYNC_STRIDE;
  if (threadIdx.x == 0) {
    long long* tk = sync + IL_SYNC_OV_TICKET + stage * IL_SYNC_STRIDE;
    long long* ep = sync + IL_SYNC_OV_EPOCH + stage * IL_SYNC_STRIDE;
    long long* mine = flags + (long long)blockIdx.x * IL_SYNC_STRIDE;
    int spins = 0, limit = 0;
    bool expired = false;
    for (;;) {
      if (leader) {
        if (__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (long long)grid) break;
        if (__hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      } else if (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      __builtin_amdgcn_s_sleep(IL_OV_POLL_SLEEP);
      if (limit == 0) { const long long own = sync[IL_SYNC_SPIN]; limit = own > 0 ? (int)(own > 0x7fffffffLL ? 0x7fffffffLL : own) : IL_SYNC_SPIN_LIMIT; }
      if (++spins > limit) { sync_timed_out(sync); expired = true; break; }
    }
    if (leader && !expired && __hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // close the stage: epoch first, ticket second
      __hip_atomic_store(ep, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tk, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  sync_acquire_all();
  if (leader)   // (also after an expired wait: the followers then proceed, poisoned like the leader, instead of each running into its own bound)
    for (int w = threadIdx.x; w < (int)gridDim.x && w < IL_OV_MAX_GRID; w += blockDim.x) __hip_atomic_store(flags + (long long)w * IL_SYNC_STRIDE, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ov_done(long long* sync, int stage) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + IL_SYNC_OV_TICKET + stage * IL_SYNC_STRIDE, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

#define LOG_SQRT_2PI 0.91893853320467274178f
#define LOG_2 0.69314718055994530942f
__device__ __forceinline__ float softplus_f(float z) { return z > 20.f ? z : log1pf(expf(z)); }
__device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + expf(-z)); }

// Developer-only phase timing (build with -DIL_PHASE_STAMPS): thread 0 of one chosen block stores s_memtime at phase boundaries.
#ifdef IL_PHASE_STAMPS
static __device__ unsigned long long il_phase_stamps[64];  // one copy per translation unit (no -fgpu-rdc); read through IL_STAMP_READER(name)
#define IL_STAMP_READER(name) extern "C" int name(unsigned long long* out_host) { return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(il_phase_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 3; }
#define IL_STAMP(cond, i) do { if ((cond) && threadIdx.x == 0) il_phase_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IL_STAMP(cond, i) do { (void)(cond); } while (0)
#define IL_STAMP_READER(name)
#endif

// Developer-only workgroup timeline (build with -DIL_TIMELINE, profiles/tools/update_timeline.py): thread 0 of every workgroup of the instrumented kernels stores
// s_memrealtime (the 100 MHz device-wide counter: comparable across XCDs, unlike s_memtime) at its phase boundaries. [kernel][workgroup][slot], one copy per translation unit.
#ifdef IL_TIMELINE
#define IL_TL_K 12
#define IL_TL_WGS 512
#define IL_TL_SLOTS 8
static __device__ unsigned long long il_tl[IL_TL_K][IL_TL_WGS][IL_TL_SLOTS];
#ifndef IL_TL_STRIDE
#define IL_TL_STRIDE 1   // sample every IL_TL_STRIDE-th workgroup (linear id) instead of the first IL_TL_WGS: a whole population launch (thousands of workgroups) in one table
#endif
#define IL_TLV(kid, slot, value) do { const unsigned tl_l = blockIdx.x + gridDim.x * blockIdx.y, tl_w = tl_l / IL_TL_STRIDE; if (threadIdx.x == 0 && tl_l % IL_TL_STRIDE == 0 && tl_w < IL_TL_WGS && blockIdx.z == 0) il_tl[kid][tl_w][slot] = (value); } while (0)
#define IL_TL(kid, slot) IL_TLV(kid, slot, __builtin_amdgcn_s_memrealtime())
#define IL_TLC(kid, slot) IL_TLV(kid, slot, __builtin_amdgcn_s_memtime())   // the shader-clock counter (clock rate = its delta / the 100 MHz counter's)
#define IL_TL_READER(name) extern "C" int name(unsigned long long* out_host) { return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(il_tl), sizeof(il_tl)) == hipSuccess ? 0 : 3; }
#define IL_TL_END(kid) do { __syncthreads(); IL_TL(kid, 7); } while (0)   // a workgroup's end = its last wave's
#else
#define IL_TL(kid, slot) do { } while (0)
#define IL_TLV(kid, slot, value) do { } while (0)
#define IL_TLC(kid, slot) do { } while (0)
#define IL_TL_END(kid) do { } while (0)
#define IL_TL_READER(name)
#endif

// ---------------------------------------------------------------------------------------------
// Always-on launch stamps of the headline schedule's kernels (il_kernel_stamps, include/il_hip.h): thread 0 of every workgroup stores s_memrealtime (the 100 MHz
// device-wide counter) when it starts and after its last wave has finished - two 8-byte fire-and-forget stores per workgroup into a table of this translation unit,
// overwritten by every launch. After a run of graph replays the host reads, per kernel, min(begin) and max(end) over the workgroups of the LAST launch: the kernel's
// duration inside the timed schedule itself (bench.py builds `roofline` from these; no HIP events, no eager re-run). One kernel id per launch of an update. The begin stamp also
// records WHERE the workgroup runs (XCD, shader engine / array, CU): tests assert from it that no workgroup of the side stream shares a CU with a pair-mode workgroup.
// ---------------------------------------------------------------------------------------------
enum { IL_ST_GAIL_GRAD = 0, IL_ST_GAIL_REDUCE = 1, IL_ST_CHAIN = 2, IL_ST_DW_CRITIC = 3, IL_ST_POLICY_CRITIC = 4, IL_ST_DW_ACTOR = 5, IL_ST_GMMIL = 6, IL_ST_PWIL = 7, IL_ST_K = 8 };
#define IL_ST_WGS 512
#define IL_ST_TABLE static __device__ unsigned long long il_st[IL_ST_K][IL_ST_WGS][4];   // {begin, end, placement, gate}: placement = XCC_ID << 16 | HW_ID[15:8] (SE / SH / CU of the workgroup's first wave)
__device__ __forceinline__ unsigned il_st_xcc() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xfu; }
__device__ __forceinline__ unsigned il_st_hwid() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
#define IL_ST_INDEX (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))
#define IL_ST_BEGIN(kid) do { const unsigned st_w = IL_ST_INDEX; if (threadIdx.x == 0 && st_w < IL_ST_WGS) { il_st[kid][st_w][0] = __builtin_amdgcn_s_memrealtime(); il_st[kid][st_w][2] = ((unsigned long long)il_st_xcc() << 16) | ((il_st_hwid() >> 8) & 0xffu); il_st[kid][st_w][3] = 0ull; } } while (0)
// (round 6) overlapped launches: when this workgroup's wait for the other stream's launch was satisfied (slot 3; 0 = the workgroup has no such wait)
#define IL_ST_GATE(kid) do { const unsigned st_w = IL_ST_INDEX; if (threadIdx.x == 0 && st_w < IL_ST_WGS) il_st[kid][st_w][3] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define IL_ST_END(kid) do { __syncthreads(); const unsigned st_w = IL_ST_INDEX; if (threadIdx.x == 0 && st_w < IL_ST_WGS) il_st[kid][st_w][1] = __builtin_amdgcn_s_memrealtime(); } while (0)   // every thread of the workgroup passes here (bodies return, never s_endpgm): a workgroup's end = its last wave's
// out_host [IL_ST_K][IL_ST_WGS][4]: only the rows of the kernel ids this translation unit owns are meaningful
#define IL_ST_READER(name) extern "C" int name(unsigned long long* out_host) { return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(il_st), sizeof(il_st)) == hipSuccess ? 0 : 3; } \
                           extern "C" int name##_clear() { static unsigned long long z[IL_ST_K][IL_ST_WGS][4]; return hipMemcpyToSymbol(HIP_SYMBOL(il_st), z, sizeof(z)) == hipSuccess ? 0 : 3; }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// Optional per-kernel timing with HIP events recorded on the launch stream (il_trace_enable / il_trace_report).
// Disabled (one branch per launch) unless bench.py / a profiling run switches it on; never enable under graph capture.
// ---------------------------------------------------------------------------------------------
struct il_trace_scope {
  hipStream_t st; int slot;
  il_trace_scope(const char* name, hipStream_t s);
  ~il_trace_scope();
};
#define IL_TRACE(name, st) il_trace_scope il_trace_scope__(name, (hipStream_t)(st))
