// Device side of the peer-window gradient exchange (include/il_hip.h il_peer_*; protocol notes at the top of peer.hip).
#pragma once
#include "il_common.hpp"

// region of one bucket inside a window: float slots[2][W][nch * CHUNK], then uint32 arrival[nch][IL_PEER_FLAG_STRIDE] (word r of a chunk's line = rank r's epoch).
// Pointers are typed as global (address space 1) from the start: window bases come out of the kernel-argument array (generic), and generic accesses become FLAT ones.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f32x4 gfloat4;
typedef __attribute__((address_space(1))) uint32_t gu32;
__host__ __device__ inline int64_t peer_chunks(int64_t n) { return (n + IL_PEER_CHUNK_FLOATS - 1) / IL_PEER_CHUNK_FLOATS; }
__device__ __forceinline__ gfloat* peer_slots(const il_peer_bucket& x, int r) { return (gfloat*)(static_cast<char*>(x.windows[r]) + x.window_offset); }
__device__ __forceinline__ gu32* peer_arrival(const il_peer_bucket& x, int r, int64_t npad) { return (gu32*)(peer_slots(x, r) + 2 * (int64_t)x.world * npad); }

#define IL_PEER_Q (IL_PEER_CHUNK_FLOATS / 4 / 256)   // 16-byte lanes per thread and chunk
// Write-through form (il_peer_bucket.flags & IL_PEER_WRITE_THROUGH; only on UNCACHED windows): the system-scope fences of the default form cost a write-back of the XCD's
// L2 (release) and an invalidate (acquire) per workgroup although no line of an uncached window is ever held in a cache. Here the payload is stored and loaded with sc0 sc1
// buffer accesses (system scope, compiler-tracked waits), every storing wave drains its stores (s_waitcnt vmcnt(0): they have been acknowledged by the owning memory) before
// the barrier that precedes the arrival words, and the consumer reads the slabs with sc0 sc1 loads after its poll matched - no fence on either side.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define IL_PEER_SYS 17   // cache-policy bits of the raw buffer intrinsics: sc0 (1) | sc1 (16)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t peer_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw buffer (stride 0), 32-bit data format: byte offsets, range-checked
}
// One workgroup (256 threads, all of them, convergent: contains barriers) exchanges chunk `c` of `bucket_`: push into slot [parity][rank] of every window, release the
// epoch, wait for all ranks' arrival words, rank-ordered sum / W. Returns the number of valid floats in the chunk; mean[j] = the mean of the 16-byte lane
// b = 4 * (tid + 256 j) of the chunk (undefined for b >= the return value). The caller stores them (k_peer_allreduce) or consumes them in place (the fused apply kernels).
__device__ __forceinline__ int peer_chunk_allreduce(const il_peer_bucket& x, const float* __restrict__ bucket_, int c, f32x4 (&mean)[IL_PEER_Q]) {
  const int tid = threadIdx.x, W = x.world, me = x.rank;
  const int64_t npad = peer_chunks(x.n) * IL_PEER_CHUNK_FLOATS, o = (int64_t)c * IL_PEER_CHUNK_FLOATS;
  const int64_t left = x.n - o;
  const int cnt = left < IL_PEER_CHUNK_FLOATS ? (int)left : IL_PEER_CHUNK_FLOATS;
  const uint32_t e = ((gu32*)x.epoch)[c] + 1u;
  const int64_t par = (int64_t)(e & 1u);
  const gfloat* bucket = (const gfloat*)bucket_ + o;

  // ---- push: this rank's chunk into slot [par][me] of every window (the slot is padded to whole chunks: whole 16-byte lanes, zero-filled past n)
  f32x4 v[IL_PEER_Q];
#pragma unroll
  for (int j = 0; j < IL_PEER_Q; ++j) {
    const int b = 4 * (tid + 256 * j);
    if (b + 3 < cnt) v[j] = *(const gfloat4*)(bucket + b);
    else { float t[4]; for (int k = 0; k < 4; ++k) t[k] = b + k < cnt ? bucket[b + k] : 0.f; v[j] = f32x4{t[0], t[1], t[2], t[3]}; }
  }
  const bool wt = (x.flags & IL_PEER_WRITE_THROUGH) != 0;
  const int64_t slot_bytes = 2 * (int64_t)W * npad * 4;
  for (int i = 1; i <= W; ++i) {   // remote windows first (each over its own link), the local one last
    const int r = (me + i) % W;
    if (wt) {
      const __amdgpu_buffer_rsrc_t rs = peer_rsrc((const void*)peer_slots(x, r), slot_bytes);
      const int off = (int)(((par * W + me) * npad + o) * 4);
#pragma unroll
      for (int j = 0; j < IL_PEER_Q; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rs, off + 16 * (tid + 256 * j), 0, IL_PEER_SYS);
    } else {
      gfloat4* dst = (gfloat4*)(peer_slots(x, r) + (par * W + me) * npad + o);
#pragma unroll
      for (int j = 0; j < IL_PEER_Q; ++j) dst[tid + 256 * j] = v[j];
    }
  }
  if (!wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope, every thread: its stores have reached their windows before the barrier below
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains (write-through form: THE ordering; fence form: restates the wait the compiler may drop behind buffer_wbl2)
  __syncthreads();
  if (tid < W) {
    uint32_t* flag = (uint32_t*)(peer_arrival(x, tid, npad) + (int64_t)c * IL_PEER_FLAG_STRIDE + me);
    if (wt) __hip_atomic_store(flag, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // an sc0 sc1 store behind the drained payload
    else __hip_atomic_store(flag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }

  // ---- wait: all W arrival words of this chunk in the own window at epoch e (or later: a peer may already have pushed e + 1 into the other parity)
  if (tid < IL_WAVE) {
    uint32_t* mine = (uint32_t*)(peer_arrival(x, me, npad) + (int64_t)c * IL_PEER_FLAG_STRIDE);
    const int limit = x.spin_limit > 0 ? x.spin_limit : IL_PEER_SPIN_LIMIT;
    int spins = 0;
    bool all = false;
    for (;;) {
      uint32_t f = e;
      if (tid < W) f = __hip_atomic_load(mine + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      all = __builtin_amdgcn_ballot_w64((int32_t)(f - e) < 0) == 0ull;
      if (all || ++spins > limit) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (!all && tid == 0) {
      const long long n = __hip_atomic_fetch_add(reinterpret_cast<long long*>(x.status), 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      const long long host = reinterpret_cast<const long long*>(x.status)[1];   // a host-mapped flag the training loop reads every step without synchronising (0 = none)
      if (host) __hip_atomic_store((__attribute__((address_space(1))) long long*)host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (!wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: the peers' stores into this window are visible to this CU from here on

  // ---- reduce: the W slabs of the chunk in rank order, then the mean
  const gfloat* slab0 = peer_slots(x, me) + par * W * npad + o;
  const float fw = (float)W;
#pragma unroll
  for (int j = 0; j < IL_PEER_Q; ++j) {
    const int b = 4 * (tid + 256 * j);
    if (b >= cnt) { mean[j] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
    f32x4 acc;
    if (wt) {
      const __amdgpu_buffer_rsrc_t rs = peer_rsrc((const void*)peer_slots(x, me), slot_bytes);
      const int off = (int)((par * W * npad + o + b) * 4);
      acc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, IL_PEER_SYS));
      for (int r = 1; r < W; ++r) {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (int)(r * npad * 4), 0, IL_PEER_SYS));
        acc[0] = __fadd_rn(acc[0], t[0]); acc[1] = __fadd_rn(acc[1], t[1]); acc[2] = __fadd_rn(acc[2], t[2]); acc[3] = __fadd_rn(acc[3], t[3]);
      }
    } else {
      acc = *(const gfloat4*)(slab0 + b);
      for (int r = 1; r < W; ++r) {
        const f32x4 t = *(const gfloat4*)(slab0 + r * npad + b);
        acc[0] = __fadd_rn(acc[0], t[0]); acc[1] = __fadd_rn(acc[1], t[1]); acc[2] = __fadd_rn(acc[2], t[2]); acc[3] = __fadd_rn(acc[3], t[3]);
      }
    }
    acc[0] = __fdiv_rn(acc[0], fw); acc[1] = __fdiv_rn(acc[1], fw); acc[2] = __fdiv_rn(acc[2], fw); acc[3] = __fdiv_rn(acc[3], fw);
    mean[j] = acc;
  }
  if (tid == 0) ((gu32*)x.epoch)[c] = e;   // every thread read epoch[c] before the first barrier
  return cnt;
}

// ---------------------------------------------------------------------------------------------
// The exchange INSIDE the kernel that produces the gradients (round 3; il_peer_bucket.n_jobs > 0). The data-parallel schedule of one update used to be the single-GPU
// one cut open at its three optimiser steps: gradients to an arena, an exchange launch, an apply launch - eight launches on the critical stream instead of four. Here the
// workgroup ("job") that holds a piece of the gradient in registers - a 32 x 32 block of a layer's dW in k_dw_adam, 256 elements of the slab sum in k_gail_reduce - pushes
// what it holds into slot [parity][rank] of every rank's window at the parameter's own offset, releases ITS arrival line, waits for the same job of the other ranks and
// averages the W slabs in rank order, then runs its AdamW epilogue as on one GPU. Same protocol, same parity argument and the same bounded wait as peer_chunk_allreduce
// (a job pushes before it waits and only waits for the same job of the other ranks: no co-residency requirement, no deadlock); the region carries n_jobs arrival
// lines instead of one per chunk. Same means as the exchange launch (rank-ordered sum / W of the same gradient values): the replicas stay bit-identical to each other
// AND to the three-launch schedule.
// ---------------------------------------------------------------------------------------------
struct PeerJob { uint32_t e; int par; int64_t npad, slot_bytes; bool wt; };
__device__ __forceinline__ PeerJob peer_job_begin(const il_peer_bucket& x, int job) {   // every thread that takes part; reads the job's epoch (advanced by peer_job_end)
  PeerJob pj;
  pj.e = ((gu32*)x.epoch)[job] + 1u; pj.par = (int)(pj.e & 1u);
  pj.npad = peer_chunks(x.n) * IL_PEER_CHUNK_FLOATS; pj.slot_bytes = 2 * (int64_t)x.world * pj.npad * 4;
  pj.wt = (x.flags & IL_PEER_WRITE_THROUGH) != 0;
  return pj;
}
// one 16-byte lane / one float of this rank's gradient at float offset o of the bucket, into every window (remote ones first, each over its own link)
__device__ __forceinline__ void peer_job_push4(const il_peer_bucket& x, const PeerJob& pj, int64_t o, const f32x4& v) {
  const int W = x.world, me = x.rank;
  const int64_t at = ((int64_t)pj.par * W + me) * pj.npad + o;
  for (int i = 1; i <= W; ++i) {
    const int r = (me + i) % W;
    if (pj.wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), peer_rsrc((const void*)peer_slots(x, r), pj.slot_bytes), (int)(at * 4), 0, IL_PEER_SYS);
    else *(gfloat4*)(peer_slots(x, r) + at) = v;
  }
}
__device__ __forceinline__ void peer_job_push1(const il_peer_bucket& x, const PeerJob& pj, int64_t o, float v) {
  const int W = x.world, me = x.rank;
  const int64_t at = ((int64_t)pj.par * W + me) * pj.npad + o;
  for (int i = 1; i <= W; ++i) {
    const int r = (me + i) % W;
    if (pj.wt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), peer_rsrc((const void*)peer_slots(x, r), pj.slot_bytes), (int)(at * 4), 0, IL_PEER_SYS);
    else peer_slots(x, r)[at] = v;
  }
}
__device__ __forceinline__ void peer_wait_expired(const il_peer_bucket& x) {
  const long long n = __hip_atomic_fetch_add(reinterpret_cast<long long*>(x.status), 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const long long host = reinterpret_cast<const long long*>(x.status)[1];
  if (host) __hip_atomic_store((__attribute__((address_space(1))) long long*)host, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ALL threads of the workgroup (barriers inside): the pushes above are complete -> arrival words of `job` in every window -> all W arrival words in the own window
__device__ __forceinline__ void peer_job_exchange(const il_peer_bucket& x, const PeerJob& pj, int job) {
  const int tid = threadIdx.x, W = x.world, me = x.rank;
  if (!pj.wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains (cf. peer_chunk_allreduce)
  __syncthreads();
  if (tid < W) {
    uint32_t* flag = (uint32_t*)(peer_arrival(x, tid, pj.npad) + (int64_t)job * IL_PEER_FLAG_STRIDE + me);
    if (pj.wt) __hip_atomic_store(flag, pj.e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(flag, pj.e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < IL_WAVE) {
    uint32_t* mine = (uint32_t*)(peer_arrival(x, me, pj.npad) + (int64_t)job * IL_PEER_FLAG_STRIDE);
    const int limit = x.spin_limit > 0 ? x.spin_limit : IL_PEER_SPIN_LIMIT;
    int spins = 0;
    bool all = false;
    for (;;) {
      uint32_t f = pj.e;
      if (tid < W) f = __hip_atomic_load(mine + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      all = __builtin_amdgcn_ballot_w64((int32_t)(f - pj.e) < 0) == 0ull;
      if (all || ++spins > limit) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (!all && tid == 0) peer_wait_expired(x);
  }
  __syncthreads();
  if (!pj.wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// the mean over the ranks (rank-ordered sum / W, the expression of peer_chunk_allreduce) of the lane / float at offset o
__device__ __forceinline__ f32x4 peer_job_mean4(const il_peer_bucket& x, const PeerJob& pj, int64_t o) {
  const int W = x.world;
  const int64_t at = (int64_t)pj.par * W * pj.npad + o;
  f32x4 acc;
  if (pj.wt) {
    const __amdgpu_buffer_rsrc_t rs = peer_rsrc((const void*)peer_slots(x, x.rank), pj.slot_bytes);
    acc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(at * 4), 0, IL_PEER_SYS));
    for (int r = 1; r < W; ++r) {
      const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((at + r * pj.npad) * 4), 0, IL_PEER_SYS));
      acc[0] = __fadd_rn(acc[0], t[0]); acc[1] = __fadd_rn(acc[1], t[1]); acc[2] = __fadd_rn(acc[2], t[2]); acc[3] = __fadd_rn(acc[3], t[3]);
    }
  } else {
    const gfloat* s0 = peer_slots(x, x.rank) + at;
    acc = *(const gfloat4*)s0;
    for (int r = 1; r < W; ++r) {
      const f32x4 t = *(const gfloat4*)(s0 + r * pj.npad);
      acc[0] = __fadd_rn(acc[0], t[0]); acc[1] = __fadd_rn(acc[1], t[1]); acc[2] = __fadd_rn(acc[2], t[2]); acc[3] = __fadd_rn(acc[3], t[3]);
    }
  }
  const float fw = (float)W;
  acc[0] = __fdiv_rn(acc[0], fw); acc[1] = __fdiv_rn(acc[1], fw); acc[2] = __fdiv_rn(acc[2], fw); acc[3] = __fdiv_rn(acc[3], fw);
  return acc;
}
__device__ __forceinline__ float peer_job_mean1(const il_peer_bucket& x, const PeerJob& pj, int64_t o) {
  const int W = x.world;
  const int64_t at = (int64_t)pj.par * W * pj.npad + o;
  float acc;
  if (pj.wt) {
    const __amdgpu_buffer_rsrc_t rs = peer_rsrc((const void*)peer_slots(x, x.rank), pj.slot_bytes);
    acc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(at * 4), 0, IL_PEER_SYS));
    for (int r = 1; r < W; ++r) acc = __fadd_rn(acc, __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)((at + r * pj.npad) * 4), 0, IL_PEER_SYS)));
  } else {
    const gfloat* s0 = peer_slots(x, x.rank) + at;
    acc = s0[0];
    for (int r = 1; r < W; ++r) acc = __fadd_rn(acc, s0[r * pj.npad]);
  }
  return __fdiv_rn(acc, (float)W);
}
__device__ __forceinline__ void peer_job_end(const il_peer_bucket& x, const PeerJob& pj, int job) { if (threadIdx.x == 0) ((gu32*)x.epoch)[job] = pj.e; }   // (every thread read it before the barriers of peer_job_exchange)
// The same for ONE float held by ONE thread (Adam(log alpha) in the tail of k_dw_adam): no barrier, the thread orders its own accesses.
__device__ __forceinline__ float peer_thread_allreduce1(const il_peer_bucket& x, int job, int64_t o, float v) {
  const PeerJob pj = peer_job_begin(x, job);
  const int W = x.world, me = x.rank;
  peer_job_push1(x, pj, o, v);
  if (!pj.wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int r = 0; r < W; ++r) {
    uint32_t* flag = (uint32_t*)(peer_arrival(x, r, pj.npad) + (int64_t)job * IL_PEER_FLAG_STRIDE + me);
    if (pj.wt) __hip_atomic_store(flag, pj.e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(flag, pj.e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  uint32_t* mine = (uint32_t*)(peer_arrival(x, me, pj.npad) + (int64_t)job * IL_PEER_FLAG_STRIDE);
  const int limit = x.spin_limit > 0 ? x.spin_limit : IL_PEER_SPIN_LIMIT;
  bool all = false;
  for (int spins = 0; spins <= limit; ++spins) {
    all = true;
    for (int r = 0; r < W; ++r) all = all && (int32_t)(__hip_atomic_load(mine + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - pj.e) >= 0;
    if (all) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (!all) peer_wait_expired(x);
  if (!pj.wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const float m = peer_job_mean1(x, pj, o);
  ((gu32*)x.epoch)[job] = pj.e;
  return m;
}
