// One-shot gradient exchange over peer-mapped windows (include/il_hip.h il_peer_*).
//
// The data-parallel update has three sync points (discriminator, critic, actor + alpha gradients) whose messages are 6.7 / 580 / 295 KB: far below the size at
// which a ring or tree collective is bandwidth-bound, so what an all-reduce costs there is its latency (protocol hops, proxy / launch overhead). xGMI is a full
// point-to-point mesh: every GPU reaches every other one over its own link. The exchange below therefore uses ONE kernel per sync point and ONE fabric crossing:
//   push    workgroup c of every rank stores chunk c of its gradient bucket into slot [rank] of EVERY rank's receive window (7 remote stores over 7 different links
//           + 1 local), then releases an arrival word (its epoch) in each of those windows;
//   wait    it polls the W arrival words of chunk c in its OWN window (one wave-load: they share a 128-byte line that no other chunk touches);
//   reduce  it sums the W slabs of chunk c in RANK ORDER and divides by W: every rank evaluates the same expression on the same bits, so replicas stay
//           bit-identical (what an all-reduce guarantees), and for W = 2 the result equals (a + b) / 2 of any collective.
// No workgroup waits for a workgroup of its own GPU, and a workgroup pushes before it waits, so the kernel needs no co-residency and cannot deadlock: rank A's
// workgroup c only needs rank B's workgroup c to be dispatched eventually.
//
// Slots are double-buffered by epoch parity: rank A can push epoch e+1 of a bucket while rank B still reads epoch e (A has seen B's push of e, not B's reads);
// A cannot push e+2 before B has pushed e+1, which B does after its epoch-e kernel has finished. Epochs are per-chunk device counters advanced by the kernel
// itself, so a captured graph replays correctly.
//
// Windows are fine-grained device allocations shared through hipIpc handles (what RCCL does for its own buffers); stores / flags / polls use system-scope
// release / acquire so that hipcc emits the cache maintenance gfx950 needs for memory another agent writes. Waits are bounded (il_peer_bucket.spin_limit): a
// rank that never arrives makes the waiters count an expiry in status[0] and carry on, the host checks it (never a hang).
#include <string.h>

#include "il_common.hpp"

static_assert(sizeof(hipIpcMemHandle_t) == IL_PEER_HANDLE_BYTES, "IL_PEER_HANDLE_BYTES must match hipIpcMemHandle_t");

static inline int64_t peer_chunks(int64_t n) { return (n + IL_PEER_CHUNK_FLOATS - 1) / IL_PEER_CHUNK_FLOATS; }

extern "C" int64_t il_peer_region_bytes(int32_t world, int64_t n) {
  if (world < 1 || world > IL_PEER_MAX_RANKS || n < 1) return -1;
  const int64_t nch = peer_chunks(n);
  const int64_t bytes = 2 * (int64_t)world * nch * IL_PEER_CHUNK_FLOATS * 4 + nch * IL_PEER_FLAG_STRIDE * 4;
  return (bytes + 255) / 256 * 256;
}

extern "C" int il_peer_window_alloc(int64_t bytes, void** window_host, unsigned char* handle_host) {
  IL_CHECK_ARG(bytes > 0 && window_host && handle_host, "il_peer_window_alloc: bad arguments");
  void* p = nullptr;
  // uncached (MTYPE UC: neither this GPU's L2 nor a peer's keeps a line of it, what RCCL uses for its own flag / LL buffers on gfx94x+), else fine-grained
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "il_peer_window_alloc: hipExtMallocWithFlags(uncached / fine-grained, %lld bytes): %s", (long long)bytes, hipGetErrorString(e));
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); return il_set_error(IL_ERR_HIP, "il_peer_window_alloc: %s", hipGetErrorString(e)); }
  memcpy(handle_host, &h, sizeof(h));
  *window_host = p;
  return IL_OK;
}

extern "C" int il_peer_window_open(const unsigned char* handle_host, void** window_host) {
  IL_CHECK_ARG(handle_host && window_host, "il_peer_window_open: bad arguments");
  hipIpcMemHandle_t h;
  memcpy(&h, handle_host, sizeof(h));
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "il_peer_window_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  *window_host = p;
  return IL_OK;
}

extern "C" int il_peer_window_close(void* window) {
  if (!window) return IL_OK;
  const hipError_t e = hipIpcCloseMemHandle(window);
  return e == hipSuccess ? IL_OK : il_set_error(IL_ERR_HIP, "il_peer_window_close: %s", hipGetErrorString(e));
}

extern "C" int il_peer_window_free(void* window) {
  if (!window) return IL_OK;
  const hipError_t e = hipFree(window);
  return e == hipSuccess ? IL_OK : il_set_error(IL_ERR_HIP, "il_peer_window_free: %s", hipGetErrorString(e));
}

// region of one bucket inside a window: float slots[2][W][nch * CHUNK], then uint32 arrival[nch][IL_PEER_FLAG_STRIDE] (word r of a chunk's line = rank r's epoch).
// Pointers are typed as global (address space 1) from the start: window bases come out of the kernel-argument array (generic), and generic accesses become FLAT ones.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f32x4 gfloat4;
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ gfloat* peer_slots(const il_peer_bucket& x, int r) { return (gfloat*)(static_cast<char*>(x.windows[r]) + x.window_offset); }
__device__ __forceinline__ gu32* peer_arrival(const il_peer_bucket& x, int r, int64_t npad) { return (gu32*)(peer_slots(x, r) + 2 * (int64_t)x.world * npad); }

#define IL_PEER_Q (IL_PEER_CHUNK_FLOATS / 4 / 256)   // 16-byte lanes per thread and chunk
__global__ __launch_bounds__(256) void k_peer_allreduce(il_peer_bucket x, float* __restrict__ bucket_) {
  const int c = blockIdx.x, tid = threadIdx.x, W = x.world, me = x.rank;
  const int64_t npad = (int64_t)gridDim.x * IL_PEER_CHUNK_FLOATS, o = (int64_t)c * IL_PEER_CHUNK_FLOATS;
  const int64_t left = x.n - o;
  const int cnt = left < IL_PEER_CHUNK_FLOATS ? (int)left : IL_PEER_CHUNK_FLOATS;
  const uint32_t e = ((gu32*)x.epoch)[c] + 1u;
  const int64_t par = (int64_t)(e & 1u);
  gfloat* bucket = (gfloat*)bucket_ + o;

  // ---- push: this rank's chunk into slot [par][me] of every window (the slot is padded to whole chunks: whole 16-byte lanes, zero-filled past n)
  f32x4 v[IL_PEER_Q];
#pragma unroll
  for (int j = 0; j < IL_PEER_Q; ++j) {
    const int b = 4 * (tid + 256 * j);
    if (b + 3 < cnt) v[j] = *(gfloat4*)(bucket + b);
    else { float t[4]; for (int k = 0; k < 4; ++k) t[k] = b + k < cnt ? bucket[b + k] : 0.f; v[j] = f32x4{t[0], t[1], t[2], t[3]}; }
  }
  for (int i = 1; i <= W; ++i) {   // remote windows first (each over its own link), the local one last
    const int r = (me + i) % W;
    gfloat4* dst = (gfloat4*)(peer_slots(x, r) + (par * W + me) * npad + o);
#pragma unroll
    for (int j = 0; j < IL_PEER_Q; ++j) dst[tid + 256 * j] = v[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope, every thread: its stores have reached their windows before the barrier below
  __syncthreads();
  if (tid < W) __hip_atomic_store((uint32_t*)(peer_arrival(x, tid, npad) + (int64_t)c * IL_PEER_FLAG_STRIDE + me), e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);

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
    if (!all && tid == 0) __hip_atomic_fetch_add(reinterpret_cast<long long*>(x.status), 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: the peers' stores into this window are visible to this CU from here on

  // ---- reduce: the W slabs of the chunk in rank order, then the mean
  const gfloat* slab0 = peer_slots(x, me) + par * W * npad + o;
  const float fw = (float)W;
#pragma unroll
  for (int j = 0; j < IL_PEER_Q; ++j) {
    const int b = 4 * (tid + 256 * j);
    if (b >= cnt) continue;
    f32x4 acc = *(const gfloat4*)(slab0 + b);
    for (int r = 1; r < W; ++r) {
      const f32x4 t = *(const gfloat4*)(slab0 + r * npad + b);
      acc[0] = __fadd_rn(acc[0], t[0]); acc[1] = __fadd_rn(acc[1], t[1]); acc[2] = __fadd_rn(acc[2], t[2]); acc[3] = __fadd_rn(acc[3], t[3]);
    }
    acc[0] = __fdiv_rn(acc[0], fw); acc[1] = __fdiv_rn(acc[1], fw); acc[2] = __fdiv_rn(acc[2], fw); acc[3] = __fdiv_rn(acc[3], fw);
    if (b + 3 < cnt) *(gfloat4*)(bucket + b) = acc;
    else { const float t[4] = {acc[0], acc[1], acc[2], acc[3]}; for (int k = 0; k < 4; ++k) if (b + k < cnt) bucket[b + k] = t[k]; }
  }
  if (tid == 0) ((gu32*)x.epoch)[c] = e;   // every thread read epoch[c] before the first barrier
}

extern "C" int il_peer_allreduce_mean(const il_peer_bucket* x, float* bucket, il_stream_t stream_) {
  IL_CHECK_ARG(x && bucket, "il_peer_allreduce_mean: null argument");
  IL_CHECK_ARG(x->world >= 1 && x->world <= IL_PEER_MAX_RANKS && x->rank >= 0 && x->rank < x->world, "il_peer_allreduce_mean: rank %d of %d", x->rank, x->world);
  IL_CHECK_ARG(x->n >= 1 && x->epoch && x->status && (x->window_offset & 255) == 0, "il_peer_allreduce_mean: bad descriptor");
  IL_CHECK_ARG((reinterpret_cast<uintptr_t>(bucket) & 15) == 0, "il_peer_allreduce_mean: the bucket must be 16-byte aligned");
  for (int r = 0; r < x->world; ++r) IL_CHECK_ARG(x->windows[r], "il_peer_allreduce_mean: window of rank %d is not mapped", r);
  hipStream_t st = (hipStream_t)stream_;
  { IL_TRACE("k_peer_allreduce", st); k_peer_allreduce<<<(unsigned)peer_chunks(x->n), 256, 0, st>>>(*x, bucket); }
  IL_CHECK_LAUNCH("il_peer_allreduce_mean");
  return IL_OK;
}
