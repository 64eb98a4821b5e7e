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
// Windows are uncached (failing that, fine-grained) device allocations shared through hipIpc handles (what RCCL does for its own buffers). Two forms (peer_device.hpp): stores /
// flags / polls with the compiler's system-scope release / acquire fences, or - uncached windows only - the payload through sc0 sc1 (write-through / cache-bypassing)
// accesses with drained stores and no fence; PeerExchange adopts a form only after a bitwise self-test with it on every rank. Waits are bounded (il_peer_bucket.spin_limit): a
// rank that never arrives makes the waiters count an expiry in status[0] and carry on, the host checks it (never a hang).
#include <string.h>

#include "peer_device.hpp"

static_assert(sizeof(hipIpcMemHandle_t) == IL_PEER_HANDLE_BYTES, "IL_PEER_HANDLE_BYTES must match hipIpcMemHandle_t");


extern "C" int64_t il_peer_region_bytes(int32_t world, int64_t n) {
  if (world < 1 || world > IL_PEER_MAX_RANKS || n < 1) return -1;
  const int64_t nch = peer_chunks(n);
  const int64_t bytes = 2 * (int64_t)world * nch * IL_PEER_CHUNK_FLOATS * 4 + nch * IL_PEER_FLAG_STRIDE * 4;
  return (bytes + 255) / 256 * 256;
}

extern "C" int64_t il_peer_job_region_bytes(int32_t world, int64_t n, int32_t n_jobs) {
  if (world < 1 || world > IL_PEER_MAX_RANKS || n < 1 || n_jobs < 1) return -1;
  const int64_t bytes = 2 * (int64_t)world * peer_chunks(n) * IL_PEER_CHUNK_FLOATS * 4 + (int64_t)n_jobs * IL_PEER_FLAG_STRIDE * 4;
  return (bytes + 255) / 256 * 256;
}

extern "C" int il_peer_window_alloc(int64_t bytes, void** window_host, unsigned char* handle_host, int32_t* kind_host) {
  IL_CHECK_ARG(bytes > 0 && window_host && handle_host && kind_host, "il_peer_window_alloc: bad arguments");
  *kind_host = 0;
  void* p = nullptr;
  // uncached (MTYPE UC: neither this GPU's L2 nor a peer's keeps a line of it, what RCCL uses for its own flag / LL buffers on gfx94x+), else fine-grained.
  // IL_PEER_WINDOW_KIND=uncached | finegrained forces one kind (tests: both kinds of window under the two-process soak; no fallback to the other kind then).
  const char* want = getenv("IL_PEER_WINDOW_KIND");
  const bool only_uc = want && want[0] == 'u', only_fg = want && want[0] == 'f';
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, only_fg ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
  if (only_fg) *kind_host = 1;
  if (e != hipSuccess && !only_uc && !only_fg) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained); *kind_host = 1; }
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

__global__ __launch_bounds__(256) void k_peer_allreduce(il_peer_bucket x, float* __restrict__ bucket_) {
  f32x4 mean[IL_PEER_Q];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int cnt = peer_chunk_allreduce(x, bucket_, c, mean);
  gfloat* bucket = (gfloat*)bucket_ + (int64_t)c * IL_PEER_CHUNK_FLOATS;
#pragma unroll
  for (int j = 0; j < IL_PEER_Q; ++j) {
    const int b = 4 * (tid + 256 * j);
    if (b >= cnt) continue;
    if (b + 3 < cnt) *(gfloat4*)(bucket + b) = mean[j];
    else { const float t[4] = {mean[j][0], mean[j][1], mean[j][2], mean[j][3]}; for (int k = 0; k < 4; ++k) if (b + k < cnt) bucket[b + k] = t[k]; }
  }
}

// The stand-alone exchange of a JOB-mode bucket (n_jobs arrival lines): workgroup j pushes, releases, waits and averages its share of the bucket through the very
// primitives the producing kernels use (peer_job_*), so PeerExchange's self-test and soak exercise that code path over the real windows before an update relies on it.
__global__ __launch_bounds__(256) void k_peer_job_allreduce(il_peer_bucket x, float* __restrict__ bucket_) {
  const int job = blockIdx.x, tid = threadIdx.x;
  gfloat* bucket = (gfloat*)bucket_;
  const int64_t n4 = x.n >> 2, per = (n4 + x.n_jobs - 1) / x.n_jobs, lo = (int64_t)job * per, hi = lo + per < n4 ? lo + per : n4;
  const PeerJob pj = peer_job_begin(x, job);
  for (int64_t i = lo + tid; i < hi; i += 256) { const f32x4 v = *(const gfloat4*)(bucket + 4 * i); peer_job_push4(x, pj, 4 * i, v); }
  const bool tail = job == x.n_jobs - 1 && tid < (int)(x.n & 3);   // the last floats of a bucket that is not a whole number of 16-byte lanes
  if (tail) peer_job_push1(x, pj, 4 * n4 + tid, bucket[4 * n4 + tid]);
  peer_job_exchange(x, pj, job);
  for (int64_t i = lo + tid; i < hi; i += 256) *(gfloat4*)(bucket + 4 * i) = peer_job_mean4(x, pj, 4 * i);
  if (tail) bucket[4 * n4 + tid] = peer_job_mean1(x, pj, 4 * n4 + tid);
  peer_job_end(x, pj, job);
}

extern "C" int il_peer_allreduce_mean(const il_peer_bucket* x, float* bucket, il_stream_t stream_) {
  IL_CHECK_ARG(x && bucket, "il_peer_allreduce_mean: null argument");
  IL_CHECK_ARG(x->world >= 1 && x->world <= IL_PEER_MAX_RANKS && x->rank >= 0 && x->rank < x->world, "il_peer_allreduce_mean: rank %d of %d", x->rank, x->world);
  IL_CHECK_ARG(x->n >= 1 && x->epoch && x->status && (x->window_offset & 255) == 0, "il_peer_allreduce_mean: bad descriptor");
  IL_CHECK_ARG((reinterpret_cast<uintptr_t>(bucket) & 15) == 0, "il_peer_allreduce_mean: the bucket must be 16-byte aligned");
  for (int r = 0; r < x->world; ++r) IL_CHECK_ARG(x->windows[r], "il_peer_allreduce_mean: window of rank %d is not mapped", r);
  hipStream_t st = (hipStream_t)stream_;
  if (x->n_jobs > 0) { IL_TRACE("k_peer_job_allreduce", st); k_peer_job_allreduce<<<(unsigned)x->n_jobs, 256, 0, st>>>(*x, bucket); }
  else { IL_TRACE("k_peer_allreduce", st); k_peer_allreduce<<<(unsigned)peer_chunks(x->n), 256, 0, st>>>(*x, bucket); }
  IL_CHECK_LAUNCH("il_peer_allreduce_mean");
  return IL_OK;
}
