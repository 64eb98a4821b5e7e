// TEST INFRASTRUCTURE: runs the HIP kernels of imitation-learning_amd/csrc on the HOST, from their own source text, so that `pytest -m "not gpu"` can execute them
// against the reference fixtures where there is no GPU. tests/host_emu/build.py rewrites `k<<<grid, block, lds, stream>>>(args)` into EMU_LAUNCH, the dynamic
// `extern __shared__` declaration into a pointer, drops address-space qualifiers and empty asm pins, and compiles the result with g++ against this header in the
// place of <hip/hip_runtime.h>.
//
// Execution model: one workgroup at a time, in blockIdx order (x fastest); its threads are fibers on ONE OS thread. A wave's 64 lanes take turns until each sits at a
// __syncthreads() or has returned, then the next wave runs; a barrier releases when every live fiber has arrived. Wave-level instructions - __shfl*, DPP, readlane,
// ballot, the wave barrier, v_mfma_f32_16x16x4_f32 - are built on one exchange: every live lane of the wave publishes its operand under a per-lane sequence number and
// yields until all of them have. Atomics are plain read-modify-writes, fences and s_waitcnt nothing: program order on one thread is stronger than any of them.
// What this can show: indexing, tile / slab / counter layouts, iteration orders, the arithmetic (MFMA = an fmaf chain over k, as on the device) - everything a parity
// test compares. What it cannot: performance, memory-ordering bugs between workgroups or waves, anything that needs two workgroups in flight at once (a workgroup that
// waits for a HIGHER-numbered one would spin out its bound here; the kernels' hand-offs all point at lower-numbered workgroups or earlier launches, which is also what
// keeps them deadlock-free on the device), and code that leans on wave lockstep without a wave barrier. Sums follow the device's association order (the DPP steps and
// readlanes are emulated lane for lane); libm is glibc's, so comparisons use the parity tests' tolerances, not bit equality.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }

// device-only builtins the emulated sources mention: scheduling hints and memory-order fences mean nothing to one OS thread running one workgroup at a time
#define __builtin_amdgcn_sched_barrier(x) do { } while (0)
#define __builtin_amdgcn_fence(...) do { } while (0)
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> inline T __hip_atomic_fetch_add(T* p, T v, int, int) { const T old = *p; *p = old + v; return old; }
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T> inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
template <class V> inline V emu_elementwise_fma(V a, V b, V c) { V r = c; for (unsigned i = 0; i < sizeof(V) / sizeof(float); ++i) r[i] = fmaf(a[i], b[i], c[i]); return r; }
#define __builtin_elementwise_fma emu_elementwise_fma

namespace emu {
enum { RUNNABLE = 0, AT_BARRIER = 1, DONE = 2 };
enum { MAX_THREADS = 1024, STACK_BYTES = 256 * 1024, SHFL_RING = 64 };
#if defined(__x86_64__)
// A fiber switch that saves the callee-saved registers and swaps stack pointers (swapcontext also saves the signal mask: two system calls per switch, and the MFMA /
// DPP emulation switches ~128 times per wave-level instruction).
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak emu_switch
.type emu_switch, @function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");
struct Fiber { void* sp; int state; };
inline void* sched_sp = nullptr;
#define EMU_FAST_SWITCH 1
#else
struct Fiber { ucontext_t ctx; int state; };
inline ucontext_t sched_ctx;
#define EMU_FAST_SWITCH 0
#endif
inline std::vector<Fiber> fibers;
inline std::vector<char> stacks;
inline int cur = 0;
inline std::function<void()> body;
inline void* dyn_smem = nullptr;
inline uint64_t shfl_val[SHFL_RING][MAX_THREADS];
inline unsigned shfl_tag[SHFL_RING][MAX_THREADS];
inline unsigned shfl_seq[MAX_THREADS];
inline float bs_buf[MAX_THREADS];
enum { COLL_RING = 8, COLL_BYTES = 16 };
inline unsigned char coll_val[COLL_RING][MAX_THREADS][COLL_BYTES];
inline unsigned coll_tag[COLL_RING][MAX_THREADS];
inline unsigned coll_seq[MAX_THREADS];
inline int block_threads = 0;
inline int schedule_mode = -1;          // 0 forward, 1 reverse, 2 random (IL_EMU_SCHEDULE, read at the first launch)
inline unsigned long long rng_state = 1;
inline unsigned next_random() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng_state >> 33); }
inline void read_schedule() {
  if (schedule_mode >= 0) return;
  const char* e = getenv("IL_EMU_SCHEDULE");
  schedule_mode = !e || !strcmp(e, "forward") ? 0 : (!strcmp(e, "reverse") ? 1 : 2);
  if (schedule_mode == 2) { const char* c = strchr(e, ':'); rng_state = c ? strtoull(c + 1, nullptr, 10) * 2 + 1 : 1; }
}

#if EMU_FAST_SWITCH
inline void to_scheduler() { emu_switch(&fibers[cur].sp, sched_sp); }
inline void to_fiber(int i) { emu_switch(&sched_sp, fibers[i].sp); }
#else
inline void to_scheduler() { swapcontext(&fibers[cur].ctx, &sched_ctx); }
inline void to_fiber(int i) { swapcontext(&sched_ctx, &fibers[i].ctx); }
#endif
inline void fiber_entry() {
  body();
  fibers[cur].state = DONE;
  to_scheduler();
  abort();   // a finished fiber is never resumed
}
inline void make_fiber(int i) {
  char* base = stacks.data() + (size_t)i * STACK_BYTES;
#if EMU_FAST_SWITCH
  void** sp = (void**)(((uintptr_t)base + STACK_BYTES) & ~(uintptr_t)15);
  *--sp = nullptr;                       // where a return address would sit: fiber_entry starts with rsp = 8 mod 16 like any called function
  *--sp = (void*)&fiber_entry;           // popped by emu_switch's `ret`
  for (int r = 0; r < 6; ++r) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
  fibers[i].sp = sp;
#else
  getcontext(&fibers[i].ctx);
  fibers[i].ctx.uc_stack.ss_sp = base;
  fibers[i].ctx.uc_stack.ss_size = STACK_BYTES;
  fibers[i].ctx.uc_link = nullptr;
  makecontext(&fibers[i].ctx, fiber_entry, 0);
#endif
}
inline int linear_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }

inline void run_block(const dim3& bdim, const std::function<void()>& fn) {
  const int n = (int)(bdim.x * bdim.y * bdim.z);
  if (n > MAX_THREADS) { fprintf(stderr, "emu: %d threads per block\n", n); abort(); }
  if ((int)fibers.size() < n) { fibers.resize(n); stacks.resize((size_t)n * STACK_BYTES); }
  body = fn;
  block_threads = n;
  for (int i = 0; i < n; ++i) {
    make_fiber(i);
    fibers[i].state = RUNNABLE;
    shfl_seq[i] = 0; coll_seq[i] = 0;
  }
  memset(shfl_tag, 0xff, sizeof(shfl_tag));
  memset(coll_tag, 0xff, sizeof(coll_tag));
  const int nwaves = (n + 63) / 64;
  int wave_order[MAX_THREADS / 64], lane_order[64];
  for (;;) {
    // one wave at a time: its lanes take turns until each of them sits at a barrier or has returned (lanes that yield inside a wave-level exchange stay runnable).
    // IL_EMU_SCHEDULE = reverse | random:<seed> changes the order of the waves between two barriers and of the lanes within a wave: a kernel whose result depends on
    // that order is missing a barrier (an LDS race the forward order happens to hide)
    for (int w = 0; w < nwaves; ++w) wave_order[w] = schedule_mode == 1 ? nwaves - 1 - w : w;
    for (int l = 0; l < 64; ++l) lane_order[l] = schedule_mode == 1 ? 63 - l : l;
    if (schedule_mode == 2) {
      for (int w = nwaves - 1; w > 0; --w) std::swap(wave_order[w], wave_order[next_random() % (unsigned)(w + 1)]);
      for (int l = 63; l > 0; --l) std::swap(lane_order[l], lane_order[next_random() % (unsigned)(l + 1)]);
    }
    for (int wi = 0; wi < nwaves; ++wi) {
      const int w0 = 64 * wave_order[wi];
      for (bool any = true; any;) {
        any = false;
        for (int li = 0; li < 64; ++li) {
          const int i = w0 + lane_order[li];
          if (i >= n || fibers[i].state != RUNNABLE) continue;
          any = true;
          cur = i;
          threadIdx = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
          to_fiber(i);
        }
      }
    }
    int waiting = 0;
    for (int i = 0; i < n; ++i) waiting += fibers[i].state == AT_BARRIER;
    if (!waiting) break;                                                  // every fiber has returned
    for (int i = 0; i < n; ++i) if (fibers[i].state == AT_BARRIER) fibers[i].state = RUNNABLE;   // barrier: all live fibers arrived
  }
}

// Wave-level exchange: every LIVE lane of the calling lane's wave publishes `bytes` of payload; returns once all of them have, with `all` pointing at the 64 payload
// slots (a lane that has already returned from the kernel counts as arrived; its slot is stale). The building block of the emulated MFMA / DPP / readlane / ballot.
inline const unsigned char* wave_exchange(const void* mine, size_t bytes) {
  const int t = linear_tid(), w0 = t & ~63, w1 = std::min(w0 + 64, block_threads);
  const unsigned k = coll_seq[t]++;
  const unsigned slot = k % COLL_RING;
  memcpy(coll_val[slot][t], mine, bytes);
  coll_tag[slot][t] = k;
  for (;;) {
    bool all = true;
    for (int i = w0; i < w1 && all; ++i) all = fibers[i].state == DONE || coll_tag[slot][i] == k || (int)(coll_tag[slot][i] - k) > 0;
    if (all) break;
    to_scheduler();
  }
  return &coll_val[slot][w0][0];
}
inline bool lane_live(int t) { return fibers[t].state != DONE; }

template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds_bytes, Args... args) {
  void* aligned = malloc(lds_bytes ? lds_bytes : 16);   // exactly the launch's LDS (16-byte aligned like every malloc): an AddressSanitizer build sees the first byte past it
  read_schedule();
  gridDim = grid; blockDim = block;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        blockIdx = dim3(x, y, z);
        dyn_smem = aligned;
        memset(aligned, 0xcd, lds_bytes);   // LDS is not zero on the GPU either: poison it so that a read of an unwritten word shows
        run_block(block, [&]() { kernel(args...); });
      }
  free(aligned);
}
}  // namespace emu

#define EMU_LAUNCH(kernel, grid, block, lds, ...) emu::launch(kernel, dim3(grid), dim3(block), (size_t)(lds), __VA_ARGS__)

inline void __syncthreads() {
  emu::fibers[emu::cur].state = emu::AT_BARRIER;
  emu::to_scheduler();
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) <= 8, "emulated __shfl: at most 8 bytes");
  const int t = emu::linear_tid();
  const unsigned k = emu::shfl_seq[t]++;
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  emu::shfl_val[k % emu::SHFL_RING][t] = raw; emu::shfl_tag[k % emu::SHFL_RING][t] = k;
  const int wave0 = t & ~63, s = wave0 + ((t - wave0) / width) * width + (src % width);
  while (emu::shfl_tag[k % emu::SHFL_RING][s] != k) emu::to_scheduler();   // the source lane has not reached this shuffle yet
  emu::to_scheduler();                                                     // let every lane publish before any lane runs ahead and recycles the ring
  T out; memcpy(&out, &emu::shfl_val[k % emu::SHFL_RING][s], sizeof(T));
  return out;
}

// ---- wave-level builtins on the exchange above (all live lanes of the wave must reach the call, as on the device)
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
inline int __builtin_amdgcn_readlane(int v, int lane) {
  const unsigned char* all = emu::wave_exchange(&v, 4);
  int out; memcpy(&out, all + (size_t)lane * emu::COLL_BYTES, 4); return out;
}
inline int __builtin_amdgcn_readfirstlane(int v) {
  const int t = emu::linear_tid(), w0 = t & ~63;
  const unsigned char* all = emu::wave_exchange(&v, 4);
  for (int l = 0; l < 64; ++l) if (w0 + l < emu::block_threads && emu::lane_live(w0 + l)) { int out; memcpy(&out, all + (size_t)l * emu::COLL_BYTES, 4); return out; }
  return v;
}
// DPP: the source lane of `ctrl` within the lane's row of 16 (row_shr / row_ror / row_shl / quad_perm are what the kernels use); invalid source + !bound_ctrl -> `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = emu::linear_tid() & 63, row = lane & ~15, i = lane & 15;
  const unsigned char* all = emu::wave_exchange(&src, 4);
  int from = -1;
  if (ctrl >= 0x121 && ctrl <= 0x12f) from = row + ((i - (ctrl & 15)) & 15);                 // row_ror:n
  else if (ctrl >= 0x111 && ctrl <= 0x11f) from = i - (ctrl & 15) >= 0 ? row + i - (ctrl & 15) : -1;   // row_shr:n
  else if (ctrl >= 0x101 && ctrl <= 0x10f) from = i + (ctrl & 15) <= 15 ? row + i + (ctrl & 15) : -1;  // row_shl:n
  else if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);            // quad_perm
  else { fprintf(stderr, "emu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
  if (from < 0) return bound_ctrl ? 0 : old;
  int out; memcpy(&out, all + (size_t)from * emu::COLL_BYTES, 4); return out;
}
typedef float emu_f32x4 __attribute__((vector_size(16)));
// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] B[4x16]; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds D[4 (l >> 4) + reg][l & 15]; an fmaf chain over k
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  const int lane = emu::linear_tid() & 63, col = lane & 15, rg = lane >> 4;
  float ab[2] = {a, b};
  const unsigned char* all = emu::wave_exchange(ab, 8);
  auto A = [&](int i, int k) { float v; memcpy(&v, all + (size_t)(i + 16 * k) * emu::COLL_BYTES, 4); return v; };
  auto B = [&](int k, int j) { float v; memcpy(&v, all + (size_t)(j + 16 * k) * emu::COLL_BYTES + 4, 4); return v; };
  for (int r = 0; r < 4; ++r) {
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(A(4 * rg + r, k), B(k, col), acc);
    c[r] = acc;
  }
  return c;
}
inline unsigned long long __ballot(int pred) {
  const int t = emu::linear_tid(), w0 = t & ~63;
  const int p = pred != 0;
  const unsigned char* all = emu::wave_exchange(&p, 4);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) if (w0 + l < emu::block_threads && emu::lane_live(w0 + l)) { int v; memcpy(&v, all + (size_t)l * emu::COLL_BYTES, 4); if (v) m |= 1ull << l; }
  return m;
}
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred); }
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (emu::linear_tid() & (width - 1)) ^ mask, width); }
template <class T> inline T __shfl_up(T v, unsigned delta, int width = 64) { const int l = emu::linear_tid() & (width - 1); const T o = __shfl(v, l >= (int)delta ? l - (int)delta : l, width); return o; }
template <class T> inline T __shfl_down(T v, unsigned delta, int width = 64) { const int l = emu::linear_tid() & (width - 1); return __shfl(v, l + (int)delta < width ? l + (int)delta : l, width); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline void __threadfence() {}
inline void __threadfence_system() {}
inline void __threadfence_block() {}
#define __builtin_amdgcn_s_sleep(x) do { } while (0)
inline void emu_wave_barrier() { const int z = 0; (void)emu::wave_exchange(&z, 4); }   // lanes of a wave run one after the other here: a wave barrier has to be a real rendezvous
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __HIP_MEMORY_SCOPE_SYSTEM 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_WAVEFRONT 3
template <class T> inline T atomicAdd(T* p, T v) { const T old = *p; *p = old + v; return old; }
template <class T> inline T atomicMax(T* p, T v) { const T old = *p; if (v > old) *p = v; return old; }
template <class T> inline T atomicMin(T* p, T v) { const T old = *p; if (v < old) *p = v; return old; }
template <class T> inline T atomicExch(T* p, T v) { const T old = *p; *p = v; return old; }
// raw buffer resources: base pointer + byte offsets (the cache-policy bits mean nothing here)
struct __amdgpu_buffer_rsrc_t { char* base; };
template <class T> inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(T* base, int, int, int) { return {(char*)base}; }
typedef unsigned emu_u32x4 __attribute__((vector_size(16)));
inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.base + voff + soff, &v, 16); }
inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.base + voff + soff, &v, 4); }
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { emu_u32x4 v; memcpy(&v, r.base + voff + soff, 16); return v; }
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { unsigned v; memcpy(&v, r.base + voff + soff, 4); return v; }
// the rest of the runtime API the emulated sources mention
typedef void* hipEvent_t;
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };
enum { hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3, hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "host-emulation"); p->multiProcessorCount = 256; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t*) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
