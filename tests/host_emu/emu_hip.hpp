// TEST INFRASTRUCTURE: runs the small plain-VALU HIP kernels of imitation-learning_amd/csrc (no MFMA / DPP / buffer intrinsics: the deep, shaped and shaped-deep GAIL
// discriminators) on the HOST, so that `pytest -m "not gpu"` can execute the very kernel sources against the reference fixtures where there is no GPU.
// tests/host_emu/build.py rewrites `k<<<grid, block, lds, stream>>>(args)` into EMU_LAUNCH and the dynamic `extern __shared__` declaration into a pointer, and compiles
// the result with g++ against this header instead of <hip/hip_runtime.h>.
//
// Execution model: one workgroup at a time; its threads are ucontext fibers on ONE OS thread, switched round-robin at __syncthreads() (a barrier releases when every
// live fiber has arrived) and at __shfl (a lane publishes its value under a per-lane sequence number and yields until the source lane has published the same number).
// Deterministic, no data races by construction; what it cannot show is anything that depends on wave-level lockstep beyond __shfl, or on memory ordering across
// workgroups (the emulated kernels have neither). Sums are not bit-identical to the GPU's (block_sum adds lanes in order, libm is glibc's): tolerances, not equality.
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
struct Fiber { ucontext_t ctx; int state; };
inline ucontext_t sched_ctx;
inline std::vector<Fiber> fibers;
inline std::vector<char> stacks;
inline int cur = 0;
inline std::function<void()> body;
inline void* dyn_smem = nullptr;
inline uint64_t shfl_val[SHFL_RING][MAX_THREADS];
inline unsigned shfl_tag[SHFL_RING][MAX_THREADS];
inline unsigned shfl_seq[MAX_THREADS];
inline float bs_buf[MAX_THREADS];

inline void to_scheduler() { swapcontext(&fibers[cur].ctx, &sched_ctx); }
inline void fiber_entry() {
  body();
  fibers[cur].state = DONE;
  to_scheduler();
}
inline int linear_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }

inline void run_block(const dim3& bdim, const std::function<void()>& fn) {
  const int n = (int)(bdim.x * bdim.y * bdim.z);
  if (n > MAX_THREADS) { fprintf(stderr, "emu: %d threads per block\n", n); abort(); }
  if ((int)fibers.size() < n) { fibers.resize(n); stacks.resize((size_t)n * STACK_BYTES); }
  body = fn;
  for (int i = 0; i < n; ++i) {
    getcontext(&fibers[i].ctx);
    fibers[i].ctx.uc_stack.ss_sp = stacks.data() + (size_t)i * STACK_BYTES;
    fibers[i].ctx.uc_stack.ss_size = STACK_BYTES;
    fibers[i].ctx.uc_link = nullptr;
    makecontext(&fibers[i].ctx, fiber_entry, 0);
    fibers[i].state = RUNNABLE;
    shfl_seq[i] = 0;
  }
  memset(shfl_tag, 0xff, sizeof(shfl_tag));
  for (;;) {
    int runnable = 0, waiting = 0;
    for (int i = 0; i < n; ++i) {
      if (fibers[i].state != RUNNABLE) continue;
      cur = i;
      threadIdx = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
      swapcontext(&sched_ctx, &fibers[i].ctx);
    }
    for (int i = 0; i < n; ++i) { runnable += fibers[i].state == RUNNABLE; waiting += fibers[i].state == AT_BARRIER; }
    if (runnable) continue;
    if (!waiting) break;                                                  // every fiber has returned
    for (int i = 0; i < n; ++i) if (fibers[i].state == AT_BARRIER) fibers[i].state = RUNNABLE;   // barrier: all live fibers arrived
  }
}

template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds_bytes, Args... args) {
  std::vector<char> smem(lds_bytes + 64);
  void* aligned = (void*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
  gridDim = grid; blockDim = block;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        blockIdx = dim3(x, y, z);
        dyn_smem = aligned;
        memset(aligned, 0xcd, lds_bytes);   // LDS is not zero on the GPU either: poison it so that a read of an unwritten word shows
        run_block(block, [&]() { kernel(args...); });
      }
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
