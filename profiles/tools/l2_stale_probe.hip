// Developer probe (round 6, profiles/r06_soak_under_load.md): which load flavours of gfx950 can return a line's PREVIOUS value after another XCD has rewritten it?
//
// One launch, two co-resident workgroups on different XCDs (block 0 = reader, block 1 = writer; the other blocks exit). Per trial t (its own 128-byte line):
//   reader: plain load of line[t] (the line is now in the reader XCD's L2 and this CU's L1), then flagA[t] = 1
//   writer: waits for flagA[t], rewrites line[t] (write-through store, drained), then flagB[t] = 1
//   reader: waits for flagB[t] (agent-scope atomic poll, like the library's waits), then reads line[t] again with the method under test and counts a STALE result.
// Methods: 0 plain load; 1 buffer load sc1; 2 buffer load sc0 sc1 (the library's sload1); 3 agent-scope atomic load; 4 acquire fence by the reading wave, then plain load;
//          5 acquire fence by WAVE 0, workgroup barrier, plain load by WAVE 1 (the round-5 form of sync_wait); 6 barrier first, then acquire fence + plain load by wave 1
//          (the round-6 form); 7 method 5 with the writer using plain stores + a release fence instead of write-through stores.
// Build: hipcc --offload-arch=gfx950 -O2 profiles/tools/l2_stale_probe.hip -o profiles/tools/_build/l2_stale_probe ; run: ./l2_stale_probe [trials]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LINE 32   // floats per 128-byte line
#define METHODS 8

__device__ __forceinline__ float load_sc(const float* base, long off, int aux) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7ffffff0, 0x00020000);
  if (aux == 16) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4), 0, 16));
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4), 0, 17));
}
// a cacheable load the compiler cannot mark (a volatile access would carry sc0 sc1)
__device__ __forceinline__ float plain_load(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void plain_store(float* p, float v) { asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(float* base, long off, float v) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)(off * 4), 0, 17);
}
__device__ __forceinline__ void wait_flag(int* f) { while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2); }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xfu; }

__global__ __launch_bounds__(128) void k_probe(float* lines, int* flagA, int* flagB, int trials, unsigned* stale, unsigned* where) {
  const int role = blockIdx.x;   // 0 reader, 1 writer
  if (role > 1) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) where[role] = xcc_id();
  __shared__ float seen;
  for (int m = 0; m < METHODS; ++m) {
    for (int t = 0; t < trials; ++t) {
      const long idx = (long)m * trials + t;
      float* line = lines + idx * LINE;
      int* fa = flagA + idx * LINE; int* fb = flagB + idx * LINE;   // flags on their own lines
      if (role == 1) {
        if (threadIdx.x == 0) {
          wait_flag(fa);
          if (m == 7) { plain_store(line, (float)(t + 1)); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
          else { store_through(line, 0, (float)(t + 1)); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
          __hip_atomic_store(fb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        continue;
      }
      // reader: every lane of both waves pulls the line into L1 / L2 first
      const float before = plain_load(line + (lane & 31));
      __syncthreads();
      if (threadIdx.x == 0) { if (before != 0.f) atomicAdd(&stale[METHODS + m], 1u); __hip_atomic_store(fa, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wait_flag(fb); }
      float got = -1.f;
      if (m <= 4) {
        if (threadIdx.x == 0) {
          if (m == 0) got = plain_load(line);
          else if (m == 1) got = load_sc(line, 0, 16);
          else if (m == 2) got = load_sc(line, 0, 17);
          else if (m == 3) got = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); got = plain_load(line); }
          if (got != (float)(t + 1)) atomicAdd(&stale[m], 1u);
        }
        __syncthreads();
      } else if (m == 5 || m == 7) {
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (wave == 1 && lane == 0) { got = plain_load(line); if (got != (float)(t + 1)) atomicAdd(&stale[m], 1u); }
        __syncthreads();
      } else {
        __syncthreads();
        if (wave == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); if (lane == 0) { got = plain_load(line); if (got != (float)(t + 1)) atomicAdd(&stale[m], 1u); } }
        __syncthreads();
      }
    }
  }
}

int main(int argc, char** argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 20000;
  const long n = (long)METHODS * trials * LINE;
  float* lines; int *fa, *fb; unsigned *stale, *where;
  hipMalloc(&lines, n * 4); hipMalloc(&fa, n * 4); hipMalloc(&fb, n * 4); hipMalloc(&stale, 2 * METHODS * 4); hipMalloc(&where, 8);
  hipMemset(lines, 0, n * 4); hipMemset(fa, 0, n * 4); hipMemset(fb, 0, n * 4); hipMemset(stale, 0, 2 * METHODS * 4);
  hipLaunchKernelGGL(k_probe, dim3(2), dim3(128), 0, 0, lines, fa, fb, trials, stale, where);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  unsigned h[2 * METHODS], w[2];
  hipMemcpy(h, stale, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(w, where, sizeof(w), hipMemcpyDeviceToHost);
  const char* names[METHODS] = {"plain load", "buffer load sc1", "buffer load sc0 sc1 (sload1)", "agent-scope atomic load", "acquire fence by the reading wave + plain load",
                                "acquire fence by wave 0, barrier, plain load by wave 1 (round-5 sync_wait)", "barrier, acquire fence + plain load by wave 1 (round-6 sync_wait)",
                                "round-5 form, writer = plain store + release fence"};
  printf("reader on XCC %u, writer on XCC %u, %d trials per method\n", w[0], w[1], trials);
  for (int m = 0; m < METHODS; ++m) printf("  method %d  stale %6u / %d   (first read not 0: %u)   %s\n", m, h[m], trials, h[METHODS + m], names[m]);
  return 0;
}
