// Developer probe (round 6, profiles/r06_soak_under_load.md): which hand-off forms of gfx950 can show a consumer a line's PREVIOUS value after another XCD has rewritten it?
//
// One launch, two co-resident workgroups on different XCDs (block 0 = reader, 2 waves; block 1 = writer, 4 waves). Trial number `seq` counts up over the whole launch and is
// the payload; the two workgroups meet through two counters (fa: reader -> writer "go", fb: writer -> reader "written"), polled with agent-scope atomic loads like the
// library's waits. A trial's slot is 256 lines (one per writer thread), slots are reused cyclically, so a stale read returns a smaller payload.
//   reader: (methods 0-6) pre-loads line 0 of the slot with a plain load (now in its XCD's L2 and its CU's L1); fa = seq; waits for fb >= seq; reads back and counts STALE results.
// Single-line methods (writer thread 0 writes line 0 through the caches' write-through path, drains, relaxed flag):
//   0 plain load   1 buffer load sc1   2 buffer load sc0 sc1 (the library's sload1)   3 agent-scope atomic load   4 acquire fence by the reading wave + plain load
//   5 acquire fence by WAVE 0, workgroup barrier, plain load by WAVE 1 (round-5 sync_wait)   6 barrier, then acquire fence + plain load by wave 1 (round-6 sync_wait)
// Whole-workgroup producer methods (every writer thread plain-stores its own line; reader reads all 256 lines below the caches, sc0 sc1):
//   7 barrier, thread 0: agent-scope RELEASE add on fb                       (round-5 sync_signal: the other waves' stores may still be in flight behind the write-back)
//   8 every wave s_waitcnt vmcnt(0), barrier, thread 0 RELEASE add            (round-6 sync_signal: sync_drain_stores)
// Build: hipcc --offload-arch=gfx950 -O2 profiles/tools/l2_stale_probe.hip -o profiles/tools/_build/l2_stale_probe ; run: ./l2_stale_probe [trials] [producer trials]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LINE 32        // dwords per 128-byte line
#define SLOT 256       // lines per slot
#define NSLOT 64
#define METHODS 9

__device__ __forceinline__ unsigned load_sc(const unsigned* base, long off, int aux) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(base), 0, 0x7ffffff0, 0x00020000);
  if (aux == 16) return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4), 0, 16);
  return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4), 0, 17);
}
// cacheable accesses the compiler cannot mark (a volatile access would carry sc0 sc1)
__device__ __forceinline__ unsigned plain_load(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void plain_store_nowait(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(unsigned* base, long off, unsigned v) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(v, rs, (int)(off * 4), 0, 17);
}
__device__ __forceinline__ void wait_ge(unsigned* f, unsigned want) { while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1); }
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xfu; }

__global__ __launch_bounds__(256) void k_probe(unsigned* lines, unsigned* fa, unsigned* fb, int trials, int ptrials, unsigned* stale, unsigned* where) {
  const int role = blockIdx.x;   // 0 reader, 1 writer
  if (role > 1) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
  if (role == 0 && tid >= 128) return;   // the reader has two waves (its barriers below are reached by exactly those: the other two have exited)
  if (tid == 0) where[role] = xcc_id();
  unsigned seq = 0;
  for (int m = 0; m < METHODS; ++m) {
    const int n = m >= 7 ? ptrials : trials;
    for (int t = 0; t < n; ++t) {
      ++seq;
      unsigned* slot = lines + (size_t)(seq % NSLOT) * SLOT * LINE;
      if (role == 1) {
        if (tid == 0) wait_ge(fa, seq);
        __syncthreads();
        if (m < 7) {
          if (tid == 0) { store_through(slot, 0, seq); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(fb, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        } else {
          plain_store_nowait(slot + (size_t)tid * LINE, seq);
          if (m == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) __hip_atomic_fetch_add(fb, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        continue;
      }
      // reader
      if (m < 7) { const unsigned before = plain_load(slot + (lane & 31)); if (before >= seq && tid == 0) atomicAdd(&stale[METHODS + m], 1u); }
      __syncthreads();
      if (tid == 0) { __hip_atomic_store(fa, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wait_ge(fb, seq); }
      unsigned got = seq;
      if (m <= 4) {
        if (tid == 0) {
          if (m == 0) got = plain_load(slot);
          else if (m == 1) got = load_sc(slot, 0, 16);
          else if (m == 2) got = load_sc(slot, 0, 17);
          else if (m == 3) got = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); got = plain_load(slot); }
          if (got != seq) atomicAdd(&stale[m], 1u);
        }
        __syncthreads();
      } else if (m == 5) {
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (wave == 1 && lane == 0) { got = plain_load(slot); if (got != seq) atomicAdd(&stale[m], 1u); }
        __syncthreads();
      } else if (m == 6) {
        __syncthreads();
        if (wave == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); if (lane == 0) { got = plain_load(slot); if (got != seq) atomicAdd(&stale[m], 1u); } }
        __syncthreads();
      } else {
        __syncthreads();
        const unsigned a = load_sc(slot, (long)tid * LINE, 17), b = load_sc(slot, (long)(tid + 128) * LINE, 17);
        const bool bad = a != seq || b != seq;
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicAdd(&stale[m], 1u);   // (per wave: a trial can count twice)
        __syncthreads();
      }
    }
  }
}

int main(int argc, char** argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 20000, ptrials = argc > 2 ? atoi(argv[2]) : 200000;
  const size_t n = (size_t)NSLOT * SLOT * LINE;
  unsigned *lines, *flags, *stale, *where;
  hipMalloc(&lines, n * 4); hipMalloc(&flags, 2 * LINE * 4); hipMalloc(&stale, 2 * METHODS * 4); hipMalloc(&where, 8);
  hipMemset(lines, 0, n * 4); hipMemset(flags, 0, 2 * LINE * 4); hipMemset(stale, 0, 2 * METHODS * 4);
  hipLaunchKernelGGL(k_probe, dim3(2), dim3(256), 0, 0, lines, flags, flags + LINE, trials, ptrials, stale, where);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  unsigned h[2 * METHODS], w[2];
  if (hipMemcpy(h, stale, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(w, where, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* names[METHODS] = {"plain load", "buffer load sc1", "buffer load sc0 sc1 (sload1)", "agent-scope atomic load", "acquire fence by the reading wave + plain load",
                                "acquire fence by wave 0, barrier, plain load by wave 1 (round-5 sync_wait)", "barrier, acquire fence + plain load by wave 1 (round-6 sync_wait)",
                                "PRODUCER: 4 waves store, barrier, thread 0 releases (round-5 sync_signal); reader below the caches",
                                "PRODUCER: 4 waves store AND drain (vmcnt 0), barrier, thread 0 releases (round-6 sync_signal)"};
  printf("reader on XCC %u, writer on XCC %u\n", w[0], w[1]);
  for (int m = 0; m < METHODS; ++m) printf("  method %d  stale %7u / %d   (pre-read already new: %u)   %s\n", m, h[m], m >= 7 ? ptrials : trials, h[METHODS + m], names[m]);
  return 0;
}
