// What does a launch boundary cost against a device-side phase hand-off inside ONE launch?  (first item: the main branch of an update is four launches -
// k_sac_chain, k_dw_adam, k_policy_critic, k_dw_adam - and the timeline puts ~12 of its 67 us at their boundaries.)
//
//   A: four launches per "update" on one stream; phase p reads what phase p - 1 wrote (another workgroup's slab, so the data crosses CUs / XCDs) and writes its own.
//   B: ONE launch of 4 x G workgroups; workgroup b belongs to phase b / G, waits until all G workgroups of the previous phase have arrived (an agent-scope
//      release / acquire pair on a monotonic counter), does the same work, arrives. Workgroups of later phases are dispatched behind the earlier ones (blockIdx order),
//      so every wait points at lower-numbered workgroups.
// Both variants must end in the same bytes (checked): B's numbers mean nothing if its hand-off is wrong. Bounded waits: a wait that gives up raises a flag every later
// wait sees, the run reports it and exits non-zero.
//
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/tools/boundary_probe profiles/tools/boundary_probe.hip     run: profiles/tools/boundary_probe [G] [floats per workgroup] [updates]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// one phase's work for workgroup w of G: out[w][i] = in[(w + 1) % G][i] * 1.0009765625f + phase   (reads ANOTHER workgroup's slab of the previous phase)
__device__ __forceinline__ void phase_work(const float* __restrict__ in, float* __restrict__ out, int w, int G, int n, int phase) {
  const float* src = in + (size_t)((w + 1) % G) * n;
  float* dst = out + (size_t)w * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i] * 1.0009765625f + (float)phase;
}

__global__ __launch_bounds__(256) void k_phase(const float* __restrict__ in, float* __restrict__ out, int n, int phase) { phase_work(in, out, blockIdx.x, gridDim.x, n, phase); }

// buffers: buf[0] -> phase 0 -> buf[1] -> phase 1 -> buf[2] -> phase 2 -> buf[3] -> phase 3 -> buf[0]
struct Bufs { float* b[4]; };

__global__ __launch_bounds__(256) void k_all_phases(Bufs bufs, int n, int G, unsigned* __restrict__ ctr, unsigned epoch, unsigned* __restrict__ gave_up) {
  const int phase = blockIdx.x / G, w = blockIdx.x - phase * G;
  if (phase > 0) {
    if (threadIdx.x == 0) {
      const unsigned target = epoch * (unsigned)G;
      int spins = 0;
      while (__hip_atomic_load(ctr + 32 * (phase - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 21) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the previous phase's stores are visible to this CU from here on
    }
    __syncthreads();
  }
  phase_work(bufs.b[phase], bufs.b[(phase + 1) & 3], w, G, n, phase);
  __syncthreads();   // all of this workgroup's stores are issued ...
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr + 32 * phase, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... and released with the arrival
}

static double checksum(const float* dev, size_t n) {
  std::vector<float> h(n);
  CHECK(hipMemcpy(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost));
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += h[i] * (double)((i % 251) + 1);
  return s;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 128, n = argc > 2 ? atoi(argv[2]) : 4096, updates = argc > 3 ? atoi(argv[3]) : 300, warm = updates / 10;
  const size_t total = (size_t)G * n;
  Bufs a, b;
  std::vector<float> init(total);
  for (size_t i = 0; i < total; ++i) init[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  for (int k = 0; k < 4; ++k) { CHECK(hipMalloc(&a.b[k], total * sizeof(float))); CHECK(hipMalloc(&b.b[k], total * sizeof(float))); CHECK(hipMemset(a.b[k], 0, total * 4)); CHECK(hipMemset(b.b[k], 0, total * 4)); }
  CHECK(hipMemcpy(a.b[0], init.data(), total * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b.b[0], init.data(), total * 4, hipMemcpyHostToDevice));
  unsigned *ctr, *gave_up;
  CHECK(hipMalloc(&ctr, 4 * 32 * sizeof(unsigned))); CHECK(hipMemset(ctr, 0, 4 * 32 * sizeof(unsigned)));
  CHECK(hipMalloc(&gave_up, sizeof(unsigned))); CHECK(hipMemset(gave_up, 0, sizeof(unsigned)));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float msA = 0, msB = 0;
  unsigned epoch = 0;
  for (int variant = 0; variant < 2; ++variant) {
    for (int u = 0; u < warm + updates; ++u) {
      if (u == warm) CHECK(hipEventRecord(e0, st));
      if (variant == 0) {
        for (int p = 0; p < 4; ++p) k_phase<<<G, 256, 0, st>>>(a.b[p], a.b[(p + 1) & 3], n, p);
      } else {
        ++epoch;
        k_all_phases<<<4 * G, 256, 0, st>>>(b, n, G, ctr, epoch, gave_up);
      }
    }
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(variant ? &msB : &msA, e0, e1));
  }
  unsigned gu = 0; CHECK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
  const double ca = checksum(a.b[0], total), cb = checksum(b.b[0], total);
  const double usA = msA * 1e3 / updates, usB = msB * 1e3 / updates;
  printf("{\"workgroups_per_phase\": %d, \"floats_per_workgroup\": %d, \"bytes_per_phase\": %zu, \"four_launches_us\": %.3f, \"one_launch_us\": %.3f, \"saved_us_per_boundary\": %.3f, "
         "\"same_result\": %s, \"expired_waits\": %u}\n", G, n, total * 8, usA, usB, (usA - usB) / 3.0, ca == cb ? "true" : "false", gu);
  return (ca == cb && !gu) ? 0 : 1;
}
