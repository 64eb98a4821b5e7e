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

// C (round 4): the fine-grained form of B - a workgroup waits only for the ONE workgroup of the previous phase whose slab it reads (a flag per workgroup and phase, written
// with the producer's release), so the phases pipeline instead of meeting at a grid-wide counter. The update path's analogue: k_policy_critic's layer-l workgroups waiting for
// the dW / AdamW blocks of layer l only (review r3 item 3).
__global__ __launch_bounds__(256) void k_all_phases_fine(Bufs bufs, int n, int G, unsigned* __restrict__ flags, unsigned epoch, unsigned* __restrict__ gave_up) {
  const int phase = blockIdx.x / G, w = blockIdx.x - phase * G;
  if (phase > 0) {
    if (threadIdx.x == 0) {
      const unsigned* f = flags + ((size_t)(phase - 1) * G + (w + 1) % G) * 32;
      int spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  phase_work(bufs.b[phase], bufs.b[(phase + 1) & 3], w, G, n, phase);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + ((size_t)phase * G + w) * 32, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// (phase 0 of update u + 1 reads what phase 3 of update u wrote: across launches, ordered by the stream)

// D (round 4): C without the fences - the mechanism of the pair-mode kernels (csrc/il_common.hpp wstore / sload): the producer stores WRITE-THROUGH (sc0 sc1: no dirty line
// is left in its XCD's L2), drains them (s_waitcnt vmcnt(0)), and raises its flag with a relaxed store; the consumer polls the flag and then reads the slab with sc0 sc1 loads
// (below its L1 and its XCD's L2). No buffer_wbl2, no buffer_inv.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_all_phases_through(Bufs bufs, int n, int G, unsigned* __restrict__ flags, unsigned epoch, unsigned* __restrict__ gave_up, int first_phase_plain) {
  const int phase = blockIdx.x / G, w = blockIdx.x - phase * G;
  if (phase > 0) {
    if (threadIdx.x == 0) {
      const unsigned* f = flags + ((size_t)(phase - 1) * G + (w + 1) % G) * 32;
      int spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21) || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
  }
  const float* src = bufs.b[phase] + (size_t)((w + 1) % G) * n;
  float* dst = bufs.b[(phase + 1) & 3] + (size_t)w * n;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7ffffff0, 0x00020000);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    // phase 0 reads what the PREVIOUS launch wrote through: the launch boundary invalidated this L2, a plain load is correct there too (same instruction kept for symmetry)
    const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, i * 4, 0, 17));
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v * 1.0009765625f + (float)phase), rd, i * 4, 0, 17);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + ((size_t)phase * G + w) * 32, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  Bufs a, b, c, dd;
  std::vector<float> init(total);
  for (size_t i = 0; i < total; ++i) init[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  for (int k = 0; k < 4; ++k) { CHECK(hipMalloc(&a.b[k], total * sizeof(float))); CHECK(hipMalloc(&b.b[k], total * sizeof(float))); CHECK(hipMalloc(&c.b[k], total * sizeof(float))); CHECK(hipMalloc(&dd.b[k], total * sizeof(float))); CHECK(hipMemset(dd.b[k], 0, total * 4)); CHECK(hipMemset(a.b[k], 0, total * 4)); CHECK(hipMemset(b.b[k], 0, total * 4)); CHECK(hipMemset(c.b[k], 0, total * 4)); }
  CHECK(hipMemcpy(a.b[0], init.data(), total * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b.b[0], init.data(), total * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(c.b[0], init.data(), total * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dd.b[0], init.data(), total * 4, hipMemcpyHostToDevice));
  unsigned* flags_d; CHECK(hipMalloc(&flags_d, (size_t)4 * G * 32 * sizeof(unsigned))); CHECK(hipMemset(flags_d, 0, (size_t)4 * G * 32 * sizeof(unsigned)));
  unsigned* flags; CHECK(hipMalloc(&flags, (size_t)4 * G * 32 * sizeof(unsigned))); CHECK(hipMemset(flags, 0, (size_t)4 * G * 32 * sizeof(unsigned)));
  unsigned *ctr, *gave_up;
  CHECK(hipMalloc(&ctr, 4 * 32 * sizeof(unsigned))); CHECK(hipMemset(ctr, 0, 4 * 32 * sizeof(unsigned)));
  CHECK(hipMalloc(&gave_up, sizeof(unsigned))); CHECK(hipMemset(gave_up, 0, sizeof(unsigned)));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float msA = 0, msB = 0, msC = 0, msD = 0;
  unsigned epoch = 0, epoch_c = 0, epoch_d = 0;
  for (int variant = 0; variant < 4; ++variant) {
    for (int u = 0; u < warm + updates; ++u) {
      if (u == warm) CHECK(hipEventRecord(e0, st));
      if (variant == 0) {
        for (int p = 0; p < 4; ++p) k_phase<<<G, 256, 0, st>>>(a.b[p], a.b[(p + 1) & 3], n, p);
      } else if (variant == 1) {
        ++epoch;
        k_all_phases<<<4 * G, 256, 0, st>>>(b, n, G, ctr, epoch, gave_up);
      } else if (variant == 2) {
        ++epoch_c;
        k_all_phases_fine<<<4 * G, 256, 0, st>>>(c, n, G, flags, epoch_c, gave_up);
      } else {
        ++epoch_d;
        k_all_phases_through<<<4 * G, 256, 0, st>>>(dd, n, G, flags_d, epoch_d, gave_up, 1);
      }
    }
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(variant == 0 ? &msA : (variant == 1 ? &msB : (variant == 2 ? &msC : &msD)), e0, e1));
  }
  unsigned gu = 0; CHECK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
  const double ca = checksum(a.b[0], total), cb = checksum(b.b[0], total), cc = checksum(c.b[0], total), cd = checksum(dd.b[0], total);
  const double usA = msA * 1e3 / updates, usB = msB * 1e3 / updates, usC = msC * 1e3 / updates, usD = msD * 1e3 / updates;
  printf("{\"workgroups_per_phase\": %d, \"floats_per_workgroup\": %d, \"bytes_per_phase\": %zu, \"four_launches_us\": %.3f, \"one_launch_us\": %.3f, \"saved_us_per_boundary\": %.3f, "
         "\"one_launch_per_producer_flags_us\": %.3f, \"saved_us_per_boundary_fine\": %.3f, \"one_launch_write_through_flags_us\": %.3f, \"saved_us_per_boundary_through\": %.3f, \"same_result\": %s, \"expired_waits\": %u}\n",
         G, n, total * 8, usA, usB, (usA - usB) / 3.0, usC, (usA - usC) / 3.0, usD, (usA - usD) / 3.0, (ca == cb && ca == cc && ca == cd) ? "true" : "false", gu);
  return (ca == cb && ca == cc && ca == cd && !gu) ? 0 : 1;
}
