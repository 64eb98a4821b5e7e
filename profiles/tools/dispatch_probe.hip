// Developer tool: how fast does the dispatcher START the workgroups of one launch, as a function of workgroup size, dynamic LDS and VGPRs per lane?
// (Question from profiles/r04_population_disc_timeline.txt: the 512 workgroups of k_gail_reward - 256 threads, 34 KB of LDS, 114 VGPRs - start over 17 us although four of
// them fit a CU, while the 512-thread tile kernels' first 512 workgroups start within 1.2 us.)
// Every workgroup stores s_memrealtime (100 MHz) and HW_ID / XCC_ID at its first instruction, then spins `spin` us.
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/dispatch_probe.hip -o profiles/tools/dispatch_probe && profiles/tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
template <int T, int VG, int PRE>
__global__ __launch_bounds__(T) void probe(Rec* out, int spin_ticks, const float* __restrict__ big, int nbig) {
  extern __shared__ float smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (VG >= 100) asm volatile("v_mov_b32 v110, 0" ::: "v110");
  if (VG >= 120) asm volatile("v_mov_b32 v125, 0" ::: "v125");
  float acc = 0.f;
  if (PRE) {   // a prologue that loads (like the real kernels' parameter staging): nbig floats per workgroup through LDS
    for (int i = threadIdx.x; i < nbig; i += T) smem[i] = big[(size_t)(blockIdx.x % 32) * nbig + i];
    __syncthreads();
    for (int i = threadIdx.x; i < nbig; i += T) acc += smem[(i * 7) % nbig];
  }
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Rec r; r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw; r.xcc = xcc & 0xf;
    out[blockIdx.x] = r;
    if (acc == 12345.f) out[0].hw = 0;
  }
}
template <int T, int VG, int PRE>
static void run(const char* name, int wgs, size_t lds, int spin_us, Rec* d, const float* big) {
  hipFuncSetAttribute((const void*)probe<T, VG, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds, (size_t)65536));
  for (int rep = 0; rep < 3; ++rep) { probe<T, VG, PRE><<<wgs, T, lds>>>(d, spin_us * 100, big, 6144); }
  hipDeviceSynchronize();
  std::vector<Rec> h(wgs); hipMemcpy(h.data(), d, wgs * sizeof(Rec), hipMemcpyDeviceToHost);
  unsigned long long m = ~0ull; for (auto& r : h) m = std::min(m, r.t0);
  std::vector<double> s; for (auto& r : h) s.push_back((r.t0 - m) / 100.0);
  std::vector<double> so = s; std::sort(so.begin(), so.end());
  auto pc = [&](double p) { return so[(size_t)(p * (wgs - 1))]; };
  // distinct CUs used: (xcc, se, sh, cu)
  std::vector<unsigned> cus; for (auto& r : h) cus.push_back((r.xcc << 16) | ((r.hw >> 8) & 0xff));
  std::sort(cus.begin(), cus.end()); const size_t ncu = std::unique(cus.begin(), cus.end()) - cus.begin();
  printf("%-44s %4d wgs x %4d thr, lds %6zu B, spin %2d us: start p10/p50/p90/max %.2f %.2f %.2f %.2f us | distinct CUs %zu | first 16 by id:", name, wgs, T, lds, spin_us, pc(.1), pc(.5), pc(.9), pc(1.0), ncu);
  for (int i = 0; i < 16; ++i) printf(" %.1f", s[i]);
  printf("\n");
}
int main() {
  Rec* d; hipMalloc(&d, 8192 * sizeof(Rec));
  float* big; hipMalloc(&big, 32 * 6144 * 4); hipMemset(big, 0, 32 * 6144 * 4);
  run<256, 0, 0>("256 thr, few VGPRs, no LDS", 512, 0, 10, d, big);
  run<256, 0, 0>("256 thr, few VGPRs, 34 KB", 512, 34 * 1024, 10, d, big);
  run<256, 110, 0>("256 thr, 111 VGPRs, 34 KB", 512, 34 * 1024, 10, d, big);
  run<256, 110, 1>("256 thr, 111 VGPRs, 34 KB, loading prologue", 512, 34 * 1024, 10, d, big);
  run<256, 110, 0>("256 thr, 111 VGPRs, 73 KB", 512, 73 * 1024, 10, d, big);
  run<256, 110, 0>("256 thr, 111 VGPRs, 73 KB, 1536 wgs", 1536, 73 * 1024, 10, d, big);
  run<512, 0, 0>("512 thr, few VGPRs, 50 KB", 512, 50 * 1024, 10, d, big);
  run<512, 0, 0>("512 thr, few VGPRs, 50 KB, 1024 wgs", 1024, 50 * 1024, 10, d, big);
  run<256, 126, 0>("256 thr, 126 VGPRs, 34 KB", 512, 34 * 1024, 10, d, big);
  run<256, 110, 0>("256 thr, 111 VGPRs, 34 KB, 1024 wgs", 1024, 34 * 1024, 10, d, big);
  return 0;
}
