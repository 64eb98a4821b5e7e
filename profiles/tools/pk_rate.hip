// Developer tool: issue rate of v_pk_fma_f32 / v_pk_add_f32 / v_fma_f32 / ds_read_b128 on gfx950, in s_memtime cycles per wave instruction, with 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_IT 2000

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * i;
  __syncthreads();
  f32x2 a[8], d = {seed + threadIdx.x * 1e-6f, seed * 2.f - threadIdx.x * 1e-6f};   // per-lane values: VGPR operands like the kernel's
  float s[16];
  unsigned long long sc[4];   // wave-uniform 64-bit values: scalar register pairs for the "s" operands
#pragma unroll
  for (int i = 0; i < 4; ++i) sc[i] = __builtin_amdgcn_readfirstlane(__float_as_uint(seed + i)) | ((unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(seed - i)) << 32);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f32x2{seed + i, seed - i};
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = seed + i;
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned lds_addr = (threadIdx.x & 63) * 16;
  // inline asm: exactly the instructions named, in this order, nothing vectorised or re-associated by the compiler
#pragma unroll 1
  for (int it = 0; it < N_IT; ++it) {
    if (KIND == 0) {        // 16 independent v_pk_fma_f32
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(d), "v"(a[(i + 1) & 7]));
    } else if (KIND == 1) {  // 8 x (v_pk_add_f32 with op_sel broadcast and negated second source + v_pk_fma_f32 of the difference): the k_gmmil_tile step
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x2 df;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(df) : "v"(d), "v"(a[(i + 3) & 7]));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(df));
      }
    } else if (KIND == 2) {  // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(d[0]), "v"(s[(i + 1) & 15]));
    } else if (KIND == 4) {  // k_gmmil_sx's order: per row two v_pk_add_f32 with a broadcast SCALAR row operand, then the two v_pk_fma_f32 that consume them (distance 2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 d0, d1;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "s"(sc[i]), "v"(a[6]));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "s"(sc[i]), "v"(a[7]));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(d0));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[(i + 4) & 7 ? i + 1 : 5]) : "v"(d1));
      }
    } else if (KIND == 5) {  // the same 16 instructions with all eight adds first (distance 8)
      f32x2 dd[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dd[2 * i]) : "s"(sc[i]), "v"(a[6]));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dd[2 * i + 1]) : "s"(sc[i]), "v"(a[7]));
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(dd[i]));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[0]) : "v"(dd[6]));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[1]) : "v"(dd[7]));
    } else if (KIND == 6) {  // KIND 4 with the row operand in a VECTOR register pair
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 d0, d1;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "v"(d), "v"(a[6]));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "v"(d), "v"(a[7]));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(d0));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a[(i + 4) & 7 ? i + 1 : 5]) : "v"(d1));
      }
    } else {                 // 4 ds_read_b128 (conflict-free, 16 B per lane, 1 KiB per wave instruction), then one wait
      f32x4 v0, v1, v2, v3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(lds_addr) : "memory");
      acc4 += v0 + v1 + v2 + v3;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = acc4[0] + acc4[1] + acc4[2] + acc4[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1];
#pragma unroll
  for (int i = 0; i < 16; ++i) r += s[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_it, int threads, unsigned long long* d, float* sink) {
  static unsigned long long h[256];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256, threads>>>(d, sink, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(e0); k<KIND><<<256, threads>>>(d, sink, 1.0001f); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  // 256 workgroups = one per CU; `threads`/256 waves per SIMD; ticks / kernel time calibrates the s_memtime rate
  printf("%-34s %4d thr/WG: %7.2f memtime ticks per wave instruction (%.0f ticks in a %.1f us kernel, %d instr per wave)\n", name, threads, avg / ((double)N_IT * per_it), avg, ms * 1e3, N_IT * per_it);
}

int main() {
  unsigned long long* d; float* sink; hipMalloc(&d, 1024 * 8); hipMalloc(&sink, 1024 * 512 * 4);
  for (int threads = 256; threads <= 1024; threads *= 2) {
    run<0>("v_pk_fma_f32 x16 independent", 16, threads, d, sink);
    run<1>("v_pk_add(op_sel) + v_pk_fma x8", 16, threads, d, sink);
    run<2>("v_fma_f32 x16 independent", 16, threads, d, sink);
    run<3>("ds_read_b128 x4 (+4 v_pk_add)", 4, threads, d, sink);
    run<4>("sx order: 2 pk_add(sgpr) + 2 pk_fma, x4", 16, threads, d, sink);
    run<5>("8 pk_add(sgpr) then 8 pk_fma", 16, threads, d, sink);
    run<6>("sx order with a VGPR row operand", 16, threads, d, sink);
  }
  return 0;
}
