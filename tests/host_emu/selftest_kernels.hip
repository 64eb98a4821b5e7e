// TEST INFRASTRUCTURE: kernels that check the host emulator itself (tests/test_kernels_host_emulation.py::test_emulator_*), built by the same translation as the product's.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = A[16x4K] B[4Kx16] by one wave with the 16x16x4 MFMA, operands and result in the lane layout the product's kernels assume:
// lane l supplies A[l & 15][k0 + (l >> 4)] and B[k0 + (l >> 4)][l & 15], and holds D[4 (l >> 4) + r][l & 15]
__global__ void k_selftest_mfma(const float* A, const float* B, float* D, int K4) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < 4 * K4; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 * K4 + k0 + g], B[(k0 + g) * 16 + i], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}

// out[0][l] = row_ror:1 of the lane id, out[1][l] = row_shr:1 (bound_ctrl off, old = -1), out[2][l] = readlane(17), out[3][l] = popcount of ballot(l % 3 == 0),
// out[4][l] = shfl_xor 5, out[5][l] = quad_perm [1, 0, 3, 2]
__global__ void k_selftest_lanes(int* out) {
  const int l = threadIdx.x;
  out[0 * 64 + l] = __builtin_amdgcn_update_dpp(0, l, 0x121, 0xf, 0xf, false);
  out[1 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, false);
  out[2 * 64 + l] = __builtin_amdgcn_readlane(l * 10, 17);
  out[3 * 64 + l] = __popcll(__ballot(l % 3 == 0));
  out[4 * 64 + l] = __shfl_xor(l, 5);
  out[5 * 64 + l] = __builtin_amdgcn_update_dpp(0, l, 0xb1, 0xf, 0xf, false);
}

// Wave 1 reads what wave 0 wrote to LDS. with_barrier = 0 is a race: the forward schedule (wave 0 runs first) hides it, the reverse schedule reads the poison.
__global__ void k_selftest_race(float* out, int with_barrier) {
  __shared__ float buf[64];
  if (threadIdx.x < 64) buf[threadIdx.x] = (float)threadIdx.x;
  if (with_barrier) __syncthreads();
  if (threadIdx.x >= 64) out[threadIdx.x - 64] = buf[127 - threadIdx.x];
}

// a load one element past a buffer: nothing notices on the GPU; the sanitised emulation does
__global__ void k_selftest_overrun(const float* in, float* out, int n) {
  const int i = threadIdx.x;
  if (i <= n) out[i < n ? i : 0] = in[i];
}

extern "C" int selftest_mfma(const float* A, const float* B, float* D, int K4) { k_selftest_mfma<<<1, 64, 0, 0>>>(A, B, D, K4); return 0; }
extern "C" int selftest_lanes(int* out) { k_selftest_lanes<<<1, 64, 0, 0>>>(out); return 0; }
extern "C" int selftest_race(float* out, int with_barrier) { k_selftest_race<<<1, 128, 0, 0>>>(out, with_barrier); return 0; }
extern "C" int selftest_overrun(const float* in, float* out, int n) { k_selftest_overrun<<<1, 64, 0, 0>>>(in, out, n); return 0; }
