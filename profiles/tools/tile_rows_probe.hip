// Developer probe (round 2): is a hidden layer of an 8-row tile on v_mfma_f32_4x4x1_16B_f32 (4 rows x 16 columns x 4 k-groups per instruction, same MAC rate as 16x16x4)
// really twice as fast as the 16-row tile on v_mfma_f32_16x16x4_f32 once the 256 KB weight panel has to come out of L2 for half the rows?
//   A  16 rows, 16x16x4 (mlp_tile.hpp tile_packed: the product kernels' layer)           grid = 96 workgroups (k_sac_chain's occupancy)
//   B  RG row groups of 4 rows on 4x4x1, same PF panel, same lane -> (column, k-group) map  grid = 96 * 4 / RG
// Each workgroup runs LAYERS hidden layers back to back (ReLU epilogue into the other LDS slab, barrier), all workgroups of a "network" stream the same PF copy.
// Prints per-layer time from s_memrealtime stamps (median over workgroups) and checks B against A on the rows they share.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I imitation-learning_amd/csrc profiles/tools/tile_rows_probe.hip -o /tmp/tile_rows_probe && /tmp/tile_rows_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mlp_tile.hpp"
int il_set_error(int code, const char*, ...) { return code; }
il_trace_scope::il_trace_scope(const char*, hipStream_t s) : st(s), slot(-1) {}
il_trace_scope::~il_trace_scope() {}

#define H 256
#define LDH (H + 4)
#define LAYERS 4
#define NETS 6

// the product's hidden layer up to round 3's broadcast form (mlp_tile.hpp then): one wave per 16 output columns on v_mfma_f32_16x16x4_f32, the wave's whole panel requested up front
template <int PANEL = 16, class Epi>
__device__ __forceinline__ void tile_packed16(const float* As, int lda, int Hh, const float* __restrict__ P, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4, nb = Hh >> 4;
  for (int t = wave; t < nb; t += nw) {
    f32x4 acc0 = zero4(), acc1 = zero4();
    const float* pp = P + (size_t)t * nb * 256 + lane * 4;
    const float* ar = As + j * lda + 4 * g;
    for (int kb = 0; kb + PANEL <= nb; kb += PANEL) {
      f32x4 b[PANEL];
#pragma unroll
      for (int u = 0; u < PANEL; ++u) b[u] = gload4(pp + (size_t)(kb + u) * 256);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < PANEL; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * (kb + u));
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
    epi(t * 16, acc0 + acc1);
  }
}
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// sum over the four 16-lane rows of the wave (the four k-groups), total in every lane: (g0 + g1) + (g2 + g3)
__device__ __forceinline__ float ksum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
// Y[4 RG x H] = Xs . W^T with W as its PF copy; lane (g, j) of wave t holds W[16 t + j][16 u + 4 g + r] in b[u][r] exactly as in tile_packed.
// epi(c0, rg, acc): acc[i] = Y[row 4 rg + i][col c0 + j] (valid in every lane)
template <int RG, class Epi>
__device__ __forceinline__ void tile_packed4(const float* As, int lda, const float* __restrict__ P, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int g = lane >> 4, nb = H >> 4;
  for (int t = wave; t < nb; t += nw) {
    f32x4 acc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[rg] = zero4();
    const float* pp = P + (size_t)t * nb * 256 + lane * 4;
    const float* ar = As + (lane & 3) * lda + 4 * g;
    f32x4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) b[u] = gload4(pp + (size_t)u * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      f32x4 a[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a[rg] = *reinterpret_cast<const f32x4*>(ar + 4 * rg * lda + 16 * u);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = mfma4(a[rg][r], b[u][r], acc[rg]);
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      f32x4 s;
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] = ksum(acc[rg][i]);
      epi(t * 16, rg, s);
    }
  }
}

__global__ __launch_bounds__(1024) void k_probe_a(const float* __restrict__ PF, const float* __restrict__ X, float* __restrict__ Y, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* S0 = smem; float* S1 = smem + 16 * LDH;
  const int wg = blockIdx.x, net = wg % NETS, tile = wg / NETS;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) S0[(i / H) * LDH + i % H] = X[(size_t)(tile * 16 + i / H) * H + i % H];
  __syncthreads();
  float* src = S0; float* dst = S1;
  for (int l = 0; l < LAYERS; ++l) {
    if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + l] = __builtin_amdgcn_s_memrealtime();
    tile_packed16(src, LDH, H, PF + (size_t)net * H * H, [&](int c0, f32x4 acc) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(4 * g + r) * LDH + c0 + j] = fmaxf(acc[r] * 0.05f, -1.f);
    });
    __syncthreads();
    float* t = src; src = dst; dst = t;
  }
  if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + LAYERS] = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) Y[(size_t)(wg * 16 + i / H) * H + i % H] = src[(i / H) * LDH + i % H];
}

// A0: variant A with the wave's panel loaded ONCE before the layers (every layer of the probe reads the same PF copy): LDS reads + MFMAs + epilogue + barrier only.
// A1: variant A with the panel loads but a single MFMA per k-block (a quarter of the matrix work): what the operand stream alone costs.
template <int MODE>
__global__ __launch_bounds__(1024) void k_probe_a01(const float* __restrict__ PF, const float* __restrict__ X, float* __restrict__ Y, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* S0 = smem; float* S1 = smem + 16 * LDH;
  const int wg = blockIdx.x, net = wg % NETS, tile = wg / NETS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4, nb = H >> 4;
  for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) S0[(i / H) * LDH + i % H] = X[(size_t)(tile * 16 + i / H) * H + i % H];
  const float* pp = PF + (size_t)net * H * H + (size_t)wave * nb * 256 + lane * 4;
  f32x4 b[16];
  if (MODE == 0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) b[u] = gload4(pp + (size_t)u * 256);
  }
  __syncthreads();
  float* src = S0; float* dst = S1;
  for (int l = 0; l < LAYERS; ++l) {
    if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + l] = __builtin_amdgcn_s_memrealtime();
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) b[u] = gload4(pp + (size_t)u * 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 acc0 = zero4(), acc1 = zero4();
    const float* ar = src + j * LDH + 4 * g;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * u);
      if (MODE == 1) { acc0 = mfma16(a[0] + a[1], b[u][0] + b[u][1] + b[u][2] + b[u][3], acc0); acc1[0] += a[2] + a[3]; }
      else {
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * g + r) * LDH + wave * 16 + j] = fmaxf(acc[r] * 0.05f, -1.f);
    __syncthreads();
    float* t = src; src = dst; dst = t;
  }
  if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + LAYERS] = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) Y[(size_t)(wg * 16 + i / H) * H + i % H] = src[(i / H) * LDH + i % H];
}

// P<RT>: the POPULATION shape of the same layer - 512-thread workgroups (8 waves, two 16-column tiles each), thousands of workgroups, every `NETS_POP` consecutive
// tiles share a weight copy - with RT row tiles of 16 rows per workgroup that share every weight fragment (RT = 1: today's kernels; RT = 2: half the L2 -> CU bytes per MFMA).
#define NETS_POP 64
template <int RT>
__global__ __launch_bounds__(512) void k_probe_p(const float* __restrict__ PF, const float* __restrict__ X, float* __restrict__ Y, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16 * RT;
  float* S0 = smem; float* S1 = smem + R * LDH;
  const int wg = blockIdx.x, net = (wg * RT / 16) % NETS_POP;   // 16 row tiles per learner-network
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4, nb = H >> 4;
  for (int i = threadIdx.x; i < R * H; i += blockDim.x) S0[(i / H) * LDH + i % H] = X[(size_t)((wg * R + i / H) % 256) * H + i % H];
  __syncthreads();
  const float* P = PF + (size_t)(net % NETS) * H * H + (size_t)(net / NETS) * 64;   // NETS distinct copies; the offset only de-aliases the addresses a little
  float* src = S0; float* dst = S1;
  for (int l = 0; l < LAYERS; ++l) {
    if (threadIdx.x == 0 && wg < 4096) stamps[wg * (LAYERS + 1) + l] = __builtin_amdgcn_s_memrealtime();
    for (int t = wave; t < nb; t += 8) {
      f32x4 b[16];
      const float* pp = P + (size_t)t * nb * 256 + lane * 4;
#pragma unroll
      for (int u = 0; u < 16; ++u) b[u] = gload4(pp + (size_t)u * 256);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { acc[rt][0] = zero4(); acc[rt][1] = zero4(); }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(src + (16 * rt + j) * LDH + 4 * g + 16 * u);
          acc[rt][0] = mfma16(a[0], b[u][0], acc[rt][0]);
          acc[rt][1] = mfma16(a[1], b[u][1], acc[rt][1]);
          acc[rt][0] = mfma16(a[2], b[u][2], acc[rt][0]);
          acc[rt][1] = mfma16(a[3], b[u][3], acc[rt][1]);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 a4 = acc[rt][0] + acc[rt][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * rt + 4 * g + r) * LDH + t * 16 + j] = fmaxf(a4[r] * 0.05f, -1.f);
      }
    }
    __syncthreads();
    float* t2 = src; src = dst; dst = t2;
  }
  if (threadIdx.x == 0 && wg < 4096) stamps[wg * (LAYERS + 1) + LAYERS] = __builtin_amdgcn_s_memrealtime();
  if (wg < NETS * 16 / RT) for (int i = threadIdx.x; i < R * H; i += blockDim.x) Y[(size_t)(wg * R + i / H) * H + i % H] = src[(i / H) * LDH + i % H];
}

// C: the product layer (16 rows, 16x16x4) in a 512-thread workgroup - 8 waves, two 16-column tiles each, 256 VGPRs per wave - with the NEXT panel (the wave's second tile,
// then its first tile of the next layer) requested before the current tile's 64 MFMAs: the L2 round trip and the epilogue of one tile hide under the other's MFMAs instead
// of under other waves (thread-level parallelism needs 4 waves per SIMD = 128 VGPRs each, which leaves no room for a second panel).
__global__ __launch_bounds__(512) void k_probe_c(const float* __restrict__ PF, const float* __restrict__ X, float* __restrict__ Y, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* S0 = smem; float* S1 = smem + 16 * LDH;
  const int wg = blockIdx.x, net = wg % NETS, tile = wg / NETS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4, nb = H >> 4;
  for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) S0[(i / H) * LDH + i % H] = X[(size_t)(tile * 16 + i / H) * H + i % H];
  const float* P = PF + (size_t)net * H * H;
  f32x4 b[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) b[u] = gload4(P + (size_t)wave * nb * 256 + lane * 4 + (size_t)u * 256);
  __syncthreads();
  float* src = S0; float* dst = S1;
#pragma unroll
  for (int l = 0; l < LAYERS; ++l) {   // straight-line like the product kernels (layer 1, layer 2, head ...): no in-flight loads across a loop back-edge
    if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + l] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const int t = wave + 8 * ti;
      const int tn = ti == 0 ? wave + 8 : wave;   // next panel: second tile of this layer, then the first tile of the next layer (same weights here: every layer reads the same PF copy)
      f32x4 nbuf[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) nbuf[u] = gload4(P + (size_t)tn * nb * 256 + lane * 4 + (size_t)u * 256);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = zero4(), acc1 = zero4();
      const float* ar = src + j * LDH + 4 * g;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * u);
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 acc = acc0 + acc1;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(4 * g + r) * LDH + t * 16 + j] = fmaxf(acc[r] * 0.05f, -1.f);
#pragma unroll
      for (int u = 0; u < 16; ++u) b[u] = nbuf[u];
    }
    __syncthreads();
    float* t2 = src; src = dst; dst = t2;
  }
  if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + LAYERS] = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) Y[(size_t)(wg * 16 + i / H) * H + i % H] = src[(i / H) * LDH + i % H];
}

template <int RG>
__global__ __launch_bounds__(1024) void k_probe_b(const float* __restrict__ PF, const float* __restrict__ X, float* __restrict__ Y, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 4 * RG;
  float* S0 = smem; float* S1 = smem + R * LDH;
  const int wg = blockIdx.x, net = wg % NETS, tile = wg / NETS;   // `tile` counts R-row tiles
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < R * H; i += blockDim.x) S0[(i / H) * LDH + i % H] = X[(size_t)(tile * R + i / H) * H + i % H];
  __syncthreads();
  float* src = S0; float* dst = S1;
  for (int l = 0; l < LAYERS; ++l) {
    if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + l] = __builtin_amdgcn_s_memrealtime();
    tile_packed4<RG>(src, LDH, PF + (size_t)net * H * H, [&](int c0, int rg, f32x4 s) {
      // every k-group holds the totals: group g stores row g of the row group (one 4-byte LDS store per lane and row group)
      dst[(4 * rg + g) * LDH + c0 + j] = fmaxf((g == 0 ? s[0] : g == 1 ? s[1] : g == 2 ? s[2] : s[3]) * 0.05f, -1.f);
    });
    __syncthreads();
    float* t = src; src = dst; dst = t;
  }
  if (threadIdx.x == 0) stamps[wg * (LAYERS + 1) + LAYERS] = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < R * H; i += blockDim.x) Y[(size_t)(wg * R + i / H) * H + i % H] = src[(i / H) * LDH + i % H];
}

// D / Q (round 3): the multi-block 4x4x1 MFMA with its A operand BROADCAST (cbsz = 4: the four A values of block `abid` feed all 16 blocks). One A VGPR then carries
// 4 rows x 16 k-values (lane (block b, i) holds X[row i][k0 + b]) and is good for 16 MFMAs; the B operand of the MFMA with abid = q is one weight per lane,
// W[column of the lane][k0 + q], so a wave computes 4 rows x 64 columns per instruction at the same MAC rate as 16x16x4 and reads its activations 16 x less often than
// the B-variants above. Same PF copy: lane l = 16 q4 + j takes the slots (g', j) of panel t = 4 cg + q4. Eight waves = 4 column groups x 2 k-halves; a k-half is two
// quarters summed one after the other, the halves meet through the output slab (each wave finalises half of the row groups). The summation order of an output does
// not depend on the number of rows per tile, so tiles of 8, 16 and 32 rows are bit-identical row by row.
template <int CB, int AB>
__device__ __forceinline__ f32x4 probe_mfma4bc(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CB, AB, 0); }
template <int RG, int Q>
__device__ __forceinline__ void bc_block(const float (&a)[RG], const f32x4 (&b)[4], f32x4 (&acc)[RG]) {   // 16 k-values: abid = 4 g' + r
#define IL_BC(GP, R_) _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) acc[rg] = probe_mfma4bc<4, 4 * GP + R_>(a[rg], b[GP][R_], acc[rg]);
  IL_BC(0, 0) IL_BC(0, 1) IL_BC(0, 2) IL_BC(0, 3) IL_BC(1, 0) IL_BC(1, 1) IL_BC(1, 2) IL_BC(1, 3)
  IL_BC(2, 0) IL_BC(2, 1) IL_BC(2, 2) IL_BC(2, 3) IL_BC(3, 0) IL_BC(3, 1) IL_BC(3, 2) IL_BC(3, 3)
#undef IL_BC
}
// UB = k-blocks (of 16) whose weight lanes are requested together (4 x 16 B per lane and block)
template <int RG, int UB, int PAIRS, class Epi>
__device__ __forceinline__ void probe_tile_packed_bc(const float* As, int lda, const float* __restrict__ P, float* Ds, int ldd, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = PAIRS ? wave >> 1 : wave & 3, kh = PAIRS ? wave & 1 : wave >> 2, q4 = lane >> 4, j = lane & 15, nb = H >> 4;
  const float* pp = P + (size_t)(4 * cg + q4) * nb * 256 + j * 4;
  const float* ar = As + (lane & 3) * lda + (lane >> 2);
  f32x4 part[RG];
#pragma unroll
  for (int quarter = 0; quarter < 2; ++quarter) {
    f32x4 acc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[rg] = zero4();
#pragma unroll
    for (int u0 = 0; u0 < 4; u0 += UB) {
      f32x4 b[UB][4];
#pragma unroll
      for (int uu = 0; uu < UB; ++uu)
#pragma unroll
        for (int gp = 0; gp < 4; ++gp) b[uu][gp] = gload4(pp + (size_t)(8 * kh + 4 * quarter + u0 + uu) * 256 + gp * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int uu = 0; uu < UB; ++uu) {
        const int u = 8 * kh + 4 * quarter + u0 + uu;
        float a[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) a[rg] = ar[4 * rg * lda + 16 * u];
        bc_block<RG, 0>(a, b[uu], acc);
      }
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) part[rg] = quarter == 0 ? acc[rg] : part[rg] + acc[rg];
  }
  // the halves meet in the output slab: wave kh keeps the row groups [kh RG/2, (kh + 1) RG/2) and leaves the others' partial sums where the owner will store its result
  const int col = 64 * cg + lane;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
    if ((rg >= RG / 2) != (kh == 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) Ds[(4 * rg + i) * ldd + col] = part[rg][i];
    }
  __syncthreads();
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
    if ((rg >= RG / 2) == (kh == 1)) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = Ds[(4 * rg + i) * ldd + col];
      epi(col, 4 * rg, kh == 0 ? part[rg] + o : o + part[rg]);   // (first half) + (second half) on both sides
    }
}
template <int RG, int UB, int NETS_, int PAIRS = 0, int OWN = 0, int TOUCH = 0>
__global__ __launch_bounds__(512) void k_probe_d(const float* __restrict__ PF, const float* __restrict__ X, float* __restrict__ Y, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 4 * RG;
  float* S0 = smem; float* S1 = smem + R * LDH;
  const int wg = blockIdx.x;
  const int net = OWN ? (wg % NETS) % 2 : (NETS_ == NETS ? wg % NETS : ((wg * R / 256) % NETS_) % NETS), tile = NETS_ == NETS ? wg / NETS : wg;   // OWN: two networks x LAYERS copies = 2 MB, inside one XCD's 4 MB L2
  for (int i = threadIdx.x; i < R * H; i += blockDim.x) S0[(i / H) * LDH + i % H] = X[(size_t)((tile * R + i / H) % 256) * H + i % H];
  if (TOUCH == 1) {   // every workgroup requests one dword of every line of every panel it is going to stream, up front (TOUCH == 2: and waits ~3 us before the layers)
    for (int l = 0; l < LAYERS; ++l)
      for (int i = threadIdx.x; i < H * H / 32; i += blockDim.x) { const float v = gload(PF + (size_t)(OWN ? l * NETS + net : net) * H * H + (size_t)i * 32); asm volatile("" ::"v"(v)); }
  }
  __syncthreads();
  float* src = S0; float* dst = S1;
  for (int l = 0; l < LAYERS; ++l) {
    if (threadIdx.x == 0 && wg < 4096) stamps[wg * (LAYERS + 1) + l] = __builtin_amdgcn_s_memrealtime();
    probe_tile_packed_bc<RG, UB, PAIRS>(src, LDH, PF + (size_t)(OWN ? l * NETS + net : net) * H * H, dst, LDH, [&](int col, int rb, f32x4 s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(rb + i) * LDH + col] = fmaxf(s[i] * 0.05f, -1.f);
    });
    __syncthreads();
    float* t = src; src = dst; dst = t;
  }
  if (threadIdx.x == 0 && wg < 4096) stamps[wg * (LAYERS + 1) + LAYERS] = __builtin_amdgcn_s_memrealtime();
  if (wg < NETS * 256 / R) for (int i = threadIdx.x; i < R * H; i += blockDim.x) Y[(size_t)(wg * R + i / H) * H + i % H] = src[(i / H) * LDH + i % H];
}

// shader clock during a burst of short kernels: s_memtime (shader-clock counter) against s_memrealtime (100 MHz) across a busy loop
__global__ void k_clock(unsigned long long* out) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc = zero4();
  for (int i = 0; i < 4000; ++i) acc = mfma16(1.0f, 1.0f, acc);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)acc[0]; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// COLD=1 in the environment: between two launches a kernel rewrites every PF copy with write-through stores from all XCDs (what the AdamW epilogue of the previous
// launch does in the product), so that a launch's FIRST layer streams its panels through the fabric instead of out of the XCD's L2
__global__ void k_rewrite(float* __restrict__ PF, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(PF + 4 * i);
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(PF + 4 * i));
  }
}
static bool g_cold = false;
static size_t g_min_lds = 0;
template <class K>
static int run(const char* name, K kern, int grid, int rows, int threads, const float* PF, const float* X, float* Y, unsigned long long* stamps, std::vector<float>& out) {
  const size_t lds = std::max(sizeof(float) * 2 * rows * LDH, g_min_lds);
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int it = 0; it < 5; ++it) kern<<<grid, threads, lds>>>(PF, X, Y, stamps);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int it = 0; it < 50; ++it) { if (g_cold) k_rewrite<<<160, 256>>>(const_cast<float*>(PF), (size_t)LAYERS * NETS * H * H / 4); kern<<<grid, threads, lds>>>(PF, X, Y, stamps); }
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st((size_t)grid * (LAYERS + 1));
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> first, later;
  for (int w = 0; w < grid; ++w) {
    first.push_back((st[w * (LAYERS + 1) + 1] - st[w * (LAYERS + 1)]) / 100.0);
    for (int l = 1; l < LAYERS; ++l) later.push_back((st[w * (LAYERS + 1) + l + 1] - st[w * (LAYERS + 1) + l]) / 100.0);
  }
  std::sort(first.begin(), first.end()); std::sort(later.begin(), later.end());
  printf("%-28s grid %3d: kernel %.2f us | layer (first touch of the panel) median %.2f us | later layers median %.2f us, p90 %.2f us\n", name, grid, ms / 50 * 1e3,
         first[first.size() / 2], later[later.size() / 2], later[later.size() * 9 / 10]);
  out.resize((size_t)std::min(grid, NETS * 256 / rows) * rows * H);
  CK(hipMemcpy(out.data(), Y, out.size() * 4, hipMemcpyDeviceToHost));
  return 0;
}

int main(int argc, char** argv) {
  g_cold = getenv("COLD") != nullptr;
  if (g_cold) printf("COLD: the PF copies are rewritten between launches; read the 'first touch' column\n");
  const int B = 256;   // batch rows: 16 tiles of 16 rows
  std::vector<float> hW((size_t)NETS * H * H), hPF(hW.size() * LAYERS), hX((size_t)B * H);
  srand(1);
  for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
  for (auto& v : hX) v = rand() / (float)RAND_MAX - 0.5f;
  for (int net = 0; net < NETS; ++net)
    for (int n = 0; n < H; ++n)
      for (int k = 0; k < H; ++k) for (int l = 0; l < LAYERS; ++l) hPF[(size_t)(l * NETS + net) * H * H + packed_fwd_index(n, k, H)] = hW[(size_t)net * H * H + (size_t)n * H + k];   // (copies 1 .. LAYERS - 1: the OWN variants)
  float *PF, *X, *Y; unsigned long long* stamps;
  CK(hipMalloc(&PF, hPF.size() * 4)); CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&Y, (size_t)NETS * B * H * 4)); CK(hipMalloc(&stamps, 8 * 4096 * (LAYERS + 1)));
  CK(hipMemcpy(PF, hPF.data(), hPF.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; ++rep) {
    for (int it = 0; it < (rep == 2 ? 2000 : 1); ++it) k_clock<<<rep == 1 ? 1024 : 96, 256>>>(stamps);
    CK(hipDeviceSynchronize());
    unsigned long long h[3]; CK(hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost));
    printf("clock probe (%s): %llu s_memtime ticks in %.2f us -> %.0f MHz if s_memtime counts shader clocks; 4000 dependent 16x16x4 MFMAs = %.1f ticks each\n",
           rep == 0 ? "96 workgroups, cold" : rep == 1 ? "1024 workgroups" : "96 workgroups after 2000 launches", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), h[0] / 4000.0);
  }
  std::vector<float> ya, yb4, yb2, yb8, yc;
  if (run("A  16 rows, 16x16x4", k_probe_a, NETS * B / 16, 16, 1024, PF, X, Y, stamps, ya)) return 1;
  if (argc > 1 && argv[1][0] == 'd') {   // round 3: only the broadcast forms against A
    auto at = [&](const std::vector<float>& y, int R, int net, int row, int col) { const int tile = row / R, wg = tile * NETS + net; return y[((size_t)wg * R + row % R) * H + col]; };
    std::vector<float> yd2, yd4, yd8, yq;
    if (run("D2  8 rows, 4x4x1 broadcast, panel 8", (k_probe_d<2, 4, NETS>), NETS * B / 8, 8, 512, PF, X, Y, stamps, yd2)) return 1;
    if (run("D2' 8 rows, 4x4x1 broadcast, panel 4", (k_probe_d<2, 2, NETS>), NETS * B / 8, 8, 512, PF, X, Y, stamps, yq)) return 1;
    { std::vector<float> y0;
      if (run("E2  D2, every layer its own panel copy", (k_probe_d<2, 4, NETS, 0, 1>), NETS * B / 8, 8, 512, PF, X, Y, stamps, y0)) return 1;
      if (run("F2  E2 + the product's wave -> item map", (k_probe_d<2, 4, NETS, 1, 1>), NETS * B / 8, 8, 512, PF, X, Y, stamps, y0)) return 1;
      if (run("T2  F2 + every line of every panel touched up front", (k_probe_d<2, 4, NETS, 1, 1, 1>), NETS * B / 8, 8, 512, PF, X, Y, stamps, y0)) return 1;
      g_min_lds = 81 * 1024;
      if (run("G2  F2 + 81 KB of LDS per workgroup", (k_probe_d<2, 4, NETS, 1, 1>), NETS * B / 8, 8, 512, PF, X, Y, stamps, y0)) return 1;
      g_min_lds = 0; }
    if (run("D4 16 rows, 4x4x1 broadcast", (k_probe_d<4, 4, NETS>), NETS * B / 16, 16, 512, PF, X, Y, stamps, yd4)) return 1;
    if (run("D8 32 rows, 4x4x1 broadcast", (k_probe_d<8, 2, NETS>), NETS * B / 32, 32, 512, PF, X, Y, stamps, yd8)) return 1;
    double d2 = 0, sc = 0; size_t dif24 = 0, dif28 = 0, dif2q = 0;
    for (int net = 0; net < NETS; ++net) for (int row = 0; row < B; ++row) for (int col = 0; col < H; ++col) {
      const double a = at(ya, 16, net, row, col);
      sc = std::max(sc, std::fabs(a)); d2 = std::max(d2, std::fabs(a - at(yd2, 8, net, row, col)));
      dif24 += at(yd2, 8, net, row, col) != at(yd4, 16, net, row, col); dif28 += at(yd2, 8, net, row, col) != at(yd8, 32, net, row, col); dif2q += at(yd2, 8, net, row, col) != at(yq, 8, net, row, col);
    }
    printf("max |D2 - A| after %d layers %.3g (scale %.3g); elements that differ between D2 and D4: %zu, D2 and D8: %zu, D2 and D2': %zu (must be 0)\n", LAYERS, d2, sc, dif24, dif28, dif2q);
    { std::vector<float> y0;
      if (run("P1 population shape, 16 rows, 16x16x4", k_probe_p<1>, 2048, 16, 512, PF, X, Y, stamps, y0)) return 1;
      if (run("Q4 population shape, 16 rows, broadcast, panel 2", (k_probe_d<4, 2, NETS_POP>), 2048, 16, 512, PF, X, Y, stamps, y0)) return 1;
      if (run("Q4 population shape, 16 rows, broadcast, panel 4", (k_probe_d<4, 4, NETS_POP>), 2048, 16, 512, PF, X, Y, stamps, y0)) return 1;
      if (run("Q8 population shape, 32 rows, broadcast, panel 2", (k_probe_d<8, 2, NETS_POP>), 1024, 32, 512, PF, X, Y, stamps, y0)) return 1;
      if (run("Q8 population shape, 32 rows, broadcast, panel 4", (k_probe_d<8, 4, NETS_POP>), 1024, 32, 512, PF, X, Y, stamps, y0)) return 1; }
    return 0;
  }
  { std::vector<float> y0, y1;
    if (run("A0 panel resident in VGPRs", k_probe_a01<0>, NETS * B / 16, 16, 1024, PF, X, Y, stamps, y0)) return 1;
    if (run("A1 panel loads, 1/4 of MFMAs", k_probe_a01<1>, NETS * B / 16, 16, 1024, PF, X, Y, stamps, y1)) return 1;
    if (run("A0 on 16 workgroups only", k_probe_a01<0>, 16, 16, 1024, PF, X, Y, stamps, y0)) return 1;
    if (run("A  on 16 workgroups only", k_probe_a, 16, 16, 1024, PF, X, Y, stamps, y0)) return 1; }
  { std::vector<float> y0;
    if (run("P1 population shape, 16 rows", k_probe_p<1>, 2048, 16, 512, PF, X, Y, stamps, y0)) return 1;
    if (run("P2 population shape, 32 rows", k_probe_p<2>, 1024, 32, 512, PF, X, Y, stamps, y0)) return 1; }
  if (run("C  16 rows, 8 waves, prefetch", k_probe_c, NETS * B / 16, 16, 512, PF, X, Y, stamps, yc)) return 1;
  { size_t bad = 0; for (size_t i = 0; i < ya.size(); ++i) bad += ya[i] != yc[i]; printf("C vs A: %zu of %zu elements differ (must be 0: same MFMA order)\n", bad, ya.size()); }
  if (run("B4 16 rows, 4x4x1 (RG = 4)", k_probe_b<4>, NETS * B / 16, 16, 1024, PF, X, Y, stamps, yb4)) return 1;
  if (run("B2  8 rows, 4x4x1 (RG = 2)", k_probe_b<2>, NETS * B / 8, 8, 1024, PF, X, Y, stamps, yb2)) return 1;
  if (run("B8 32 rows, 4x4x1 (RG = 8)", k_probe_b<8>, NETS * B / 32, 32, 1024, PF, X, Y, stamps, yb8)) return 1;
  // the output of workgroup (net, tile) holds rows tile * R .. of network `net`: compare by (net, row, col)
  auto at = [&](const std::vector<float>& y, int R, int net, int row, int col) { const int tile = row / R, wg = tile * NETS + net; return y[((size_t)wg * R + row % R) * H + col]; };
  double d4 = 0, d2 = 0, d8 = 0, sc = 0;
  for (int net = 0; net < NETS; ++net) for (int row = 0; row < B; ++row) for (int col = 0; col < H; ++col) {
    const double a = at(ya, 16, net, row, col);
    sc = std::max(sc, std::fabs(a));
    d4 = std::max(d4, std::fabs(a - at(yb4, 16, net, row, col))); d2 = std::max(d2, std::fabs(a - at(yb2, 8, net, row, col))); d8 = std::max(d8, std::fabs(a - at(yb8, 32, net, row, col)));
  }
  printf("max |B - A| after %d layers: RG=4 %.3g, RG=2 %.3g, RG=8 %.3g (scale %.3g); RG variants among themselves bit-identical: %s\n", LAYERS, d4, d2, d8, sc,
         (d4 == d2 && d2 == d8) ? "same max" : "differ");
  return 0;
}
