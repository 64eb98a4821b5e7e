// LDS-staged MLP tile primitives for one workgroup = one tile of IL_TILE_R (16) batch rows.
//
// Design (MI355X, wave64, fp32 parity => exact-fp32 MFMA 16x16x4):
//  * activations of the tile live in LDS ([16][ld] fp32, ld = K + 4 keeps 16-B alignment and spreads rows over banks);
//  * weights are streamed straight from L2 into MFMA B operands (each weight element is used exactly once per
//    workgroup, so staging it through LDS would only add traffic); rows of a torch [N][K] weight are read as
//    16-B lanes (64 contiguous bytes per 4-lane k-group);
//  * each wave owns 16 output columns; a workgroup is H/16 waves (16 waves = 4 per SIMD at H = 256) so L2 latency of the
//    weight lanes is hidden by the other waves on the SIMD;
//  * the k-index of an MFMA step is permuted (lane group g covers k0+4g..k0+4g+3 over four steps) so that both
//    operands are single 16-byte loads. Summation order within a dot product changes, results stay exact-fp32 FMAs.
#pragma once
#include "il_common.hpp"

#ifndef IL_L1_PAIR
#define IL_L1_PAIR 1   // tile_fwd: k-blocks of a narrow first layer fetched pairwise (0 = the round-2 one-at-a-time loop, for A/B builds); same MFMA order, same bits
#endif
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// Operand loads are never wrapped in a branch or a select (hipcc turns `cond ? load : 0` into an exec-masked branch that
// serialises the loop): out-of-range columns/rows CLAMP their address instead. That is exact because the other MFMA operand
// is zero there (LDS tiles are zero-padded) or the corresponding outputs are discarded by the epilogue.
//   MODE 0: row + k is 16-B aligned and k + 3 < kvalid always (hidden layers)
//   MODE 1: 16-B aligned rows, kvalid % 4 == 0
//   MODE 2: arbitrary row stride / kvalid (first layer with odd input widths): four clamped dword loads
template <int MODE>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, int k, int kvalid) {
  if (MODE == 0) return gload4(p + k);
  if (MODE == 1) return gload4(p + min(k, kvalid - 4));
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = gload(p + min(k + i, kvalid - 1));
  return r;
}

// Latency hiding is done with THREAD-level parallelism, not software pipelining (hipcc re-rolls a hand-pipelined loop into
// load -> s_waitcnt -> MFMA): a workgroup runs one wave per 16 output columns (16 waves = 4 per SIMD at H = 256), so while
// one wave waits for its weight lanes from L2 the other three keep the SIMD's matrix pipe busy.

// ---------------------------------------------------------------------------------------------
// Y[16 x N] = Xs[16 x Kpad] . W^T      W: global row-major [N][ldw], columns >= Kw read as zero (Xs is zero-padded too)
// N % 16 == 0.  epi(c0, acc): acc[reg] = Y[row 4g+reg][col c0 + j],  j = lane&15, g = lane>>4.
// ---------------------------------------------------------------------------------------------
template <int MODE, int PANEL, class Epi>
__device__ __forceinline__ void tile_fwd_impl(const float* Xs, int ldx, int Kpad, const float* __restrict__ W, int ldw, int Kw, int N, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
#if IL_L1_PAIR
  if (Kpad == 32 && N == 32 * nw) {   // (round 4) a narrow first layer with TWO column tiles per wave (the 8-wave population kernels): the weight lanes of both tiles are
    f32x4 b[2][2];                    // requested before the first MFMA - the tile loop below made the second tile's round trip wait for the first tile's epilogue
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) b[t][u] = load4<MODE>(W + (size_t)((wave + t * nw) * 16 + j) * ldw, 16 * u + 4 * g, Kw);
    __builtin_amdgcn_sched_barrier(0);
    const float* xr = Xs + j * ldx + 4 * g;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * u);
        acc0 = mfma16(a[0], b[t][u][0], acc0);
        acc1 = mfma16(a[1], b[t][u][1], acc1);
        acc0 = mfma16(a[2], b[t][u][2], acc0);
        acc1 = mfma16(a[3], b[t][u][3], acc1);
      }
      f32x4 acc = acc0 + acc1;
      epi((wave + t * nw) * 16, acc);
    }
    return;
  }
#endif
  for (int c0 = wave * 16; c0 < N; c0 += nw * 16) {
    f32x4 acc0 = zero4(), acc1 = zero4();  // two accumulators: the 16x16x4 f32 MFMA has a 40-cycle dependent latency vs 32-cycle issue
    const float* wr = W + (size_t)(c0 + j) * ldw + 4 * g;
    const float* xr = Xs + j * ldx + 4 * g;
    int k0 = 0;
    if (MODE == 0 && Kpad == 256 && PANEL >= 16) {  // the H = 256 hidden layer: the wave's whole weight panel (16 lanes of 16 B) is requested up front, MFMAs
      f32x4 b[16];                   // start as soon as the first lane lands and the rest stream in underneath them
#pragma unroll
      for (int u = 0; u < 16; ++u) b[u] = load4<MODE>(wr - 4 * g, 16 * u + 4 * g, Kw);
      __builtin_amdgcn_sched_barrier(0);  // keep all 16 requests ahead of the MFMAs (the scheduler otherwise sinks them to 2 in flight)
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * u);
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
      k0 = 256;
    }
    for (; k0 + 128 <= Kpad; k0 += 128) {  // 8 k-blocks per trip: eight 16-B weight lanes in flight before the first MFMA needs one
      f32x4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = load4<MODE>(wr - 4 * g, k0 + 16 * u + 4 * g, Kw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0 + 16 * u);
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
    for (; k0 + 64 <= Kpad; k0 += 64) {
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = load4<MODE>(wr - 4 * g, k0 + 16 * u + 4 * g, Kw);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0 + 16 * u);
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
#if IL_L1_PAIR
    for (; k0 + 32 <= Kpad; k0 += 32) {   // the first layers (K = 17 .. 32 -> Kpad = 32): both k-blocks' weight lanes requested before the first MFMA instead of two dependent L2 round trips
      f32x4 b[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) b[u] = load4<MODE>(wr - 4 * g, k0 + 16 * u + 4 * g, Kw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0 + 16 * u);
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
#endif
    for (; k0 < Kpad; k0 += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0);
      const f32x4 b = load4<MODE>(wr - 4 * g, k0 + 4 * g, Kw);
      acc0 = mfma16(a[0], b[0], acc0);
      acc1 = mfma16(a[1], b[1], acc1);
      acc0 = mfma16(a[2], b[2], acc0);
      acc1 = mfma16(a[3], b[3], acc1);
    }
    f32x4 acc = acc0 + acc1;
    epi(c0, acc);
  }
}
template <int PANEL = 16, class Epi>
__device__ __forceinline__ void tile_fwd(const float* Xs, int ldx, int Kpad, const float* __restrict__ W, int ldw, int Kw, int N, Epi epi) {
  const bool aligned = ((ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  if (aligned && Kw == Kpad) tile_fwd_impl<0, PANEL>(Xs, ldx, Kpad, W, ldw, Kw, N, epi);
  else if (aligned && (Kw & 3) == 0 && Kw >= 4) tile_fwd_impl<1, PANEL>(Xs, ldx, Kpad, W, ldw, Kw, N, epi);
  else tile_fwd_impl<2, PANEL>(Xs, ldx, Kpad, W, ldw, Kw, N, epi);
}

// ---------------------------------------------------------------------------------------------
// dX[16 x K] = dYs[16 x Npad] . W      W: global row-major [Nvalid][ldw], rows >= Nvalid read as zero; columns kb + j >= K clamp
// their address (the epilogue must ignore them when K % 16 != 0).  epi(kb, acc): acc[reg] = dX[row 4g+reg][col kb + j]
// ---------------------------------------------------------------------------------------------
template <bool FULL, class Epi>
__device__ __forceinline__ void tile_bwd_dx_impl(const float* dYs, int ldy, int Npad, int Nvalid, const float* __restrict__ W, int ldw, int K, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  for (int kb = wave * 16; kb < K; kb += nw * 16) {
    f32x4 acc0 = zero4(), acc1 = zero4();
    const float* wp = W + min(kb + j, K - 1);
    const float* yr = dYs + j * ldy + 4 * g;
    int n0 = 0;
    // 16 weight dwords in flight per lane (32 would need 64 address VGPRs: with 1024-thread workgroups the 128-VGPR budget spills)
    for (; n0 + 64 <= Npad; n0 += 64) {
      float b[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) { const int n = n0 + 16 * u + 4 * g + s; b[u][s] = gload(wp + (size_t)(FULL ? n : min(n, Nvalid - 1)) * ldw); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(yr + n0 + 16 * u);
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
    for (; n0 < Npad; n0 += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(yr + n0);
      float b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { const int n = n0 + 4 * g + s; b[s] = gload(wp + (size_t)(FULL ? n : min(n, Nvalid - 1)) * ldw); }  // rows >= Nvalid: dYs columns are zero there
      acc0 = mfma16(a[0], b[0], acc0);
      acc1 = mfma16(a[1], b[1], acc1);
      acc0 = mfma16(a[2], b[2], acc0);
      acc1 = mfma16(a[3], b[3], acc1);
    }
    f32x4 acc = acc0 + acc1;
    epi(kb, acc);
  }
}
template <class Epi>
__device__ __forceinline__ void tile_bwd_dx(const float* dYs, int ldy, int Npad, int Nvalid, const float* __restrict__ W, int ldw, int K, Epi epi) {
  if (Nvalid == Npad) tile_bwd_dx_impl<true>(dYs, ldy, Npad, Nvalid, W, ldw, K, epi);
  else tile_bwd_dx_impl<false>(dYs, ldy, Npad, Nvalid, W, ldw, K, epi);
}

// ---------------------------------------------------------------------------------------------
// Columns [c_lo, c_hi) of dX[16 x K] = dYs[16 x N] . W with the N-REDUCTION split over the waves (wave w owns n in [16w, 16w + 16), ...): for a
// narrow dX (dQ/da: A <= 8 columns of a K = S + A wide product) tile_bwd_dx keeps one or two waves busy with N/4 dependent MFMAs each; here
// every wave issues 4 and the partial tiles meet in LDS (`part` >= (N/16) * 256 floats), summed in n-block order (deterministic).
// Contains __syncthreads(); every thread of the block must call. epi(col, row, value) once per element of the 16-column tiles that overlap the range.
// ---------------------------------------------------------------------------------------------
// The weight operands of tile_bwd_dx_cols / tile_fwd_small depend on nothing the kernel computes: `*_prefetch` requests this wave's lanes (first column tile; up to two
// n- / k-blocks per wave, i.e. every block of a 16- or 8-wave workgroup at H = 256) at the top of the kernel, so that the small GEMM at the END of a dependent chain of
// layers starts from registers instead of from an L2 / fabric round trip behind a barrier. Same values, same MFMA order: same bits.
#ifndef IL_SMALL_PREFETCH
#define IL_SMALL_PREFETCH 1
#endif
struct ColsPre { float b[2][4]; };
__device__ __forceinline__ ColsPre tile_bwd_dx_cols_prefetch(const float* __restrict__ W, int ldw, int K, int N, int c_lo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const float* wp = W + min(((c_lo >> 4) << 4) + j, K - 1);
  ColsPre p;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n0 = min((wave + q * nw) * 16, N - 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) p.b[q][s] = gload(wp + (size_t)(n0 + 4 * g + s) * ldw);
  }
  return p;
}
template <class Epi>
__device__ __forceinline__ void tile_bwd_dx_cols(const float* dYs, int ldy, int N, const float* __restrict__ W, int ldw, int K, int c_lo, int c_hi, float* part, Epi epi,
                                                 const ColsPre* pre = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  // One partial tile per 16-wide n-block (N / 16 <= 16 of them), summed in n-block order: the result does not depend on how many waves the workgroup
  // has (a 512-thread launch of the population path gives bit-identical values to the 1024-thread launch of a single learner).
  for (int kb = (c_lo >> 4) << 4; kb < c_hi; kb += 16) {
    const float* wp = W + min(kb + j, K - 1);
    for (int n0 = wave * 16; n0 < N; n0 += nw * 16) {
      f32x4 acc0 = zero4(), acc1 = zero4();
      const f32x4 a = *reinterpret_cast<const f32x4*>(dYs + j * ldy + n0 + 4 * g);
      float b[4];
      const int q = (n0 >> 4) - wave;   // 0, nw, 2 nw, ...
      if (pre && kb == ((c_lo >> 4) << 4) && q <= nw) {
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = q == 0 ? pre->b[0][s] : pre->b[1][s];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = gload(wp + (size_t)(n0 + 4 * g + s) * ldw);
      }
      acc0 = mfma16(a[0], b[0], acc0);
      acc1 = mfma16(a[1], b[1], acc1);
      acc0 = mfma16(a[2], b[2], acc0);
      acc1 = mfma16(a[3], b[3], acc1);
      const f32x4 acc = acc0 + acc1;
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(n0 >> 4) * 256 + (4 * g + r) * 16 + j] = acc[r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
      float s = 0.f;
      for (int w = 0; w < (N >> 4); ++w) s += part[w * 256 + i];
      epi(kb + (i & 15), i >> 4, s);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Packed hidden-layer weights. Reading a torch [N][K] matrix as MFMA operands touches 16 different rows per wave-load (64 useful
// bytes of 16 separate 128-B lines): measured 14 B/clk/CU, and the MFMAs starve behind it (19.8k cycles per 16x256x256 layer vs
// 10.7k without the loads). The same bytes as contiguous 1 KiB wave-loads run at 12.3k. So the H x H layers are read from two
// lane-ordered copies kept in the workspace (written by k_repack at the start of an update and by the Adam epilogue afterwards):
//   PF (forward,  B[k][n] = W[n][k]):  PF[((n/16 * H/16 + k/16) * 64 + ((k%16)/4)*16 + n%16) * 4 + k%4]
//   PB (backward, B[n][k] = W[n][k]):  PB[((k/16 * H/16 + n/16) * 64 + ((n%16)/4)*16 + k%16) * 4 + n%4]
// i.e. wave-load (tile, block) = 64 lanes x 16 B contiguous, and a wave's whole panel (H/16 blocks) is one contiguous H*16*4-byte run.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t packed_fwd_index(int n, int k, int H) { return ((size_t)((n >> 4) * (H >> 4) + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3); }
__host__ __device__ inline size_t packed_bwd_index(int n, int k, int H) { return ((size_t)((k >> 4) * (H >> 4) + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (k & 15)) * 4 + (n & 3); }

// shared body: acc += A(LDS rows, K = H) . panel, panel = P + tile * (H/16) * 256 floats
// PANEL = weight blocks (16-byte lanes per thread) requested ahead of a tile's MFMAs: 16 = the whole H = 256 panel (64 VGPRs; the single learner's 16-wave workgroups, one per
// CU), 8 = half of it per round (the population launches: ~80 VGPRs let THREE 8-wave workgroups share a CU instead of two). Same MFMA order either way: same bits.
template <int PANEL = 16, class Epi>
__device__ __forceinline__ void tile_packed(const float* As, int lda, int H, const float* __restrict__ P, Epi epi, int t0 = 0, int t1 = -1) {   // output tiles [t0, t1): default all
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4, nb = H >> 4;
  if (t1 < 0) t1 = nb;
  for (int t = t0 + wave; t < t1; t += nw) {
    f32x4 acc0 = zero4(), acc1 = zero4();
    const float* pp = P + (size_t)t * nb * 256 + lane * 4;
    const float* ar = As + j * lda + 4 * g;
    int kb = 0;
    for (; kb + PANEL <= nb; kb += PANEL) {
      f32x4 b[PANEL];
#pragma unroll
      for (int u = 0; u < PANEL; ++u) b[u] = gload4(pp + (size_t)(kb + u) * 256);
      __builtin_amdgcn_sched_barrier(0);  // all 16 KiB of the panel (PANEL = 16) requested before the first MFMA
#pragma unroll
      for (int u = 0; u < PANEL; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * (kb + u));
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
    for (; kb + 4 <= nb; kb += 4) {
      f32x4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = gload4(pp + (size_t)(kb + u) * 256);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * (kb + u));
        acc0 = mfma16(a[0], b[u][0], acc0);
        acc1 = mfma16(a[1], b[u][1], acc1);
        acc0 = mfma16(a[2], b[u][2], acc0);
        acc1 = mfma16(a[3], b[u][3], acc1);
      }
    }
    f32x4 acc = acc0 + acc1;
    epi(t * 16, acc);
  }
}
// (round 6, the general-shape tile engine) Two output tiles per wave (an 8-wave workgroup at H = 256, 256 VGPRs per wave): BOTH 16 KB panels are requested before the
// first MFMA - the whole layer's 256 KB is in flight at once - instead of tile_packed's load, 64 MFMAs, load, 64 MFMAs (measured: 6.1 - 6.9 us per 16 x 256 x 256 layer
// against 3.4 us of MFMA issue). Same MFMA order per tile: same bits. Falls back to tile_packed for any other shape.
template <class Epi>
__device__ __forceinline__ void tile_packed_x2(const float* As, int lda, int H, const float* __restrict__ P, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4, nb = H >> 4;
  if (nb != 16 || nw != 8) { tile_packed<16>(As, lda, H, P, epi); return; }
  const float* p0 = P + (size_t)wave * nb * 256 + lane * 4;
  const float* p1 = P + (size_t)(wave + 8) * nb * 256 + lane * 4;
  const float* ar = As + j * lda + 4 * g;
  f32x4 b0[16], b1[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) b0[u] = gload4(p0 + (size_t)u * 256);
#pragma unroll
  for (int u = 0; u < 16; ++u) b1[u] = gload4(p1 + (size_t)u * 256);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * u);
      acc0 = mfma16(a[0], b0[u][0], acc0); acc1 = mfma16(a[1], b0[u][1], acc1); acc0 = mfma16(a[2], b0[u][2], acc0); acc1 = mfma16(a[3], b0[u][3], acc1);
    }
    epi(wave * 16, acc0 + acc1);
  }
  {
    f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * u);
      acc0 = mfma16(a[0], b1[u][0], acc0); acc1 = mfma16(a[1], b1[u][1], acc1); acc0 = mfma16(a[2], b1[u][2], acc0); acc1 = mfma16(a[3], b1[u][3], acc1);
    }
    epi((wave + 8) * 16, acc0 + acc1);
  }
}
// (Measured and dropped in round 2: an early "touch" of a workgroup's packed panels - one dword per 128-byte line requested at the top of the kernel so that the
// lines rewritten by the previous Adam kernel are already in this XCD's L2 when the layer starts. Same-box A/B 14.24k -> 13.39k updates/s: loads return in order,
// so the rows and first-layer operands issued behind the cold touches wait for them; profiles/r02_update_timeline.md.)
// Y[16 x H] = Xs[16 x H] . W^T with W given as its PF copy;  epi(c0, acc): acc[reg] = Y[row 4g+reg][col c0 + j]
template <int PANEL = 16, class Epi>
__device__ __forceinline__ void tile_fwd_packed(const float* Xs, int ldx, int H, const float* __restrict__ PF, Epi epi) { tile_packed<PANEL>(Xs, ldx, H, PF, epi); }
// dX[16 x H] = dYs[16 x H] . W with W given as its PB copy;  epi(kb, acc): acc[reg] = dX[row 4g+reg][col kb + j]
template <int PANEL = 16, class Epi>
__device__ __forceinline__ void tile_bwd_packed(const float* dYs, int ldy, int H, const float* __restrict__ PB, Epi epi, int t0 = 0, int t1 = -1) { tile_packed<PANEL>(dYs, ldy, H, PB, epi, t0, t1); }

// ---------------------------------------------------------------------------------------------
// Pair mode (round 4): a 16-row tile of a 16 x 256 x 256 layer is 8,192 MFMA clocks on ONE CU however many CUs idle, and its 256 KB weight panel reaches that CU at
// ~75 GB/s - both 3.4 us. A PAIR of 8-wave workgroups (512 threads, <= 256 VGPRs per wave) splits the layer by OUTPUT COLUMNS: wave w of half p owns output tile 8 p + w,
// i.e. the same 64 MFMAs in the same k order on the same operands as wave 8 p + w of the 16-wave workgroup - every element keeps its summation order, the results are
// bit-identical - while each CU issues half the MFMAs and pulls half the panel. What the two halves need of each other (the next layer's K dimension) crosses through an
// 8 KB slab in the workspace: write-through (sc0 sc1) 16-byte stores, a drained flag, sc0 sc1 loads on the consumer - no L2 write-back, no L1 invalidate (MI355X guide,
// "Valid forms": `sc1` payload -> vmcnt(0) -> flag; `sc1` loads replace the acquire when the producer stored `sc1`). The narrow first layer (K <= 64) is computed by both
// halves in full - cheaper than a hop. With 256 VGPRs a wave parks its whole 16 KB panel of the NEXT big layer in registers while the current phase runs.
// ---------------------------------------------------------------------------------------------
struct Panel16 { f32x4 b[16]; };
// Between the small requests of a tile's prologue (row indices, W1, biases, rows) and its panel requests: a bare s_barrier (no fence, no wait). A CU returns its loads
// in request order ACROSS its waves, so a wave that runs a little behind the others (the one that announced, the head threads) would otherwise find its few small loads
// queued behind seven other waves' 16 KB panels (224 KB at ~75 GB/s = 3 us: measured as a 4.3 us prologue of k_policy_critic_pair).
__device__ __forceinline__ void issue_fence() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
// the 16 lane-ordered weight blocks of output tile t (H = 256: nb = 16), requested at once
__device__ __forceinline__ void panel_prefetch(Panel16& p, const float* __restrict__ P, int t) {
  const float* pp = P + (size_t)t * 16 * 256 + (threadIdx.x & 63) * 4;
#pragma unroll
  for (int u = 0; u < 16; ++u) p.b[u] = gload4(pp + (size_t)u * 256);
  __builtin_amdgcn_sched_barrier(0);
}
// The same request in two halves (blocks [0, 8) and [8, 16) - the order the MFMAs consume them in): the first half goes out EARLY, behind a prologue's small loads - 8 KB per
// wave holds a wave at the issue stage for ~0.4 us, inside the latency of those small loads - and streams in under the first layer; the second half is requested right
// before the MFMAs and streams in under the first half's. (Measured round 4: a hidden layer with its whole panel requested at its start 3.0 us, 1.7 us of it MFMA issue.)
__device__ __forceinline__ void panel_prefetch_lo(Panel16& p, const float* __restrict__ P, int t) {
  const float* pp = P + (size_t)t * 16 * 256 + (threadIdx.x & 63) * 4;
#pragma unroll
  for (int u = 0; u < 8; ++u) p.b[u] = gload4(pp + (size_t)u * 256);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void panel_prefetch_hi(Panel16& p, const float* __restrict__ P, int t) {
  const float* pp = P + (size_t)t * 16 * 256 + (threadIdx.x & 63) * 4;
#pragma unroll
  for (int u = 8; u < 16; ++u) p.b[u] = gload4(pp + (size_t)u * 256);
  __builtin_amdgcn_sched_barrier(0);
}
// output tile t of As[16 x 256] . panel from registers; the MFMA order of tile_packed.  epi(t * 16, acc)
template <class Epi>
__device__ __forceinline__ void tile_packed_regs(const float* As, int lda, const Panel16& p, int t, Epi epi) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const float* ar = As + j * lda + 4 * g;
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ar + 16 * u);
    acc0 = mfma16(a[0], p.b[u][0], acc0);
    acc1 = mfma16(a[1], p.b[u][1], acc1);
    acc0 = mfma16(a[2], p.b[u][2], acc0);
    acc1 = mfma16(a[3], p.b[u][3], acc1);
  }
  const f32x4 acc = acc0 + acc1;
  epi(t * 16, acc);
}
// The narrow first layer (Kpad <= 64) of a pair-mode tile. Read as MFMA operands straight from the torch [N][Kw] matrix it is 16 scattered dword loads per lane and column
// tile when the rows are not 16-byte aligned (Kw = 18: 1.7 us of texture-address time per CU, measured as the first version's prologue): here the matrix is pulled in as
// whole 16-byte lanes (N * Kw / 4 of them, contiguous; <= 8 per thread), left in LDS as W1s[N][Kpad + 4] with zero columns behind Kw, and the operands come from there.
// Zero weights against the zero-padded rows give the same exact +0 products as the clamped addresses of load4<MODE>: same bits. Everything is REQUESTED first (w1_issue,
// rows_idx / rows_issue, then the hidden layer's panel) and committed to LDS afterwards, so the small loads are at the head of the CU's in-order return queue.
struct W1Pre { f32x4 v[8]; };
__device__ __forceinline__ void w1_issue(W1Pre& w, const float* __restrict__ W, int nlanes) {   // W 16-byte aligned, nlanes = N * Kw / 4
  const int bd = blockDim.x, tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) if (q * bd < nlanes) w.v[q] = gload4(W + 4 * (size_t)min(q * bd + tid, nlanes - 1));
}
__device__ __forceinline__ void w1_commit(const W1Pre& w, float* W1s, int ldw1, int Kw, int Kpad, int N, int nlanes) {
  const int bd = blockDim.x, tid = threadIdx.x;
  const unsigned mk = fastdiv_magic(Kw);
#pragma unroll
  for (int q = 0; q < 8; ++q) if (q * bd < nlanes) {
    const int i = q * bd + tid;
    if (i < nlanes) {
      int n = fastdiv(4 * i, mk), k = 4 * i - n * Kw;
#pragma unroll
      for (int c = 0; c < 4; ++c) { W1s[n * ldw1 + k] = w.v[q][c]; if (++k == Kw) { k = 0; ++n; } }
    }
  }
  const int pad = Kpad - Kw;
  if (pad > 0) { const unsigned mp = fastdiv_magic(pad); for (int i = tid; i < N * pad; i += bd) { const int n = fastdiv(i, mp), k = i - n * pad; W1s[n * ldw1 + Kw + k] = 0.f; } }
}
template <class Epi>
__device__ __forceinline__ void l1_compute_lds(const float* W1s, int ldw1, const float* Xs, int ldx, int Kpad, int N, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, j = lane & 15, g = lane >> 4;
  const float* xr = Xs + j * ldx + 4 * g;
  for (int c0 = wave * 16; c0 < N; c0 += nw * 16) {
    const float* wr = W1s + (c0 + j) * ldw1 + 4 * g;
    f32x4 acc0 = zero4(), acc1 = zero4();
    for (int k0 = 0; k0 < Kpad; k0 += 16) {   // the per-k-block MFMA order of tile_fwd_impl
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k0), bq = *reinterpret_cast<const f32x4*>(wr + k0);
      acc0 = mfma16(a[0], bq[0], acc0);
      acc1 = mfma16(a[1], bq[1], acc1);
      acc0 = mfma16(a[2], bq[2], acc0);
      acc1 = mfma16(a[3], bq[3], acc1);
    }
    const f32x4 acc = acc0 + acc1;
    epi(c0, acc);
  }
}
// The same first layer when the rows of W1 ARE 16-byte aligned (Kw % 4 == 0: the critics at HalfCheetah dims, IN = 24): two column tiles x Kpad / 16 k-blocks of plain
// 16-byte operand lanes per wave, requested with the rest of the prologue and held in registers - no LDS staging to pay for (w1_commit is ~150 VALU instructions per
// thread at two waves per SIMD: 0.6 us). load4<MODE> with tile_fwd_impl's arguments, tile_fwd_impl's MFMA order: same bits.
struct L1Pre { f32x4 b[2][4]; };
template <int MODE>
__device__ __forceinline__ void l1_prefetch_impl(L1Pre& w, const float* __restrict__ W, int ldw, int Kw, int Kpad, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float* wr = W + (size_t)(min((wave + q * nw) * 16, N - 16) + j) * ldw;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (16 * u < Kpad) w.b[q][u] = load4<MODE>(wr, 16 * u + 4 * g, Kw);
  }
}
__device__ __forceinline__ bool l1_rows_aligned(const float* W, int Kw) { return (Kw & 3) == 0 && Kw >= 4 && (reinterpret_cast<uintptr_t>(W) & 15) == 0; }
__device__ __forceinline__ void l1_prefetch(L1Pre& w, const float* __restrict__ W, int Kw, int Kpad, int N) {   // l1_rows_aligned(W, Kw)
  if (Kw == Kpad) l1_prefetch_impl<0>(w, W, Kw, Kw, Kpad, N); else l1_prefetch_impl<1>(w, W, Kw, Kw, Kpad, N);
}
template <class Epi>
__device__ __forceinline__ void l1_compute_regs(const L1Pre& w, const float* Xs, int ldx, int Kpad, int N, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, j = lane & 15, g = lane >> 4;
  const float* xr = Xs + j * ldx + 4 * g;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c0 = (wave + q * nw) * 16;
    if (c0 < N) {
      f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
      for (int u = 0; u < 4; ++u) if (16 * u < Kpad) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * u);
        acc0 = mfma16(a[0], w.b[q][u][0], acc0);
        acc1 = mfma16(a[1], w.b[q][u][1], acc1);
        acc0 = mfma16(a[2], w.b[q][u][2], acc0);
        acc1 = mfma16(a[3], w.b[q][u][3], acc1);
      }
      const f32x4 acc = acc0 + acc1;
      epi(c0, acc);
    }
  }
}
// The tile's [16][Kpad] input rows (load_rows_cat's values: cat(f1, f2), zero-padded; through il_batch.gather when `gather`), two elements per thread at most
// (16 * Kpad <= 2 * blockDim.x): index, then element, requested into registers; rows_commit writes them to LDS.
struct RowsPre { int64_t sr[2]; float v[2]; };
__device__ __forceinline__ void rows_idx(RowsPre& p, int Kpad, int row0, const int32_t* __restrict__ gather) {
  const int bd = blockDim.x, tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 2; ++q) { const int r = fastdiv(min(q * bd + tid, IL_TILE_R * Kpad - 1), fastdiv_magic(Kpad)); p.sr[q] = gather ? (int64_t)gload(gather + row0 + r) : (int64_t)(row0 + r); }
}
__device__ __forceinline__ void rows_issue(RowsPre& p, int Kpad, const float* __restrict__ f1, int ld1, int K1, const float* __restrict__ f2, int ld2, int K2, int row0,
                                           bool gathered, int64_t capacity, bool f2_dense) {
  const int bd = blockDim.x, tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = min(q * bd + tid, IL_TILE_R * Kpad - 1), r = fastdiv(i, fastdiv_magic(Kpad)), k = i - r * Kpad;
    int64_t sr = p.sr[q];
    if (gathered) sr = sr < 0 ? 0 : (sr >= capacity ? capacity - 1 : sr);
    const float* a = f1 + (size_t)sr * ld1 + min(k, K1 - 1);
    if (f2 && k >= K1) a = f2 + (size_t)(f2_dense ? (int64_t)(row0 + r) : sr) * ld2 + min(k - K1, K2 - 1);
    p.v[q] = gload(a);
  }
}
__device__ __forceinline__ void rows_commit(const RowsPre& p, float* Xs, int ldx, int Kpad, int Kvalid) {
  const int bd = blockDim.x, tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = q * bd + tid;
    if (i < IL_TILE_R * Kpad) { const int r = fastdiv(i, fastdiv_magic(Kpad)), k = i - r * Kpad; Xs[r * ldx + k] = k < Kvalid ? p.v[q] : 0.f; }
  }
}
// The hop between the halves of a pair. Slab: [128 columns][16 rows] floats, one 16-byte lane = 4 rows of one column (what an MFMA accumulator lane holds).
// Flag line (128 bytes per slab): word 0 = the flag (producer: 1; consumer: back to 0), word 1 = the CONSUMER's XCD + 1 (announced at its start, cleared with the flag).
// Producer: pair_store lanes, then pair_publish (every wave drains its stores; barrier; ONE relaxed agent-scope flag store). Consumer: pair_announce at its start;
// pair_receive (one polling lane, bounded like every device-side wait; barrier; sc0 sc1 loads: never served by this CU's L1; thread 0 clears the line).
// Two store flavours, chosen per wave (IL_PAIR_L2_HOP): if the consumer has announced the producer's own XCD (HW_REG_XCC_ID - checked, not assumed from the block id), the
// halves share an L2: plain stores are complete for every CU of the XCD once vmcnt says so (the vector L1 writes through), and the consumer's L1-bypassing loads hit that
// L2 - no trip to HBM on either side (measured on the update's timeline: publish 1.3 -> 0.6-0.9 us, receive 1.0-1.2 -> 0.7-1.0 us). Otherwise (other XCD, or not announced yet): write-through (sc0 sc1) stores,
// the form that is valid under any placement (MI355X guide, "Valid forms").
#ifndef IL_PAIR_L2_HOP
#define IL_PAIR_L2_HOP 1
#endif
__device__ __forceinline__ unsigned il_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xfu; }
__device__ __forceinline__ f32x4 pair_load4(const float* base, int64_t off) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7ffffff0, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4), 0, 17));   // sc0 | sc1
}
__device__ __forceinline__ void pair_announce(unsigned* line) {   // consumer, first thing
  if (threadIdx.x == 0) __hip_atomic_store(line + 1, il_xcc_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool pair_same_xcd(unsigned* line) {   // producer, a little before its stores (the load's latency hides under the MFMAs); wave-uniform
#if IL_PAIR_L2_HOP
  return __hip_atomic_load(line + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == il_xcc_id() + 1u;
#else
  return false;
#endif
}
__device__ __forceinline__ void pair_store(float* slab, int64_t off, const f32x4& v, bool same_xcd) {
  if (same_xcd) *reinterpret_cast<f32x4*>(slab + off) = v;
  else wstore4<true>(slab, off, v);
}
__device__ __forceinline__ void pair_publish(unsigned* line) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// copies the partner's 16 x 128 half out of `slab` into columns [c_base, c_base + 128) of the [16][ld] LDS tile; every thread calls (barriers inside)
template <class Timeout>
__device__ __forceinline__ void pair_receive(unsigned* line, const float* slab, float* Ts, int ld, int c_base, Timeout timed_out) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > IL_SYNC_SPIN_LIMIT) { timed_out(); break; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 128 * 4; i += blockDim.x) {
    const int c = i >> 2, r4 = (i & 3) * 4;
    const f32x4 v = pair_load4(slab, (int64_t)c * 16 + r4);
#pragma unroll
    for (int q = 0; q < 4; ++q) Ts[(r4 + q) * ld + c_base + c] = v[q];
  }
  if (threadIdx.x == 0) {   // consumed: ready for the next launch (which follows this one in stream order)
    __hip_atomic_store(line, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(line + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Small output layer: Os[16][16] = Xs[16 x K] . W^T + b, W [N][ldw] with N <= 16 (actor head 2A, critic head 1).
// The K/16 k-blocks are dealt round-robin to the waves; partial tiles are reduced through LDS (`part` >= (K/16)*256 floats; K % 16 == 0).
// Contains __syncthreads(); every thread of the block must call. Result valid after return.
// ---------------------------------------------------------------------------------------------
struct SmallPre { f32x4 b[2]; };
__device__ __forceinline__ SmallPre tile_fwd_small_prefetch(const float* __restrict__ W, int ldw, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const float* wr = W + (size_t)min(j, N - 1) * ldw;
  SmallPre p;
#pragma unroll
  for (int q = 0; q < 2; ++q) p.b[q] = gload4(wr + min((wave + q * nw) * 16, K - 16) + 4 * g);
  return p;
}
__device__ __forceinline__ void tile_fwd_small(const float* Xs, int ldx, int K, const float* __restrict__ W, int ldw, int N, const float* __restrict__ bias,
                                               float* Os, float* part, const SmallPre* pre = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const float* wr = W + (size_t)min(j, N - 1) * ldw;  // clamped row: no divergent branch around the load
  for (int k0 = wave * 16; k0 < K; k0 += nw * 16) {   // one partial tile per k-block (K / 16 <= 16), summed in k-block order below: independent of the wave count
    f32x4 acc = zero4();
    const f32x4 a = *reinterpret_cast<const f32x4*>(Xs + j * ldx + k0 + 4 * g);
    const int q = (k0 >> 4) - wave;
    const f32x4 b = (pre && q <= nw) ? (q == 0 ? pre->b[0] : pre->b[1]) : gload4(wr + k0 + 4 * g);  // columns j >= N produce garbage that Os below discards
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a[s], b[s], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(k0 >> 4) * 256 + (4 * g + r) * 16 + j] = acc[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < (K >> 4); ++w) s += part[w * 256 + i];
    const int col = i & 15;
    Os[i] = (col < N) ? s + bias[col] : 0.f;
  }
  __syncthreads();
}

// Load a [16 x (K1+K2)] tile of concatenated fields (e.g. cat(state, action)) into LDS, zero-padded to Kpad columns.
// gather != NULL: batch row r of f1 (and of f2 unless f2_dense) is source row gather[r], clamped to [0, capacity) (il_batch.gather)
__device__ __forceinline__ void load_rows_cat(float* Xs, int ldx, int Kpad, const float* __restrict__ f1, int ld1, int K1, const float* __restrict__ f2,
                                              int ld2, int K2, int row0, int nrows_valid, const int32_t* __restrict__ gather = nullptr, int64_t capacity = 0,
                                              bool f2_dense = false) {
  for (int i = threadIdx.x; i < IL_TILE_R * Kpad; i += blockDim.x) {
    const int r = i / Kpad, k = i - r * Kpad;
    float v = 0.f;
    if (r < nrows_valid) {
      size_t sr = (size_t)(row0 + r);
      if (gather) { const int64_t g = gather[row0 + r]; sr = (size_t)(g < 0 ? 0 : (g >= capacity ? capacity - 1 : g)); }
      if (k < K1) v = f1[sr * ld1 + k];
      else if (k < K1 + K2) v = f2[(f2_dense ? (size_t)(row0 + r) : sr) * ld2 + (k - K1)];
    }
    Xs[r * ldx + k] = v;
  }
}

// XCD-aware (tile, net) decode. Workgroup b is observed to run on XCD b % 8 and every XCD has a private L2, so all workgroups that
// stream the SAME network's weights are placed on the same 8/n_nets XCDs: each XCD then pulls one network through the fabric
// instead of all of them (the weights were just rewritten by the Adam kernel, so the first touch per XCD is a fabric fetch).
// Placement is a speed heuristic only; any (tile, net) bijection is correct. Falls back to net-major order when it does not divide.
__device__ __forceinline__ void xcd_tile_net(int b, int nt, int n_nets, int& tile, int& net) {
  const int xpn = 8 / n_nets;  // XCDs per network (n_nets in {1, 2, 4, 8})
  if (n_nets <= 8 && (8 % n_nets) == 0 && (nt % xpn) == 0 && ((nt * n_nets) % 8) == 0) {
    const int x = b & 7, q = b >> 3;
    net = x / xpn;
    tile = (x % xpn) * (nt / xpn) + q;
  } else {
    net = b / nt; tile = b - net * nt;
  }
}

// Population launches: grid = (workgroups per learner, learners). The dispatcher hands out workgroups in linear order (x fastest) round-robin over the 8 XCDs (workgroup g
// runs on XCD g % 8, confirmed with s_getreg XCC_ID), so with the natural decode (learner = blockIdx.y) every learner's workgroups are spread over all eight private L2s:
// each of them pulls that learner's weight panels through the fabric, and an L2 sees the weights of every learner in flight (16 at a time: 5x its capacity). Re-decoding
// the linear id puts learner l on XCD l % 8 - groups of 8 learners are interleaved - so an L2 holds the panels of the two learners it is working on and the 16 tiles of a
// network re-use them. A learner's workgroups keep their relative order (a role that waits for lower-numbered workgroups of its learner still does). Learners beyond the
// last full group of 8 keep the natural decode. Pure re-labelling: results are bit-identical.
#ifndef IL_POP_XCD
#define IL_POP_XCD 1
#endif
__device__ __forceinline__ void pop_ids(int& bx, int& by) {
#if IL_POP_XCD
  const int nx = gridDim.x, g = by * nx + bx, full = ((int)gridDim.y >> 3) * 8 * nx;
  if (g < full) { const int grp = g / (8 * nx), r = g - grp * 8 * nx; by = grp * 8 + (r & 7); bx = r >> 3; }
#endif
}

struct MlpView {  // flat torch-order parameter vector of a depth-2 MLP
  const float *W1, *b1, *W2, *b2, *W3, *b3;
};
__host__ __device__ inline int64_t mlp_numel(int in, int H, int out) { return (int64_t)H * in + H + (int64_t)H * H + H + (int64_t)out * H + out; }
__device__ __forceinline__ MlpView mlp_view(const float* p, int in, int H, int out) {
  MlpView v;
  v.W1 = p; v.b1 = v.W1 + (size_t)H * in; v.W2 = v.b1 + H; v.b2 = v.W2 + (size_t)H * H; v.W3 = v.b2 + H; v.b3 = v.W3 + (size_t)out * H;
  return v;
}
__host__ __device__ inline int round_up16(int x) { return (x + 15) & ~15; }
