// GMMIL pairwise-RBF reward (reference models.py:25-44, 183-201) for gfx950.
//
// reward_i = sum_gamma w~_i sum_j exp(-gamma d(x_i, e_j)) w~e_j  -  w~_i sum_j exp(-gamma d(x_i, x_j)) w~_j,   d(x, y) = (1/D) sum_k (x_k - y_k)^2.
// Round 6: the reward launch for D <= 128 is k_gmmil_mfma (below): the distances as a CENTRED Gram product on the matrix pipes - as close to float64 as the direct
// difference form whatever the data's offset. The kernels that follow first are the DIRECT difference forms of rounds 1-5 (3 VALU flop per pair-feature): they still
// compute the distance matrix the bandwidths' medians come from (il_gmmil_sqdist), rewards for D > 128, and everything under IL_GMMIL_MFMA=0. (The UNcentred
// ||x||^2+||y||^2-2xy form loses ~3 digits to cancellation once the data has an offset, and the reward is itself a difference of near-equal sums: never used.)
// The reference materialises [B,B,D] temporaries (0.5 GB at B=1024, D=120); here nothing larger than a tile of pair distances ever exists, and it lives in registers:
//   k_gmmil_pack   feature-major copies XT[D][B1], ET[D][B2] (so LDS tiles load coalesced and read conflict-free) and
//                  the normalised weights;
//   k_gmmil_tile   grid (i-tile, j-tile, matrix): 64x64 pairs per workgroup, 4x4 per thread, features streamed through
//                  LDS in chunks of 32 (a ring of register slots keeps every chunk of D <= 128 in flight from the start;
//                  LDS operands double-buffered in registers), two ds_read_b128 per 16 pair updates; epilogue exp +
//                  weighted row sums; the row tile's last-arriving workgroup then sums the per-column-tile partials in
//                  tile order (deterministic).
// Measured anatomy at B = 1024, D = 120 (profiles/r02_gmmil_timeline.md; s_memtime per workgroup): the packed-op loop runs at the VALU issue rate
// (4.3-4.9 ticks per v_pk instruction per SIMD); what the kernel time holds beyond it is the first fabric round trip, the exp epilogue, the
// release ticket and the last arriver's sums.
#include "il_common.hpp"
IL_ST_TABLE

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define GT 64   // tile edge (pairs): columns of a pair tile, rows of a pack tile
#define GKC 32  // feature chunk
#ifndef GMMIL_RB
#define GMMIL_RB 4   // rows of a thread's register block (x 4 columns); a pair tile is GTR rows x GT columns
#endif
#ifndef GMMIL_GG
#define GMMIL_GG 2   // features per double-buffered group of LDS operand reads
#endif
#ifndef GMMIL_PF
#define GMMIL_PF 4   // feature chunks in flight (register slots of the global -> LDS ring)
#endif
#define GTR (16 * GMMIL_RB)
#define GCTR 32   // floats between two arrival counters: same-line atomics from 8 XCDs serialise at the memory side

// exp(-g1 d) + exp(-g2 d), d = ssq / D (models.py:25-33), for one pair from its sum of squared differences: 2^(ssq * c), c = -g * log2(e) / D folded into one scalar per
// bandwidth (round 5). The straight form - an IEEE division, two expf() of the device library with their range reductions: ~50 instructions per pair - made the epilogue of a
// tile as long as its feature loop (7.5 of 21 us, profiles/r05_gmmil_sx_timeline.txt); this is two multiplies and two v_exp_f32. Error: the exponent's argument carries two
// roundings (|arg| 2^-23) and v_exp_f32 one ulp, so a term e^-x is off by <= (1.2e-7 x + 6e-8) e^-x <= 1e-7 absolute for every x >= 0: four decades inside the 1e-5 bound
// the rewards are tested at (tests/test_timed_sizes.py, test_gpu_parity.py). Every launch form shares it: they stay bit-identical to each other.
struct GmmilExp { float c1, c2; };
__host__ __device__ inline GmmilExp gmmil_exp_consts(float g1, float g2, int D) { return GmmilExp{-g1 * 1.44269504088896340736f / (float)D, -g2 * 1.44269504088896340736f / (float)D}; }
__device__ __forceinline__ float gmmil_pair_kernel(float ssq, const GmmilExp& e) { return __builtin_amdgcn_exp2f(ssq * e.c1) + __builtin_amdgcn_exp2f(ssq * e.c2); }

struct GmmilWs { int64_t xt, et, wn, wen, part, ctr, part2, total; int b1p, b2p, njt; };
__host__ __device__ inline GmmilWs gmmil_ws(int n1, int n2, int D) {
  GmmilWs w; w.b1p = (n1 + GTR - 1) / GTR * GTR; w.b2p = (n2 + GT - 1) / GT * GT;   // policy rows: whole row tiles (GTR is a multiple of GT); padded rows / columns carry weight 0
  const int nj1 = w.b2p / GT, nj2 = w.b1p / GT; w.njt = nj1 > nj2 ? nj1 : nj2;
  int64_t o = 0;
  w.xt = o; o += (int64_t)D * w.b1p; w.et = o; o += (int64_t)D * w.b2p; w.wn = o; o += w.b1p; w.wen = o; o += w.b2p;
  w.part = o; o += (int64_t)2 * w.njt * w.b1p; w.ctr = o; o += (int64_t)(w.b1p / 32) * GCTR;   // arrival counter per row tile (64 rows; k_gmmil_sx: 32 rows), one 128-byte line each (zeroed by k_gmmil_pack)
  w.part2 = o; o += (int64_t)((w.b2p + 31) / 32 + (w.b1p + 31) / 32) * w.b1p;   // k_gmmil_mfma: one partial row sum per (column block of >= 32, row)
  w.total = o;
  return w;
}
extern "C" int64_t il_gmmil_workspace_floats(int32_t n1, int32_t n2, int32_t D) { return gmmil_ws(n1, n2, D).total; }

__device__ __forceinline__ float cat_at(const il_batch& b, int S, int r, int k) {
  return k < S ? b.states[(size_t)r * b.ld_states + k] : b.actions[(size_t)r * b.ld_actions + (k - S)];
}

// grid = (tiles of 64 rows over both sets: policy tiles first, then expert tiles; feature chunks of GKC): one workgroup transposes ONE 64 x 32 chunk, so the 1 MB of
// inputs is spread over ~128 workgroups instead of 32 that each looped over four chunks behind barriers (12.7 us -> see profiles/r02_*); the chunk-0 workgroup of a
// tile also normalises its 64 weights (every such workgroup sums the whole weight column: B strided loads, a few per thread).
__global__ __launch_bounds__(256) void k_gmmil_pack(il_batch pol, il_batch exp, int S, int D, float* __restrict__ ws_) {
  __shared__ float tile[GT][GKC + 1];
  __shared__ float red[32];
  const GmmilWs w = gmmil_ws(pol.n, exp.n, D);
  const int nt1 = w.b1p / GT;
  if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < w.b1p / GTR; i += blockDim.x) reinterpret_cast<unsigned*>(ws_ + w.ctr)[i * GCTR] = 0u;
  const bool is_exp = (int)blockIdx.x >= nt1;
  const il_batch& b = is_exp ? exp : pol;
  const int n = b.n, np = is_exp ? w.b2p : w.b1p, row0 = ((int)blockIdx.x - (is_exp ? nt1 : 0)) * GT;
  float* T = ws_ + (is_exp ? w.et : w.xt);
  const int k0 = (int)blockIdx.y * GKC;
  float s = 0.f;
  if (blockIdx.y == 0) for (int i = threadIdx.x; i < n; i += blockDim.x) s += b.weights[(size_t)i * b.ld_weights];   // requested before the transpose: in flight under it
  for (int i = threadIdx.x; i < GT * GKC; i += blockDim.x) {
    const int r = i / GKC, k = i - r * GKC;
    tile[r][k] = (row0 + r < n && k0 + k < D) ? cat_at(b, S, row0 + r, k0 + k) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GT * GKC; i += blockDim.x) {
    const int k = i / GT, r = i - k * GT;
    if (k0 + k < D) T[(size_t)(k0 + k) * np + row0 + r] = tile[r][k];
  }
  if (blockIdx.y != 0) return;
  s = block_sum(s, red);
  float* wn = ws_ + (is_exp ? w.wen : w.wn);
  for (int r = threadIdx.x; r < GT; r += blockDim.x) wn[row0 + r] = (row0 + r < n) ? b.weights[(size_t)(row0 + r) * b.ld_weights] / s : 0.f;
}

// MODE 0: reward partials; MODE 1: write the distance matrix (out [n1][n2])
template <int MODE>
__global__ __launch_bounds__(256) void k_gmmil_tile(int n1, int n2, int D, float g1, float g2, float* __restrict__ ws_, float* __restrict__ dist_out, int self_second,
                                                    float* __restrict__ out_r, float* __restrict__ out_sim, float* __restrict__ out_self) {
  constexpr int RB = GMMIL_RB, RQ = RB / 4;   // a thread's block: RB rows (RQ 16-byte LDS reads) x 4 columns (one read)
  __shared__ __attribute__((aligned(16))) float Xs[GKC][GTR];
  __shared__ __attribute__((aligned(16))) float Ys[GKC][GT];
  const GmmilWs w = gmmil_ws(n1, n2, D);
  const int it = blockIdx.x, jt = blockIdx.y, mat = blockIdx.z;  // mat 0: policy vs expert, 1: policy vs policy
  const bool vs_self = (mat == 1) || (MODE == 1 && self_second);
  const int npy = vs_self ? w.b1p : w.b2p;
  if (jt * GT >= npy) return;
  const float* XT = ws_ + w.xt; const float* YT = ws_ + (vs_self ? w.xt : w.et);
  const float* wy = ws_ + (vs_self ? w.wn : w.wen);
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  f32x2 acc2[RB][2];
#pragma unroll
  for (int a = 0; a < RB; ++a) { acc2[a][0] = f32x2{0.f, 0.f}; acc2[a][1] = f32x2{0.f, 0.f}; }
  // Feature chunks of GKC through a ring of NPF register slots: the loads of chunks c+1 .. c+NPF-1 are in flight while chunk c is consumed, and a slot is
  // refilled (chunk c+NPF) as soon as its values are in LDS. XT / ET were written by k_gmmil_pack a moment ago, mostly on other XCDs, so a load is a trip to
  // the fabric: measured 1.7 us per chunk against 1 us of packed ops - with one chunk of look-ahead the loop waited 60 % of its time. D <= NPF * GKC (every
  // shipped environment): all of a tile's operands are requested before the first barrier.
  constexpr int PERX = GKC * GTR / 256, PERY = GKC * GT / 256, NPF = GMMIL_PF;
  float xr[NPF][PERX], yr[NPF][PERY];
  auto fetch = [&](float* xs_, float* ys_, int k0) {
#pragma unroll
    for (int u = 0; u < PERX; ++u) {
      const int i = threadIdx.x + u * 256, k = i / GTR, c = i - k * GTR;
      xs_[u] = XT[(size_t)min(k0 + k, D - 1) * w.b1p + it * GTR + c];   // clamped address, zeroed when it is stored to LDS: a predicated load is an exec-masked branch each,
                                                                        // a select right here would wait for the load
    }
#pragma unroll
    for (int u = 0; u < PERY; ++u) {
      const int i = threadIdx.x + u * 256, k = i / GT, c = i - k * GT;
      ys_[u] = YT[(size_t)min(k0 + k, D - 1) * npy + jt * GT + c];
    }
  };
  const bool stamp = blockIdx.x == 3 && blockIdx.y == 5 && blockIdx.z == 0;
  IL_STAMP(stamp, 0);
#pragma unroll
  for (int sl = 0; sl < NPF; ++sl) if (sl * GKC < D) fetch(xr[sl], yr[sl], sl * GKC);
  for (int kk = 0; kk < D; kk += NPF * GKC) {
#pragma unroll
   for (int sl = 0; sl < NPF; ++sl) {
    const int k0 = kk + sl * GKC;
    if (k0 >= D) break;
#pragma unroll
    for (int u = 0; u < PERX; ++u) { const int i = threadIdx.x + u * 256, k = i / GTR, c = i - k * GTR; Xs[k][c] = (k0 + k < D) ? xr[sl][u] : 0.f; }
#pragma unroll
    for (int u = 0; u < PERY; ++u) { const int i = threadIdx.x + u * 256, k = i / GT, c = i - k * GT; Ys[k][c] = (k0 + k < D) ? yr[sl][u] : 0.f; }
    __syncthreads();
    if (k0 + NPF * GKC < D) fetch(xr[sl], yr[sl], k0 + NPF * GKC);
    // all GKC features of the chunk (features >= D are staged as zeros on both sides: they add (0 - 0)^2), so the loop has a fixed trip count.
    // The LDS operands are double-buffered in registers in groups of GG features: the ds_read_b128 of the NEXT group are issued (and pinned there by a
    // scheduling barrier) before the packed ops of the current one. Left to itself hipcc sinks every read to its first use: read - s_waitcnt lgkmcnt(0) -
    // 16 ops per feature, i.e. a full LDS round trip exposed per feature step with one or two waves per SIMD (measured 200 cycles per step against 68 of VALU issue).
    constexpr int GG = GMMIL_GG;
    f32x4 xa_[GG][RQ], ya_[GG], xb_[GG][RQ], yb_[GG];
    auto lds_group = [&](f32x4 (*xg)[RQ], f32x4* yg, int kb) {
#pragma unroll
      for (int u = 0; u < GG; ++u) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) xg[u][q] = *reinterpret_cast<const f32x4*>(&Xs[kb + u][ti * RB + 4 * q]);
        yg[u] = *reinterpret_cast<const f32x4*>(&Ys[kb + u][tj * 4]);
      }
    };
    auto fma_group = [&](const f32x4 (*xg)[RQ], const f32x4* yg) {
#pragma unroll
      for (int u = 0; u < GG; ++u) {
        const f32x2 y01 = {yg[u][0], yg[u][1]}, y23 = {yg[u][2], yg[u][3]};
#pragma unroll
        for (int a = 0; a < RB; ++a) {   // two pairs per instruction: v_pk_add_f32 + v_pk_fma_f32 (same roundings as the scalar sub + fma)
          const float xs = xg[u][a >> 2][a & 3];
          const f32x2 xa = {xs, xs};
          const f32x2 d0 = xa - y01, d1 = xa - y23;
          acc2[a][0] = __builtin_elementwise_fma(d0, d0, acc2[a][0]);
          acc2[a][1] = __builtin_elementwise_fma(d1, d1, acc2[a][1]);
        }
      }
    };
    lds_group(xa_, ya_, 0);
#pragma unroll 1
    for (int kb = 0; kb < GKC; kb += 2 * GG) {   // a rolled loop: unrolled, the copies do not share registers (296 VGPRs with four chunks in flight)
      lds_group(xb_, yb_, kb + GG);
      __builtin_amdgcn_sched_barrier(0);
      fma_group(xa_, ya_);
      __builtin_amdgcn_sched_barrier(0);
      lds_group(xa_, ya_, (kb + 2 * GG) & (GKC - 1));   // the last trip re-reads group 0 (discarded) instead of branching
      __builtin_amdgcn_sched_barrier(0);
      fma_group(xb_, yb_);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
   }
  }
  IL_STAMP(stamp, 1);
  float acc[RB][4];
#pragma unroll
  for (int a = 0; a < RB; ++a) { acc[a][0] = acc2[a][0][0]; acc[a][1] = acc2[a][0][1]; acc[a][2] = acc2[a][1][0]; acc[a][3] = acc2[a][1][1]; }
  const float fD = (float)D;
  const GmmilExp gex = gmmil_exp_consts(g1, g2, D);
  if (MODE == 1) {
    const int n2e = vs_self ? n1 : n2;
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = it * GTR + ti * RB + a, j = jt * GT + tj * 4 + b;
        if (i < n1 && j < n2e) dist_out[(size_t)i * n2e + j] = acc[a][b] / fD;
      }
    return;
  }
  const f32x4 wv = *reinterpret_cast<const f32x4*>(wy + jt * GT + tj * 4);
  float* part = ws_ + w.part + ((size_t)mat * w.njt + jt) * w.b1p + it * GTR;
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) s += wv[b] * gmmil_pair_kernel(acc[a][b], gex);
    s = group16_sum(s);
    if (tj == 0) part[ti * RB + a] = s;
  }
  IL_STAMP(stamp, 2);
  if (!out_r) return;
  // The row tile's reward needs the partial sums of every column tile of BOTH matrices: the workgroup that arrives last (one agent-scope release
  // ticket per workgroup, after a barrier; only the last arriver pays for the acquire) adds them up in tile order - the separate, launch-bound
  // "final" kernel this replaces cost 11 us.
  __shared__ unsigned last;
  sync_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned expect = (unsigned)(w.b2p / GT + w.b1p / GT);
    last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(ws_ + w.ctr) + it * GCTR, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1u == expect;
  }
  __syncthreads();
  if (last) sync_acquire_all();   // (every wave of the last arriver: il_common.hpp)
  IL_STAMP(stamp, 3);
  if (!last || threadIdx.x >= GTR) return;
  const int i = it * GTR + threadIdx.x;
  if (i >= n1) return;
  // all partials of a matrix requested before the first add (a dependent load-add chain is one fabric round trip per column tile); added in tile order
  auto ordered_sum = [&](const float* p, int nq) {
    float s = 0.f;
    for (int q0 = 0; q0 < nq; q0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(size_t)min(q0 + u, nq - 1) * w.b1p];
#pragma unroll
      for (int u = 0; u < 16; ++u) if (q0 + u < nq) s += v[u];
    }
    return s;
  };
  const float s0 = ordered_sum(ws_ + w.part + i, w.b2p / GT);
  const float s1 = ordered_sum(ws_ + w.part + (size_t)w.njt * w.b1p + i, w.b1p / GT);
  const float wi = ws_[w.wn + i];
  const float sim = wi * s0, self = wi * s1;
  out_r[i] = sim - self;
  if (out_sim) out_sim[i] = sim;
  if (out_self) out_self[i] = self;
}

// ---------------------------------------------------------------------------------------------
// k_gmmil_direct (round 4): k_gmmil_tile reading its operands straight from the two batches. k_gmmil_pack only existed to give the tile kernel feature-major operands; it
// cost a 6 us launch plus a kernel boundary per reward call, and its output - written an instant earlier, mostly on other XCDs - made every operand load of the tile
// kernel a fabric round trip. Here a thread loads 16-byte lanes ALONG a row (8 lanes cover the 32 features of a chunk: whole 128-byte lines of the row-major batch),
// and transposes on its way into LDS: Xs[k][r ^ 4 ((k / 4) % 8)] - the XOR swizzle keeps every group of four consecutive rows aligned and contiguous (the inner loop's
// ds_read_b128) and spreads the eight lanes that write one row's 32 features over eight banks (an unswizzled transposed store is an 8-way conflict). Each workgroup sums
// the weight columns itself (k_gmmil_pack's chunk-0 workgroups did the same sums in the same order: same bits). Same pair arithmetic, same partial sums, same
// last-arriver reduction: bit-identical rewards. The arrival counters are left at zero by the last arriver of each row tile: zero-initialise the workspace ONCE per shape
// (include/il_hip.h). Rows that are not whole 16-byte lanes (S, A, strides or pointers not multiples of 4 floats) take the element-wise loads of cat_at.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 cat_lane(const il_batch& b, int S, int D, int r, int k, bool lanes) {   // features k .. k + 3 of row r (k % 4 == 0); beyond D: don't care (zeroed at the LDS store)
  if (lanes) {
    const int kc = min(k, D - 4);
    return kc < S ? gload4(b.states + (size_t)r * b.ld_states + kc) : gload4(b.actions + (size_t)r * b.ld_actions + (kc - S));
  }
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = cat_at(b, S, r, min(k + q, D - 1));
  return v;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_gmmil_direct(il_batch pol, il_batch exp, int S, int D, float g1, float g2, float* __restrict__ ws_, float* __restrict__ dist_out, int self_second,
                                                      float* __restrict__ out_r, float* __restrict__ out_sim, float* __restrict__ out_self, int lanes) {
  constexpr int RB = GMMIL_RB, RQ = RB / 4;
  __shared__ __attribute__((aligned(16))) float Xs[GKC][GTR];
  __shared__ __attribute__((aligned(16))) float Ys[GKC][GT];
  __shared__ float red[32];
  globalize(pol); globalize(exp);
  const int n1 = pol.n, n2 = exp.n;
  const GmmilWs w = gmmil_ws(n1, n2, D);
  const int it = blockIdx.x, jt = blockIdx.y, mat = blockIdx.z;  // mat 0: policy vs expert, 1: policy vs policy
  const bool vs_self = (mat == 1) || (MODE == 1 && self_second);
  const int npy = vs_self ? w.b1p : w.b2p;
  if (jt * GT >= npy) return;
  const il_batch& yb = vs_self ? pol : exp;
  const int ny = yb.n;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  // weight-column sums (MODE 0): requested first, reduced after the feature loop
  float sx = 0.f, sy = 0.f;
  if (MODE == 0) {
    for (int i = threadIdx.x; i < n1; i += blockDim.x) sx += pol.weights[(size_t)i * pol.ld_weights];
    if (!vs_self) for (int i = threadIdx.x; i < ny; i += blockDim.x) sy += yb.weights[(size_t)i * yb.ld_weights];
  }
  f32x2 acc2[RB][2];
#pragma unroll
  for (int a = 0; a < RB; ++a) { acc2[a][0] = f32x2{0.f, 0.f}; acc2[a][1] = f32x2{0.f, 0.f}; }
  constexpr int PX4 = GKC * GTR / 4 / 256, PY4 = GKC * GT / 4 / 256, NPF = GMMIL_PF;   // 16-byte lanes per thread and chunk
  f32x4 xr[NPF][PX4], yr[NPF][PY4];
  auto fetch = [&](f32x4* xs_, f32x4* ys_, int k0) {
#pragma unroll
    for (int u = 0; u < PX4; ++u) { const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7; xs_[u] = cat_lane(pol, S, D, min(it * GTR + r, n1 - 1), k0 + 4 * kq, lanes != 0); }
#pragma unroll
    for (int u = 0; u < PY4; ++u) { const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7; ys_[u] = cat_lane(yb, S, D, min(jt * GT + r, ny - 1), k0 + 4 * kq, lanes != 0); }
  };
#pragma unroll
  for (int sl = 0; sl < NPF; ++sl) if (sl * GKC < D) fetch(xr[sl], yr[sl], sl * GKC);
  for (int kk = 0; kk < D; kk += NPF * GKC) {
#pragma unroll
   for (int sl = 0; sl < NPF; ++sl) {
    const int k0 = kk + sl * GKC;
    if (k0 >= D) break;
#pragma unroll
    for (int u = 0; u < PX4; ++u) {
      const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7, col = r ^ (4 * kq);
      const bool rv = it * GTR + r < n1;
#pragma unroll
      for (int q = 0; q < 4; ++q) Xs[4 * kq + q][col] = (rv && k0 + 4 * kq + q < D) ? xr[sl][u][q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PY4; ++u) {
      const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7, col = r ^ (4 * kq);
      const bool rv = jt * GT + r < ny;
#pragma unroll
      for (int q = 0; q < 4; ++q) Ys[4 * kq + q][col] = (rv && k0 + 4 * kq + q < D) ? yr[sl][u][q] : 0.f;
    }
    __syncthreads();
    if (k0 + NPF * GKC < D) fetch(xr[sl], yr[sl], k0 + NPF * GKC);
    constexpr int GG = GMMIL_GG;   // (2: both features of a group share the swizzle of their group of four)
    f32x4 xa_[GG][RQ], ya_[GG], xb_[GG][RQ], yb_[GG];
    auto lds_group = [&](f32x4 (*xg)[RQ], f32x4* yg, int kb) {
      const int sw = ((kb >> 2) & 7) << 2;
#pragma unroll
      for (int u = 0; u < GG; ++u) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) xg[u][q] = *reinterpret_cast<const f32x4*>(&Xs[kb + u][(ti * RB + 4 * q) ^ sw]);
        yg[u] = *reinterpret_cast<const f32x4*>(&Ys[kb + u][(tj * 4) ^ sw]);
      }
    };
    auto fma_group = [&](const f32x4 (*xg)[RQ], const f32x4* yg) {
#pragma unroll
      for (int u = 0; u < GG; ++u) {
        const f32x2 y01 = {yg[u][0], yg[u][1]}, y23 = {yg[u][2], yg[u][3]};
#pragma unroll
        for (int a = 0; a < RB; ++a) {
          const float xs = xg[u][a >> 2][a & 3];
          const f32x2 xa = {xs, xs};
          const f32x2 d0 = xa - y01, d1 = xa - y23;
          acc2[a][0] = __builtin_elementwise_fma(d0, d0, acc2[a][0]);
          acc2[a][1] = __builtin_elementwise_fma(d1, d1, acc2[a][1]);
        }
      }
    };
    lds_group(xa_, ya_, 0);
#pragma unroll 1
    for (int kb = 0; kb < GKC; kb += 2 * GG) {
      lds_group(xb_, yb_, kb + GG);
      __builtin_amdgcn_sched_barrier(0);
      fma_group(xa_, ya_);
      __builtin_amdgcn_sched_barrier(0);
      lds_group(xa_, ya_, (kb + 2 * GG) & (GKC - 1));
      __builtin_amdgcn_sched_barrier(0);
      fma_group(xb_, yb_);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
   }
  }
  float acc[RB][4];
#pragma unroll
  for (int a = 0; a < RB; ++a) { acc[a][0] = acc2[a][0][0]; acc[a][1] = acc2[a][0][1]; acc[a][2] = acc2[a][1][0]; acc[a][3] = acc2[a][1][1]; }
  const float fD = (float)D;
  const GmmilExp gex = gmmil_exp_consts(g1, g2, D);
  if (MODE == 1) {
    const int n2e = vs_self ? n1 : n2;
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = it * GTR + ti * RB + a, j = jt * GT + tj * 4 + b;
        if (i < n1 && j < n2e) dist_out[(size_t)i * n2e + j] = acc[a][b] / fD;
      }
    return;
  }
  sx = block_sum(sx, red);
  sy = vs_self ? sx : block_sum(sy, red);
  f32x4 wv;
#pragma unroll
  for (int b = 0; b < 4; ++b) { const int j = jt * GT + tj * 4 + b; wv[b] = j < ny ? yb.weights[(size_t)j * yb.ld_weights] / sy : 0.f; }
  float* part = ws_ + w.part + ((size_t)mat * w.njt + jt) * w.b1p + it * GTR;
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) s += wv[b] * gmmil_pair_kernel(acc[a][b], gex);
    s = group16_sum(s);
    if (tj == 0) part[ti * RB + a] = s;
  }
  if (!out_r) return;
  __shared__ unsigned last;
  sync_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned expect = (unsigned)(w.b2p / GT + w.b1p / GT);
    unsigned* ctr = reinterpret_cast<unsigned*>(ws_ + w.ctr) + it * GCTR;
    last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1u == expect;
    if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next call
  }
  __syncthreads();
  if (last) sync_acquire_all();
  if (!last || threadIdx.x >= GTR) return;
  const int i = it * GTR + threadIdx.x;
  if (i >= n1) return;
  auto ordered_sum = [&](const float* p, int nq) {
    float s = 0.f;
    for (int q0 = 0; q0 < nq; q0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(size_t)min(q0 + u, nq - 1) * w.b1p];
#pragma unroll
      for (int u = 0; u < 16; ++u) if (q0 + u < nq) s += v[u];
    }
    return s;
  };
  const float s0 = ordered_sum(ws_ + w.part + i, w.b2p / GT);
  const float s1 = ordered_sum(ws_ + w.part + (size_t)w.njt * w.b1p + i, w.b1p / GT);
  const float wi = pol.weights[(size_t)i * pol.ld_weights] / sx;
  const float sim = wi * s0, self = wi * s1;
  out_r[i] = sim - self;
  if (out_sim) out_sim[i] = sim;
  if (out_self) out_self[i] = self;
}
// ---------------------------------------------------------------------------------------------
// k_gmmil_resident (round 5): k_gmmil_direct with the WHOLE feature range of both operand tiles resident in LDS (64 rows x D features each: 2 x 30 KB at Ant dims, two
// workgroups per CU) instead of a ring of 32-feature chunks. The chunked form paid two workgroup barriers and a burst of transposing LDS stores per chunk - phases in which
// the SIMDs issue no pair arithmetic - and its feature loop ran at ~64 % of the packed-op issue rate with two co-resident workgroups (profiles/r02_gmmil_timeline.md: 27.5k
// ticks for 2 x 4 chunks of 2.2k). Here: every operand lane is requested up front (as before: D <= 128 always had all chunks in flight), stored transposed with the same
// XOR swizzle, ONE barrier, then a barrier-free loop over all features with the LDS operands double-buffered in registers. The arrival of a tile's partial row sums is
// fence-free (the pair-mode kernels' mechanism, mlp_tile.hpp): partials written THROUGH (sc0 sc1), stores drained, barrier, one relaxed ticket; the last arriver reads them
// below the caches - no agent-scope release (an L2 write-back per workgroup) and no acquire. Same pair arithmetic in the same feature order, same 64-column partial sums,
// same tile-ordered final sums: bit-identical to k_gmmil_direct / k_gmmil_pack + k_gmmil_tile.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int gmmil_kp4(int D) { return (D + 3) & ~3; }
static size_t gmmil_resident_lds(int D) { return ((size_t)gmmil_kp4(D) * (GTR + GT) + 64) * sizeof(float); }
template <int MODE>
__global__ __launch_bounds__(256) void k_gmmil_resident(il_batch pol, il_batch exp, int S, int D, float g1, float g2, float* __restrict__ ws_, float* __restrict__ dist_out, int self_second,
                                                        float* __restrict__ out_r, float* __restrict__ out_sim, float* __restrict__ out_self, int lanes) {
  constexpr int RB = GMMIL_RB, RQ = RB / 4, GG = GMMIL_GG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  IL_ST_BEGIN(IL_ST_GMMIL);
  const int KP = gmmil_kp4(D);
  float* Xs = smem;                       // [KP][GTR], column r of feature k at (r ^ 4 ((k / 4) % 8))
  float* Ys = Xs + (size_t)KP * GTR;      // [KP][GT]
  float* red = Ys + (size_t)KP * GT;      // [32] block_sum scratch, [32] = "last arriver" flag
  globalize(pol); globalize(exp);
  const int n1 = pol.n, n2 = exp.n;
  const GmmilWs w = gmmil_ws(n1, n2, D);
  const int it = blockIdx.x, jt = blockIdx.y, mat = blockIdx.z;  // mat 0: policy vs expert, 1: policy vs policy
  const bool vs_self = (mat == 1) || (MODE == 1 && self_second);
  const int npy = vs_self ? w.b1p : w.b2p;
  if (jt * GT >= npy) { IL_ST_END(IL_ST_GMMIL); return; }
  const il_batch& yb = vs_self ? pol : exp;
  const int ny = yb.n;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  float sx = 0.f, sy = 0.f;
  if (MODE == 0) {
    for (int i = threadIdx.x; i < n1; i += blockDim.x) sx += pol.weights[(size_t)i * pol.ld_weights];
    if (!vs_self) for (int i = threadIdx.x; i < ny; i += blockDim.x) sy += yb.weights[(size_t)i * yb.ld_weights];
  }
  // every 16-byte operand lane of both tiles, requested before anything is consumed (chunks of 32 features = 8 lanes along a row, like k_gmmil_direct); NCH chunks in registers
  constexpr int PX4 = GKC * GTR / 4 / 256, PY4 = GKC * GT / 4 / 256, NCH = 5;   // D <= 160 (the launcher checks)
  f32x4 xr[NCH][PX4], yr[NCH][PY4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * GKC < D) {
#pragma unroll
      for (int u = 0; u < PX4; ++u) { const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7; xr[c][u] = cat_lane(pol, S, D, min(it * GTR + r, n1 - 1), c * GKC + 4 * kq, lanes != 0); }
#pragma unroll
      for (int u = 0; u < PY4; ++u) { const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7; yr[c][u] = cat_lane(yb, S, D, min(jt * GT + r, ny - 1), c * GKC + 4 * kq, lanes != 0); }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * GKC < D) {
#pragma unroll
      for (int u = 0; u < PX4; ++u) {
        const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7, col = r ^ (4 * kq);
        const bool rv = it * GTR + r < n1;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = c * GKC + 4 * kq + q; if (k < KP) Xs[(size_t)k * GTR + col] = (rv && k < D) ? xr[c][u][q] : 0.f; }
      }
#pragma unroll
      for (int u = 0; u < PY4; ++u) {
        const int i = threadIdx.x + u * 256, r = i >> 3, kq = i & 7, col = r ^ (4 * kq);
        const bool rv = jt * GT + r < ny;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = c * GKC + 4 * kq + q; if (k < KP) Ys[(size_t)k * GT + col] = (rv && k < D) ? yr[c][u][q] : 0.f; }
      }
    }
  }
  __syncthreads();
  f32x2 acc2[RB][2];
#pragma unroll
  for (int a = 0; a < RB; ++a) { acc2[a][0] = f32x2{0.f, 0.f}; acc2[a][1] = f32x2{0.f, 0.f}; }
  {
    f32x4 xa_[GG][RQ], ya_[GG], xb_[GG][RQ], yb_[GG];
    auto lds_group = [&](f32x4 (*xg)[RQ], f32x4* yg, int kb) {   // (both features of a group of GG = 2 share the swizzle of their group of four)
      const int sw = ((kb >> 2) & 7) << 2;
#pragma unroll
      for (int u = 0; u < GG; ++u) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) xg[u][q] = *reinterpret_cast<const f32x4*>(&Xs[(size_t)(kb + u) * GTR + ((ti * RB + 4 * q) ^ sw)]);
        yg[u] = *reinterpret_cast<const f32x4*>(&Ys[(size_t)(kb + u) * GT + ((tj * 4) ^ sw)]);
      }
    };
    auto fma_group = [&](const f32x4 (*xg)[RQ], const f32x4* yg) {
#pragma unroll
      for (int u = 0; u < GG; ++u) {
        const f32x2 y01 = {yg[u][0], yg[u][1]}, y23 = {yg[u][2], yg[u][3]};
#pragma unroll
        for (int a = 0; a < RB; ++a) {   // two pairs per instruction: v_pk_add_f32 + v_pk_fma_f32 (same roundings as the scalar sub + fma)
          const float xs = xg[u][a >> 2][a & 3];
          const f32x2 xa = {xs, xs};
          const f32x2 d0 = xa - y01, d1 = xa - y23;
          acc2[a][0] = __builtin_elementwise_fma(d0, d0, acc2[a][0]);
          acc2[a][1] = __builtin_elementwise_fma(d1, d1, acc2[a][1]);
        }
      }
    };
    lds_group(xa_, ya_, 0);
#pragma unroll 1
    for (int kb = 0; kb < KP; kb += 2 * GG) {   // features kb .. kb + 3 (features >= D are zeros on both sides: they add (0 - 0)^2)
      lds_group(xb_, yb_, kb + GG);
      __builtin_amdgcn_sched_barrier(0);
      fma_group(xa_, ya_);
      __builtin_amdgcn_sched_barrier(0);
      lds_group(xa_, ya_, min(kb + 2 * GG, KP - GG));   // (the last trip re-reads a group that is discarded instead of branching)
      __builtin_amdgcn_sched_barrier(0);
      fma_group(xb_, yb_);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float acc[RB][4];
#pragma unroll
  for (int a = 0; a < RB; ++a) { acc[a][0] = acc2[a][0][0]; acc[a][1] = acc2[a][0][1]; acc[a][2] = acc2[a][1][0]; acc[a][3] = acc2[a][1][1]; }
  const float fD = (float)D;
  const GmmilExp gex = gmmil_exp_consts(g1, g2, D);
  if (MODE == 1) {
    const int n2e = vs_self ? n1 : n2;
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = it * GTR + ti * RB + a, j = jt * GT + tj * 4 + b;
        if (i < n1 && j < n2e) dist_out[(size_t)i * n2e + j] = acc[a][b] / fD;
      }
    IL_ST_END(IL_ST_GMMIL);
    return;
  }
  sx = block_sum(sx, red);
  sy = vs_self ? sx : block_sum(sy, red);
  f32x4 wv;
#pragma unroll
  for (int b = 0; b < 4; ++b) { const int j = jt * GT + tj * 4 + b; wv[b] = j < ny ? yb.weights[(size_t)j * yb.ld_weights] / sy : 0.f; }
  float* part = ws_ + w.part + ((size_t)mat * w.njt + jt) * w.b1p + it * GTR;
#pragma unroll
  for (int a = 0; a < RB; ++a) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) s += wv[b] * gmmil_pair_kernel(acc[a][b], gex);
    s = group16_sum(s);
    if (tj == 0) wstore1(part, ti * RB + a, s);   // written through: the row tile's last arriver reads it below the caches
  }
  if (!out_r) { IL_ST_END(IL_ST_GMMIL); return; }
  // The row tile's reward needs the partial sums of every column tile of BOTH matrices: the workgroup that arrives last adds them up in tile order. Every wave drains its
  // write-through stores, barrier, ONE relaxed ticket (no release: nothing of this workgroup that another one reads sits in a cache); the counter is left at zero.
  unsigned* lastp = reinterpret_cast<unsigned*>(red + 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned expect = (unsigned)(w.b2p / GT + w.b1p / GT);
    unsigned* ctr = reinterpret_cast<unsigned*>(ws_ + w.ctr) + it * GCTR;
    const unsigned last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == expect;
    if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next call
    *lastp = last;
  }
  __syncthreads();
  const bool last = *lastp != 0u;
  const int i = it * GTR + threadIdx.x;
  if (last && threadIdx.x < GTR && i < n1) {
    // all partials of a matrix requested before the first add (a dependent load-add chain is one fabric round trip per column tile); added in tile order
    auto ordered_sum = [&](const float* p, int nq) {
      float s = 0.f;
      for (int q0 = 0; q0 < nq; q0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sload1(p, (int64_t)min(q0 + u, nq - 1) * w.b1p + i);
#pragma unroll
        for (int u = 0; u < 16; ++u) if (q0 + u < nq) s += v[u];
      }
      return s;
    };
    const float s0 = ordered_sum(ws_ + w.part, w.b2p / GT);
    const float s1 = ordered_sum(ws_ + w.part + (size_t)w.njt * w.b1p, w.b1p / GT);
    const float wi = pol.weights[(size_t)i * pol.ld_weights] / sx;
    const float sim = wi * s0, self = wi * s1;
    out_r[i] = sim - self;
    if (out_sim) out_sim[i] = sim;
    if (out_self) out_self[i] = self;
  }
  IL_ST_END(IL_ST_GMMIL);
}
static int gmmil_resident_on(int D) {   // IL_GMMIL_RESIDENT=0: the chunked k_gmmil_direct (developer A/B; same bits). D <= 160: five chunks of operand lanes in registers
  static const int on = [] { const char* e = getenv("IL_GMMIL_RESIDENT"); return e && e[0] == '0' ? 0 : 1; }();
  return on != 0 && D >= 4 && D <= 160;
}
template <class K>
static int gmmil_ensure_lds(K fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  const hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? IL_OK : il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
}
// ---------------------------------------------------------------------------------------------
// k_gmmil_sx (round 5): the pair arithmetic with the ROW operand in scalar registers. In k_gmmil_direct / _resident a thread's 4 x 4 pairs need two ds_read_b128 per
// feature (its 4 rows, its 4 columns) for 16 packed instructions: with two workgroups per CU the LDS pipe (8 waves x 16 clocks per feature) is as busy as the SIMDs
// (2 waves x 64 clocks): neither reaches its rate (round-2 timeline: 64 % of the packed-op issue rate; 8 x 4 blocks halve the reads but leave one wave per SIMD and
// lose, profiles/r05_gmmil_rb8_ab.txt). Here a WAVE owns 4 rows x 256 columns: its rows' features are wave-uniform, so they come through the scalar cache
// (s_load_dwordx8 from the row-major batch, constant address space) into SGPRs and feed v_pk_add_f32 as broadcast scalar operands (op_sel) - no LDS read, no VALU
// instruction; the lane's 4 columns stay ONE ds_read_b128 per feature. Workgroup = 8 waves = 32 rows x 256 columns with all D features of the 256 columns resident in
// LDS (120 KB at Ant dims: one workgroup per CU, two waves per SIMD, 256 workgroups at B = 1024). Same pair arithmetic in the same feature order, the same 64-column
// partial sums (a 16-lane group = one 64-column tile), the same tile-ordered final sums, the weight sums by the first 256 threads in the old order: bit-identical.
// Needs whole 16-byte lanes, S and D multiples of 8 (a group of 8 features never straddles states | actions) and D <= 156 (LDS); other shapes keep k_gmmil_resident.
// ---------------------------------------------------------------------------------------------
#define GSX_ROWS 32
#define GSX_COLS 256
#define AS4 __attribute__((address_space(4)))
typedef float f32x8 __attribute__((ext_vector_type(8)));
static size_t gmmil_sx_lds(int D) { return ((size_t)D * GSX_COLS + 64 + 4 * GSX_ROWS) * sizeof(float); }
template <int MODE>
__global__ __launch_bounds__(512) void k_gmmil_sx(il_batch pol, il_batch exp, int S, int D, float g1, float g2, float* __restrict__ ws_, float* __restrict__ dist_out, int self_second,
                                                  float* __restrict__ out_r, float* __restrict__ out_sim, float* __restrict__ out_self) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  IL_ST_BEGIN(IL_ST_GMMIL);
  IL_TL(0, 0);
  float* Ys = smem;                          // [D][256], column c of feature k at (c ^ 4 ((k / 4) % 8))
  float* red = Ys + (size_t)D * GSX_COLS;    // [32] block_sum scratch, [32] = "last arriver" flag
  // (the row operand's base pointers as the kernel arguments carry them - scalar registers; globalize() launders the descriptors through vector registers)
  const AS4 float* x_states = (const AS4 float*)pol.states; const AS4 float* x_actions = (const AS4 float*)pol.actions;
  const int x_ld_s = pol.ld_states, x_ld_a = pol.ld_actions;
  globalize(pol); globalize(exp);
  const int n1 = pol.n, n2 = exp.n;
  const GmmilWs w = gmmil_ws(n1, n2, D);
  const int it = blockIdx.x, jt4 = blockIdx.y, mat = blockIdx.z;  // mat 0: policy vs expert, 1: policy vs policy
  const bool vs_self = (mat == 1) || (MODE == 1 && self_second);
  const int npy = vs_self ? w.b1p : w.b2p;
  if (jt4 * GSX_COLS >= npy) { IL_ST_END(IL_ST_GMMIL); return; }
  const il_batch& yb = vs_self ? pol : exp;
  const int ny = yb.n, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // weight-column sums (MODE 0): the first 256 threads in the order of the 256-thread kernels (the other waves add exact zeros in block_sum)
  float sx = 0.f, sy = 0.f;
  if (MODE == 0 && tid < 256) {
    for (int i = tid; i < n1; i += 256) sx += pol.weights[(size_t)i * pol.ld_weights];
    if (!vs_self) for (int i = tid; i < ny; i += 256) sy += yb.weights[(size_t)i * yb.ld_weights];
  }
  f32x4 wv = {0.f, 0.f, 0.f, 0.f};   // this lane's four column weights, requested with everything else (normalised once the sums are known)
  if (MODE == 0) {
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int j = jt4 * GSX_COLS + lane * 4 + b; wv[b] = j < ny ? yb.weights[(size_t)j * yb.ld_weights] : 0.f; }
  }
  // the 256 columns' features: every 16-byte lane requested before anything is consumed (chunks of 32 features = 8 lanes along a row, 4 lanes per thread and chunk)
  constexpr int NCH = 5, PY4 = GKC * GSX_COLS / 4 / 512;
  f32x4 yr[NCH][PY4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * GKC < D) {
#pragma unroll
      for (int u = 0; u < PY4; ++u) { const int i = tid + u * 512, r = i >> 3, kq = i & 7; yr[c][u] = cat_lane(yb, S, D, min(jt4 * GSX_COLS + r, ny - 1), c * GKC + 4 * kq, true); }
    }
  }
  // this wave's 4 rows through the scalar cache: 8 features of a row = one s_load_dwordx8 (uniform addresses in the constant address space)
  const AS4 float* xrow[4]; const AS4 float* xact[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = min(it * GSX_ROWS + wave * 4 + a, n1 - 1);
    xrow[a] = x_states + (size_t)r * x_ld_s;
    xact[a] = x_actions + (size_t)r * x_ld_a;
  }
  auto xload = [&](f32x8* x8, int k0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) x8[a] = k0 < S ? *(const AS4 f32x8*)(xrow[a] + k0) : *(const AS4 f32x8*)(xact[a] + (k0 - S));
  };
  f32x8 xr[4], xn[4];
  xload(xr, 0);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c * GKC < D) {
#pragma unroll
      for (int u = 0; u < PY4; ++u) {
        const int i = tid + u * 512, r = i >> 3, kq = i & 7, col = r ^ (4 * kq);
        const bool rv = jt4 * GSX_COLS + r < ny;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = c * GKC + 4 * kq + q; if (k < D) Ys[(size_t)k * GSX_COLS + col] = rv ? yr[c][u][q] : 0.f; }
      }
    }
  }
  IL_TL(0, 1);
  if (MODE == 0) {   // block_sum's arithmetic (wave sums, then the waves' partials in wave order) with its barrier folded into the one the operands need
    sx = wave_sum(sx);
    if (!vs_self) sy = wave_sum(sy);
    if (lane == 0) { red[wave] = sx; red[8 + wave] = sy; }
  }
  __syncthreads();
  if (MODE == 0) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s0 += red[i]; s1 += red[8 + i]; }
    sx = s0; sy = vs_self ? s0 : s1;
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int j = jt4 * GSX_COLS + lane * 4 + b; wv[b] = j < ny ? wv[b] / sy : 0.f; }
  }
  IL_TL(0, 2); IL_TLC(1, 2);
  f32x2 acc2[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a) { acc2[a][0] = f32x2{0.f, 0.f}; acc2[a][1] = f32x2{0.f, 0.f}; }
  {
    // One phase per group of eight features: its column operands (8 ds_read_b128) and row operands (4 s_load_dwordx8) are requested a whole phase - 128 packed
    // instructions, 512 clocks of this wave's issue - ahead and pinned in registers before the phase starts. Why so far ahead: the rows are streamed once, so every scalar
    // load misses the scalar cache and comes from L2 (~600 clocks); and scalar loads return out of order, so while one is outstanding every LDS wait the compiler emits
    // is lgkmcnt(0) - the only wait of a phase therefore sits at its start, behind requests that are a phase old. (Round-5 timeline with phases of four features: the
    // second wave of each SIMD finished 4.5 us after the first: alone on the SIMD it issued a packed instruction every 15 clocks, stalled on these loads.)
    f32x4 y0[8], y1[8];
    auto lds8 = [&](f32x4* yg, int kb) {   // features kb .. kb + 7 (kb % 8 == 0: two swizzle groups)
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int sw = (((kb + u) >> 2) & 7) << 2; yg[u] = *reinterpret_cast<const f32x4*>(&Ys[(size_t)(kb + u) * GSX_COLS + ((4 * lane) ^ sw)]); }
    };
#define GSX_FMA8(X8, YG)                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                     \
      const f32x2 y01 = {YG[u][0], YG[u][1]}, y23 = {YG[u][2], YG[u][3]};                                               \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                                   \
        const float xs = X8[a][u];   /* (a v_mov into a vector register first: measured slower, 13.7 vs 12.1 us for the loop) */               \
        const f32x2 xa = {xs, xs};                                                                                      \
        const f32x2 d0 = xa - y01, d1 = xa - y23;                                                                       \
        acc2[a][0] = __builtin_elementwise_fma(d0, d0, acc2[a][0]);                                                     \
        acc2[a][1] = __builtin_elementwise_fma(d1, d1, acc2[a][1]);                                                     \
      }                                                                                                                 \
    }
    lds8(y0, 0);
#pragma unroll 1
    for (int k0 = 0; k0 < D; k0 += 16) {
      // (register pins: a phase's operands have landed BEFORE the next requests go out - behind them the wait would also drain those)
      asm volatile("" :: "v"(y0[0]), "v"(y0[1]), "v"(y0[2]), "v"(y0[3]), "v"(y0[4]), "v"(y0[5]), "v"(y0[6]), "v"(y0[7]));
      __builtin_amdgcn_sched_barrier(0);
      xload(xn, min(k0 + 8, D - 8));
      lds8(y1, min(k0 + 8, D - 8));
      __builtin_amdgcn_sched_barrier(0);
      GSX_FMA8(xr, y0)
      __builtin_amdgcn_sched_barrier(0);
      if (k0 + 8 >= D) break;   // (D / 8 odd: the second phase of the last trip does not exist)
      asm volatile("" :: "v"(y1[0]), "v"(y1[1]), "v"(y1[2]), "v"(y1[3]), "v"(y1[4]), "v"(y1[5]), "v"(y1[6]), "v"(y1[7]));
      __builtin_amdgcn_sched_barrier(0);
      xload(xr, min(k0 + 16, D - 8));
      lds8(y0, min(k0 + 16, D - 8));
      __builtin_amdgcn_sched_barrier(0);
      GSX_FMA8(xn, y1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef GSX_FMA8
  }
  IL_TL(0, 3); IL_TLC(1, 3);
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) { acc[a][0] = acc2[a][0][0]; acc[a][1] = acc2[a][0][1]; acc[a][2] = acc2[a][1][0]; acc[a][3] = acc2[a][1][1]; }
  const float fD = (float)D;
  const GmmilExp gex = gmmil_exp_consts(g1, g2, D);
  const int row0 = it * GSX_ROWS + wave * 4;
  if (MODE == 1) {
    const int n2e = vs_self ? n1 : n2;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = row0 + a, j = jt4 * GSX_COLS + lane * 4 + b;
        if (i < n1 && j < n2e) dist_out[(size_t)i * n2e + j] = acc[a][b] / fD;
      }
    IL_ST_END(IL_ST_GMMIL);
    return;
  }
  // The workgroup's 32 rows x 4 column tiles of partial row sums are collected in LDS and handed over by one wave (below).
  float* ps = red + 64;   // [4 tiles][32 rows]
  const int g16 = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) s += wv[b] * gmmil_pair_kernel(acc[a][b], gex);
    s = group16_sum(s);
    if ((lane & 15) == 0) ps[g16 * GSX_ROWS + wave * 4 + a] = s;
  }
  IL_TL(0, 4);
  __syncthreads();
  IL_TL(0, 5);
  if (tid < 64) {   // 4 tiles x 16 pairs of floats, handed over as agent-scope atomic exchanges: they execute at the memory side like the arrival ticket (0.5 us there and
    // back on the timeline), where write-through stores of the same 512 bytes took 4.8 us to drain - per workgroup, with 256 workgroups arriving together
    const int tl = tid >> 4, l16 = tid & 15, q = jt4 * 4 + tl;
    if (q * GT < npy) {
      const unsigned long long v = *reinterpret_cast<const unsigned long long*>(ps + tl * GSX_ROWS + 2 * l16);
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(ws_ + w.part + (((int64_t)mat * w.njt + q) * w.b1p + it * GSX_ROWS + 2 * l16));
      (void)__hip_atomic_exchange(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!out_r) { IL_ST_END(IL_ST_GMMIL); return; }
  unsigned* lastp = reinterpret_cast<unsigned*>(red + 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  IL_TL(0, 6);
  if (tid == 0) {
    const unsigned expect = (unsigned)((w.b2p + GSX_COLS - 1) / GSX_COLS + (w.b1p + GSX_COLS - 1) / GSX_COLS);
    unsigned* ctr = reinterpret_cast<unsigned*>(ws_ + w.ctr) + it * GCTR;
    const unsigned last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == expect;
    if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next call
    *lastp = last;
  }
  __syncthreads();
  IL_TL(0, 7);
  const bool last = *lastp != 0u;
  const int i = it * GSX_ROWS + tid;
  if (last && tid < GSX_ROWS && i < n1) {
    auto ordered_sum = [&](const float* p, int nq) {
      float s = 0.f;
      for (int q0 = 0; q0 < nq; q0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sload1(p, (int64_t)min(q0 + u, nq - 1) * w.b1p + i);
#pragma unroll
        for (int u = 0; u < 16; ++u) if (q0 + u < nq) s += v[u];
      }
      return s;
    };
    const float s0 = ordered_sum(ws_ + w.part, w.b2p / GT);
    const float s1 = ordered_sum(ws_ + w.part + (size_t)w.njt * w.b1p, w.b1p / GT);
    const float wi = pol.weights[(size_t)i * pol.ld_weights] / sx;
    const float sim = wi * s0, self = wi * s1;
    out_r[i] = sim - self;
    if (out_sim) out_sim[i] = sim;
    if (out_self) out_self[i] = self;
  }
  IL_ST_END(IL_ST_GMMIL);
}
static int gmmil_sx_on(int S, int A, int D, int state_only, int lanes) {   // IL_GMMIL_SX=0: k_gmmil_resident (developer A/B; same bits)
  static const int on = [] { const char* e = getenv("IL_GMMIL_SX"); return e && e[0] == '0' ? 0 : 1; }();
  return on != 0 && lanes && D >= 8 && D % 8 == 0 && S % 8 == 0 && gmmil_sx_lds(D) <= (size_t)160 * 1024 && D <= 5 * GKC;
}
// ---------------------------------------------------------------------------------------------
// k_gmmil_mfma (round 6): the pair distances as a CENTRED Gram product on the matrix pipes - the default reward launch for D <= 128.
//   ssq(x, y) = |x - c|^2 + |y - c|^2 - 2 (x - c).(y - c),   c = the mean of 8 rows spread over the expert batch, the same for every operand of a workgroup.
// Why this is allowed now: the plain |x|^2 + |y|^2 - 2 x.y form loses the digits the data's OFFSET takes (observations around 50: rewards off by 7e-8 against a bound of
// 1e-8; around 1000: 2e-5), which is why rounds 1-5 stayed on the direct difference form - three VALU flops per pair-feature at the packed instructions' issue rate,
// 31 % of fp32 at best. Centred, every term is of the size of the data's SPREAD, of which the median pair distance - the kernel's own length scale,
// gamma = 1 / median - is a fixed multiple: the exponent's absolute error is ~1e-7 whatever the offset, and the rewards are as close to float64 as the direct form's
// (measured on the B = 1024 Ant case at offsets 0 / 50 / 1000: 2.0e-10 / 2.3e-10 / 2.6e-10 against 2.6e-10 / 2.3e-10 / 2.1e-10 for the direct form; bound 1.0e-8;
// tests/test_gmmil_centred_form.py keeps that experiment). The distance matrix the bandwidths' medians are taken from (il_gmmil_sqdist, first call only) stays on the
// direct form. A workgroup = 4 waves = 64 rows x 128 columns; wave w keeps its 16 rows' centred features in registers as MFMA A fragments (one 16-byte lane per 16
// features: lane l holds features 16 q + 4 (l >> 4) .. + 3 of row l & 15, so one register quad feeds four v_mfma_f32_16x16x4_f32 and A and B agree on the k order), the
// 128 columns' centred features sit in LDS row-major with a 4-float skew (16 lanes' 16-byte reads hit 64 distinct banks), two column tiles are accumulated side by
// side (independent MFMA chains) while the previous two tiles' exponentials run on the VALU. 2 flop per pair-feature on the pipe that delivers them: 0.54 GFLOP at
// B = 1024, D = 120 (padded to 128) against 0.755 on the VALU. Partial row sums, arrival ticket and the last arriver's block-ordered sums as in k_gmmil_sx.
// ---------------------------------------------------------------------------------------------
#define GMF_ROWS 64
template <int NKQ, int COLS> struct GmfLds {   // floats
  static constexpr int DP = 16 * NKQ, LD = DP + 4;
  static constexpr int ys = 0, cs = ys + COLS * LD, cpart = cs + DP, nyh = cpart + 8 * DP, wys = nyh + 256, nxs = wys + COLS, red = nxs + GMF_ROWS, ps = red + 64, total = ps + GMF_ROWS;
};
// features k .. k + 3 (k % 4 == 0) of row r of the concatenated [states | actions] batch as ONE branch-free request per 16-byte lane (LANES: whole lanes along the rows, S and A
// multiples of 4) or four dword requests: the side of the concatenation is a select on the ADDRESS (a select on the loaded value makes hipcc branch around each load and wait
// for it - the first build of this kernel spent 4.7 us requesting its operands one after the other); beyond D the address is clamped and the caller zeroes the value.
template <bool LANES>
__device__ __forceinline__ f32x4 gmf_load(const il_batch& b, int S, int D, int r, int k) {
  if (LANES) {
    const int kc = min(k, D - 4);
    const float* p = kc < S ? b.states + ((size_t)r * b.ld_states + kc) : b.actions + ((size_t)r * b.ld_actions + (kc - S));
    return gload4(p);
  }
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kc = min(k + q, D - 1);
    const float* p = kc < S ? b.states + ((size_t)r * b.ld_states + kc) : b.actions + ((size_t)r * b.ld_actions + (kc - S));
    v[q] = gload(p);
  }
  return v;
}
// a thread's share of a weight column's sum: the first 1,024 rows as four independent requests issued with the operands (gmf_weight_req; a plain strided loop waits for
// each load before it issues the next - and for everything requested before it), longer columns in a loop behind them
struct GmfW4 { float a[4]; };
__device__ __forceinline__ GmfW4 gmf_weight_req(const il_batch& b, int tid) {
  GmfW4 r;
#pragma unroll
  for (int u = 0; u < 4; ++u) r.a[u] = gload(b.weights + (size_t)min(tid + 256 * u, b.n - 1) * b.ld_weights);
  return r;
}
__device__ __forceinline__ float gmf_weight_share(const il_batch& b, int tid, const GmfW4& first) {
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) s += (tid + 256 * u < b.n) ? first.a[u] : 0.f;
  for (int i = 1024 + tid; i < b.n; i += 256) s += gload(b.weights + (size_t)i * b.ld_weights);
  return s;
}
template <int NKQ, bool LANES, int COLS>
__global__ __launch_bounds__(256) void k_gmmil_mfma(il_batch pol, il_batch exp, int S, int D, float g1, float g2, float* __restrict__ ws_,
                                                    float* __restrict__ out_r, float* __restrict__ out_sim, float* __restrict__ out_self) {
  using L = GmfLds<NKQ, COLS>;
  constexpr int DP = L::DP, LD = L::LD, NQ = DP / 4, YP = 256 / COLS, HQ = NQ / YP;   // a column's features are split over YP threads, HQ 16-byte lanes each
  extern __shared__ __attribute__((aligned(16))) float smem[];
  IL_ST_BEGIN(IL_ST_GMMIL);
  IL_TL(2, 0);
  float* Ys = smem + L::ys; float* cs = smem + L::cs; float* cpart = smem + L::cpart; float* nyh = smem + L::nyh; float* wys = smem + L::wys;
  float* nxs = smem + L::nxs; float* red = smem + L::red; float* ps = smem + L::ps;
  globalize(pol); globalize(exp);
  const int n1 = pol.n, n2 = exp.n;
  const GmmilWs w = gmmil_ws(n1, n2, D);
  const int nE = (w.b2p + COLS - 1) / COLS, nX = (w.b1p + COLS - 1) / COLS;
  // workgroup -> (row block, column block): consecutive workgroup ids go to the eight XCDs in turn, so the 32 workgroups an XCD hosts of every 256 are given an 8 x 4 patch
  // of the block grid - its L2 then fetches 8 row blocks + 4 column blocks instead of 2 row blocks + every column block (1 MB -> 0.5 MB per XCD at B = 1024)
  const int nI = w.b1p / GMF_ROWS, nJ = nE + nX;
  int it = (int)blockIdx.x % nI, jy = (int)blockIdx.x / nI;
  if (nI % 8 == 0 && nJ % 4 == 0 && ((nI / 8) * (nJ / 4)) % 8 == 0) {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, patch = (slot >> 5) * 8 + xcd, in = slot & 31;
    it = (patch % (nI / 8)) * 8 + (in & 7); jy = (patch / (nI / 8)) * 4 + (in >> 3);
  }
  const int mat = jy >= nE ? 1 : 0, jb = jy - (mat ? nE : 0);   // mat 0: policy vs expert, 1: policy vs policy
  const bool vs_self = mat == 1;
  const il_batch& yb = vs_self ? pol : exp;
  const int ny = yb.n, tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto zero_tail = [&](f32x4 v, int k, bool row_ok) {   // features >= D (and rows beyond the batch) are exact zeros in both operands; whole lanes: a quad is inside D or outside
    if (LANES) { const bool ok = row_ok && k < D; return f32x4{ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f}; }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (row_ok && k + e < D) ? v[e] : 0.f;
    return v;
  };
  auto sumsq = [](float acc, const f32x4& v) { return __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], acc)))); };
  // ---- everything this workgroup reads from memory is requested here
  f32x4 cv = {0.f, 0.f, 0.f, 0.f};   // centre: 8 expert rows spread over the batch (thread group cg reads row cg n2 / 8), one 16-byte feature lane per thread
  const int cq = tid & 31, cg = tid >> 5;
  if (cq < NQ) cv = gmf_load<LANES>(exp, S, D, (int)(((long long)cg * n2) >> 3), 4 * cq);
  const GmfW4 wx4 = gmf_weight_req(pol, tid), wy4 = gmf_weight_req(yb, tid);
  float wv = gload(yb.weights + (size_t)min(jb * COLS + (tid & (COLS - 1)), ny - 1) * yb.ld_weights);
  wv = jb * COLS + (tid & (COLS - 1)) < ny ? wv : 0.f;
  const int ycol = tid & (COLS - 1), yhalf = tid / COLS, yrow = jb * COLS + ycol;
  f32x4 yq[HQ];
#pragma unroll
  for (int u = 0; u < HQ; ++u) yq[u] = gmf_load<LANES>(yb, S, D, min(yrow, ny - 1), 4 * (yhalf * HQ + u));
  const int xrow = it * GMF_ROWS + wave * 16 + l16;
  f32x4 xq[NKQ];
#pragma unroll
  for (int q = 0; q < NKQ; ++q) xq[q] = gmf_load<LANES>(pol, S, D, min(xrow, n1 - 1), 16 * q + 4 * g);
  // ---- centre and weight sums
  float sx = gmf_weight_share(pol, tid, wx4), sy = 0.f;
  if (!vs_self) sy = gmf_weight_share(yb, tid, wy4);
  if (cq < NQ) *reinterpret_cast<f32x4*>(&cpart[cg * DP + 4 * cq]) = zero_tail(cv, 4 * cq, true);
  sx = wave_sum(sx);
  if (!vs_self) sy = wave_sum(sy);
  if (lane == 0) { red[wave] = sx; red[4 + wave] = sy; }
  IL_TL(2, 1);
  __syncthreads();
  if (tid < NQ) {
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) c4 += *reinterpret_cast<const f32x4*>(&cpart[u * DP + 4 * tid]);
    *reinterpret_cast<f32x4*>(&cs[4 * tid]) = c4 * (1.f / 8.f);
  }
  {
    const float s0 = (red[0] + red[1]) + (red[2] + red[3]), s1 = (red[4] + red[5]) + (red[6] + red[7]);
    sx = s0; sy = vs_self ? s0 : s1;
  }
  __syncthreads();
  IL_TL(2, 2);
  // ---- centred operands: the columns into LDS with their squared norms, the rows in registers (the centre's lanes are read in one batch: a read per quad, each waited
  // for, was an LDS round trip per quad)
  {
    f32x4 cy[HQ], cx[NKQ];
#pragma unroll
    for (int u = 0; u < HQ; ++u) cy[u] = *reinterpret_cast<const f32x4*>(&cs[4 * (yhalf * HQ + u)]);
#pragma unroll
    for (int q = 0; q < NKQ; ++q) cx[q] = *reinterpret_cast<const f32x4*>(&cs[16 * q + 4 * g]);
    __builtin_amdgcn_sched_barrier(0);
    float nyp = 0.f;
    const bool ok = yrow < ny;
#pragma unroll
    for (int u = 0; u < HQ; ++u) {
      const int k = 4 * (yhalf * HQ + u);
      const f32x4 v = zero_tail(yq[u] - cy[u], k, ok);
      nyp = sumsq(nyp, v);
      *reinterpret_cast<f32x4*>(&Ys[ycol * LD + k]) = v;
    }
    nyh[yhalf * COLS + ycol] = nyp;
    if (tid < COLS) wys[tid] = wv / sy;
    float nxp = 0.f;
#pragma unroll
    for (int q = 0; q < NKQ; ++q) {
      xq[q] = zero_tail(xq[q] - cx[q], 16 * q + 4 * g, true);
      nxp = sumsq(nxp, xq[q]);
    }
    nxp += __shfl_xor(nxp, 16);
    nxp += __shfl_xor(nxp, 32);
    if (g == 0) nxs[wave * 16 + l16] = nxp;
  }
  IL_TL(2, 3);
  __syncthreads();
  IL_TL(2, 4);
  // ---- the Gram tiles: two column tiles per trip on independent accumulators; a trip's exponentials are written after the next trip's MFMAs
  const f32x4 nx4 = *reinterpret_cast<const f32x4*>(&nxs[wave * 16 + 4 * g]);
  const GmmilExp gex = gmmil_exp_consts(g1, g2, D);
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  auto gram2 = [&](int ct, f32x4& acc0, f32x4& acc1) {
    const float* y0 = Ys + (ct * 16 + l16) * LD + 4 * g;
    const float* y1 = y0 + 16 * LD;
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NKQ; ++q) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(y0 + 16 * q), b1 = *reinterpret_cast<const f32x4*>(y1 + 16 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc0 = mfma16(xq[q][e], b0[e], acc0); acc1 = mfma16(xq[q][e], b1[e], acc1); }
    }
  };
  auto finish = [&](int ct, const f32x4& acc) {   // lane: rows 4 g + r of the wave's 16, column 16 ct + l16 of the workgroup's 128
    const int c = ct * 16 + l16;
    float nyv = nyh[c];
#pragma unroll
    for (int q = 1; q < YP; ++q) nyv += nyh[q * COLS + c];
    const float wyv = wys[c];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ssq = fmaxf(__builtin_fmaf(-2.f, acc[r], nx4[r] + nyv), 0.f);
      rs[r] = __builtin_fmaf(wyv, gmmil_pair_kernel(ssq, gex), rs[r]);
    }
  };
  f32x4 a0, a1, p0, p1;
  gram2(0, p0, p1);
#pragma unroll
  for (int ct = 2; ct < COLS / 16; ct += 2) {
    gram2(ct, a0, a1);
    finish(ct - 2, p0); finish(ct - 1, p1);
    p0 = a0; p1 = a1;
  }
  finish(COLS / 16 - 2, p0); finish(COLS / 16 - 1, p1);
  IL_TL(2, 5);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s = group16_sum(rs[r]);
    if (l16 == 0) ps[wave * 16 + 4 * g + r] = s;
  }
  __syncthreads();
  if (tid < GMF_ROWS / 2) {   // the workgroup's 64 partial row sums leave as agent-scope atomic exchanges (executed at the memory side, like the arrival ticket)
    const unsigned long long v = *reinterpret_cast<const unsigned long long*>(ps + 2 * tid);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(ws_ + w.part2 + ((int64_t)jy * w.b1p + it * GMF_ROWS + 2 * tid));
    (void)__hip_atomic_exchange(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned* lastp = reinterpret_cast<unsigned*>(red + 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  IL_TL(2, 6);
  if (tid == 0) {
    unsigned* ctr = reinterpret_cast<unsigned*>(ws_ + w.ctr) + (2 * it) * GCTR;
    const unsigned last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)(nE + nX);
    if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next call
    *lastp = last;
  }
  __syncthreads();
  IL_TL(2, 7);
  const int i = it * GMF_ROWS + tid;
  if (*lastp != 0u && tid < GMF_ROWS && i < n1) {
    auto ordered_sum = [&](const float* p, int nq) {
      float s = 0.f;
      for (int q0 = 0; q0 < nq; q0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sload1(p, (int64_t)min(q0 + u, nq - 1) * w.b1p + i);
#pragma unroll
        for (int u = 0; u < 16; ++u) if (q0 + u < nq) s += v[u];
      }
      return s;
    };
    const float* p0 = ws_ + w.part2; const float* p1 = p0 + (size_t)nE * w.b1p;
    float s0, s1;
    if (nE <= 8 && nX <= 8) {   // both matrices' partials requested together: one trip to memory instead of two behind each other
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v0[u] = sload1(p0, (int64_t)min(u, nE - 1) * w.b1p + i); v1[u] = sload1(p1, (int64_t)min(u, nX - 1) * w.b1p + i); }
      s0 = 0.f; s1 = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) { if (u < nE) s0 += v0[u]; if (u < nX) s1 += v1[u]; }
    } else {
      s0 = ordered_sum(p0, nE);
      s1 = ordered_sum(p1, nX);
    }
    const float wi = pol.weights[(size_t)i * pol.ld_weights] / sx;
    const float sim = wi * s0, self = wi * s1;
    out_r[i] = sim - self;
    if (out_sim) out_sim[i] = sim;
    if (out_self) out_self[i] = self;
  }
  IL_ST_END(IL_ST_GMMIL);
}
static int gmmil_mfma_on(int D) {   // IL_GMMIL_MFMA=0: the direct-difference launches (k_gmmil_sx and the forms behind it)
  static const int on = [] { const char* e = getenv("IL_GMMIL_MFMA"); return e && e[0] == '0' ? 0 : 1; }();
  return on != 0 && D >= 1 && D <= 128;
}
template <int NKQ, bool LANES, int COLS>
static int gmmil_mfma_launch_(const il_batch* pol, const il_batch* exp, int S, int D, float g1, float g2, float* workspace, float* out_r, float* out_sim, float* out_self, hipStream_t st) {
  const GmmilWs w = gmmil_ws(pol->n, exp->n, D);
  const size_t lds = (size_t)GmfLds<NKQ, COLS>::total * sizeof(float);
  auto kern = k_gmmil_mfma<NKQ, LANES, COLS>;   // (one identifier: the host emulator's launch macro splits its arguments at commas)
  if (int rc = gmmil_ensure_lds(kern, lds)) return rc;
  const int nE = (w.b2p + COLS - 1) / COLS, nX = (w.b1p + COLS - 1) / COLS;
  IL_TRACE("k_gmmil_tile", st);
  kern<<<dim3((w.b1p / GMF_ROWS) * (nE + nX), 1, 1), 256, lds, st>>>(*pol, *exp, S, D, g1, g2, workspace, out_r, out_sim, out_self);
  IL_CHECK_LAUNCH("il_gmmil_reward");
  return IL_OK;
}
// column block: 128, halved while the grid stays at <= 128 workgroups (B = 1024: 128 -> 256 workgroups; B = 512: 32 -> 256; B = 256: 32 -> 64). IL_GMMIL_COLS=32|64|128 forces one.
static int gmmil_mfma_cols(const il_batch* pol, const il_batch* exp, int D) {
  static const int forced = [] { const char* e = getenv("IL_GMMIL_COLS"); const int v = e ? atoi(e) : 0; return (v == 32 || v == 64 || v == 128) ? v : 0; }();
  if (forced) return forced;
  const GmmilWs w = gmmil_ws(pol->n, exp->n, D);
  int cols = 128;
  while (cols > 32 && (w.b1p / GMF_ROWS) * ((w.b2p + cols / 2 - 1) / (cols / 2) + (w.b1p + cols / 2 - 1) / (cols / 2)) <= 256 &&
         (w.b1p / GMF_ROWS) * ((w.b2p + cols - 1) / cols + (w.b1p + cols - 1) / cols) <= 128) cols >>= 1;
  return cols;
}
template <int NKQ, bool LANES>
static int gmmil_mfma_launch_l(const il_batch* pol, const il_batch* exp, int S, int D, float g1, float g2, float* workspace, float* out_r, float* out_sim, float* out_self, hipStream_t st) {
  switch (gmmil_mfma_cols(pol, exp, D)) {
    case 32: return gmmil_mfma_launch_<NKQ, LANES, 32>(pol, exp, S, D, g1, g2, workspace, out_r, out_sim, out_self, st);
    case 64: return gmmil_mfma_launch_<NKQ, LANES, 64>(pol, exp, S, D, g1, g2, workspace, out_r, out_sim, out_self, st);
    default: return gmmil_mfma_launch_<NKQ, LANES, 128>(pol, exp, S, D, g1, g2, workspace, out_r, out_sim, out_self, st);
  }
}
template <int NKQ>
static int gmmil_mfma_launch(const il_batch* pol, const il_batch* exp, int S, int D, float g1, float g2, float* workspace, float* out_r, float* out_sim, float* out_self, int lanes, hipStream_t st) {
  return lanes && D >= 4 ? gmmil_mfma_launch_l<NKQ, true>(pol, exp, S, D, g1, g2, workspace, out_r, out_sim, out_self, st)
                         : gmmil_mfma_launch_l<NKQ, false>(pol, exp, S, D, g1, g2, workspace, out_r, out_sim, out_self, st);
}
static bool gmmil_direct() { static const int on = [] { const char* e = getenv("IL_GMMIL_DIRECT"); return e && e[0] == '0' ? 0 : 1; }(); return on != 0; }   // IL_GMMIL_DIRECT=0: k_gmmil_pack + k_gmmil_tile (developer A/B; same bits)
static int gmmil_lanes(const il_batch* a, const il_batch* b, int S, int A, int state_only) {   // whole 16-byte lanes along the rows of both batches?
  auto ok = [&](const il_batch* x) {
    const bool st = (reinterpret_cast<uintptr_t>(x->states) & 15) == 0 && x->ld_states % 4 == 0 && S % 4 == 0;
    const bool ac = state_only || ((reinterpret_cast<uintptr_t>(x->actions) & 15) == 0 && x->ld_actions % 4 == 0 && A % 4 == 0);
    return st && ac;
  };
  return ok(a) && ok(b) ? 1 : 0;
}

extern "C" int il_gmmil_reward(const il_batch* pol, const il_batch* exp, int32_t S, int32_t A, int32_t state_only, float g1, float g2, float* out_rewards,
                               float* out_sim, float* out_self, float* workspace, int64_t workspace_floats, il_stream_t stream_) {
  IL_NO_GATHER(pol, "il_gmmil_reward"); IL_NO_GATHER(exp, "il_gmmil_reward");
  IL_CHECK_ARG(pol && exp && out_rewards && workspace && pol->n > 0 && exp->n > 0, "il_gmmil_reward: bad arguments");
  const int D = S + (state_only ? 0 : A);
  const GmmilWs w = gmmil_ws(pol->n, exp->n, D);
  if (workspace_floats < w.total) return il_set_error(IL_ERR_WORKSPACE, "il_gmmil_reward: workspace too small (%lld < %lld floats)", (long long)workspace_floats, (long long)w.total);
  hipStream_t st = (hipStream_t)stream_;
  if (gmmil_mfma_on(D)) {
    const int lanes = gmmil_lanes(pol, exp, S, A, state_only);
    if (D <= 32) return gmmil_mfma_launch<2>(pol, exp, S, D, g1, g2, workspace, out_rewards, out_sim, out_self, lanes, st);
    if (D <= 64) return gmmil_mfma_launch<4>(pol, exp, S, D, g1, g2, workspace, out_rewards, out_sim, out_self, lanes, st);
    return gmmil_mfma_launch<8>(pol, exp, S, D, g1, g2, workspace, out_rewards, out_sim, out_self, lanes, st);
  }
  if (gmmil_direct() && gmmil_sx_on(S, A, D, state_only, gmmil_lanes(pol, exp, S, A, state_only))) {
    const size_t lds = gmmil_sx_lds(D);
    if (int rc = gmmil_ensure_lds(k_gmmil_sx<0>, lds)) return rc;
    IL_TRACE("k_gmmil_tile", st);
    k_gmmil_sx<0><<<dim3(w.b1p / GSX_ROWS, (w.njt * GT + GSX_COLS - 1) / GSX_COLS, 2), 512, lds, st>>>(*pol, *exp, S, D, g1, g2, workspace, nullptr, 0, out_rewards, out_sim, out_self);
    IL_CHECK_LAUNCH("il_gmmil_reward");
    return IL_OK;
  }
  if (gmmil_direct() && gmmil_resident_on(D)) {
    const size_t lds = gmmil_resident_lds(D);
    if (int rc = gmmil_ensure_lds(k_gmmil_resident<0>, lds)) return rc;
    IL_TRACE("k_gmmil_tile", st);
    k_gmmil_resident<0><<<dim3(w.b1p / GTR, w.njt, 2), 256, lds, st>>>(*pol, *exp, S, D, g1, g2, workspace, nullptr, 0, out_rewards, out_sim, out_self, gmmil_lanes(pol, exp, S, A, state_only));
    IL_CHECK_LAUNCH("il_gmmil_reward");
    return IL_OK;
  }
  if (gmmil_direct() && D >= 4) {
    IL_TRACE("k_gmmil_tile", st);
    k_gmmil_direct<0><<<dim3(w.b1p / GTR, w.njt, 2), 256, 0, st>>>(*pol, *exp, S, D, g1, g2, workspace, nullptr, 0, out_rewards, out_sim, out_self, gmmil_lanes(pol, exp, S, A, state_only));
    IL_CHECK_LAUNCH("il_gmmil_reward");
    return IL_OK;
  }
  { IL_TRACE("k_gmmil_pack", st); k_gmmil_pack<<<dim3(w.b1p / GT + w.b2p / GT, (D + GKC - 1) / GKC), 256, 0, st>>>(*pol, *exp, S, D, workspace); }
  { IL_TRACE("k_gmmil_tile", st); k_gmmil_tile<0><<<dim3(w.b1p / GTR, w.njt, 2), 256, 0, st>>>(pol->n, exp->n, D, g1, g2, workspace, nullptr, 0, out_rewards, out_sim, out_self); }
  IL_CHECK_LAUNCH("il_gmmil_reward");
  return IL_OK;
}

// distance matrix between a and b ([na][nb]); needs a workspace of il_gmmil_workspace_floats(na, nb, D) floats appended after `out`?
// No: to keep the ABI allocation-free the caller passes the same kind of workspace as for il_gmmil_reward.
extern "C" int il_gmmil_sqdist(const il_batch* a, const il_batch* b, int32_t S, int32_t A, int32_t state_only, float* out, float* workspace,
                                  int64_t workspace_floats, il_stream_t stream_) {
  IL_NO_GATHER(a, "il_gmmil_sqdist"); IL_NO_GATHER(b, "il_gmmil_sqdist");
  IL_CHECK_ARG(a && b && out && workspace && a->n > 0 && b->n > 0, "il_gmmil_sqdist: bad arguments");
  const int D = S + (state_only ? 0 : A);
  const GmmilWs w = gmmil_ws(a->n, b->n, D);
  if (workspace_floats < w.total) return il_set_error(IL_ERR_WORKSPACE, "il_gmmil_sqdist: workspace too small");
  hipStream_t st = (hipStream_t)stream_;
  if (gmmil_direct() && gmmil_sx_on(S, A, D, state_only, gmmil_lanes(a, b, S, A, state_only))) {
    const size_t lds = gmmil_sx_lds(D);
    if (int rc = gmmil_ensure_lds(k_gmmil_sx<1>, lds)) return rc;
    IL_TRACE("k_gmmil_tile", st);
    k_gmmil_sx<1><<<dim3(w.b1p / GSX_ROWS, (w.b2p + GSX_COLS - 1) / GSX_COLS, 1), 512, lds, st>>>(*a, *b, S, D, 0.f, 0.f, workspace, out, 0, nullptr, nullptr, nullptr);
    IL_CHECK_LAUNCH("il_gmmil_sqdist");
    return IL_OK;
  }
  if (gmmil_direct() && gmmil_resident_on(D)) {
    const size_t lds = gmmil_resident_lds(D);
    if (int rc = gmmil_ensure_lds(k_gmmil_resident<1>, lds)) return rc;
    IL_TRACE("k_gmmil_tile", st);
    k_gmmil_resident<1><<<dim3(w.b1p / GTR, w.b2p / GT, 1), 256, lds, st>>>(*a, *b, S, D, 0.f, 0.f, workspace, out, 0, nullptr, nullptr, nullptr, gmmil_lanes(a, b, S, A, state_only));
    IL_CHECK_LAUNCH("il_gmmil_sqdist");
    return IL_OK;
  }
  if (gmmil_direct() && D >= 4) {
    IL_TRACE("k_gmmil_tile", st);
    k_gmmil_direct<1><<<dim3(w.b1p / GTR, w.b2p / GT, 1), 256, 0, st>>>(*a, *b, S, D, 0.f, 0.f, workspace, out, 0, nullptr, nullptr, nullptr, gmmil_lanes(a, b, S, A, state_only));
    IL_CHECK_LAUNCH("il_gmmil_sqdist");
    return IL_OK;
  }
  { IL_TRACE("k_gmmil_pack", st); k_gmmil_pack<<<dim3(w.b1p / GT + w.b2p / GT, (D + GKC - 1) / GKC), 256, 0, st>>>(*a, *b, S, D, workspace); }
  { IL_TRACE("k_gmmil_tile", st); k_gmmil_tile<1><<<dim3(w.b1p / GTR, w.b2p / GT, 1), 256, 0, st>>>(a->n, b->n, D, 0.f, 0.f, workspace, out, 0, nullptr, nullptr, nullptr); }
  IL_CHECK_LAUNCH("il_gmmil_sqdist");
  return IL_OK;
}

IL_STAMP_READER(il_debug_stamps_gmmil)
IL_ST_READER(il_stamps_gmmil)
IL_TL_READER(il_debug_timeline_gmmil)
