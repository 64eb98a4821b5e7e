// PWIL greedy Wasserstein-coupling reward (reference models.py:216-249) for gfx950.
//
// Two paths. Small coupling sets (the usual case: at most 256 atoms can be consumed per step) take k_pwil_select + k_pwil_merge, see below.
// Fallback: one workgroup of 1024 threads per environment step. The reference deletes consumed rows (three O(N D) copies per
// deletion); here a consumed atom is marked by a negative weight, each thread keeps the running minimum of the atoms it
// owns (strided ownership), and one greedy iteration is a 1024-way (distance, index) arg-min -- ties resolve to the lowest
// index, which is what argmin on the reference's order-preserving shrunk tensor returns -- followed by a rescan by the
// single owner of the consumed atom. Cost / remaining weight are accumulated in double like the reference's Python floats.
#include <float.h>
#include <limits.h>

#include "il_common.hpp"

__global__ __launch_bounds__(1024) void k_pwil_reset(il_pwil d, unsigned* ticket) {
  const float w = (float)(1.0 / (double)d.n_atoms);
  if (ticket && blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;   // k_pwil_step's arrival counter (it also resets itself at the end of every step)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.n_atoms; i += gridDim.x * blockDim.x) d.weights[i] = w;
}

__device__ __forceinline__ void argmin_combine(float& dmin, int& imin, float od, int oi) {
  if (od < dmin || (od == dmin && oi < imin)) { dmin = od; imin = oi; }
}

__global__ __launch_bounds__(1024) void k_pwil_reward(il_pwil d, const float* __restrict__ state, const float* __restrict__ action, float* __restrict__ out) {
  __shared__ float z[512];
  __shared__ float wd[16];
  __shared__ int wi[16];
  const int tid = threadIdx.x, N = d.n_atoms, D = d.dim, S = d.state_dim;
  for (int k = tid; k < D; k += blockDim.x) {
    const float x = k < S ? state[k] : action[k - S];
    z[k] = d.scale[k] * (x + d.offset[k]);
  }
  __syncthreads();
  float lmin = FLT_MAX; int lidx = INT_MAX;
  for (int i = tid; i < N; i += blockDim.x) {
    float dist = FLT_MAX;
    if (d.weights[i] >= 0.f) {
      float s = 0.f;
      for (int k = 0; k < D; ++k) { const float df = d.atoms[(size_t)i * D + k] - z[k]; s += df * df; }
      dist = sqrtf(s);
      if (dist < lmin) { lmin = dist; lidx = i; }
    }
    d.dists[i] = dist;
  }
  double weight = d.agent_weight, cost = 0.0;
  for (int iter = 0; iter <= N && weight > 0.0; ++iter) {
    // ---- block arg-min
    float bd = lmin; int bi = lidx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float od = __shfl_xor(bd, o, 64); const int oi = __shfl_xor(bi, o, 64); argmin_combine(bd, bi, od, oi); }
    __syncthreads();
    if ((tid & 63) == 0) { wd[tid >> 6] = bd; wi[tid >> 6] = bi; }
    __syncthreads();
    bd = wd[0]; bi = wi[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) argmin_combine(bd, bi, wd[w], wi[w]);
    if (bi == INT_MAX) break;  // every atom consumed
    const double ew = (double)d.weights[bi], dist = (double)bd;
    __syncthreads();  // everyone has read weights[bi] before its owner rewrites it
    const bool owner = (bi % (int)blockDim.x) == tid;
    if (weight >= ew) {
      cost += ew * dist; weight -= ew;
      if (owner) {  // consume the atom and rescan the atoms this thread owns
        d.weights[bi] = -1.f; d.dists[bi] = FLT_MAX;
        lmin = FLT_MAX; lidx = INT_MAX;
        for (int i = tid; i < N; i += blockDim.x) { const float dd = d.dists[i]; if (dd < lmin) { lmin = dd; lidx = i; } }
      }
    } else {
      cost += weight * dist;
      if (owner) d.weights[bi] = (float)ew - (float)weight;
      weight = 0.0;
    }
  }
  if (tid == 0) out[0] = (float)(d.reward_scale * exp(-d.reward_bandwidth * cost));
}

// ---------------------------------------------------------------------------------------------
// Parallel path. The greedy coupling consumes atoms in ascending distance, and one step can consume at most m = ceil(agent_weight * N) + 2
// of them (every live atom weighs 1/N except one partially consumed one), so a step only needs the m smallest live distances:
//   k_pwil_select  one workgroup per 256 atoms: distances (2.4 MB of atoms at N = 25k spread over ~100 CUs instead of one), then each atom's rank
//                  inside its chunk by counting (256 broadcast LDS reads; ties by index, like argmin); ranks < K write (distance, index) to the
//                  chunk's ascending candidate list.
//   k_pwil_merge   one workgroup: G-way merge of the lists (a thread owns lists t, t + 256, ...; a round = block arg-min over the heads), consuming
//                  atoms exactly like the one-workgroup kernel: same order, same double-precision cost accumulation => same reward bit for bit.
// ---------------------------------------------------------------------------------------------
#define PW_CHUNK 256
struct __attribute__((aligned(16))) PwCand { float dist; int idx; float w; float pad; };   // w = the atom's remaining weight at selection time

// (distance, index) as one ordered 64-bit key: distances are >= 0, so their IEEE bits order like the values; ties go to the lower index (argmin).
__device__ __forceinline__ unsigned long long pw_key(float dist, int idx) { return ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)idx; }
typedef unsigned pw_u32x4 __attribute__((ext_vector_type(4)));
// Rank of this thread's key among the (up to) 256 DISTINCT keys of a 256-thread workgroup (every thread calls; one barrier inside). Each wave ranks its own 64 keys with
// v_readlane broadcasts (no LDS), leaves them in `sk` in ascending order - padded to 128 slots with the largest key, so that the binary searches need no bounds -, and a
// key's rank is its rank in its wave plus, per other wave, the number of smaller keys found by a binary search in that wave's sorted run: 64 register broadcasts + 21 LDS
// reads per thread where counting the 256 keys out of LDS took 256 broadcast reads + compares (4.9 of a 20 us step).
__device__ __forceinline__ int pw_rank256(unsigned long long key, unsigned long long (*sk)[128]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int klo = (int)(unsigned)key, khi = (int)(unsigned)(key >> 32);
  int rw = 0;
#pragma unroll 16
  for (int l = 0; l < 64; ++l) {
    const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(khi, l) << 32) | (unsigned)__builtin_amdgcn_readlane(klo, l);
    rw += o < key ? 1 : 0;
  }
  sk[wave][rw] = key; sk[wave][64 + lane] = ~0ull;
  __syncthreads();
  int rank = rw;
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    const unsigned long long* a = sk[(wave + q) & 3];
    int pos = 0;
#pragma unroll
    for (int step = 64; step > 0; step >>= 1) pos += a[pos + step - 1] < key ? step : 0;
    rank += pos;
  }
  return rank;
}
// WT: the candidates leave with sc0 sc1 (write-through) stores - k_pwil_step's arrival ticket then needs no release (an L2 write-back per workgroup), only drained stores
template <bool WT>
__device__ __forceinline__ void pwil_select_block(const il_pwil& d, const float* __restrict__ state, const float* __restrict__ action, int K, PwCand* __restrict__ cand) {
  __shared__ float z[512];
  __shared__ unsigned long long sk[4][128];   // each wave's 64 keys in ascending order, padded with the largest key (binary searches over 128 slots need no bounds)
  const int tid = threadIdx.x, N = d.n_atoms, D = d.dim, S = d.state_dim;
  IL_TL(0, 0);
  for (int k = tid; k < D; k += PW_CHUNK) {
    const float x = k < S ? state[k] : action[k - S];
    z[k] = d.scale[k] * (x + d.offset[k]);
  }
  __syncthreads();
  IL_TL(0, 1);
  const int i = (int)blockIdx.x * PW_CHUNK + tid;
  float dist = FLT_MAX;
  const int ic = i < N ? i : N - 1;
  const float wi = gload(d.weights + ic);
  if (i < N && wi >= 0.f) {   // (the row of a consumed atom is not read: over an episode half of the rows are)
    const float* a = d.atoms + (size_t)i * D;
    float s = 0.f;
    int k = 0;
    if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(d.atoms) & 15) == 0) {
      // the scalar loop below compiled to load -> wait -> fma once per feature: D dependent round trips per atom. Eight 16-byte lanes of the row are requested first
      // (addresses past the row clamp to its last lane and are not used); the adds keep their order k = 0, 1, 2, ..., so the distance keeps its bits.
      for (; k < D; k += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gload4(a + min(k + 4 * u, D - 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k + 4 * u < D) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float df = v[u][c] - z[k + 4 * u + c]; s += df * df; }
          }
      }
    }
    for (; k < D; ++k) { const float df = a[k] - z[k]; s += df * df; }
    dist = sqrtf(s);
  }
  // rank of every atom inside its chunk by (distance, index) - ties to the lower index, like argmin (the in-chunk index is the key's low word)
  const int rank = pw_rank256(pw_key(dist, tid), sk);
  IL_TL(0, 2);
  if (rank < K) {
    PwCand c; c.dist = dist; c.idx = dist < FLT_MAX ? i : INT_MAX; c.w = dist < FLT_MAX ? wi : 0.f; c.pad = 0.f;
    if (WT) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(cand, 0, 0x7ffffff0, 0x00020000);   // raw buffer, byte offsets
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pw_u32x4, c), rs, (int)(((size_t)blockIdx.x * K + rank) * sizeof(PwCand)), 0, 17);   // sc0 | sc1
    } else cand[(size_t)blockIdx.x * K + rank] = c;
  }
  IL_TL(0, 3);
}
__global__ __launch_bounds__(PW_CHUNK) void k_pwil_select(il_pwil d, const float* __restrict__ state, const float* __restrict__ action, int K, PwCand* __restrict__ cand) {
  pwil_select_block<false>(d, state, action, K, cand);
}

#define PW_LDS_CAND 4096   // candidates staged in LDS (64 KB): the merge loop then touches no global memory
#define PW_MAXQ 16         // lists per lane of the merging wave: G <= 64 * PW_MAXQ chunks (262k atoms); larger sets use the one-workgroup kernel

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {   // all 64 lanes active; result in every lane
  v = dpp_min_u64<0x128>(v); v = dpp_min_u64<0x124>(v); v = dpp_min_u64<0x122>(v); v = dpp_min_u64<0x121>(v);   // row_ror 8, 4, 2, 1: min of each 16-lane row
  unsigned long long m = v;
#pragma unroll
  for (int r = 0; r < 64; r += 16) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, r), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), r);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    m = o < m ? o : m;
  }
  return m;
}

// 256 threads stage the candidate lists in LDS, then ONE wave merges: a round is a local min over the lane's own list heads, a DPP min over
// the wave and a readlane of the winner's weight - no barrier and no LDS round trip between lanes (the 4-value __shfl_xor arg-min of a
// 256-thread version cost 1.6 us per round, i.e. 44 us for the 26 atoms of a step at N = 25k, T = 1000).
// The serial merge of one wave over the per-chunk candidate lists `cl` (instantiated once for the LDS copy and once for the lists in
// HBM: a pointer that may be either is generic, and every head fetch of this latency-bound loop would be a flat_load).
__device__ __forceinline__ void pwil_merge_wave(const il_pwil& d, int G, int K, const PwCand* __restrict__ cl, float* __restrict__ out) {
  const int lane = threadIdx.x;
  int ptr[PW_MAXQ];
#pragma unroll
  for (int q = 0; q < PW_MAXQ; ++q) ptr[q] = 0;
  const unsigned long long EMPTY = ~0ull;
  double weight = d.agent_weight, cost = 0.0;
  for (int iter = 0; iter <= d.n_atoms && weight > 0.0; ++iter) {
    unsigned long long key = EMPTY; float kw = 0.f; int kq = 0;
#pragma unroll
    for (int q = 0; q < PW_MAXQ; ++q) {
      const int l = lane + q * 64;
      if (q * 64 < G && l < G && ptr[q] < K) {
        const PwCand c = cl[(size_t)l * K + ptr[q]];
        const unsigned long long k2 = c.idx == INT_MAX ? EMPTY : pw_key(c.dist, c.idx);
        if (k2 < key) { key = k2; kw = c.w; kq = q; }
      }
    }
    const unsigned long long best = wave_min_u64(key);
    if (best == EMPTY) break;   // every atom consumed
    const int winner = __ffsll((long long)__ballot(key == best)) - 1;   // indices are unique: exactly one lane
    const double ew = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(kw), winner));   // met at most once per step: the weight read at selection time is current
    const double dist = (double)__uint_as_float((unsigned)(best >> 32));
    const int bi = (int)(unsigned)best;
    if (weight >= ew) {
      cost += ew * dist; weight -= ew;
      if (lane == 0) d.weights[bi] = -1.f;
#pragma unroll
      for (int q = 0; q < PW_MAXQ; ++q) if (lane == winner && kq == q) ptr[q] += 1;   // next head of that list (static register indices)
    } else {
      cost += weight * dist;
      if (lane == 0) d.weights[bi] = (float)ew - (float)weight;
      weight = 0.0;
    }
  }
  if (lane == 0) out[0] = (float)(d.reward_scale * exp(-d.reward_bandwidth * cost));
}

// ---------------------------------------------------------------------------------------------
// k_pwil_merge (round 3): the step's coupling WITHOUT a serial G-way merge. The greedy coupling consumes candidates in ascending (distance, index) order until the
// agent's weight 1/T is spent - so all it needs is a sorted PREFIX of the candidates whose weights add up to at least that. (i) A threshold that provably covers the
// prefix: sort the G list heads (the chunk minima) by rank counting and walk them until their cumulative weight reaches the agent's weight (plus one head of slack):
// every candidate of the prefix is <= that head's key. With distances spread over ~100 chunks this leaves ~1.2 x the consumed atoms of the G K candidates. (ii) The
// survivors are compacted into LDS (order irrelevant), (iii) ranked by counting (keys are unique: the atom index is part of the key), (iv) walked in rank order by the
// same greedy loop as before - cost and remaining weight in double, consumed in ascending key order: the same operations on the same values in the same order as the
// serial merge and the one-workgroup kernel, so the reward and the weights keep their bits - (v) the consumed atoms are marked in parallel. If the survivors run out
// before the weight does (partially consumed atoms of earlier steps are lighter than 1/N, rounding at the threshold) and candidates above the threshold exist, the step
// is redone with every candidate (pass 2: exact by construction, rare). The single wave's G-way merge this replaces took ~1 us per consumed atom (25.5 of the 35 us of
// a step at N = 25k, T = 1000). Needs G K <= PW_LDS_CAND; larger sets keep k_pwil_merge_serial.
// ---------------------------------------------------------------------------------------------
#define PW_PER_THREAD (PW_LDS_CAND / 256)
__device__ __forceinline__ void pwil_merge_block(const il_pwil& d, int G, int K, const PwCand* __restrict__ cand, float* __restrict__ out) {
  __shared__ unsigned long long skey[PW_LDS_CAND];   // survivors (unsorted); before that: [0, G) the list heads, [2048, 2048 + G) the heads in ascending order
  __shared__ float sw[PW_LDS_CAND];                  // their weights, same indexing
  __shared__ unsigned short order[PW_LDS_CAND];      // order[rank] = survivor slot
  __shared__ int nsurv, nvalid;
  const int tid = threadIdx.x, total = G * K;
  const unsigned long long EMPTY = ~0ull;
  // every candidate this thread owns, and the heads of the lists it owns, requested together: ONE round trip to the lists k_pwil_select just wrote
  PwCand c[PW_PER_THREAD];
#pragma unroll
  for (int u = 0; u < PW_PER_THREAD; ++u) c[u] = cand[min(tid + 256 * u, total - 1)];
  PwCand hd[4];   // lists tid, tid + 256, ... (G <= 1024)
#pragma unroll
  for (int q = 0; q < 4; ++q) hd[q] = cand[(size_t)min(tid + 256 * q, G - 1) * K];
  __builtin_amdgcn_sched_barrier(0);
  IL_TL(1, 0);
  if (tid == 0) { nsurv = 0; nvalid = 0; }
  if (G <= 256) {   // the usual case: the heads in ascending order through the wave-level ranking of the selecting workgroups (an exhausted list sorts last, by list index)
    const bool mine = tid < G;
    // (hd[0] = the head of list tid; threads past the last list carry distinct keys above every list's, so that each wave still fills its 64 sorted slots)
    const unsigned long long h = !mine ? (0xFFFFFFFFull << 32) | (0x80000000u | (unsigned)tid) : (hd[0].idx == INT_MAX ? (0xFFFFFFFFull << 32) | (unsigned)tid : pw_key(hd[0].dist, hd[0].idx));
    IL_TL(1, 1);   // the heads have arrived (the first use of the candidate loads)
    const int rank = pw_rank256(h, reinterpret_cast<unsigned long long(*)[128]>(skey + 1024));   // scratch between the two halves this function uses
    if (mine) { skey[2048 + rank] = hd[0].idx == INT_MAX ? EMPTY : h; sw[2048 + rank] = hd[0].w; }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int l = tid + 256 * q;
      if (l < G) { skey[l] = hd[q].idx == INT_MAX ? EMPTY : pw_key(hd[q].dist, hd[q].idx); sw[l] = hd[q].w; }
    }
    __syncthreads();
    IL_TL(1, 1);
    for (int l = tid; l < G; l += 256) {   // heads in ascending order (EMPTY heads tie: give them distinct ranks by list index, they sort last)
      const unsigned long long h = skey[l];
      int rank = 0;
#pragma unroll 8
      for (int j = 0; j < G; ++j) { const unsigned long long o = skey[j]; rank += (o < h || (o == h && j < l)) ? 1 : 0; }
      skey[2048 + rank] = h; sw[2048 + rank] = sw[l];
    }
  }
  __syncthreads();
  IL_TL(1, 2);   // heads sorted
  // The threshold: walk the sorted heads until their weights cover the agent's weight, then one more head of slack. The bound only has to be safe (pass 2 below makes the
  // step exact whatever it is), so it is a wave-parallel float prefix sum over the first 64 sorted heads - every wave computes the same value - instead of a loop of
  // dependent LDS reads (round 3, second step: that loop and the greedy loop below were ~6 us of a 25 us step).
  unsigned long long T = EMPTY;
  {
    const int lane = tid & 63;
    const bool live = lane < G && skey[2048 + lane] != EMPTY;
    float cum = live ? sw[2048 + lane] : 0.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float up = __shfl_up(cum, o, 64); if (lane >= o) cum += up; }
    const unsigned long long reach = __ballot(live && (double)cum >= d.agent_weight * 1.0001);
    if (reach) {
      const int j = __ffsll((long long)reach) - 1;
      if (j + 1 < 64 && j + 1 < G && skey[2048 + j + 1] != EMPTY) T = skey[2048 + j + 1];   // otherwise every candidate survives
    }
  }
  __syncthreads();   // the heads have been read by everyone: the survivors may overwrite them
  IL_TL(1, 3);   // threshold known
  __shared__ int again;
  for (int pass = 0; pass < 2; ++pass) {
    int mine = 0;
#pragma unroll
    for (int u = 0; u < PW_PER_THREAD; ++u) {
      if (tid + 256 * u < total && c[u].idx != INT_MAX) {
        const unsigned long long k2 = pw_key(c[u].dist, c[u].idx);
        ++mine;
        if (k2 <= T) { const int pos = atomicAdd(&nsurv, 1); skey[pos] = k2; sw[pos] = c[u].w; }
      }
    }
    if (pass == 0) {   // one LDS atomic per wave (256 threads adding to one word serialise: ~2 us of the first version of this merge)
      int m = mine;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
      if ((tid & 63) == 0 && m) atomicAdd(&nvalid, m);
    }
    __syncthreads();
    if (pass == 0) IL_TL(1, 4);   // survivors compacted
    const int n = nsurv;
    for (int s = tid; s < n; s += 256) {
      const unsigned long long k2 = skey[s];
      int rank = 0;
#pragma unroll 8
      for (int j = 0; j < n; ++j) rank += skey[j] < k2 ? 1 : 0;
      order[rank] = (unsigned short)s;
    }
    __syncthreads();
    if (pass == 0) IL_TL(1, 5);   // survivors ranked
    if (tid < 64) {   // wave 0: the greedy coupling over the sorted survivors - cost and remaining weight in double, ascending key order, exactly the serial merge's operations
      const int lane = tid;
      double weight = d.agent_weight, cost = 0.0;
      int consumed = 0, part_idx = -1; float part_w = 0.f;
      if (n <= 64) {   // the usual case (~1.2 x the consumed atoms): survivor `it` lives in lane `it`, the loop reads it with v_readlane instead of three dependent LDS reads
        const int s = lane < n ? order[lane] : 0;
        const unsigned long long key = lane < n ? skey[s] : 0ull;
        const int kdist = (int)(unsigned)(key >> 32), kidx = (int)(unsigned)key, kw = __float_as_int(lane < n ? sw[s] : 0.f);
        if (pass == 0) IL_TL(2, 0);
        // (Measured, round 3: the conversions and the product ew dist hoisted out of the loop as per-lane doubles broadcast with two v_readlane each: 2.8 us for the ~26
        // atoms of a step against 2.2 us for this form; a branch-free form with selects: 2.2 us. A trip costs ~200 clocks of dependent double-precision latency and
        // VALU -> scalar-branch hand-offs on a single wave, not instruction issue.)
        for (int it = 0; it < n && weight > 0.0; ++it) {
          const double ew = (double)__int_as_float(__builtin_amdgcn_readlane(kw, it)), dist = (double)__int_as_float(__builtin_amdgcn_readlane(kdist, it));
          if (weight >= ew) { cost += ew * dist; weight -= ew; ++consumed; }
          else { cost += weight * dist; part_idx = __builtin_amdgcn_readlane(kidx, it); part_w = (float)ew - (float)weight; weight = 0.0; }
        }
        if (pass == 0) IL_TL(2, 1);
        if (!(weight > 0.0) || n == nvalid || T == EMPTY) { if (lane < consumed) d.weights[kidx] = -1.f; }
        if (pass == 0) IL_TL(2, 2);
      } else {
        for (int it = 0; it < n && weight > 0.0; ++it) {
          const int s = order[it];
          const unsigned long long best = skey[s];
          const double ew = (double)sw[s], dist = (double)__uint_as_float((unsigned)(best >> 32));
          if (weight >= ew) { cost += ew * dist; weight -= ew; ++consumed; }
          else { cost += weight * dist; part_idx = (int)(unsigned)best; part_w = (float)ew - (float)weight; weight = 0.0; }
        }
        if (!(weight > 0.0) || n == nvalid || T == EMPTY)
          for (int t = lane; t < consumed; t += 64) d.weights[(int)(unsigned)skey[order[t]]] = -1.f;
      }
      const bool done = !(weight > 0.0) || n == nvalid || T == EMPTY;   // otherwise: the survivors ran out before the weight did and candidates above the threshold exist
      if (lane == 0) {
        again = done ? 0 : 1;
        if (done) {
          if (part_idx >= 0) d.weights[part_idx] = part_w;
          out[0] = (float)(d.reward_scale * exp(-d.reward_bandwidth * cost));
        }
      }
      if (pass == 0) IL_TL(2, 3);
    }
    __syncthreads();
    if (pass == 0) IL_TL(1, 6);   // greedy coupling done
    if (!again) break;
    if (tid == 0) nsurv = 0;   // pass 2: every candidate (exact by construction; rare)
    T = EMPTY;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_pwil_merge(il_pwil d, int G, int K, const PwCand* __restrict__ cand, float* __restrict__ out) {
  pwil_merge_block(d, G, K, cand, out);
}

// One launch per environment step: every workgroup selects its chunk's candidates, takes a ticket, and the LAST one to arrive runs the merge (nobody waits: no
// co-residency requirement). A step was two launches of ~5 us of work each; at 19 us per step the launches themselves were what was left.
__global__ __launch_bounds__(PW_CHUNK) void k_pwil_step(il_pwil d, const float* __restrict__ state, const float* __restrict__ action, int K, PwCand* __restrict__ cand,
                                                        unsigned* __restrict__ ticket, float* __restrict__ out) {
  __shared__ unsigned last;
  pwil_select_block<true>(d, state, action, K, cand);
  // The candidates left with write-through stores: once every wave has drained its own (acknowledged by the memory side) and the workgroup has met, a RELAXED ticket is
  // enough - no release, i.e. no write-back of the XCD's L2 per workgroup; only the last arriver pays an acquire (an invalidate) before it reads the lists.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = t == gridDim.x - 1 ? 1u : 0u;
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next step's launch (stream-ordered behind this one)
    }
  }
  IL_TL(0, 4);   // ticket taken
  __syncthreads();
  if (last) sync_acquire_all();
  if (!last) return;
  pwil_merge_block(d, (int)gridDim.x, K, cand, out);
  IL_TL(1, 7);
}

__global__ __launch_bounds__(256) void k_pwil_merge_serial(il_pwil d, int G, int K, const PwCand* __restrict__ cand, float* __restrict__ out) {
  __shared__ PwCand sc[PW_LDS_CAND];
  const int tid = threadIdx.x;
  const bool staged = G * K <= PW_LDS_CAND;
  if (staged) for (int i = tid; i < G * K; i += 256) sc[i] = cand[i];
  __syncthreads();
  if (tid >= 64) return;
  if (staged) pwil_merge_wave(d, G, K, sc, out);
  else pwil_merge_wave(d, G, K, cand, out);
}

static int pwil_take(const il_pwil* d) { return (int)ceil(d->agent_weight * (double)d->n_atoms) + 2; }   // most atoms one step can consume
extern "C" int64_t il_pwil_scratch_floats(int32_t n_atoms, double agent_weight) {
  const int m = (int)ceil(agent_weight * (double)n_atoms) + 2, K = m < PW_CHUNK ? m : PW_CHUNK;
  const int64_t lists = ((int64_t)n_atoms + PW_CHUNK - 1) / PW_CHUNK * K * 4;
  return (lists > n_atoms ? lists : n_atoms) + 4;   // + one 16-byte slot behind everything: k_pwil_step's arrival counter
}
static unsigned* pwil_ticket(const il_pwil* d) {
  const int64_t n = il_pwil_scratch_floats(d->n_atoms, d->agent_weight);
  return reinterpret_cast<unsigned*>(d->dists + n - 4);
}

extern "C" int il_pwil_reset(const il_pwil* d, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->weights && d->n_atoms > 0, "il_pwil_reset: bad arguments");
  { IL_TRACE("k_pwil_reset", (hipStream_t)stream_); k_pwil_reset<<<ceil_div(d->n_atoms, 1024) < 256 ? ceil_div(d->n_atoms, 1024) : 256, 1024, 0, (hipStream_t)stream_>>>(*d, d->dists ? pwil_ticket(d) : nullptr); }
  IL_CHECK_LAUNCH("il_pwil_reset");
  return IL_OK;
}

extern "C" int il_pwil_reward(const il_pwil* d, const float* state, const float* action, float* out_reward, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->atoms && d->weights && d->dists && d->scale && d->offset && state && out_reward, "il_pwil_reward: bad arguments");
  IL_CHECK_ARG(d->dim >= 1 && d->dim <= 512 && d->n_atoms > 0, "il_pwil_reward: dim=%d out of range [1,512]", d->dim);
  IL_CHECK_ARG(d->state_dim == d->dim || action, "il_pwil_reward: action pointer missing");
  const int m = pwil_take(d), G = ceil_div(d->n_atoms, PW_CHUNK);
  static const bool one_wg = getenv("IL_PWIL_ONE_WORKGROUP") != nullptr;   // developer A/B switch
  if (m <= PW_CHUNK && G <= 64 * PW_MAXQ && !one_wg) {
    PwCand* cand = reinterpret_cast<PwCand*>(d->dists);   // >= il_pwil_scratch_floats(n_atoms, agent_weight) floats
    static const bool serial = getenv("IL_PWIL_SERIAL_MERGE") != nullptr;   // developer A/B switches: the round-2 one-wave merge; select and merge as two launches
    static const bool two = getenv("IL_PWIL_TWO_LAUNCHES") != nullptr;
    if (G * m <= PW_LDS_CAND && !serial && !two) {
      IL_TRACE("k_pwil_step", (hipStream_t)stream_); k_pwil_step<<<G, PW_CHUNK, 0, (hipStream_t)stream_>>>(*d, state, action, m, cand, pwil_ticket(d), out_reward);
      IL_CHECK_LAUNCH("il_pwil_reward");
      return IL_OK;
    }
    { IL_TRACE("k_pwil_select", (hipStream_t)stream_); k_pwil_select<<<G, PW_CHUNK, 0, (hipStream_t)stream_>>>(*d, state, action, m, cand); }
    if (G * m <= PW_LDS_CAND && !serial) { IL_TRACE("k_pwil_merge", (hipStream_t)stream_); k_pwil_merge<<<1, 256, 0, (hipStream_t)stream_>>>(*d, G, m, cand, out_reward); }
    else { IL_TRACE("k_pwil_merge_serial", (hipStream_t)stream_); k_pwil_merge_serial<<<1, 256, 0, (hipStream_t)stream_>>>(*d, G, m, cand, out_reward); }
  } else {
    IL_TRACE("k_pwil_reward", (hipStream_t)stream_); k_pwil_reward<<<1, 1024, 0, (hipStream_t)stream_>>>(*d, state, action, out_reward);
  }
  IL_CHECK_LAUNCH("il_pwil_reward");
  return IL_OK;
}

IL_TL_READER(il_debug_timeline_pwil)
