// MT19937 (Matsumoto & Nishimura) on the device: the index stream of reference memory.py:51-56 (numpy legacy RandomState.randint = masked rejection on 32-bit
// outputs, plus the reference's rejection of slot (idx - 1) % size), drawn by ONE workgroup. Shared by replay.hip (k_sample2) and gail.hip (the sampler workgroup
// that rides in the resident discriminator launch).  state[0..623] = mt words, state[624] = position.
#pragma once
#include "il_common.hpp"

#define MT_N 624
#define MT_M 397
__host__ __device__ static inline uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
  return y;
}
__host__ __device__ static inline uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7FFFFFFFu);
  return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
}
// Device version: one workgroup. The draw is a stream compaction of the tempered MT output (every candidate consumes
// exactly one 32-bit word; rejected ones are skipped), so 256 candidates are tested in parallel and compacted in order.
struct MtShared { uint32_t mt[MT_N]; uint32_t nx[MT_N]; int wave_cnt[4]; int pos, count, last, have_next; };

// The NEXT block of 624 words (the twist of sh.mt) into sh.nx. A pure function of the current block, so the resident sampler computes it while it waits for the previous
// update: when the draw crosses the block boundary (about once per update: two batches of 256 consume ~670 words) the new block is already there.
__device__ __forceinline__ void mt_next_block(MtShared& sh) {
  const int tid = threadIdx.x;
  for (int k = tid; k < 227; k += 256) sh.nx[k] = mt_mix(sh.mt[k], sh.mt[k + 1], sh.mt[k + MT_M]);
  __syncthreads();
  for (int k = 227 + tid; k < 454; k += 256) sh.nx[k] = mt_mix(sh.mt[k], sh.mt[k + 1], sh.nx[k - 227]);
  __syncthreads();
  for (int k = 454 + tid; k < 623; k += 256) sh.nx[k] = mt_mix(sh.mt[k], sh.mt[k + 1], sh.nx[k - 227]);
  __syncthreads();
  if (tid == 0) { sh.nx[623] = mt_mix(sh.mt[623], sh.nx[0], sh.nx[396]); sh.have_next = 1; }
  __syncthreads();
}

__device__ __forceinline__ void mt_draw(MtShared& sh, const int64_t* __restrict__ ring_state, int n, int32_t* __restrict__ out) {
  const int tid = threadIdx.x;
  const int64_t idx = ring_state[0], full = ring_state[1], size = ring_state[2];
  const int64_t high = full ? size : idx - 1;
  const int64_t excl = ((idx - 1) % size + size) % size;
  const uint32_t rng = high > 0 ? (uint32_t)(high - 1) : 0u;
  uint32_t mask = rng; mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  if (tid == 0) sh.count = 0;
  __syncthreads();
  if (rng == 0) {  // degenerate range: numpy returns 0 without consuming the stream
    for (int i = tid; i < n; i += 256) wstore1(reinterpret_cast<float*>(out), i, 0.f);
    return;
  }
  for (int guard = 0; guard < 100000; ++guard) {
    int pos = sh.pos, count = sh.count;
    if (count >= n) break;
    if (pos >= MT_N && sh.have_next) {   // the block prepared by mt_next_block
      for (int k = tid; k < MT_N; k += 256) sh.mt[k] = sh.nx[k];
      __syncthreads();
      if (tid == 0) { sh.pos = 0; sh.have_next = 0; }
      __syncthreads();
      pos = 0;
    } else if (pos >= MT_N) {  // twist in place: four dependency-free phases
      uint32_t nv[3]; int q = 0;
      for (int k = tid; k < 227; k += 256) nv[q++] = mt_mix(sh.mt[k], sh.mt[k + 1], sh.mt[k + MT_M]);
      __syncthreads(); q = 0;
      for (int k = tid; k < 227; k += 256) sh.mt[k] = nv[q++];
      __syncthreads(); q = 0;
      for (int k = 227 + tid; k < 454; k += 256) nv[q++] = mt_mix(sh.mt[k], sh.mt[k + 1], sh.mt[k - 227]);
      __syncthreads(); q = 0;
      for (int k = 227 + tid; k < 454; k += 256) sh.mt[k] = nv[q++];
      __syncthreads(); q = 0;
      for (int k = 454 + tid; k < 623; k += 256) nv[q++] = mt_mix(sh.mt[k], sh.mt[k + 1], sh.mt[k - 227]);
      __syncthreads(); q = 0;
      for (int k = 454 + tid; k < 623; k += 256) sh.mt[k] = nv[q++];
      __syncthreads();
      if (tid == 0) { sh.mt[623] = mt_mix(sh.mt[623], sh.mt[0], sh.mt[396]); sh.pos = 0; }
      __syncthreads();
      pos = 0;
    }
    const int avail = MT_N - pos, take = avail < 256 ? avail : 256;
    bool ok = false; uint32_t v = 0;
    if (tid < take) { v = mt_temper(sh.mt[pos + tid]) & mask; ok = (v <= rng) && ((int64_t)v != excl); }
    const unsigned long long bal = __ballot(ok);
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) sh.wave_cnt[w] = __popcll(bal);
    __syncthreads();
    int before = __popcll(bal & ((1ull << lane) - 1ull));
    for (int i = 0; i < w; ++i) before += sh.wave_cnt[i];
    const int tot = sh.wave_cnt[0] + sh.wave_cnt[1] + sh.wave_cnt[2] + sh.wave_cnt[3];
    const int need = n - count;
    // candidates are consumed up to and including the one that completes the batch
    // written THROUGH (round 6, profiles/r06_soak_under_load.md): the resident draw hands the indices to launches of another stream that are already running; like every other
    // in-launch hand-off of the schedule they are in memory once the store is acknowledged (sync_signal drains every wave before the arrival), not after a later L2 write-back
    if (ok && before < need) wstore1(reinterpret_cast<float*>(out), count + before, __uint_as_float(v));
    if (tot >= need) { if (ok && before == need - 1) sh.last = tid; }
    __syncthreads();
    if (tid == 0) {
      if (tot >= need) { sh.pos = pos + sh.last + 1; sh.count = n; }
      else { sh.pos = pos + take; sh.count = count + tot; }
    }
    __syncthreads();
  }
}


// The whole draw of an update by one workgroup of 256 threads (`sh` in LDS): generator state in, the next block prepared, [wait for the previous update: resident],
// agent batch then expert batch (train.py:173), [IL_SYNC_INDICES] signalled, state out.
// `stage` (round 5, il_gail_disc_step_draw_staged): after the draw the workgroup also copies the n drawn AGENT rows (row4 16-byte lanes each, ring rows at `stage_ring`,
// indices clamped like il_replay_gather) into the dense slab `stage` BEFORE it signals: with the early draw (IL_SYNC_CHAIN_DONE) this happens tens of microseconds ahead
// of the next update, whose forward / critic-loss launch then reads dense rows - one global trip in its prologues instead of index -> row.
struct MtStage { const float* ring; float* rows; long long capacity; int row4; };
__device__ __forceinline__ void mt_sample_update(MtShared& sh, uint32_t* __restrict__ state, int n, const int64_t* __restrict__ rs_a, int32_t* __restrict__ idx_a,
                                                 const int64_t* __restrict__ rs_b, int32_t* __restrict__ idx_b, long long* __restrict__ sync, int resident, MtStage stage = MtStage{}) {
  const int tid = threadIdx.x;
  for (int i = tid; i < MT_N; i += 256) sh.mt[i] = state[i];
  if (tid == 0) { sh.pos = (int)state[MT_N]; sh.have_next = 0; }
  __syncthreads();
  mt_next_block(sh);   // before the ring state is needed (and, resident, before the wait): off the path from "previous update done" to "indices drawn"
  // resident: launched with NO dependency on the update's main stream, i.e. while the previous update is still running. The draw itself must wait until that update
  // is over (its kernels read the index arrays this one overwrites, and the ring cursor may still move): [IL_SYNC_MAIN_EPOCH] has to reach the number of draws made
  // so far (= [IL_SYNC_INDICES]).
  IL_TL(0, 1);
  if (resident) {
    // (round 5) the index arrays are free as soon as the previous update's forward / critic-loss launch - their last reader on the main stream - has retired all its
    // workgroups ([IL_SYNC_CHAIN_DONE], [IL_SYNC_CHAIN_WGS] published by that launch): the draw then runs ~30 us ahead of the next update instead of between two updates.
    // No such launch seen yet (first update, schedules without it): the previous update's end, as before.
    const long long cw = __hip_atomic_load(sync + IL_SYNC_CHAIN_WGS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cw > 0) sync_wait(sync, IL_SYNC_CHAIN_DONE, sync_read(sync, IL_SYNC_INDICES) * cw);
    else sync_wait(sync, IL_SYNC_MAIN_EPOCH, sync_read(sync, IL_SYNC_INDICES));
  }
  IL_TL(0, 2);
  __syncthreads();
  mt_draw(sh, rs_a, n, idx_a);
  if (rs_b) { __syncthreads(); mt_draw(sh, rs_b, n, idx_b); }
  IL_TL(0, 3);
  if (stage.rows) {
    __syncthreads();   // idx_a as written by this workgroup's threads above
    typedef float st_f32x4 __attribute__((ext_vector_type(4)));
    const st_f32x4* src = reinterpret_cast<const st_f32x4*>(stage.ring);
    st_f32x4* dst = reinterpret_cast<st_f32x4*>(stage.rows);
    const int lanes = n * stage.row4;
    for (int i0 = tid; i0 < lanes; i0 += 4 * 256) {   // four lanes per thread and trip: indices, then rows, requested together
      long long sr[4]; st_f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = min(i0 + u * 256, lanes - 1); const long long s_ = idx_a[i / stage.row4]; sr[u] = s_ < 0 ? 0 : (s_ >= stage.capacity ? stage.capacity - 1 : s_); }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = min(i0 + u * 256, lanes - 1); v[u] = src[sr[u] * stage.row4 + i % stage.row4]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; if (i < lanes) dst[i] = v[u]; }
    }
  }
  if (sync) sync_signal(sync + IL_SYNC_INDICES);   // both index arrays (and the staged rows) are in place (a consumer that gathers its own rows need not wait for k_gather2)
  __syncthreads();
  for (int i = tid; i < MT_N; i += 256) state[i] = sh.mt[i];
  if (tid == 0) state[MT_N] = (uint32_t)sh.pos;
  IL_TL(0, 7);
}
