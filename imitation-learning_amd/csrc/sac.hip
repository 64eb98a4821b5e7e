// SAC update kernels (reference training.py:14-54) for gfx950.
//
// One update = 6 dependent launches, cut only where the algorithm has a global dependency:
//   k_actor_fwd      tiles of 16 rows: actor(s') -> a', logp' (no grad) and actor(s) -> a~, logp (activations kept)
//   k_critic_fwd     (tile, net) : critic_1/2(s,a) with saved activations, target_1/2(s', a')
//   k_critic_bwd     (tile, net) : y, dQ, back-propagation to the layer-1 pre-activations
//   k_dw_adam        output-stationary dW = dZ^T.X over the whole batch on MFMA + fused AdamW (critic)
//   k_policy_critic  (tile, net) : updated critic on (s, a~), dQ/da~; the pair's second workgroup to finish continues with the policy
//                                  backward of the tile (min-Q selection, tanh-Gaussian backward, back-propagation through the actor)
//   k_dw_adam        actor dW + AdamW, Adam(log_alpha), polyak as tail blocks
// Activations cross kernels through an L2-resident workspace (~3 MB at B=256, H=256) stored FEATURE-MAJOR ([H][B]): an MFMA
// accumulator lane holds 4 consecutive batch rows of one feature, so producers store and consumers load whole 16-byte lanes
// (row masks in the backward epilogues, and both operands of the dW kernel, whose reduction index is the batch row).
// Parameters, Adam moments and the target network are each read and written exactly once per update (24 B/param + 8 B/param).
#include "il_common.hpp"
#include "mlp_tile.hpp"
#include "peer_device.hpp"
#include "disc_reward.hpp"
IL_ST_TABLE
#ifdef IL_EXP_CHECK   // developer build (profiles/tools/direct_soak_matrix.py): what the consumers of the fence-free hand-offs READ against what their producers finally WROTE
static __device__ unsigned il_chk[16];
static __device__ float il_chk_rew[2][4096], il_chk_tq[2][3][4096], il_chk_a2[4][4096 * 8], il_chk_rel[4096];
extern "C" int il_debug_check(unsigned* out_host) { return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(il_chk), sizeof(il_chk)) == hipSuccess ? 0 : 3; }
#endif

// Measured (round 3, same box, three interleaved rounds): write-through stores in the dW / AdamW epilogue ONLY: 14.71k -> 14.90k updates/s; ALSO for the activations and dZ
// the tile kernels leave in the workspace: 14.59k - the next launch reads those, and a written-through line is not left behind in the L2 for the readers of its own XCD.
#ifndef IL_WT_TILE_STORES
#define IL_WT_TILE_STORES 0
#endif


struct SacWs {  // float offsets into il_sac.workspace
  int64_t a_h1, a_h2, a_xpre, a_eps, a_lsraw, a_anew, a_logp, n_a2, n_logp2, a_x0;
  int64_t c_x0, c_h1, c_h2, c_q, t_q, c_dz3, c_dz2, c_dz1;
  int64_t p_q, p_g, a_dz3, a_dz2, a_dz1, alpha_part, pair_ctr, chain_ctr;
  int64_t pk_af, pk_ab, pk_cf, pk_cb, pk_tf, pk_tb;  // lane-ordered copies of the H x H layers (mlp_tile.hpp "Packed hidden-layer weights")
  int64_t x_slab, x_flag, c_rew, p_part;   // p_part: pair-mode k_policy_critic: dQ/da partial tiles [2][nt][16 n-blocks][16 rows][A]   // pair mode (mlp_tile.hpp): 6 * nt hop slabs of 16 x H/2 floats, their flags (one 128-byte line each), the relabel role's rewards [B]
  int64_t total;
};
__host__ __device__ inline SacWs sac_ws(int S, int A, int H, int B) {
  SacWs w; int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += (n + 3) & ~(int64_t)3; return r; };
  const int64_t BH = (int64_t)B * H, BA = (int64_t)B * A;
  w.a_h1 = take(BH); w.a_h2 = take(BH); w.a_xpre = take(BA); w.a_eps = take(BA); w.a_lsraw = take(BA); w.a_anew = take(BA); w.a_logp = take(B);
  w.n_a2 = take(BA); w.n_logp2 = take(B); w.a_x0 = take((int64_t)B * S);   // a_x0: s^T [S][B] of the sampled rows (written by the actor(s) tiles; the actor's layer-1 dW reads it like every other [feature][B] operand)
  w.c_x0 = take((int64_t)B * (S + A)); w.c_h1 = take(2 * BH); w.c_h2 = take(2 * BH); w.c_q = take(2 * B); w.t_q = take(2 * B);
  w.c_dz3 = take(2 * B); w.c_dz2 = take(2 * BH); w.c_dz1 = take(2 * BH);
  w.p_q = take(2 * B); w.p_g = take(2 * BA); w.a_dz3 = take((int64_t)B * 16); w.a_dz2 = take(BH); w.a_dz1 = take(BH); w.alpha_part = take(B / IL_TILE_R + 4); w.pair_ctr = take((int64_t)(B / IL_TILE_R) * IL_CTR_STRIDE + 4); w.chain_ctr = take((int64_t)(B / IL_TILE_R) * IL_CTR_STRIDE + 4);
  const int64_t HH = (int64_t)H * H;
  w.pk_af = take(HH); w.pk_ab = take(HH); w.pk_cf = take(2 * HH); w.pk_cb = take(2 * HH); w.pk_tf = take(2 * HH); w.pk_tb = take(2 * HH);
  o = (o + 31) & ~(int64_t)31;   // slabs and flag lines start on 128-byte lines of their own
  w.x_slab = take((int64_t)6 * (B / IL_TILE_R) * IL_TILE_R * (H / 2)); o = (o + 31) & ~(int64_t)31; w.x_flag = take((int64_t)6 * (B / IL_TILE_R) * IL_CTR_STRIDE + 32); w.c_rew = take(B); w.p_part = take((int64_t)2 * (B / IL_TILE_R) * 16 * IL_TILE_R * A);
  w.total = o;
  return w;
}
__host__ __device__ inline int64_t net_stride(int in, int H, int out) { return (mlp_numel(in, H, out) + 3) & ~(int64_t)3; }

// one wave per 16 hidden columns (4 waves per SIMD at H = 256), never fewer than 4 waves
static inline int tile_threads(int H) { return H * 4 < 256 ? 256 : H * 4; }
// LDS bytes needed by the tile kernels
static inline size_t tile_lds_bytes(int in_pad, int H) {
  return sizeof(float) * ((size_t)IL_TILE_R * (in_pad + 4) + 2 * (size_t)IL_TILE_R * (H + 4) + (size_t)(tile_threads(H) / 64) * 256 + 256 + 64);
}

// ---------------------------------------------------------------------------------------------
// tanh-Gaussian head for one (row, action component); op order follows torch.distributions (see oracle/nets.py)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void head_sample(float mean, float ls_raw, float eps, float& x, float& a, float& nlp, float& ladj) {
  const float ls = fminf(fmaxf(ls_raw, -20.f), 2.f);
  const float sd = expf(ls);
  x = __fadd_rn(__fmul_rn(eps, sd), mean);
  a = tanhf(x);
  const float d = __fsub_rn(x, mean);
  nlp = -(d * d) / (2.f * (sd * sd)) - logf(sd) - LOG_SQRT_2PI;
  ladj = 2.f * (LOG_2 - x - softplus_f(-2.f * x));
}

// il_sac.debug_masks (tests only; include/il_hip.h): hv = 4 consecutive rows (row .. row + 3) of hidden column `col` after the ReLU
__device__ __forceinline__ void debug_mask4(const il_sac& d, int slot, int row, int col, const f32x4& hv) {
  if (d.debug_masks) {
#pragma unroll
    for (int r = 0; r < 4; ++r) d.debug_masks[((size_t)slot * d.batch + row + r) * d.hidden + col] = hv[r] > 0.f ? 1.f : 0.f;
  }
}

// k_repack: PF / PB copies of W2 for actor (net 0), critic_1,2 (1,2) and target_1,2 (3,4). `mask` selects nets (bit per net).
// Every public entry point derives the copies it reads from the parameters inside the same call, so they can never be stale.
__global__ __launch_bounds__(256) void k_repack(il_sac d, unsigned mask, const il_sac* __restrict__ dL) {
  if (dL) d = dL[blockIdx.z];  // population axis: one descriptor per learner (wave-uniform scalar loads)
  globalize(d);
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  const int net = blockIdx.y;
  const SacWs ws = sac_ws(S, A, H, d.batch);
  if (blockIdx.x == 0 && blockIdx.y == 0)   // arrival counters of k_policy_critic's tile pairs and of k_sac_chain's tiles (they reset themselves; this covers a reused arena)
  {
    for (int i = threadIdx.x; i < d.batch / IL_TILE_R; i += blockDim.x) { reinterpret_cast<unsigned*>(d.workspace + ws.pair_ctr)[i * IL_CTR_STRIDE] = 0u; reinterpret_cast<unsigned*>(d.workspace + ws.chain_ctr)[i * IL_CTR_STRIDE] = 0u; }
    if (threadIdx.x == 0) reinterpret_cast<unsigned*>(d.workspace + ws.chain_ctr)[(d.batch / IL_TILE_R) * IL_CTR_STRIDE + 1] = 0u;   // il_sac_handoff_timeouts counts from here
    for (int i = threadIdx.x; i < 6 * (d.batch / IL_TILE_R); i += blockDim.x) { reinterpret_cast<unsigned*>(d.workspace + ws.x_flag)[i * IL_CTR_STRIDE] = 0u; reinterpret_cast<unsigned*>(d.workspace + ws.x_flag)[i * IL_CTR_STRIDE + 1] = 0u; }   // pair-mode hop flags (their consumers clear them; this covers a reused arena)
  }
  if (!((mask >> net) & 1u)) return;
  const int64_t HH = (int64_t)H * H, ns = net_stride(IN, H, 1);
  const float* W2; float* pf; float* pb;
  if (net == 0) { W2 = d.actor + (size_t)H * S + H; pf = d.workspace + ws.pk_af; pb = d.workspace + ws.pk_ab; }
  else {
    const int k = (net - 1) & 1; const bool tgt = net >= 3;
    W2 = (tgt ? d.target : d.critic) + k * ns + (size_t)H * IN + H;
    pf = d.workspace + (tgt ? ws.pk_tf : ws.pk_cf) + k * HH; pb = d.workspace + (tgt ? ws.pk_tb : ws.pk_cb) + k * HH;
  }
  // one thread = a 4 x 4 block (rows n..n+3, columns k..k+3): four 16-B row reads, transposed in registers, eight 16-B stores
  const int kq = H / 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kq * kq; i += gridDim.x * blockDim.x) {
    const int n = (i / kq) * 4, k = (i - (i / kq) * kq) * 4;
    f32x4 w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(W2 + (size_t)(n + r) * H + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(pf + packed_fwd_index(n + r, k, H)) = w[r];   // k..k+3 of row n+r: one lane of PF
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 c; c[0] = w[0][q]; c[1] = w[1][q]; c[2] = w[2][q]; c[3] = w[3][q];
      *reinterpret_cast<f32x4*>(pb + packed_bwd_index(n, k + q, H)) = c;                                   // rows n..n+3 of column k+q: one lane of PB
    }
  }
}

// mode: 0 = next rows then current rows (grid 2*nt), 1 = next only, 2 = current only
template <int PANEL = 16>
__device__ __forceinline__ void actor_fwd_tile(const il_sac& d, const il_batch& b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, bool is_cur, int tile,
                                               float* smem) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch;
  const int row0 = tile * IL_TILE_R;
  const int Sp = round_up16(S), ldx = Sp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* part = H2s + IL_TILE_R * ldh; float* Os = part + (blockDim.x >> 6) * 256;
  float* red = Os + 256;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView net = mlp_view(d.actor, S, H, 2 * A);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, tid = threadIdx.x;
  // Requested / computed before the first barrier, consumed after the MFMA loops: the biases of this wave's 16 columns, and for the head
  // threads their noise sample (Philox + Box-Muller is ~1 us of dependent ALU work that needs nothing from the MLP) and absorbing flag.
  const int pc = min(wave * 16 + j, H - 1);
  const float pb1 = gload(net.b1 + pc), pb2 = gload(net.b2 + pc);
#if IL_SMALL_PREFETCH
  const SmallPre w3pre = tile_fwd_small_prefetch(net.W3, H, 2 * A, H);   // the head's weight lanes: consumed after both hidden layers
#endif
  const uint32_t ctr = d.noise_counter ? *d.noise_counter : 0u;
  float e_pre = 0.f, absorb_pre = 0.f;
  if (tid < IL_TILE_R * A) {
    const int r = tid / A, c = tid - r * A, row = row0 + r;
    const float* ep = is_cur ? eps_cur : eps_next;
    e_pre = ep ? ep[(size_t)row * A + c] : philox_normal(d.noise_seed, ctr, is_cur ? IL_STREAM_EPS_CUR : IL_STREAM_EPS_NEXT, (uint32_t)(row * A + c));
    if (!is_cur) absorb_pre = b.absorbing[brow(b, row) * b.ld_absorbing];
  }
  const float* src = is_cur ? b.states : b.next_states;
  const int ld = is_cur ? b.ld_states : b.ld_next_states;
  IL_TL(is_cur ? 6 : 5, 0);
  load_rows_cat(Xs, ldx, Sp, src, ld, S, nullptr, 0, 0, row0, IL_TILE_R, b.gather, b.gather_capacity);
  __syncthreads();
  IL_TL(is_cur ? 6 : 5, 1);
  if (is_cur)
    for (int i = tid; i < IL_TILE_R * S; i += blockDim.x) { const int c = i >> 4, r = i & 15; W[ws.a_x0 + (size_t)c * B + row0 + r] = Xs[r * ldx + c]; }   // x0^T [S][B]
  tile_fwd<PANEL>(Xs, ldx, Sp, net.W1, S, S, H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb1 : net.b1[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H1s[(4 * g + r) * ldh + col] = hv[r]; }
    if (is_cur) { wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.a_h1 + (int64_t)col * B + row0 + 4 * g, hv); debug_mask4(d, 0, row0 + 4 * g, col, hv); }
  });
  __syncthreads();
  IL_TL(is_cur ? 6 : 5, 2);
  tile_fwd_packed<PANEL>(H1s, ldh, H, W + ws.pk_af, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb2 : net.b2[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H2s[(4 * g + r) * ldh + col] = hv[r]; }
    if (is_cur) { wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.a_h2 + (int64_t)col * B + row0 + 4 * g, hv); debug_mask4(d, 1, row0 + 4 * g, col, hv); }
  });
  __syncthreads();
  IL_TL(is_cur ? 6 : 5, 3);
#if IL_SMALL_PREFETCH
  tile_fwd_small(H2s, ldh, H, net.W3, H, 2 * A, net.b3, Os, part, &w3pre);
#else
  tile_fwd_small(H2s, ldh, H, net.W3, H, 2 * A, net.b3, Os, part);
#endif
  IL_TL(is_cur ? 6 : 5, 4);
  // head: one thread per (row, action component); per-row sums through LDS (sequential over A like torch's sum(-1))
  float* nl = part; float* la = part + 256;
  if (tid < IL_TILE_R * A) {
    const int r = tid / A, c = tid - r * A, row = row0 + r;
    const float e = e_pre;
    const float mean = Os[r * 16 + c], lsr = Os[r * 16 + A + c];
    float x, a, nlp, ladj;
    head_sample(mean, lsr, e, x, a, nlp, ladj);
    nl[r * 16 + c] = nlp; la[r * 16 + c] = ladj;
    if (is_cur) {
      W[ws.a_xpre + (size_t)row * A + c] = x; W[ws.a_eps + (size_t)row * A + c] = e; W[ws.a_lsraw + (size_t)row * A + c] = lsr;
      W[ws.a_anew + (size_t)row * A + c] = a;
    } else {
      W[ws.n_a2 + (size_t)row * A + c] = (1.f - absorb_pre) * a;
    }
  }
  __syncthreads();
  if (tid < IL_TILE_R) {
    float sn = 0.f, sl = 0.f;
    for (int c = 0; c < A; ++c) { sn += nl[tid * 16 + c]; sl += la[tid * 16 + c]; }
    const float logp = (0.f - sl) + sn;
    W[(is_cur ? ws.a_logp : ws.n_logp2) + row0 + tid] = logp;
  }
  IL_TL(is_cur ? 6 : 5, 5);
  (void)red;
}

template <int PANEL>
__device__ __forceinline__ void k_actor_fwd_body(il_sac d, il_batch b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, int mode,
                                                    const il_sac* __restrict__ dL, const il_batch* __restrict__ bL, float* smem) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (dL) { pop_ids(bx, by); d = dL[by]; b = bL[by]; }
  globalize(d); globalize(b);
  const int nt = d.batch / IL_TILE_R;
  const bool is_cur = (mode == 2) || (mode == 0 && bx >= nt);
  actor_fwd_tile<PANEL>(d, b, eps_next, eps_cur, is_cur, bx % nt, smem);
  IL_TL_END(is_cur ? 6 : 5);
}

// ---------------------------------------------------------------------------------------------
// Critic / target forward (critic_fwd_tile; k_critic_fwd: grid = 4 * nt) and the per-tile chaining primitives of k_sac_chain.
// ---------------------------------------------------------------------------------------------
// Per-tile arrival counter of the fused forward + critic-loss launch (k_sac_chain): 0 -> 1 (actor on s' done) -> 3 (both target critics done)
// -> 5 (both critics have read the targets; the one that sees 4 resets it to 0 for the next launch). Producer side: every thread's stores,
// barrier, ONE agent-scope release; consumer side: ONE polling lane, an agent-scope acquire, barrier (cf. sync_signal / sync_wait).
__device__ __forceinline__ void tile_arrive(unsigned* ctr) {
  sync_drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait that gives up is counted twice: in the workspace slot il_sac_handoff_timeouts() reads, and in [IL_SYNC_TIMEOUTS] when the learner has il_sync counters
// (what UpdatePlan.sync_timeouts() checks). Neither can happen while the launch is co-resident or dispatched in block order; both must read 0.
struct TileTimeouts { unsigned* slot; long long* sync; };
// [IL_SYNC_CHAIN_WGS] / [IL_SYNC_CHAIN_DONE] (include/il_hip.h): the forward / critic-loss launch tells the resident sampler how many workgroups it has and when each has
// retired - every read of the update's index arrays on the main stream happens in this launch, so the next update's draw may start once all of them are gone.
// chain_done: after a workgroup barrier (every wave's loads have returned: their values were used); relaxed - the sampler only OVERWRITES what these workgroups read.
__device__ __forceinline__ void chain_publish_grid(const il_sac& d, int on) {
  if (on && d.sync && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<long long*>(d.sync) + IL_SYNC_CHAIN_WGS, (long long)gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_done(const il_sac& d, int on) {
  if (on && d.sync && threadIdx.x == 0) __hip_atomic_fetch_add(reinterpret_cast<long long*>(d.sync) + IL_SYNC_CHAIN_DONE, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ TileTimeouts tile_timeouts(const il_sac& d) {
  const SacWs ws = sac_ws(d.state_dim, d.action_dim, d.hidden, d.batch);
  return {reinterpret_cast<unsigned*>(d.workspace + ws.chain_ctr) + (d.batch / IL_TILE_R) * IL_CTR_STRIDE + 1, reinterpret_cast<long long*>(d.sync)};
}
__device__ __forceinline__ void tile_await(unsigned* ctr, unsigned target, const TileTimeouts& timeouts) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > IL_SYNC_SPIN_LIMIT) {
        __hip_atomic_fetch_add(timeouts.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (timeouts.sync) sync_timed_out(timeouts.sync);
        break;
      }
    }
  }
  __syncthreads();
  sync_acquire_all();
}

// Pair-mode chain: arrivals in the low four bits of the tile counter, flag bits above them (the relabel role arrives with += 16, so that it cannot be mistaken for
// the tile's actor(s') or a target); waits until (value & 15) >= low and every bit of `bits` is set. Same bounded poll + one agent acquire as tile_await.
template <bool ACQUIRE = true>   // ACQUIRE = false: everything the waiter reads afterwards was written through by its producer and is read below the L1 (sload1)
__device__ __forceinline__ void tile_await_bits(unsigned* ctr, unsigned low, unsigned bits, const TileTimeouts& timeouts) {
  if (threadIdx.x == 0) {
    int spins = 0;
    for (;;) {
      const unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v & 15u) >= low && (v & bits) == bits) break;
      __builtin_amdgcn_s_sleep(4);
      if (++spins > IL_SYNC_SPIN_LIMIT) {
        __hip_atomic_fetch_add(timeouts.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (timeouts.sync) sync_timed_out(timeouts.sync);
        break;
      }
    }
  }
  __syncthreads();
  if (ACQUIRE) sync_acquire_all();
}
// the producer side of the same: every wave drains its write-through stores, barrier, ONE relaxed arrival
__device__ __forceinline__ void tile_arrive_through(unsigned* ctr, unsigned inc) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Forward of one critic-shaped network on one 16-row tile. net 0,1: critic_k(s, a) keeping h1, h2 (and x0 for net 0) for the weight gradients;
// net 2,3: target_k(s', a'). Leaves H1s / H2s (post-ReLU activations) and q16[r] = Q in LDS. `await` != NULL (target networks inside k_sac_chain):
// a' of this tile is still being produced by another workgroup of the same launch; everything that does not need it is done first.
template <int PANEL = 16>
__device__ __forceinline__ void critic_fwd_tile(const il_sac& d, const il_batch& b, int net, int tile, float* smem, unsigned* await) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int row0 = tile * IL_TILE_R;
  const bool is_target = net >= 2; const int k = net & 1;
  const int INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* q16 = H2s + IL_TILE_R * ldh;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const int64_t ns = net_stride(IN, H, 1);
  const MlpView p = mlp_view((is_target ? d.target : d.critic) + k * ns, IN, H, 1);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // biases of this wave's 16 columns and this lane's slice of w3: requested before the first barrier, consumed after the MFMA loops
  const int pc = min(wave * 16 + j, H - 1);
  const float pb1 = gload(p.b1 + pc), pb2 = gload(p.b2 + pc), pb3 = gload(p.b3);
  float w3v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) w3v[u] = gload(p.W3 + min(lane + 64 * u, H - 1));
  IL_TL(8, 0);
  if (is_target && await) {
    load_rows_cat(Xs, ldx, INp, b.next_states, b.ld_next_states, S, nullptr, 0, 0, row0, IL_TILE_R, b.gather, b.gather_capacity);   // s' columns, zero elsewhere
    tile_await(await, 1u, tile_timeouts(d));
    IL_TL(0, 2);
    for (int i = threadIdx.x; i < IL_TILE_R * A; i += blockDim.x) { const int r = i / A, c = i - r * A; Xs[r * ldx + S + c] = W[ws.n_a2 + (size_t)(row0 + r) * A + c]; }
  } else if (is_target) load_rows_cat(Xs, ldx, INp, b.next_states, b.ld_next_states, S, W + ws.n_a2, A, A, row0, IL_TILE_R, b.gather, b.gather_capacity, true);
  else load_rows_cat(Xs, ldx, INp, b.states, b.ld_states, S, b.actions, b.ld_actions, A, row0, IL_TILE_R, b.gather, b.gather_capacity);
  __syncthreads();
  IL_TL(8, 1);
  if (net == 0)
    for (int i = threadIdx.x; i < IL_TILE_R * IN; i += blockDim.x) { const int c = i >> 4, r = i & 15; W[ws.c_x0 + (size_t)c * B + row0 + r] = Xs[r * ldx + c]; }  // x0^T [IN][B]
  tile_fwd<PANEL>(Xs, ldx, INp, p.W1, IN, IN, H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb1 : p.b1[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H1s[(4 * g + r) * ldh + col] = hv[r]; }
    if (!is_target) { wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.c_h1 + (int64_t)k * B * H + (int64_t)col * B + row0 + 4 * g, hv); debug_mask4(d, 2 + 2 * k, row0 + 4 * g, col, hv); }
  });
  __syncthreads();
  IL_TL(8, 2);
  tile_fwd_packed<PANEL>(H1s, ldh, H, W + (is_target ? ws.pk_tf : ws.pk_cf) + (size_t)k * H * H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb2 : p.b2[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H2s[(4 * g + r) * ldh + col] = hv[r]; }
    if (!is_target) { wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.c_h2 + (int64_t)k * B * H + (int64_t)col * B + row0 + 4 * g, hv); debug_mask4(d, 3 + 2 * k, row0 + 4 * g, col, hv); }
  });
  __syncthreads();
  IL_TL(8, 3);
  for (int r = wave; r < IL_TILE_R; r += nw) {   // Q = h2 . w3 + b3: one wave per row, w3 from the registers loaded at the top
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int n = lane + 64 * u; if (n < H) sq += H2s[r * ldh + n] * w3v[u]; }
    sq = wave_sum(sq);
    if (lane == 0) { W[(is_target ? ws.t_q : ws.c_q) + (size_t)k * B + row0 + r] = sq + pb3; q16[r] = sq + pb3; }
  }
  IL_TL_END(8);
}

template <int PANEL>
__device__ __forceinline__ void k_critic_fwd_body(il_sac d, il_batch b, const il_sac* __restrict__ dL, const il_batch* __restrict__ bL, float* smem) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (dL) { pop_ids(bx, by); d = dL[by]; b = bL[by]; }
  globalize(d); globalize(b);
  int net, tile;
  xcd_tile_net(bx, d.batch / IL_TILE_R, 4, tile, net);
  critic_fwd_tile<PANEL>(d, b, net, tile, smem, nullptr);
}

// ---------------------------------------------------------------------------------------------
// critic backward (training.py:24-30): y, dQ_k = w * 2 (Q_k - y) / B, dz2 = dQ w3 [h2>0], dz1 = (dz2 . W2) [h1>0].  grid = nt * 2
// ---------------------------------------------------------------------------------------------
template <int PANEL>
__device__ __forceinline__ void k_critic_bwd_body(il_sac d, il_batch b, const il_sac* __restrict__ dL, const il_batch* __restrict__ bL, float* smem) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (dL) { pop_ids(bx, by); d = dL[by]; b = bL[by]; }
  globalize(d); globalize(b);
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int nt = B / IL_TILE_R;
  int k, tile;
  xcd_tile_net(bx, nt, 2, tile, k);
  IL_TL(9, 0);
  const int row0 = tile * IL_TILE_R;
  const int ldh = H + 4;
  float* DZ2s = smem; float* dz3s = DZ2s + IL_TILE_R * ldh;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView p = mlp_view(d.critic + k * net_stride(IN, H, 1), IN, H, 1);
  const float* h2 = W + ws.c_h2 + (size_t)k * B * H; const float* h1 = W + ws.c_h1 + (size_t)k * B * H;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  // Operands that do not depend on this kernel's own results are requested up front, so their latency overlaps the dQ phase and the
  // MFMA loop instead of following a barrier: this thread's (feature, 4 rows) lane of h2 with its w3, and its epilogue lane of h1.
  const int pn = threadIdx.x >> 2, pr4 = (threadIdx.x & 3) * 4;
  const bool pre = blockDim.x == 4 * H;   // one (feature, row group) item per thread: true for every supported hidden size
  f32x4 hv2 = zero4(); float w3p = 0.f;
  if (pre) { hv2 = gload4(h2 + (size_t)pn * B + row0 + pr4); w3p = gload(p.W3 + pn); }
  const f32x4 hv1 = gload4(h1 + (size_t)min(wave * 16 + j, H - 1) * B + row0 + 4 * g);
  if (d.sync) {   // the rewards come from the discriminator branch on another stream: ready once every reward workgroup of THIS update has signalled
    long long* sy = reinterpret_cast<long long*>(d.sync);
    sync_wait_leader(sy, IL_SYNC_REWARDS, (sync_read(sy, IL_SYNC_MAIN_EPOCH) + 1) * (long long)nt);   // (leader: nothing on the device reads the rewards array between this launch's start and the relabel kernel's stores)
  }
  if (threadIdx.x < IL_TILE_R) {
    const int row = row0 + threadIdx.x;
    const float alpha = expf(d.log_alpha[0]);
    const float m = 1.f - b.absorbing[(size_t)row * b.ld_absorbing];
    const float tv = fminf(W[ws.t_q + row], W[ws.t_q + B + row]) - m * alpha * W[ws.n_logp2 + row];
    const float y = b.rewards[(size_t)row * b.ld_rewards] + (1.f - b.terminals[(size_t)row * b.ld_terminals]) * d.discount * tv;
    const float q = W[ws.c_q + (size_t)k * B + row];
    const float dq = (b.weights[(size_t)row * b.ld_weights] * (2.f * (q - y))) / (float)B;
    dz3s[threadIdx.x] = dq;
    W[ws.c_dz3 + (size_t)k * B + row] = dq;
  }
  if (bx == 0 && threadIdx.x == 64) adam_tick(d.critic_opt);  // consumed by the following k_dw_adam / il_adam_step (a lane that is idle in this phase)
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) {  // (feature n, 4 consecutive rows) per thread: 16-byte lanes of the [H][B] layout
    const int n = i >> 2, r4 = (i & 3) * 4;
    const f32x4 hv = pre ? hv2 : *reinterpret_cast<const f32x4*>(h2 + (size_t)n * B + row0 + r4);
    const float w3 = pre ? w3p : p.W3[n];
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float m = hv[q] > 0.f ? w3 : 0.f; DZ2s[(r4 + q) * ldh + n] = m; o[q] = dz3s[r4 + q] * m; }
    wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.c_dz2 + (int64_t)k * B * H + (int64_t)n * B + row0 + r4, o);
  }
  __syncthreads();
  // dz1 = dQ * ([h1 > 0] (m . W2)) with m = [h2 > 0] w3: the row factor dQ is applied AFTER the GEMM, so that k_sac_chain can run the GEMM
  // before the rewards (hence dQ) exist; every path uses this order, which keeps them bit-identical to each other.
  tile_bwd_packed<PANEL>(DZ2s, ldh, H, W + ws.pk_cb + (size_t)k * H * H, [&](int kb, f32x4 acc) {
    const size_t off = (size_t)(kb + j) * B + row0 + 4 * g;
    const f32x4 hv = (kb == wave * 16) ? hv1 : *reinterpret_cast<const f32x4*>(h1 + off);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = dz3s[4 * g + r] * (hv[r] > 0.f ? acc[r] : 0.f);
    wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.c_dz1 + (int64_t)k * B * H + (int64_t)off, o);
  });
  IL_TL_END(9);
}

// Critic-loss backward of (critic k, tile) continuing from critic_fwd_tile in the SAME workgroup: h1, h2 and Q are still in LDS, so nothing is
// re-read from HBM. Same arithmetic, in the same order, as k_critic_bwd, in two parts: `_gemm` needs neither the targets nor the rewards
// (G = [h1 > 0] ((w3 [h2 > 0]) . W2), left in LDS over h1; m = w3 [h2 > 0] over h2) and runs before the waits; `_scale` forms dQ and scales.
__device__ __forceinline__ void critic_bwd_resident_gemm(const il_sac& d, int k, float* smem) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh;
  const SacWs ws = sac_ws(S, A, H, B);
  const MlpView p = mlp_view(d.critic + k * net_stride(IN, H, 1), IN, H, 1);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const bool pre = blockDim.x == 4 * H;
  const float w3p = pre ? gload(p.W3 + (threadIdx.x >> 2)) : 0.f;
  __syncthreads();   // the Q pass of critic_fwd_tile has read h2
  for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) {
    const int n = i >> 2, r4 = (i & 3) * 4;
    const float w3 = pre ? w3p : p.W3[n];
#pragma unroll
    for (int q = 0; q < 4; ++q) { float* h = H2s + (r4 + q) * ldh + n; *h = *h > 0.f ? w3 : 0.f; }
  }
  __syncthreads();
  tile_bwd_packed(H2s, ldh, H, d.workspace + ws.pk_cb + (size_t)k * H * H, [&](int kb, f32x4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { float* h = H1s + (4 * g + r) * ldh + kb + j; *h = *h > 0.f ? acc[r] : 0.f; }   // each element owned by one lane
  });
}
// relabel: the rewards of this tile are the discriminator `dd`'s prediction on (s, a) - the rows still sit in Xs - computed here once its AdamW step of
// this update is complete ([IL_SYNC_PARAMS], n_reduce workgroups per step); otherwise dense `rewards` or the batch's own reward field.
struct ChainRelabel { il_disc dd; int on, n_reduce; float* out; int fwd_only; int wait_indices; int local_rewards; int xcd_nets, gather_wgs; int early_draw; int overlap; long long ov_n; };   // overlap: 0 = off, else the grid size of the actor optimiser launch this launch waits for   // overlap (k_sac_chain_pair, il_sac_update_gather_overlap): the previous update's actor optimiser launch may still be running on the other stream - ov_n = this stage's own epoch (ov_own, set by the kernel) stands in for [IL_SYNC_MAIN_EPOCH] and the roles wait for [IL_SYNC_OV_EPOCH + IL_OV_DWA] >= ov_n where they first need what that launch writes   // early_draw: publish [IL_SYNC_CHAIN_WGS] (IL_EARLY_DRAW=0: the resident sampler waits for the previous update's end, as in round 4)   // xcd_nets: grid = 8 * nt, one network per XCD (chain_decode_xcd); gather_wgs: row-copy workgroups among the blocks of XCDs 6, 7   // wait_indices: IL_FLAG_SAC_WAIT_INDICES; local_rewards: il_sac_update_gather without `rewards` / `relabel` (the ring's reward field: nothing to wait for)   // fwd_only: the forward kernels only (IL_FLAG_SAC_FORWARD_ONLY / data-parallel phase 0): no critic backward
// Runs between the critic's own work and its wait for the targets: the discriminator's step usually lands while the targets are still being computed,
// so the relabel stays off the critical path. Leaves the tile's rewards in LDS (rew16) for critic_bwd_resident_scale.
__device__ __forceinline__ void critic_relabel_tile(const il_sac& d, const ChainRelabel& rl, int k, int tile, float* smem) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  const int row0 = tile * IL_TILE_R, INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* q16 = Xs + IL_TILE_R * ldx + 2 * IL_TILE_R * ldh; float* rew16 = q16 + 2 * IL_TILE_R;
  long long* sy = reinterpret_cast<long long*>(d.sync);
  sync_wait_leader(sy, IL_SYNC_PARAMS, (sync_read(sy, IL_SYNC_MAIN_EPOCH) + 1) * (long long)rl.n_reduce);   // (leader: the stepped parameters are read below the caches, disc_reward_tile<.., true>)
  IL_TL(4, 0);   // [4]: the moment the discriminator's step became visible to this critic workgroup
  const RewardLds R = reward_carve(q16 + 64, rl.dd.state_dim + rl.dd.action_dim, rl.dd.hidden);
  disc_reward_tile<3, true>(rl.dd, R, Xs, ldx, IL_TILE_R, nullptr, row0, [&](int r, float reward, float) {
    rew16[r] = reward;
    if (k == 0 && rl.out) rl.out[row0 + r] = reward;
  });
}
// per-row scalars of the critic loss that depend on neither the targets nor the rewards: requested before the wait for the targets (through il_batch.gather they
// are two dependent global loads, index then ring row)
struct RowScalars { float m, not_done, weight, reward; };
__device__ __forceinline__ RowScalars critic_row_scalars(const il_batch& b, bool reward_ready, int tile) {
  RowScalars rs = {0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < IL_TILE_R) {
    const int row = tile * IL_TILE_R + threadIdx.x;
    const size_t sr = brow(b, row);
    rs.m = 1.f - b.absorbing[sr * b.ld_absorbing];
    rs.not_done = 1.f - b.terminals[sr * b.ld_terminals];
    rs.weight = b.weights[sr * b.ld_weights];
    if (reward_ready) rs.reward = b.rewards[sr * b.ld_rewards];   // (rewards that another stream is still relabelling are read after their wait)
  }
  return rs;
}
__device__ __forceinline__ void critic_bwd_resident_scale(const il_sac& d, const il_batch& b, const float* __restrict__ rewards, const ChainRelabel& rl, const RowScalars& rs, int k, int tile,
                                                          float* smem, const float* __restrict__ rew_ws = nullptr, bool through = false) {   // through: the targets' Q, log pi(a'|s') and the rewards were written THROUGH by their producers in this launch (pair mode): read below the L1, no acquire needed   // rew_ws: the tile's rewards were predicted by the relabel role of this launch (pair mode), visible behind the tile counter's acquire
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int nt = B / IL_TILE_R, row0 = tile * IL_TILE_R;
  const int INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* q16 = H2s + IL_TILE_R * ldh; float* dz3s = q16 + IL_TILE_R;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  float* rew16 = dz3s + IL_TILE_R;   // filled by critic_relabel_tile when rl.on
  if (!rl.on && d.sync && !rl.local_rewards) {   // rewards come from the discriminator branch on another stream (see k_critic_bwd); the ring's own rewards (SAC / PWIL plans) need no hand-off
    long long* sy = reinterpret_cast<long long*>(d.sync);
    sync_wait_leader(sy, IL_SYNC_REWARDS, (sync_read(sy, IL_SYNC_MAIN_EPOCH) + 1) * (long long)nt);   // (leader: nothing on the device reads the rewards array between this launch's start and the relabel kernel's stores)
  }
  if (threadIdx.x < IL_TILE_R) {
    const int row = row0 + threadIdx.x;
    const float alpha = expf(d.log_alpha[0]);
    const float m = rs.m;
    const float tq1 = through ? sload1(W, ws.t_q + row) : W[ws.t_q + row], tq2 = through ? sload1(W, ws.t_q + B + row) : W[ws.t_q + B + row];
    const float lp2 = through ? sload1(W, ws.n_logp2 + row) : W[ws.n_logp2 + row];
    const float tv = fminf(tq1, tq2) - m * alpha * lp2;
    const float rew = rew_ws ? (through ? sload1(rew_ws, row) : rew_ws[row]) : (rl.on ? rew16[threadIdx.x] : (rewards ? rewards[row] : (d.sync ? b.rewards[brow(b, row) * b.ld_rewards] : rs.reward)));
    const float y = rew + rs.not_done * d.discount * tv;
    const float q = q16[threadIdx.x];
    const float dq = (rs.weight * (2.f * (q - y))) / (float)B;
    dz3s[threadIdx.x] = dq;
    W[ws.c_dz3 + (size_t)k * B + row] = dq;
#ifdef IL_EXP_CHECK
    if (through && row < 4096) { il_chk_rew[k][row] = rew; il_chk_tq[k][0][row] = tq1; il_chk_tq[k][1][row] = tq2; il_chk_tq[k][2][row] = lp2; }
#endif
  }
  if (k == 0 && tile == 0 && threadIdx.x == 64) adam_tick(d.critic_opt);
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) {  // (feature n, 4 consecutive rows) per thread: 16-byte lanes of the [H][B] layout
    const int n = i >> 2, r4 = (i & 3) * 4;
    f32x4 o2, o1;
#pragma unroll
    for (int q = 0; q < 4; ++q) { o2[q] = dz3s[r4 + q] * H2s[(r4 + q) * ldh + n]; o1[q] = dz3s[r4 + q] * H1s[(r4 + q) * ldh + n]; }
    wstore4<(IL_WT_TILE_STORES != 0)>(W, ws.c_dz2 + (int64_t)k * B * H + (int64_t)n * B + row0 + r4, o2);
    wstore4<(IL_WT_TILE_STORES != 0)>(W, ws.c_dz1 + (int64_t)k * B * H + (int64_t)n * B + row0 + r4, o1);
  }
}

// Measured in round 2 and NOT kept (profiles/r02_update_timeline.md): a hidden layer of a tile takes 5.2 us against 3.4 us of MFMA issue at the nominal clock, and that gap is
// not a cold first touch of the weights the previous Adam kernel rewrote: (a) warmer workgroups on idle CUs that pull every panel of an XCD's roles into its L2 at launch
// (2 - 4 per XCD): layer times and updates/s unchanged; (b) the same lines requested early by the role workgroups themselves: slower (their first loads queue behind them);
// (c) the panel (or half of it) in registers across the first layer: over the 128-VGPR budget of a 16-wave workgroup, spills.
// Also measured and not kept: consecutive updates on two ALTERNATING main streams, the first kernel of update n waiting on the device for every workgroup of update n-1's
// last kernel (a counter) instead of following it in stream order, to take the ~5 us hipGraph boundary between two launches on one stream off the critical path:
// 12.9-13.0k against 14.03k updates/s on the same box (the resident, spinning launch of the next update costs the running one more than the boundary it hides).
// k_sac_chain: both actor forwards, the four critic / target forwards and the critic-loss backward of ONE learner in one launch of 6 * nt
// workgroups, chained per 16-row TILE instead of per kernel:   actor(s') [tile] -> target_1,2(s', a') [tile] -> critic_1,2 backward [tile],
// while actor(s) and the critic forwards, which depend on nothing, run beside them. Versus the three launches (k_actor_fwd, k_critic_fwd,
// k_critic_bwd) the critical path loses two kernel boundaries, the waiting workgroups have their weights' first lanes and biases in
// flight already, and the critic backward never re-reads h1 / h2. Roles are laid out in block order actor(s'), targets, critics, actor(s):
// a workgroup only waits for lower-numbered ones, so in-order dispatch keeps the waits deadlock-free (the grid is co-resident anyway:
// 6 * nt <= number of CUs is checked by the caller).
__device__ __forceinline__ void chain_decode(int bid, int nt, int& role, int& net, int& tile) {
  if ((nt & 7) == 0) {   // XCD-aware (workgroup b runs on XCD b % 8): each critic-shaped network's workgroups share 4 XCDs
    const int x = bid & 7, q = bid >> 3, ra = nt >> 3, rc = nt >> 2;
    if (q < ra) { role = 0; net = 0; tile = q * 8 + x; }
    else if (q < ra + 2 * rc) { role = q < ra + rc ? 1 : 2; net = x >> 2; tile = (x & 3) * rc + (q - ra) % rc; }
    else { role = 3; net = 0; tile = (q - ra - 2 * rc) * 8 + x; }
  } else {
    role = bid < nt ? 0 : (bid < 3 * nt ? 1 : (bid < 5 * nt ? 2 : 3));
    const int l = bid - (role == 0 ? 0 : (role == 1 ? nt : (role == 2 ? 3 * nt : 5 * nt)));
    net = l / nt; tile = l - net * nt;
  }
}
// One network per XCD (grid = 8 * nt; workgroup b runs on XCD b % 8): the nt tile workgroups of a role-network all sit on the same XCD, so its weights cross the fabric
// once instead of once per XCD that hosts one of its tiles (chain_decode: 8x for the actor roles, 4x for the critic-shaped ones - 11 MB of reads per launch against
// 1.7 MB of weights). XCD 0: actor(s'), 1-2: targets, 3-4: critics, 5: actor(s), 6-7: the row-copy workgroups (the rest of their blocks exit at once). A role still only
// waits for lower-numbered workgroups (tile q: 8q < 8q + 1, 8q + 2 < 8q + 3, 8q + 4).
__device__ __forceinline__ void chain_decode_xcd(int bid, int& role, int& net, int& tile) {
  const int x = bid & 7;
  tile = bid >> 3;
  if (x == 0) { role = 0; net = 0; }
  else if (x <= 2) { role = 1; net = x - 1; }
  else if (x <= 4) { role = 2; net = x - 3; }
  else { role = 3; net = 0; }
}
// b.gather != NULL (il_sac_update_gather): the batch has been drawn but not gathered. Every role reads its rows straight from the ring through the
// indices, and the workgroups behind the 6 * nt chain roles copy the rows to `rows_out` for the later kernels of the update (one 16-byte
// lane per thread, [IL_SYNC_ROWS] += 1 per workgroup) - they wait for nothing and nobody in this launch waits for them.
__device__ __forceinline__ void sac_chain_body(il_sac& d, il_batch& b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, const float* __restrict__ rewards,
                                               float* __restrict__ rows_out, ChainRelabel& rl, float* smem) {
  const int nt = d.batch / IL_TILE_R;
  // resident sampler (il_replay_draw_resident on the other stream): this update's indices are signalled, not stream-ordered. This launch follows the previous
  // update's last kernel in its stream, so [IL_SYNC_MAIN_EPOCH] already counts that update; the draw usually finished while this launch was being dispatched.
  const int bid = blockIdx.x;
  int gw = -1, G = 0;   // row-copy workgroup index / count
  if (rl.xcd_nets) { if ((bid & 7) >= 6) { gw = 2 * (bid >> 3) + (bid & 7) - 6; G = rl.gather_wgs; if (gw >= G) return; } }
  else if (bid >= 6 * nt) { gw = bid - 6 * nt; G = (int)gridDim.x - 6 * nt; }
  IL_TL(0, 0);
  if (rl.wait_indices) { long long* sy = reinterpret_cast<long long*>(d.sync); sync_wait_leader(sy, IL_SYNC_INDICES, sync_read(sy, IL_SYNC_MAIN_EPOCH) + 1); }   // (leader: see k_sac_chain_pair)
  IL_TL(0, 1);
  if (gw >= 0) {
    const int row4 = b.ld_states / 4, lanes = d.batch * row4;
    const f32x4* src = reinterpret_cast<const f32x4*>(b.states);
    f32x4* dst = reinterpret_cast<f32x4*>(as_global(rows_out));
    for (int i = gw * blockDim.x + threadIdx.x; i < lanes; i += G * blockDim.x) {
      const int r = i / row4, c = i - r * row4;
      dst[i] = src[brow(b, r) * row4 + c];
    }
    if (d.sync) sync_signal(reinterpret_cast<long long*>(d.sync) + IL_SYNC_ROWS);
    IL_TL(0, 7);
    return;
  }
  int role, net, tile;
  if (rl.xcd_nets) chain_decode_xcd(bid, role, net, tile); else chain_decode(bid, nt, role, net, tile);
  const SacWs ws = sac_ws(d.state_dim, d.action_dim, d.hidden, d.batch);
  unsigned* ctr = reinterpret_cast<unsigned*>(d.workspace + ws.chain_ctr) + tile * IL_CTR_STRIDE;
  if (role == 0) { actor_fwd_tile(d, b, eps_next, eps_cur, false, tile, smem); IL_TL(0, 6); tile_arrive(ctr); IL_TL(0, 7); }
  else if (role == 1) {
    critic_fwd_tile(d, b, 2 + net, tile, smem, ctr);
    IL_TL(0, 6);
    if (!rl.fwd_only) { tile_arrive(ctr); IL_TL(0, 7); }
    else {   // nobody waits for the targets in this launch: the second target workgroup of the tile leaves the counter at 0 for the next one
      sync_drain_stores();
      __syncthreads();
      if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == 2u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  else if (role == 2) {
    critic_fwd_tile(d, b, net, tile, smem, nullptr);
    IL_TL(0, 2);
    if (rl.fwd_only) return;
    critic_bwd_resident_gemm(d, net, smem);
    IL_TL(0, 3);
    if (rl.on) critic_relabel_tile(d, rl, net, tile, smem);
    IL_TL(0, 4);
    const RowScalars rs = critic_row_scalars(b, !rl.on && !rewards && !d.sync, tile);
    tile_await(ctr, 3u, tile_timeouts(d));
    IL_TL(0, 5);
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 4u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    critic_bwd_resident_scale(d, b, rewards, rl, rs, net, tile, smem);
    IL_TL(0, 7);
  } else { actor_fwd_tile(d, b, eps_next, eps_cur, true, tile, smem); IL_TL(0, 7); }
}

__global__ __launch_bounds__(1024) void k_sac_chain(il_sac d, il_batch b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, const float* __restrict__ rewards,
                                                    float* __restrict__ rows_out, ChainRelabel rl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  globalize(d); globalize(b);
  if (rl.on) { globalize(rl.dd); rl.out = as_global(rl.out); }
  chain_publish_grid(d, rl.early_draw);
  sac_chain_body(d, b, eps_next, eps_cur, rewards, rows_out, rl, smem);
  __syncthreads();
  chain_done(d, rl.early_draw);
}

// Population launch of the chain: grid (6 * nt, learners). Workgroups are dispatched in linear order (x fastest), a role only waits for lower-numbered workgroups of ITS
// learner (block order actor(s') < targets < critics < actor(s)), so the waits are deadlock-free without the whole grid being co-resident: a producer is always dispatched
// before its consumers. Replaces k_actor_fwd + k_critic_fwd + k_critic_bwd of the population path (two launches and the h1 / h2 round trip of the critics less).
__global__ __launch_bounds__(1024) void k_sac_chain_pop(const il_sac* __restrict__ dL, const il_batch* __restrict__ bL) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  il_sac d = dL[blockIdx.y]; il_batch b = bL[blockIdx.y];
  globalize(d); globalize(b);
  ChainRelabel rl = {};
  sac_chain_body(d, b, nullptr, nullptr, nullptr, nullptr, rl, smem);
}

// ---------------------------------------------------------------------------------------------
// Pair mode of the chained launch (round 4; mlp_tile.hpp "Pair mode"). k_sac_chain's critical path is actor(s') -> targets -> critic-loss scale, two 16 x 256 x 256
// layers of 5 us each on one CU apiece, with the inline relabel (5 us behind the discriminator's step) on the critic workgroups' own path. Here, with 512-thread
// workgroups (H = 256, round_up16(S + A) <= 64):
//   * actor(s') and each target critic of a tile are a PAIR of workgroups that split the hidden layer by output columns (half p = 1 sends its 16 x 128 half of h2 and
//     exits; half 0 continues with the head / Q). Narrow first layers are computed by both halves; their weight lanes are requested before the rows are in LDS, the hidden
//     layer's panel before the first layer runs;
//   * the reward relabel of a tile is a role of its own (it waits for [IL_SYNC_PARAMS] beside the critics instead of inside them) and arrives on the tile counter with
//     += 16; the critic-loss workgroups wait ONCE, for both targets and the rewards;
//   * critics and actor(s) run the same tile functions as k_sac_chain with 8 waves (two column tiles per wave; off the critical path).
// Block order: actor(s') p = 1, p = 0 | targets p = 1, p = 0 | relabel | critics | actor(s) | row copies: a workgroup still only waits for lower-numbered ones, and a
// pair's halves share block id mod 8 (same XCD: the hop stays short; placement is speed only). Every element keeps its summation order: bit-identical to k_sac_chain.
// ---------------------------------------------------------------------------------------------
struct PairIds { int role, net, tile, half; };   // role 0 actor(s'), 1 target, 2 critic, 3 actor(s), 4 relabel, 5 row copy
__device__ __forceinline__ void seg_decode(int l, int nt, int n_nets, int& net, int& tile) {   // l in [0, n_nets * nt): XCD-aware when nt % 8 == 0 (a network's tiles share 8 / n_nets XCDs)
  if ((nt & 7) == 0 && n_nets == 2) { const int x = l & 7, q = l >> 3, rc = nt >> 2; net = x >> 2; tile = (x & 3) * rc + q; }
  else if ((nt & 7) == 0) { net = 0; tile = l; }
  else { net = l / nt; tile = l - net * nt; }
}
__device__ __forceinline__ PairIds chain_pair_decode(int bid, int nt, int relabel) {
  PairIds r = {5, 0, 0, 0};
  int l = bid;
  if (l < 2 * nt) { r.role = 0; r.half = l < nt ? 1 : 0; seg_decode(l % nt, nt, 1, r.net, r.tile); return r; }
  l -= 2 * nt;
  if (l < 4 * nt) { r.role = 1; r.half = l < 2 * nt ? 1 : 0; seg_decode(l % (2 * nt), nt, 2, r.net, r.tile); return r; }
  l -= 4 * nt;
  if (relabel) { if (l < nt) { r.role = 4; r.tile = l; return r; } l -= nt; }
  if (l < 2 * nt) { r.role = 2; seg_decode(l, nt, 2, r.net, r.tile); return r; }
  l -= 2 * nt;
  if (l < nt) { r.role = 3; r.tile = l; return r; }
  r.tile = l - nt;   // row-copy workgroup index
  return r;
}
__host__ __device__ static inline int chain_pair_workgroups(int nt, int relabel, int G) { return (relabel ? 10 : 9) * nt + G; }

// LDS of a pair-mode tile: the tile kernels' carve (tile_lds_bytes) followed by W1s[H][Kpad + 4]
__device__ __forceinline__ float* pair_w1s(float* smem, int in_pad, int H) { return smem + IL_TILE_R * (in_pad + 4) + 2 * IL_TILE_R * (H + 4) + (H >> 4) * 256 + 256 + 64; }
// actor(s') of one tile as a pair (reference models.py:90-94 on next_states; training.py:21): the arithmetic of actor_fwd_tile(is_cur = false)
// ov >= 0 (overlapped launches): the previous update's actor optimiser launch may still be running. The row indices and the rows are requested first; everything that launch
// writes (the actor's parameters and lane-ordered copy, the Philox counter) is requested behind the wait for its epoch.
__device__ __forceinline__ void actor_next_pair(const il_sac& d, const il_batch& b, const float* __restrict__ eps_next, int tile, int half, float* smem, float* slab, unsigned* flag, long long ov = -1, int ov_grid = 0) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch;
  const int row0 = tile * IL_TILE_R;
  const int Sp = round_up16(S), ldx = Sp + 4, ldh = H + 4, ldw1 = Sp + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* part = H2s + IL_TILE_R * ldh; float* Os = part + (H >> 4) * 256;
  float* W1s = pair_w1s(smem, Sp, H);
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView net = mlp_view(d.actor, S, H, 2 * A);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, tid = threadIdx.x;
  const int t2 = 8 * half + wave;   // this wave's output tile of the hidden layer
  if (half == 0) pair_announce(flag);
  // requests, smallest first: row indices, W1, biases, the rows, then the hidden layer's panel
  RowsPre rp; rows_idx(rp, Sp, row0, b.gather);
  const bool head_thread = half == 0 && tid < IL_TILE_R * A;
  const int hr = tid / A, hc = tid - hr * A;
  int64_t hidx = row0 + min(hr, IL_TILE_R - 1);
  if (head_thread && b.gather) hidx = gload(b.gather + row0 + hr);
  float absorb_pre = 0.f;
  if (ov >= 0) {   // rows first, then the wait; the weights behind it
    rows_issue(rp, Sp, b.next_states, b.ld_next_states, S, nullptr, 0, 0, row0, b.gather != nullptr, b.gather_capacity, false);
    if (head_thread) { if (b.gather) hidx = hidx < 0 ? 0 : (hidx >= b.gather_capacity ? b.gather_capacity - 1 : hidx); absorb_pre = gload(b.absorbing + (size_t)hidx * b.ld_absorbing); }
    ov_wait(reinterpret_cast<long long*>(d.sync), IL_OV_DWA, ov, ov_grid);
    IL_ST_GATE(IL_ST_CHAIN);
  }
  const int w1_lanes = H * S / 4;
  const bool w1_regs = l1_rows_aligned(net.W1, S);
  W1Pre w1; L1Pre w1r;
  if (w1_regs) l1_prefetch(w1r, net.W1, S, Sp, H); else w1_issue(w1, net.W1, w1_lanes);
  const float pb1a = gload(net.b1 + wave * 16 + j), pb1b = gload(net.b1 + min((wave + nw) * 16 + j, H - 1)), pb2 = gload(net.b2 + t2 * 16 + j);
  const uint32_t nctr = (half == 0 && d.noise_counter) ? gload(d.noise_counter) : 0u;
  __builtin_amdgcn_sched_barrier(0);
  if (ov < 0) {
    rows_issue(rp, Sp, b.next_states, b.ld_next_states, S, nullptr, 0, 0, row0, b.gather != nullptr, b.gather_capacity, false);
    if (head_thread) { if (b.gather) hidx = hidx < 0 ? 0 : (hidx >= b.gather_capacity ? b.gather_capacity - 1 : hidx); absorb_pre = gload(b.absorbing + (size_t)hidx * b.ld_absorbing); }
  }
  SmallPre w3pre = {};
  if (half == 0) w3pre = tile_fwd_small_prefetch(net.W3, H, 2 * A, H);
  // Only the FIRST HALF of the hidden layer's panel is requested here (panel_prefetch_lo), behind every wave's small loads (issue_fence): a wave that issues 16 KB of
  // loads is held at the issue stage until the CU's memory pipeline has taken them - with eight waves doing so ~0.85 us per panel, during which it cannot commit its rows
  // (two whole panels parked in the prologue of the first pair-mode k_policy_critic: a 4.8 us prologue). The second half goes out right before the MFMAs and streams in
  // under the first half's.
  issue_fence();
  Panel16 pn; panel_prefetch_lo(pn, W + ws.pk_af, t2);
  IL_TL(10, 1);
  if (!w1_regs) w1_commit(w1, W1s, ldw1, S, Sp, H, w1_lanes);
  rows_commit(rp, Xs, ldx, Sp, S);
  __syncthreads();
  IL_TL(10, 2);
  {
    auto epi1 = [&](int c0, f32x4 acc) {
      const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb1a : pb1b;
#pragma unroll
      for (int r = 0; r < 4; ++r) H1s[(4 * g + r) * ldh + col] = fmaxf(acc[r] + bb, 0.f);
    };
    if (w1_regs) l1_compute_regs(w1r, Xs, ldx, Sp, H, epi1); else l1_compute_lds(W1s, ldw1, Xs, ldx, Sp, H, epi1);
  }
  const bool near = half == 1 && pair_same_xcd(flag);
  __syncthreads();
  IL_TL(10, 3);
  panel_prefetch_hi(pn, W + ws.pk_af, t2);
  tile_packed_regs(H1s, ldh, pn, t2, [&](int c0, f32x4 acc) {
    const int col = c0 + j;
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = fmaxf(acc[r] + pb2, 0.f);
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) H2s[(4 * g + r) * ldh + col] = hv[r];
    } else pair_store(slab, (int64_t)(col - 128) * 16 + 4 * g, hv, near);
  });
  IL_TL(10, 4);
  if (half == 1) { pair_publish(flag); IL_TL(10, 7); return; }
  // the noise of the head (Philox + Box-Muller: ~1 us of dependent ALU work) is drawn while the partner's half is on its way
  float e_pre = 0.f;
  if (head_thread) e_pre = eps_next ? eps_next[(size_t)(row0 + hr) * A + hc] : philox_normal(d.noise_seed, nctr, IL_STREAM_EPS_NEXT, (uint32_t)((row0 + hr) * A + hc));
  const TileTimeouts tmo = tile_timeouts(d);
  pair_receive(flag, slab, H2s, ldh, 128, [&] { __hip_atomic_fetch_add(tmo.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (tmo.sync) sync_timed_out(tmo.sync); });
  IL_TL(10, 5);
  tile_fwd_small(H2s, ldh, H, net.W3, H, 2 * A, net.b3, Os, part, &w3pre);
  float* nl = part; float* la = part + 256;
  if (head_thread) {
    const int row = row0 + hr;
    const float mean = Os[hr * 16 + hc], lsr = Os[hr * 16 + A + hc];
    float x, a, nlp, ladj;
    head_sample(mean, lsr, e_pre, x, a, nlp, ladj);
    nl[hr * 16 + hc] = nlp; la[hr * 16 + hc] = ladj;
    wstore1(W, ws.n_a2 + (int64_t)row * A + hc, (1.f - absorb_pre) * a);   // (written through: the targets and the critic-loss workgroups read these without an acquire)
  }
  __syncthreads();
  if (tid < IL_TILE_R) {
    float sn = 0.f, sl = 0.f;
    for (int c = 0; c < A; ++c) { sn += nl[tid * 16 + c]; sl += la[tid * 16 + c]; }
    wstore1(W, ws.n_logp2 + row0 + tid, (0.f - sl) + sn);
  }
  IL_TL(10, 6);
}

// target_k(s', a') of one tile as a pair (training.py:22): the arithmetic of critic_fwd_tile(net = 2 + k, await)
__device__ __forceinline__ void target_pair(const il_sac& d, const il_batch& b, int k, int tile, int half, float* smem, float* slab, unsigned* flag, unsigned* ctr, long long ov = -1, int ov_grid = 0) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int row0 = tile * IL_TILE_R;
  const int INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* W1s = pair_w1s(smem, INp, H);
  const int ldw1 = INp + 4, w1_lanes = H * IN / 4;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView p = mlp_view(d.target + k * net_stride(IN, H, 1), IN, H, 1);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int t2 = 8 * half + wave;
  if (half == 0) pair_announce(flag);
  RowsPre rp; rows_idx(rp, INp, row0, b.gather);
  if (ov >= 0) {   // overlapped launches: the target network is being stepped by the previous update's tail - rows first, its parameters behind the wait
    rows_issue(rp, INp, b.next_states, b.ld_next_states, S, nullptr, 0, 0, row0, b.gather != nullptr, b.gather_capacity, false);
    ov_wait(reinterpret_cast<long long*>(d.sync), IL_OV_DWA, ov, ov_grid);
    IL_ST_GATE(IL_ST_CHAIN);
  }
  const bool w1_regs = l1_rows_aligned(p.W1, IN);   // (wave-uniform) aligned rows: operand lanes straight into registers; otherwise through LDS
  W1Pre w1; L1Pre w1r;
  if (w1_regs) l1_prefetch(w1r, p.W1, IN, INp, H); else w1_issue(w1, p.W1, w1_lanes);
  const float pb1a = gload(p.b1 + wave * 16 + j), pb1b = gload(p.b1 + min((wave + nw) * 16 + j, H - 1)), pb2 = gload(p.b2 + t2 * 16 + j), pb3 = gload(p.b3);
  float w3v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) w3v[u] = gload(p.W3 + min(lane + 64 * u, H - 1));
  __builtin_amdgcn_sched_barrier(0);
  if (ov < 0) rows_issue(rp, INp, b.next_states, b.ld_next_states, S, nullptr, 0, 0, row0, b.gather != nullptr, b.gather_capacity, false);   // s' columns, zero elsewhere
  issue_fence();
  Panel16 pn; panel_prefetch(pn, W + ws.pk_tf + (size_t)k * H * H, t2);   // (this workgroup is about to wait for its tile's actor(s'): being held at the issue stage costs nothing here)
  if (!w1_regs) w1_commit(w1, W1s, ldw1, IN, INp, H, w1_lanes);
  rows_commit(rp, Xs, ldx, INp, S);
  IL_TL(10, 1);
  tile_await_bits<false>(ctr, 1u, 0u, tile_timeouts(d));   // a' of this tile, written through by its actor(s') workgroup (the barrier also covers the LDS writes above)
  IL_TL(10, 2);
  for (int i = threadIdx.x; i < IL_TILE_R * A; i += blockDim.x) { const int r = i / A, c = i - r * A; Xs[r * ldx + S + c] = sload1(W, ws.n_a2 + (int64_t)(row0 + r) * A + c); }
#ifdef IL_EXP_CHECK
  if (A <= 8) for (int i = threadIdx.x; i < IL_TILE_R * A; i += blockDim.x) { const int r = i / A, c = i - r * A; if (row0 + r < 4096) il_chk_a2[2 * k + half][(row0 + r) * 8 + c] = Xs[r * ldx + S + c]; }
#endif
  __syncthreads();
  {
    auto epi1 = [&](int c0, f32x4 acc) {
      const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb1a : pb1b;
#pragma unroll
      for (int r = 0; r < 4; ++r) H1s[(4 * g + r) * ldh + col] = fmaxf(acc[r] + bb, 0.f);
    };
    if (w1_regs) l1_compute_regs(w1r, Xs, ldx, INp, H, epi1); else l1_compute_lds(W1s, ldw1, Xs, ldx, INp, H, epi1);
  }
  __syncthreads();
  IL_TL(10, 3);
  const bool near = half == 1 && pair_same_xcd(flag);
  tile_packed_regs(H1s, ldh, pn, t2, [&](int c0, f32x4 acc) {
    const int col = c0 + j;
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = fmaxf(acc[r] + pb2, 0.f);
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) H2s[(4 * g + r) * ldh + col] = hv[r];
    } else pair_store(slab, (int64_t)(col - 128) * 16 + 4 * g, hv, near);
  });
  IL_TL(10, 4);
  if (half == 1) { pair_publish(flag); IL_TL(10, 7); return; }
  const TileTimeouts tmo = tile_timeouts(d);
  pair_receive(flag, slab, H2s, ldh, 128, [&] { __hip_atomic_fetch_add(tmo.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (tmo.sync) sync_timed_out(tmo.sync); });
  IL_TL(10, 5);
  for (int r = wave; r < IL_TILE_R; r += nw) {   // Q = h2 . w3 + b3: one wave per row, the lane / DPP order of critic_fwd_tile
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int n = lane + 64 * u; if (n < H) sq += H2s[r * ldh + n] * w3v[u]; }
    sq = wave_sum(sq);
    if (lane == 0) wstore1(W, ws.t_q + (int64_t)k * B + row0 + r, sq + pb3);
  }
  IL_TL(10, 6);
}

// the reward relabel of one tile as a role of its own (models.py:177-180 through disc_reward_tile: the code, thread mapping and bits of k_gail_reward)
#ifdef IL_EXP_CHECK   // the same tile once more, AFTER the role has arrived (its timing up to the arrival is the product's): rows and parameters re-read behind an acquire of every wave
__device__ __forceinline__ void relabel_verify(const il_sac& d, const il_batch& b, const ChainRelabel& rl, int tile, float* smem) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  const int row0 = tile * IL_TILE_R, INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* q16 = Xs + IL_TILE_R * ldx + 2 * IL_TILE_R * ldh;
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  load_rows_cat(Xs, ldx, INp, b.states, b.ld_states, S, b.actions, b.ld_actions, A, row0, IL_TILE_R, b.gather, b.gather_capacity);
  __syncthreads();
  const RewardLds R = reward_carve(q16 + 64, rl.dd.state_dim + rl.dd.action_dim, rl.dd.hidden);
  disc_reward_tile<3, true>(rl.dd, R, Xs, ldx, IL_TILE_R, nullptr, row0, [&](int r, float reward, float) {
    if (row0 + r < 4096 && __float_as_uint(il_chk_rel[row0 + r]) != __float_as_uint(reward)) atomicAdd(&il_chk[4], 1u);
  });
}
#endif
__device__ __forceinline__ void relabel_role(const il_sac& d, const il_batch& b, const ChainRelabel& rl, int tile, float* smem) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  const int row0 = tile * IL_TILE_R, INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* q16 = Xs + IL_TILE_R * ldx + 2 * IL_TILE_R * ldh;
  const SacWs ws = sac_ws(S, A, H, d.batch);
  float* W = d.workspace;
  load_rows_cat(Xs, ldx, INp, b.states, b.ld_states, S, b.actions, b.ld_actions, A, row0, IL_TILE_R, b.gather, b.gather_capacity);
  long long* sy = reinterpret_cast<long long*>(d.sync);
  IL_TL(10, 1);
  // (leader acquire: the stepped parameters - the one thing this role reads that another stream wrote during this launch - are read below the caches, disc_reward_tile<.., true>;
  // the barrier also covers the rows above)
  sync_wait_leader(sy, IL_SYNC_PARAMS, ((rl.overlap ? rl.ov_n : sync_read(sy, IL_SYNC_MAIN_EPOCH)) + 1) * (long long)rl.n_reduce);
  IL_TL(10, 2);
  const RewardLds R = reward_carve(q16 + 64, rl.dd.state_dim + rl.dd.action_dim, rl.dd.hidden);
#ifdef IL_EXP_CHECK
  {
    const long long want = ((rl.overlap ? rl.ov_n : sync_read(sy, IL_SYNC_MAIN_EPOCH)) + 1) * (long long)rl.n_reduce;
    if (threadIdx.x == 0) {
      const long long pv = __hip_atomic_load(sy + IL_SYNC_PARAMS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), sv = __hip_atomic_load(sy + IL_SYNC_SIDE_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pv < want) atomicAdd(&il_chk[1], 1u);
      if (pv > want) atomicAdd(&il_chk[2], 1u);
      if (sv * rl.n_reduce > want) atomicAdd(&il_chk[3], 1u);
    }
  }
#endif
  disc_reward_tile<3, true>(rl.dd, R, Xs, ldx, IL_TILE_R, nullptr, row0, [&](int r, float reward, float) {
    wstore1(W, ws.c_rew + row0 + r, reward);
    if (rl.out) rl.out[row0 + r] = reward;
#ifdef IL_EXP_CHECK
    if (row0 + r < 4096) il_chk_rel[row0 + r] = reward;
#endif
  });
  IL_TL(10, 6);
}

__device__ __forceinline__ void sac_chain_pair_body(il_sac& d, il_batch& b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, const float* __restrict__ rewards,
                                                    float* __restrict__ rows_out, ChainRelabel& rl, float* smem) {
  const int nt = d.batch / IL_TILE_R, H = d.hidden;
  const PairIds id = chain_pair_decode(blockIdx.x, nt, rl.on);
  IL_TL(10, 0);
  long long* sy = reinterpret_cast<long long*>(d.sync);
  const long long ov = rl.overlap ? rl.ov_n : -1;   // >= 0: [IL_SYNC_MAIN_EPOCH] may not count the previous update yet (its last launch is still running on the other stream)
  // (leader acquire: the index arrays are read by no kernel between this launch's start and the draw - the discriminator step that shares them waits for the same counter -
  // so no L2 can hold a pre-draw copy fetched during this launch; 163 workgroups x 8 waves of invalidates here cost the launch 7 us: profiles/r06_soak_under_load.md)
  if (rl.wait_indices) sync_wait_leader(sy, IL_SYNC_INDICES, (ov >= 0 ? ov : sync_read(sy, IL_SYNC_MAIN_EPOCH)) + 1);
  if (id.role == 5) {
    const int G = (int)gridDim.x - chain_pair_workgroups(nt, rl.on, 0), gw = id.tile;
    const int row4 = b.ld_states / 4, lanes = d.batch * row4;
    const f32x4* src = reinterpret_cast<const f32x4*>(b.states);
    f32x4* dst = reinterpret_cast<f32x4*>(as_global(rows_out));
    for (int i = gw * blockDim.x + threadIdx.x; i < lanes; i += G * blockDim.x) {
      const int r = i / row4, c = i - r * row4;
      dst[i] = src[brow(b, r) * row4 + c];
    }
    if (d.sync) sync_signal(reinterpret_cast<long long*>(d.sync) + IL_SYNC_ROWS);
    IL_TL(10, 7);
    return;
  }
  const SacWs ws = sac_ws(d.state_dim, d.action_dim, H, d.batch);
  unsigned* ctr = reinterpret_cast<unsigned*>(d.workspace + ws.chain_ctr) + id.tile * IL_CTR_STRIDE;
  const int slot = id.role == 0 ? id.tile : nt + id.net * nt + id.tile;   // hop slabs / flags: [0, nt) actor(s'), [nt, 3 nt) the targets
  float* slab = d.workspace + ws.x_slab + (size_t)slot * IL_TILE_R * (H / 2);
  unsigned* flag = reinterpret_cast<unsigned*>(d.workspace + ws.x_flag) + slot * IL_CTR_STRIDE;
  if (id.role == 0) {
    actor_next_pair(d, b, eps_next, id.tile, id.half, smem, slab, flag, ov, rl.overlap);
    if (id.half == 0) { tile_arrive_through(ctr, 1u); IL_TL(10, 7); }
  } else if (id.role == 1) {
    target_pair(d, b, id.net, id.tile, id.half, smem, slab, flag, ctr, ov, rl.overlap);
    if (id.half == 0) { tile_arrive_through(ctr, 1u); IL_TL(10, 7); }
  } else if (id.role == 4) {
    relabel_role(d, b, rl, id.tile, smem);
    tile_arrive_through(ctr, 16u);
#ifdef IL_EXP_CHECK
    relabel_verify(d, b, rl, id.tile, smem);
#endif
    IL_TL(10, 7);
  } else if (id.role == 2) {
    // overlapped launches: the critics could run their forward and the backward GEMM ahead of the wait (their parameters were stepped two launches ago), but measured
    // (profiles/r06_overlap_timeline.txt) that work slowed down under the other workgroups' acquires and left dirty lines in the L2s that the previous launch's releases
    // then had to write back: they wait first - they are not on the launch's critical path (actor(s') -> targets)
    if (ov >= 0) { ov_wait(sy, IL_OV_DWA, ov, rl.overlap); IL_ST_GATE(IL_ST_CHAIN); }
    critic_fwd_tile(d, b, id.net, id.tile, smem, nullptr);
    IL_TL(10, 2);
    critic_bwd_resident_gemm(d, id.net, smem);
    IL_TL(10, 3);
    const RowScalars rs = critic_row_scalars(b, !rl.on && !rewards && !d.sync, id.tile);
    tile_await_bits<false>(ctr, 3u, rl.on ? 16u : 0u, tile_timeouts(d));   // (no acquire: what this workgroup reads of the tile's producers was written through, critic_bwd_resident_scale reads it below the L1)
    IL_TL(10, 5);
    if (threadIdx.x == 0 && (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 15u) == 4u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    critic_bwd_resident_scale(d, b, rewards, rl, rs, id.net, id.tile, smem, rl.on ? d.workspace + ws.c_rew : nullptr, true);
    IL_TL(10, 7);
  } else {
    if (ov >= 0) { ov_wait(sy, IL_OV_DWA, ov, rl.overlap); IL_ST_GATE(IL_ST_CHAIN); }   // actor(s): the previous actor optimiser launch reads the buffers this role writes, and writes the weights it reads
    actor_fwd_tile(d, b, eps_next, eps_cur, true, id.tile, smem); IL_TL(10, 7);
  }
}

__global__ __launch_bounds__(512) void k_sac_chain_pair(il_sac d, il_batch b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, const float* __restrict__ rewards,
                                                        float* __restrict__ rows_out, ChainRelabel rl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  globalize(d); globalize(b);
  if (rl.on) { globalize(rl.dd); rl.out = as_global(rl.out); }
  IL_ST_BEGIN(IL_ST_CHAIN);
  chain_publish_grid(d, rl.early_draw);
  if (rl.overlap) rl.ov_n = ov_own(reinterpret_cast<long long*>(d.sync), IL_OV_CHAIN);
  sac_chain_pair_body(d, b, eps_next, eps_cur, rewards, rows_out, rl, smem);
  IL_ST_END(IL_ST_CHAIN);
  chain_done(d, rl.early_draw);
  if (rl.overlap) ov_done(reinterpret_cast<long long*>(d.sync), IL_OV_CHAIN);
}

// policy-loss backward of one 16-row tile: min-Q selection, tanh-Gaussian backward, actor back-prop (dz3, dz2, dz1 for the dW kernel), alpha partial.
// Runs as the tail of k_policy_critic in the workgroup that finishes the tile's second critic.
// (part, nparts): the last GEMM, whose result only goes to HBM for the weight-gradient kernel, is split by output columns over `nparts` workgroups
// that each run everything before it redundantly (k_policy_critic helpers); part 0 writes the shared outputs. nparts = 1: the whole tail.
// `wait` is called once everything that does not depend on the critics of this launch has been requested / computed.
// PARK (pair-mode helpers: 512 threads, <= 256 VGPRs, H = 256, at most one output tile of the last GEMM per wave): that tile's 16 KB panel is requested before the wait
// for the critics and parked in registers - the helper idles ~8 us there, so being held at the issue stage is free, and the GEMM at the very end of the update's
// longest dependent chain starts from registers.
// PARTS (pair-mode k_policy_critic): dQ/da arrives as its 16 per-n-block partial tiles per critic (written through by the critic halves, read below the L1, summed here in
// n-block order - the order tile_bwd_dx_cols sums them in), and Q through the same kind of store: the wait in front needs no acquire.
template <int PANEL = 16, bool PARK = false, bool PARTS = false, class Wait>
__device__ __forceinline__ void actor_bwd_tile(const il_sac& d, const il_batch& b, int tile, float* __restrict__ out_logp, float* __restrict__ out_q, float* smem, int part,
                                               int nparts, Wait wait) {
  if (!out_logp) out_logp = d.out_logp;
  if (!out_q) out_q = d.out_q;
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch;
  const int row0 = tile * IL_TILE_R;
  const int ldh = H + 4, ldz = 20;
  float* DZ2s = smem; float* DZ3s = DZ2s + IL_TILE_R * ldh; float* red = DZ3s + IL_TILE_R * ldz;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView net = mlp_view(d.actor, S, H, 2 * A);
  const float alpha = expf(d.log_alpha[0]);
  const int tid = threadIdx.x;
  const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const float* h2 = W + ws.a_h2; const float* h1 = W + ws.a_h1;
  // epilogue operands of the two back-propagation GEMMs (this lane's 4 rows of one feature of h2 / h1): requested now, used after the MFMA loops
  const int nb = H >> 4, pt0 = part * nb / nparts, pt1 = (part + 1) * nb / nparts;   // this part's output tiles of the last GEMM
  const size_t poff = (size_t)min(wave * 16 + j, H - 1) * B + row0 + 4 * g;
  const f32x4 hv2p = gload4(h2 + poff), hv1p = gload4(h1 + (size_t)min((pt0 + wave) * 16 + j, H - 1) * B + row0 + 4 * g);
  const bool stamp = tile == 0 && part == 0;
  IL_STAMP(stamp, 24);
  for (int i = tid; i < IL_TILE_R * ldz; i += blockDim.x) DZ3s[i] = 0.f;
  float apart = 0.f;
  // operands of the head backward that do not depend on the critics (training.py:38-46)
  float cc = 0.f, th2 = 0.f, omaa = 0.f, e = 0.f, lsr = 0.f, sd = 1.f;
  const int hr = tid / A, hc = tid - hr * A, hrow = row0 + hr;
  if (tid < IL_TILE_R * A) {
    const float wgt = b.weights[(size_t)hrow * b.ld_weights], m = 1.f - b.absorbing[(size_t)hrow * b.ld_absorbing];
    cc = (wgt * m * alpha) / (float)B;
    const float x = W[ws.a_xpre + (size_t)hrow * A + hc], an = W[ws.a_anew + (size_t)hrow * A + hc];
    e = W[ws.a_eps + (size_t)hrow * A + hc]; lsr = W[ws.a_lsraw + (size_t)hrow * A + hc];
    sd = expf(fminf(fmaxf(lsr, -20.f), 2.f));
    th2 = 2.f * tanhf(x); omaa = 1.f - an * an;
  }
  if (tid < IL_TILE_R) {
    const int row = row0 + tid;
    const float lp = W[ws.a_logp + row];
    apart = b.weights[(size_t)row * b.ld_weights] * (1.f - b.absorbing[(size_t)row * b.ld_absorbing]) * (lp + d.entropy_target);
    if (part != 0) { out_logp = nullptr; out_q = nullptr; }
    if (out_logp) out_logp[row] = lp;
    if (out_q) out_q[row] = fminf(W[ws.c_q + row], W[ws.c_q + B + row]);   // training.py:54 Q_values = min of the critics on the sampled (s, a), before their step
  }
  IL_STAMP(stamp, 25);
  apart = block_sum(apart, red);  // contains barriers: the zeroing of DZ3s is complete afterwards
  IL_STAMP(stamp, 26);
  if (tid == 0 && part == 0) {
    W[ws.alpha_part + tile] = apart;
  }
  Panel16 parked;
  const bool own_tile = pt0 + wave < pt1;
  if (PARK && own_tile) panel_prefetch(parked, W + ws.pk_ab, pt0 + wave);
  if (nparts > 1) {   // a waiting helper pulls the operands of its two GEMMs into this XCD's L2 (they were rewritten by the previous update's Adam kernel on other XCDs)
    if (!PARK && pt0 + wave < pt1) {
      const float* pp = W + ws.pk_ab + (size_t)(pt0 + wave) * nb * 256 + lane * 4;
#pragma unroll
      for (int u = 0; u < 16; ++u) { if (u < nb) { f32x4 v = gload4(pp + (size_t)u * 256); asm volatile("" ::"v"(v)); } }
    }
    for (int i = tid * 4; i < 2 * A * H; i += blockDim.x * 4) { f32x4 v = gload4(net.W3 + i); asm volatile("" ::"v"(v)); }
  }
  wait();
  if (tid < IL_TILE_R * A) {
    float q1, q2, g1, g2;
    if (PARTS) {
      q1 = sload1(W, ws.p_q + hrow); q2 = sload1(W, ws.p_q + B + hrow);
      const int nt_ = B / IL_TILE_R;
      float v1[16], v2[16];
#pragma unroll
      for (int w = 0; w < 16; ++w) {   // all 32 requested before the first add
        v1[w] = sload1(W, ws.p_part + (((int64_t)(0 * nt_ + tile) * 16 + w) * IL_TILE_R + hr) * A + hc);
        v2[w] = sload1(W, ws.p_part + (((int64_t)(1 * nt_ + tile) * 16 + w) * IL_TILE_R + hr) * A + hc);
      }
      g1 = 0.f; g2 = 0.f;
      const int nbk = H >> 4;
#pragma unroll
      for (int w = 0; w < 16; ++w) if (w < nbk) { g1 += v1[w]; g2 += v2[w]; }
    } else {
      q1 = W[ws.p_q + hrow]; q2 = W[ws.p_q + B + hrow];
      g1 = W[ws.p_g + (size_t)hrow * A + hc]; g2 = W[ws.p_g + ((size_t)B + hrow) * A + hc];
    }
    const float s1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
    const float da = (-(s1) / (float)B) * g1 + (-(1.f - s1) / (float)B) * g2;
    const float dxp = cc * th2 + da * omaa;
    const float dsd = dxp * e - cc / sd;
    const float dls = (lsr >= -20.f && lsr <= 2.f) ? dsd * sd : 0.f;
    DZ3s[hr * ldz + hc] = dxp; DZ3s[hr * ldz + A + hc] = dls;
  }
  __syncthreads();
  if (part == 0)
    for (int i = tid; i < IL_TILE_R * 16; i += blockDim.x) W[ws.a_dz3 + (size_t)(i >> 4) * B + row0 + (i & 15)] = DZ3s[(i & 15) * ldz + (i >> 4)];  // dz3^T [16][B]
  IL_STAMP(stamp, 27);
  // dz2 = (dz3 . W3) [h2 > 0]
  tile_bwd_dx(DZ3s, ldz, 16, 2 * A, net.W3, H, H, [&](int kb, f32x4 acc) {
    const size_t off = (size_t)(kb + j) * B + row0 + 4 * g;
    const f32x4 hv = (kb == wave * 16) ? hv2p : *reinterpret_cast<const f32x4*>(h2 + off);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = hv[r] > 0.f ? acc[r] : 0.f; DZ2s[(4 * g + r) * ldh + kb + j] = o[r]; }
    if (part == 0) wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.a_dz2 + (int64_t)off, o);
  });
  __syncthreads();
  IL_STAMP(stamp, 28);
  auto epi_dz1 = [&](int kb, f32x4 acc) {
    const size_t off = (size_t)(kb + j) * B + row0 + 4 * g;
    const f32x4 hv = (kb == (pt0 + wave) * 16) ? hv1p : *reinterpret_cast<const f32x4*>(h1 + off);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = hv[r] > 0.f ? acc[r] : 0.f;
    wstore4<(IL_WT_TILE_STORES && PANEL >= 16)>(W, ws.a_dz1 + (int64_t)off, o);
  };
  if (PARK) { if (own_tile) tile_packed_regs(DZ2s, ldh, parked, pt0 + wave, epi_dz1); }
  else tile_bwd_packed<PANEL>(DZ2s, ldh, H, W + ws.pk_ab, epi_dz1, pt0, pt1);
  IL_STAMP(stamp, 29);
}

// helpers > 0 (single learner, 2 * nt + helpers * nt co-resident workgroups): the policy backward of a tile is not run by the pair's second
// arriver but by `helpers` extra workgroups per tile that wait for both critics (tile counter) with their own operands already requested,
// and split the last GEMM between them by output columns. Same arithmetic per element, so the result is bit-identical to helpers = 0.
#define IL_PC_HELPERS 4
#define IL_PC_XCD_NETS 0x100   // flag bit in k_policy_critic's `helpers` argument
#define IL_PC_NO_TAIL 0x200    // flag bit: critic workgroups only, the policy backward is a launch of its own (k_actor_bwd: population path, IL_POP_SPLIT_TAIL)
template <int PANEL>
__device__ __forceinline__ void k_policy_critic_body(il_sac d, il_batch b, float* __restrict__ out_logp, float* __restrict__ out_q, const il_sac* __restrict__ dL,
                                                        const il_batch* __restrict__ bL, int helpers, float* smem) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (dL) { pop_ids(bx, by); d = dL[by]; b = bL[by]; }
  globalize(d); globalize(b);
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int nt = B / IL_TILE_R;
  // helpers & IL_PC_XCD_NETS (single learner, grid = 8 * nt): one critic per XCD (workgroup b runs on XCD b % 8) - XCD 0 / 1: the nt tiles of critic 0 / 1, XCD 2 ..
  // 2 + helpers - 1: helper part p of every tile (its column slice of the actor's backward panel is read by that XCD alone), the other blocks exit. Tile q: blocks
  // 8q, 8q + 1 (critics) < 8q + 2 + p (helpers): a helper still only waits for lower-numbered workgroups.
  const bool xcd_nets = (helpers & IL_PC_XCD_NETS) != 0;
  const bool no_tail = (helpers & IL_PC_NO_TAIL) != 0;
  helpers &= ~(IL_PC_XCD_NETS | IL_PC_NO_TAIL);
  if (xcd_nets && (bx & 7) >= 2 + helpers) return;
  if (xcd_nets ? (bx & 7) >= 2 : bx >= 2 * nt) {   // helper: block order keeps it behind both critics of its tile (it only waits for lower-numbered workgroups)
    const int h = xcd_nets ? ((bx & 7) - 2) * nt + (bx >> 3) : bx - 2 * nt, tile = h % nt, part = h / nt;
    const SacWs ws = sac_ws(S, A, H, B);
    unsigned* ctr = reinterpret_cast<unsigned*>(d.workspace + ws.pair_ctr) + tile * IL_CTR_STRIDE;
    if (h == 0 && threadIdx.x == 0) { adam_tick(d.actor_opt); adam_tick(d.alpha_opt); }   // consumed by the next kernel
    IL_TL(3, 0);
    actor_bwd_tile<PANEL>(d, b, tile, out_logp, out_q, smem, part, helpers, [&] {
      IL_TL(3, 1);
      tile_await(ctr, 2u, tile_timeouts(d));
      IL_TL(3, 2);
      if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u + (unsigned)helpers)
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // last helper through: ready for the next launch
    });
    IL_TL_END(3);
    return;
  }
  int k, tile;
  if (xcd_nets) { k = bx & 7; tile = bx >> 3; } else xcd_tile_net(bx, nt, 2, tile, k);
  const int row0 = tile * IL_TILE_R;
  const int INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* q16 = H2s + IL_TILE_R * ldh;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView p = mlp_view(d.critic + k * net_stride(IN, H, 1), IN, H, 1);
  const bool stamp = bx == 0;
  IL_STAMP(stamp, 16);
  IL_TL(3, 0);
  IL_TL(7, 0);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // biases of this wave's 16 columns and this lane's slice of w3: requested before the first barrier, used after the MFMA loops
  const int pc = min(wave * 16 + j, H - 1);
  const float pb1 = gload(p.b1 + pc), pb2 = gload(p.b2 + pc), pb3 = gload(p.b3);
  float w3v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) w3v[u] = gload(p.W3 + min(lane + 64 * u, H - 1));
  constexpr bool cols_pre = IL_SMALL_PREFETCH && PANEL >= 16;   // (the 80-VGPR population build has no registers to park them in)
  ColsPre w1pre = {};
  if (cols_pre) w1pre = tile_bwd_dx_cols_prefetch(p.W1, IN, IN, H, S);   // the action columns of W1 for dQ/da, the last GEMM of this workgroup
  load_rows_cat(Xs, ldx, INp, b.states, b.ld_states, S, W + ws.a_anew, A, A, row0, IL_TILE_R);
  __syncthreads();
  IL_STAMP(stamp, 17);
  IL_TL(7, 1);
  tile_fwd<PANEL>(Xs, ldx, INp, p.W1, IN, IN, H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb1 : p.b1[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H1s[(4 * g + r) * ldh + col] = hv[r]; }
    debug_mask4(d, 6 + 2 * k, row0 + 4 * g, col, hv);
  });
  __syncthreads();
  IL_STAMP(stamp, 18);
  IL_TL(7, 2);
  tile_fwd_packed<PANEL>(H1s, ldh, H, W + ws.pk_cf + (size_t)k * H * H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb2 : p.b2[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H2s[(4 * g + r) * ldh + col] = hv[r]; }
    debug_mask4(d, 7 + 2 * k, row0 + 4 * g, col, hv);
  });
  __syncthreads();
  IL_STAMP(stamp, 19);
  IL_TL(7, 3);
  // Q = h2 . w3 + b3 and, in the same pass, dQ/dh2 with upstream 1 (scaling and min-selection happen in the tail): dz2 = w3 [h2 > 0], in place.
  // One wave per row; w3 comes from the registers loaded at the top.
  for (int r = wave; r < IL_TILE_R; r += nw) {
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = lane + 64 * u;
      if (n < H) { const float h = H2s[r * ldh + n]; sq += h * w3v[u]; H2s[r * ldh + n] = h > 0.f ? w3v[u] : 0.f; }
    }
    sq = wave_sum(sq);
    if (lane == 0) W[ws.p_q + (size_t)k * B + row0 + r] = sq + pb3;
  }
  __syncthreads();
  IL_STAMP(stamp, 21);
  IL_TL(7, 5);
  tile_bwd_packed<PANEL>(H2s, ldh, H, W + ws.pk_cb + (size_t)k * H * H, [&](int kb, f32x4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* h = H1s + (4 * g + r) * ldh + kb + j;
      *h = *h > 0.f ? acc[r] : 0.f;  // dz1 in place (each element owned by exactly one lane)
    }
  });
  __syncthreads();
  IL_STAMP(stamp, 22);
  IL_TL(7, 6);
  // dQ/da = the action columns of dz1 . W1 on MFMA, the H-reduction split over the 16 waves (tile_bwd_dx_cols)
  float* gout = W + ws.p_g + ((size_t)k * B + row0) * A;
  tile_bwd_dx_cols(H1s, ldh, H, p.W1, IN, IN, S, S + A, q16 + 64, [&](int col, int row, float v) {
    const int c = col - S;
    if (c >= 0 && c < A) gout[(size_t)row * A + c] = v;
  }, cols_pre ? &w1pre : nullptr);
  IL_STAMP(stamp, 23);
  IL_TL(7, 7);
  // The policy backward of this tile needs Q and dQ/da of BOTH critics, i.e. of two workgroups. Instead of a kernel boundary, the workgroup
  // that arrives second continues with it. The barrier orders every wave's stores before thread 0's agent-scope acq_rel ticket, which
  // is the only L2 write-back / invalidate of the hand-off (a __threadfence() per wave costs 16 of them per workgroup: measured -8 %).
  unsigned* ctr = reinterpret_cast<unsigned*>(W + ws.pair_ctr) + tile * IL_CTR_STRIDE;
  if (helpers > 0) { IL_TL(3, 6); tile_arrive(ctr); IL_TL(3, 7); return; }
  if (no_tail) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // second arriver: ready for the next launch
    q16[0] = __uint_as_float(ticket);
  }
  __syncthreads();
  if (__float_as_uint(q16[0]) == 0u) {   // first arriver: done, after ticking the two optimisers the next kernel steps (off everyone's critical path)
    if (tile == 0 && threadIdx.x == 0) { adam_tick(d.actor_opt); adam_tick(d.alpha_opt); }
    return;
  }
  actor_bwd_tile<PANEL>(d, b, tile, out_logp, out_q, smem, 0, 1, [] {});
}

// The policy backward of one tile as a workgroup of a launch of its own (population path): in k_policy_critic_pop the second of a tile's two critic workgroups to arrive
// runs it as its tail, so half the workgroups of that launch are 1.7 x as long as the other half and the launch ends on the long ones. grid = (nt, learners).
template <int PANEL>
__device__ __forceinline__ void k_actor_bwd_body(il_sac d, il_batch b, const il_sac* __restrict__ dL, const il_batch* __restrict__ bL, float* smem) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (dL) { pop_ids(bx, by); d = dL[by]; b = bL[by]; }
  globalize(d); globalize(b);
  if (bx == 0 && threadIdx.x == 0) { adam_tick(d.actor_opt); adam_tick(d.alpha_opt); }   // consumed by the next kernel
  actor_bwd_tile<PANEL>(d, b, bx, nullptr, nullptr, smem, 0, 1, [] {});
}

// The tile kernels, twice: 1024-thread workgroups with the whole weight panel of a tile in flight (single learner, data-parallel and per-function paths: one workgroup
// per CU, latency hidden inside the workgroup), and `_pop` = 512-thread workgroups with half panels compiled for six waves per SIMD (<= 80 VGPRs), so that THREE workgroups
// of the population launches share a CU instead of two and the latency-bound small phases of one hide under the MFMA phases of the others. Same arithmetic, same order.
#define IL_TILE_KERNELS(SUFFIX, PANEL, BOUNDS)                                                                                                                            \
  __global__ BOUNDS void k_actor_fwd##SUFFIX(il_sac d, il_batch b, const float* __restrict__ eps_next, const float* __restrict__ eps_cur, int mode,                      \
                                             const il_sac* __restrict__ dL, const il_batch* __restrict__ bL) {                                                           \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                                                                         \
    k_actor_fwd_body<PANEL>(d, b, eps_next, eps_cur, mode, dL, bL, smem);                                                                                                \
  }                                                                                                                                                                      \
  __global__ BOUNDS void k_critic_fwd##SUFFIX(il_sac d, il_batch b, const il_sac* __restrict__ dL, const il_batch* __restrict__ bL) {                                    \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                                                                         \
    k_critic_fwd_body<PANEL>(d, b, dL, bL, smem);                                                                                                                        \
  }                                                                                                                                                                      \
  __global__ BOUNDS void k_critic_bwd##SUFFIX(il_sac d, il_batch b, const il_sac* __restrict__ dL, const il_batch* __restrict__ bL) {                                    \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                                                                         \
    k_critic_bwd_body<PANEL>(d, b, dL, bL, smem);                                                                                                                        \
  }                                                                                                                                                                      \
  __global__ BOUNDS void k_policy_critic##SUFFIX(il_sac d, il_batch b, float* __restrict__ out_logp, float* __restrict__ out_q, const il_sac* __restrict__ dL,           \
                                                 const il_batch* __restrict__ bL, int helpers) {                                                                         \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                                                                         \
    k_policy_critic_body<PANEL>(d, b, out_logp, out_q, dL, bL, helpers, smem);                                                                                           \
  }                                                                                                                                                                      \
  __global__ BOUNDS void k_actor_bwd##SUFFIX(il_sac d, il_batch b, const il_sac* __restrict__ dL, const il_batch* __restrict__ bL) {                                     \
    extern __shared__ __attribute__((aligned(16))) float smem[];                                                                                                         \
    k_actor_bwd_body<PANEL>(d, b, dL, bL, smem);                                                                                                                         \
  }
IL_TILE_KERNELS(, 16, __launch_bounds__(1024))
#ifndef IL_POP_PANEL
#define IL_POP_PANEL 8
#endif
#ifndef IL_POP_WAVES_PER_EU
#define IL_POP_WAVES_PER_EU 6
#endif
IL_TILE_KERNELS(_pop, IL_POP_PANEL, __launch_bounds__(512, IL_POP_WAVES_PER_EU))

// ---------------------------------------------------------------------------------------------
// Pair mode of k_policy_critic (round 4): (critic k, tile) is a pair of 512-thread workgroups. Both compute the narrow first layer in full; the hidden layer's forward
// and its backward are split by output columns (wave w of half p: output tile 8 p + w, both panels parked in registers - the backward one is requested while the forward
// MFMAs run); after the forward the halves SWAP their 16 x 128 halves of h2 (Q and the mask w3 [h2 > 0] need whole rows; both compute them), after the backward half 1
// sends its half of dz1 and exits, half 0 runs the dQ/da columns and arrives on the tile's counter. Helpers as in k_policy_critic. Block order: critics p = 1, p = 0,
// helpers. A pair waits for each other: both halves must be resident - (4 + helpers) * nt <= CUs is checked by the caller. Bit-identical to k_policy_critic.
// ---------------------------------------------------------------------------------------------
// ov >= 0 (overlapped launches): the critic optimiser launch of this update may still be running on the other stream - the rows ((s, a~): written by the forward launch that
// precedes this one in its stream) are requested first, the critic's parameters behind the wait for that launch's epoch.
__device__ __forceinline__ void policy_critic_pair(const il_sac& d, const il_batch& b, int k, int tile, int half, float* smem, long long ov = -1, int ov_grid = 0) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, B = d.batch, IN = S + A;
  const int nt = B / IL_TILE_R, row0 = tile * IL_TILE_R;
  const int INp = round_up16(IN), ldx = INp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* W1s = pair_w1s(smem, INp, H);
  const int ldw1 = INp + 4, w1_lanes = H * IN / 4;
  const SacWs ws = sac_ws(S, A, H, B);
  float* W = d.workspace;
  const MlpView p = mlp_view(d.critic + k * net_stride(IN, H, 1), IN, H, 1);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int t2 = 8 * half + wave;
  // hop slots (h2 halves, both directions): ((k nt + tile) 2 + half) in [0, 4 nt)
  const size_t slab_floats = (size_t)IL_TILE_R * (H / 2);
  const int sa = (k * nt + tile) * 2;
  float* slabs = W + ws.x_slab; unsigned* flags = reinterpret_cast<unsigned*>(W + ws.x_flag);
  const TileTimeouts tmo = tile_timeouts(d);
  auto timed_out = [&] { __hip_atomic_fetch_add(tmo.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (tmo.sync) sync_timed_out(tmo.sync); };
  IL_TL(11, 0);
  pair_announce(flags + (size_t)(sa + 1 - half) * IL_CTR_STRIDE);   // consumer of the partner's h2 half ...
  RowsPre rp; rows_idx(rp, INp, row0, nullptr);
  if (ov >= 0) {
    rows_issue(rp, INp, b.states, b.ld_states, S, W + ws.a_anew, A, A, row0, false, 0, true);
    ov_wait(reinterpret_cast<long long*>(d.sync), IL_OV_DWC, ov + 1, ov_grid);
    IL_ST_GATE(IL_ST_POLICY_CRITIC);
  }
  const bool w1_regs = l1_rows_aligned(p.W1, IN);
  W1Pre w1; L1Pre w1r;
  if (w1_regs) l1_prefetch(w1r, p.W1, IN, INp, H); else w1_issue(w1, p.W1, w1_lanes);
  if (ov < 0) rows_issue(rp, INp, b.states, b.ld_states, S, W + ws.a_anew, A, A, row0, false, 0, true);
  const float pb1a = gload(p.b1 + wave * 16 + j), pb1b = gload(p.b1 + min((wave + nw) * 16 + j, H - 1)), pb2 = gload(p.b2 + t2 * 16 + j), pb3 = gload(p.b3);
  float w3v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) w3v[u] = gload(p.W3 + min(lane + 64 * u, H - 1));
  const ColsPre w1pre = tile_bwd_dx_cols_prefetch(p.W1, IN, IN, H, S);   // (slot `half` holds this wave's n-block 8 half + wave of the dQ/da columns)
  issue_fence();
  Panel16 pf; panel_prefetch_lo(pf, W + ws.pk_cf + (size_t)k * H * H, t2);
  if (!w1_regs) w1_commit(w1, W1s, ldw1, IN, INp, H, w1_lanes);
  rows_commit(rp, Xs, ldx, INp, IN);
  __syncthreads();
  IL_TL(11, 1);
  {
    auto epi1 = [&](int c0, f32x4 acc) {
      const int col = c0 + j; const float bb = (c0 == wave * 16) ? pb1a : pb1b;
      f32x4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H1s[(4 * g + r) * ldh + col] = hv[r]; }
      if (half == 0) debug_mask4(d, 6 + 2 * k, row0 + 4 * g, col, hv);
    };
    if (w1_regs) l1_compute_regs(w1r, Xs, ldx, INp, H, epi1); else l1_compute_lds(W1s, ldw1, Xs, ldx, INp, H, epi1);
  }
  __syncthreads();
  IL_TL(11, 2);
  const bool near_a = pair_same_xcd(flags + (size_t)(sa + half) * IL_CTR_STRIDE);
  panel_prefetch_hi(pf, W + ws.pk_cf + (size_t)k * H * H, t2);
  tile_packed_regs(H1s, ldh, pf, t2, [&](int c0, f32x4 acc) {
    const int col = c0 + j;
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + pb2, 0.f); H2s[(4 * g + r) * ldh + col] = hv[r]; }
    debug_mask4(d, 7 + 2 * k, row0 + 4 * g, col, hv);
    pair_store(slabs + (size_t)(sa + half) * slab_floats, (int64_t)(col - 128 * half) * 16 + 4 * g, hv, near_a);
  });
  pair_publish(flags + (size_t)(sa + half) * IL_CTR_STRIDE);
  IL_TL(11, 3);
  pair_receive(flags + (size_t)(sa + 1 - half) * IL_CTR_STRIDE, slabs + (size_t)(sa + 1 - half) * slab_floats, H2s, ldh, 128 * (1 - half), timed_out);
  IL_TL(11, 4);
  Panel16 pk; panel_prefetch_lo(pk, W + ws.pk_cb + (size_t)k * H * H, t2);   // streams in under the Q pass
  // Q = h2 . w3 + b3 and dz2 = w3 [h2 > 0] in place: whole rows, computed by both halves (half 0 stores Q)
  for (int r = wave; r < IL_TILE_R; r += nw) {
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = lane + 64 * u;
      if (n < H) { const float h = H2s[r * ldh + n]; sq += h * w3v[u]; H2s[r * ldh + n] = h > 0.f ? w3v[u] : 0.f; }
    }
    sq = wave_sum(sq);
    if (lane == 0 && half == 0) wstore1(W, ws.p_q + (int64_t)k * B + row0 + r, sq + pb3);   // (written through: the helpers read it without an acquire)
  }
  __syncthreads();
  IL_TL(11, 5);
  panel_prefetch_hi(pk, W + ws.pk_cb + (size_t)k * H * H, t2);
  tile_packed_regs(H2s, ldh, pk, t2, [&](int kb, f32x4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { float* h = H1s + (4 * g + r) * ldh + kb + j; *h = *h > 0.f ? acc[r] : 0.f; }   // dz1 in place, this half's columns (each element owned by one lane)
  });
  __syncthreads();
  IL_TL(11, 6);
  // dQ/da = the action columns of dz1 . W1: a sum over the 16 n-blocks of hidden units, in n-block order. Each half holds the dz1 columns of EIGHT of them: wave w forms
  // the partial tile of block 8 half + w (the four MFMAs of tile_bwd_dx_cols) and writes it through; the tile's helpers add the sixteen partials in n-block order, as
  // tile_bwd_dx_cols' own reduction does - no second hop between the halves, no release here, no acquire there (round 4: -1.5 us on the critical path of this launch).
  {
    const int nb = 8 * half + wave, n0 = nb * 16;
    float* pp = W + ws.p_part + ((int64_t)(k * nt + tile) * 16 + nb) * IL_TILE_R * A;
    const f32x4 a = *reinterpret_cast<const f32x4*>(H1s + j * ldh + n0 + 4 * g);
    const int kb0 = (S >> 4) << 4;
    for (int kb = kb0; kb < S + A; kb += 16) {
      float bq[4];
      if (IL_SMALL_PREFETCH && kb == kb0) {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bq[s_] = half == 0 ? w1pre.b[0][s_] : w1pre.b[1][s_];
      } else {
        const float* wp = p.W1 + min(kb + j, IN - 1);
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bq[s_] = gload(wp + (size_t)(n0 + 4 * g + s_) * IN);
      }
      f32x4 acc0 = zero4(), acc1 = zero4();
      acc0 = mfma16(a[0], bq[0], acc0);
      acc1 = mfma16(a[1], bq[1], acc1);
      acc0 = mfma16(a[2], bq[2], acc0);
      acc1 = mfma16(a[3], bq[3], acc1);
      const f32x4 acc = acc0 + acc1;
      const int c = kb + j - S;
      if (c >= 0 && c < A) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wstore1(pp, (int64_t)(4 * g + r) * A + c, acc[r]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* ctr = reinterpret_cast<unsigned*>(W + ws.pair_ctr) + tile * IL_CTR_STRIDE;
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  IL_TL(11, 7);
}

__global__ __launch_bounds__(512) void k_policy_critic_pair(il_sac d, il_batch b, float* __restrict__ out_logp, float* __restrict__ out_q, int helpers, int overlap) {   // overlap: 0 = off, else the grid size of the critic optimiser launch this launch waits for
  extern __shared__ __attribute__((aligned(16))) float smem[];
  globalize(d); globalize(b);
  IL_ST_BEGIN(IL_ST_POLICY_CRITIC);
  const int nt = d.batch / IL_TILE_R, bx = blockIdx.x;
  long long* osy = reinterpret_cast<long long*>(d.sync);
#ifdef IL_EXP_CHECK
  if (bx == 0) {
    const SacWs cws = sac_ws(d.state_dim, d.action_dim, d.hidden, d.batch);
    const float* CW = d.workspace; const int CB = min(d.batch, 4096), CA = d.action_dim;
    for (int row = threadIdx.x; row < CB; row += blockDim.x) {
      const unsigned fr = __float_as_uint(sload1(CW, cws.c_rew + row)), t1 = __float_as_uint(sload1(CW, cws.t_q + row)), t2 = __float_as_uint(sload1(CW, cws.t_q + d.batch + row)), lp = __float_as_uint(sload1(CW, cws.n_logp2 + row));
      for (int k = 0; k < 2; ++k) {
        if (__float_as_uint(il_chk_rew[k][row]) != fr) atomicAdd(&il_chk[5], 1u);
        if (__float_as_uint(il_chk_tq[k][0][row]) != t1 || __float_as_uint(il_chk_tq[k][1][row]) != t2) atomicAdd(&il_chk[6], 1u);
        if (__float_as_uint(il_chk_tq[k][2][row]) != lp) atomicAdd(&il_chk[7], 1u);
      }
      if (CA <= 8) for (int c = 0; c < CA; ++c) { const unsigned a2 = __float_as_uint(sload1(CW, cws.n_a2 + (int64_t)row * CA + c)); for (int q = 0; q < 4; ++q) if (__float_as_uint(il_chk_a2[q][row * 8 + c]) != a2) atomicAdd(&il_chk[8], 1u); }
    }
    if (threadIdx.x == 0) atomicAdd(&il_chk[0], 1u);   // checks made
  }
#endif
  const long long ov = overlap ? ov_own(osy, IL_OV_PC) : -1;   // (helpers need nothing of the critic optimiser launch: they wait for this launch's critic workgroups)
  if (bx >= 4 * nt) {   // helper: behind both critics' pairs of its tile in block order
    const int h = bx - 4 * nt, tile = h % nt, part = h / nt;
    const SacWs ws = sac_ws(d.state_dim, d.action_dim, d.hidden, d.batch);
    unsigned* ctr = reinterpret_cast<unsigned*>(d.workspace + ws.pair_ctr) + tile * IL_CTR_STRIDE;
    if (h == 0 && threadIdx.x == 0) { adam_tick(d.actor_opt); adam_tick(d.alpha_opt); }
    IL_TL(3, 0);
    const auto wait = [&] {   // both halves of both critics have written Q and their dQ/da partial tiles THROUGH (sc0 sc1) before arriving: one relaxed poll, no acquire
      IL_TL(3, 1);
      if (threadIdx.x == 0) {
        const TileTimeouts tmo = tile_timeouts(d);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4u) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > IL_SYNC_SPIN_LIMIT) { __hip_atomic_fetch_add(tmo.slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (tmo.sync) sync_timed_out(tmo.sync); break; }
        }
      }
      __syncthreads();
      IL_TL(3, 2);
      if (threadIdx.x == 0 && __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3u + (unsigned)helpers)
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if ((d.hidden >> 4) <= helpers * (int)(blockDim.x >> 6)) actor_bwd_tile<16, true, true>(d, b, tile, out_logp, out_q, smem, part, helpers, wait);   // one output tile of the last GEMM per wave at most
    else actor_bwd_tile<16, false, true>(d, b, tile, out_logp, out_q, smem, part, helpers, wait);
    IL_TL_END(3);
    IL_ST_END(IL_ST_POLICY_CRITIC);
    if (overlap) ov_done(osy, IL_OV_PC);
    return;
  }
  const int half = bx < 2 * nt ? 1 : 0;
  int k, tile;
  seg_decode(bx % (2 * nt), nt, 2, k, tile);
  policy_critic_pair(d, b, k, tile, half, smem, ov, overlap);
  IL_ST_END(IL_ST_POLICY_CRITIC);
  if (overlap) ov_done(osy, IL_OV_PC);
}

// ---------------------------------------------------------------------------------------------
// actor backward (training.py:38-46): L = mean(w m alpha logp - min Q).  grid = nt
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// k_dw_adam: output-stationary weight gradients on MFMA with a fused AdamW epilogue.
//   dW_l[n][k] = sum_r dZ_l[r][n] X_l[r][k]  (reduction index = batch row, 4 per MFMA step), db_l[n] = sum_r dZ_l[r][n]
// One wave = one job: a 16(n) x 16(k) tile of some layer's weight, or 16 bias elements. Operands come from the feature-major
// workspace ([feature][B]): lane (j, g) reads 4 consecutive batch rows of feature n0+j / k0+j as ONE 16-byte load, eight such
// loads per operand are in flight before the first MFMA. Gradients never touch HBM unless grads_only (data-parallel: they are
// all-reduced first).  Tail blocks: Adam(log_alpha), polyak, Philox counter.
// ---------------------------------------------------------------------------------------------
#include "dw_block.hpp"
template <bool STAMP = false>   // STAMP: the single-learner kernel records when the wait was satisfied (IL_ST_GATE)
__device__ __forceinline__ void dw_ov_wait(const DwArgs& a) {   // all threads of the workgroup
  const int stage = __builtin_amdgcn_readfirstlane(a.ov_stage);   // (wave-uniform by construction; a population launch builds its DwArgs from a descriptor it loaded)
  if (stage) {
    long long* sy = reinterpret_cast<long long*>(a.sync); const int st = stage - 1;
    ov_wait(sy, st - 1, ov_own(sy, st) + 1, a.ov_grid);
    if (STAMP) IL_ST_GATE(st == IL_OV_DWA ? IL_ST_DW_ACTOR : IL_ST_DW_CRITIC);
  }
}

__device__ __forceinline__ void adam_store(const DwArgs& a, const adam_consts& ac, int64_t o, float gr) {
  if (a.grads_only) { a.grads[o] = gr; return; }
  float pp = a.params[o], mm = a.opt.m[o], vv = a.opt.v[o];
  adam_update(pp, gr, mm, vv, ac);
  a.params[o] = pp; a.opt.m[o] = mm; a.opt.v[o] = vv;
}

#ifndef IL_DW_PREFETCH
#define IL_DW_PREFETCH 1
#endif
#ifndef IL_DW_SCHED_BARRIER
#define IL_DW_SCHED_BARRIER 1   // dw_tile: all operand loads of a chunk ahead of its MFMAs (0 = the round-2 schedule, for A/B builds)
#endif
#ifndef IL_TAIL_BLOCKS
#define IL_TAIL_BLOCKS 69       // single learner: tail blocks of the actor launch (block 0: Adam(log alpha) + counters; all: polyak over the target arena and its lane-ordered copies, one trip each at H = 256)
#endif
#ifndef IL_DW_XCD_BLOCKS
#define IL_DW_XCD_BLOCKS 1      // dw_block_job: the H x H layer's 64 blocks dealt to the XCDs as 2 x 4 rectangles (fabric traffic; 0 = row-major job order)
#endif
#ifndef IL_DW_BLOCK32
#define IL_DW_BLOCK32 1         // single learner: the H x H layers' dW as 32 x 32 blocks staged through LDS (dw_block32); 0 = a wave per 16 x 16 tile straight from global memory
#endif
#ifndef IL_DW_U
#define IL_DW_U 16              // single-learner k_dw_adam: 16-row operand lanes in flight per operand and chunk (16 = the whole batch of 256 rows in one round)
#endif
// XT: x is feature-major [Kvalid][B]; otherwise row-major [B][ldx] (the actor's layer-1 input = the states field of the batch)
// U = 16-row operand lanes in flight per operand: 8 for the single learner (one block per CU: latency hiding has to come from the wave itself),
// 4 for the population launch (half the registers -> four waves per SIMD instead of two hide the latency across blocks).
template <bool XT, int U>
__device__ __forceinline__ void dw_tile(const DwArgs& a, const adam_consts& ac, const float* __restrict__ dzT, int Nvalid, const float* __restrict__ x, int ldx, int Kvalid,
                                        int n0, int kb, int64_t poff, float* __restrict__ pkf = nullptr, float* __restrict__ pkb = nullptr) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int B = a.batch;
  f32x4 acc0 = zero4(), acc1 = zero4();
  // out-of-range features clamp their address: the rows / columns of dW they produce are discarded by the epilogue
  const float* dzp = dzT + (size_t)min(n0 + j, Nvalid - 1) * B + 4 * g;
  const int kc = min(kb + j, Kvalid - 1);
  const float* xp = XT ? x + (size_t)kc * B + 4 * g : x + (size_t)(4 * g) * ldx + kc;
  auto ldx4 = [&](int r0) -> f32x4 {
    if (XT) return gload4(xp + r0);
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = gload(xp + (size_t)(r0 + s) * ldx);
    return v;
  };
  // The Adam operands of this lane's four dW elements do not depend on the products: fetch them before the MFMA loop so that their
  // HBM latency (they were last touched one update ago) hides under it instead of following it.
  const int k = kb + j, kk = min(k, Kvalid - 1);
  float pp[4], mm[4], vv[4];
  if (IL_DW_PREFETCH && !a.grads_only) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t o = poff + (int64_t)min(n0 + 4 * g + r, Nvalid - 1) * Kvalid + kk;
      pp[r] = gload(a.params + o); mm[r] = gload(a.opt.m + o); vv[r] = gload(a.opt.v + o);
    }
  }
  int r0 = 0;
  // Every operand lane of a chunk is REQUESTED before its first MFMA: without the scheduling barrier hipcc sinks the loads back between the MFMAs (round 3, ISA of the
  // round-2 build: four loads in flight, `s_waitcnt vmcnt(2)` in front of every group of four MFMAs - sixteen dependent L2 round trips per tile, which is what made this
  // launch 6-8 us for 0.85 us of MFMA issue). Chunks of 16 U, then 64, then 16 rows; the MFMA order (row groups ascending, k-steps 0, 2 -> acc0 and 1, 3 -> acc1) does not
  // depend on the chunking, so every chunk size gives the same bits.
#define IL_DW_CHUNK(UU)                                                                                            \
  for (; r0 + 16 * (UU) <= B; r0 += 16 * (UU)) {                                                                   \
    f32x4 av[UU], bv[UU];                                                                                          \
    _Pragma("unroll") for (int u = 0; u < (UU); ++u) { av[u] = gload4(dzp + r0 + 16 * u); bv[u] = ldx4(r0 + 16 * u); } \
    if (IL_DW_SCHED_BARRIER) __builtin_amdgcn_sched_barrier(0);                                                    \
    _Pragma("unroll") for (int u = 0; u < (UU); ++u) {                                                             \
      acc0 = mfma16(av[u][0], bv[u][0], acc0);                                                                     \
      acc1 = mfma16(av[u][1], bv[u][1], acc1);                                                                     \
      acc0 = mfma16(av[u][2], bv[u][2], acc0);                                                                     \
      acc1 = mfma16(av[u][3], bv[u][3], acc1);                                                                     \
    }                                                                                                              \
  }
  IL_DW_CHUNK(U)
  if (U > 4) { IL_DW_CHUNK(4) }
  IL_DW_CHUNK(1)
#undef IL_DW_CHUNK
  const f32x4 acc = acc0 + acc1;
  if (k >= Kvalid) return;
  if (a.grads_only) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * g + r;
      if (n < Nvalid) a.grads[poff + (int64_t)n * Kvalid + k] = acc[r];
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + 4 * g + r;
    if (n < Nvalid) {
      const int64_t o = poff + (int64_t)n * Kvalid + k;
      if (!IL_DW_PREFETCH) { pp[r] = a.params[o]; mm[r] = a.opt.m[o]; vv[r] = a.opt.v[o]; }
      adam_update(pp[r], acc[r], mm[r], vv[r], ac);
      a.params[o] = pp[r]; a.opt.m[o] = mm[r]; a.opt.v[o] = vv[r];
    }
  }
  if (pkf) {  // the updated W2 values, in both lane orders (this tile owns rows n0..n0+15, columns kb..kb+15: always full for an H x H layer)
    f32x4 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) { w[r] = pp[r]; pkf[packed_fwd_index(n0 + 4 * g + r, k, Kvalid)] = w[r]; }
    *reinterpret_cast<f32x4*>(pkb + packed_bwd_index(n0 + 4 * g, k, Kvalid)) = w;   // rows n0+4g..+3 of column k: one 16-byte lane of PB
  }
}

// 16 bias elements per wave: lane (j, g) sums rows 16i + 4g .. +3 of feature n0 + j, the four row groups meet through shuffles
__device__ __forceinline__ void dw_bias(const DwArgs& a, const adam_consts& ac, const float* __restrict__ dzT, int Nvalid, int n0, int64_t poff) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int B = a.batch;
  const float* p = dzT + (size_t)min(n0 + j, Nvalid - 1) * B + 4 * g;
  f32x4 s4 = zero4();
  int r0 = 0;
#if IL_DW_SCHED_BARRIER
  // (round 3) as a plain loop this was load -> s_waitcnt vmcnt(0) -> add, once per 16 rows: sixteen DEPENDENT L2 round trips at B = 256 - the bias jobs, not the MFMA
  // tiles, were the long pole of the launch. All lanes of a chunk are requested first; the adds keep their order (ascending rows), so the sums keep their bits.
  for (; r0 + 256 <= B; r0 += 256) {
    f32x4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = gload4(p + r0 + 16 * u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; ++u) s4 += t[u];
  }
  for (; r0 + 64 <= B; r0 += 64) {
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = gload4(p + r0 + 16 * u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) s4 += t[u];
  }
#endif
  for (; r0 < B; r0 += 16) s4 += *reinterpret_cast<const f32x4*>(p + r0);
  float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (g == 0 && n0 + j < Nvalid) adam_store(a, ac, poff + n0 + j, s);
}

template <bool PEER>
__device__ __forceinline__ void dw_tail_alpha(const DwArgs& a, const DwPeer* pp, bool poisoned = false) {   // tail block 0: Adam(log alpha), the Philox counter, the end-of-update signal (one thread); poisoned: no store to log alpha or its moments
  if (a.log_alpha && threadIdx.x == 0) {
    float s = 0.f;
    int i0 = 0;
#if IL_DW_SCHED_BARRIER
    for (; i0 + 16 <= a.n_alpha_part; i0 += 16) {   // one thread, B / 16 partials: requested together, added in index order (as a plain loop: one dependent round trip per partial)
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = gload(a.alpha_part + i0 + u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) s += t[u];
    }
#endif
    for (int i = i0; i < a.n_alpha_part; ++i) s += a.alpha_part[i];
    const float alpha = expf(a.log_alpha[0]);
    float gr = -(alpha) * (s / (float)a.batch);
    if (PEER) gr = peer_thread_allreduce1(pp->x, a.n_big_blocks, pp->alpha_at, gr);   // data-parallel: the mean over the ranks (its own arrival line behind the block jobs')
    if (a.grads_only) a.alpha_grad[0] = gr;
    else if (!poisoned) {
      const adam_consts ac = load_adam_consts(a.alpha_opt);
      float pp = a.log_alpha[0], mm = a.alpha_opt.m[0], vv = a.alpha_opt.v[0];
      adam_update(pp, gr, mm, vv, ac);
      a.log_alpha[0] = pp; a.alpha_opt.m[0] = mm; a.alpha_opt.v[0] = vv;
    }
    // (with il_sync counters the bumped counter is read by the NEXT update's discriminator launch, resident on the other stream: written through and drained like every
    // other in-launch hand-off, profiles/r06_soak_under_load.md)
    if (a.noise_counter) { if (a.sync) { wstore1(reinterpret_cast<float*>(a.noise_counter), 0, __uint_as_float(a.noise_counter[0] + 1u)); sync_drain_stores(); } else a.noise_counter[0] += 1; }
    // this update's SAC half is done. Release: the resident sampler and, behind it, the discriminator kernels of the NEXT update start from this signal and read the noise counter bumped above
    if (a.sync) __hip_atomic_fetch_add(reinterpret_cast<long long*>(a.sync) + IL_SYNC_MAIN_EPOCH, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void dw_tail_polyak(const DwArgs& a, const int tb, const int ntb) {   // every tail block: target <- tau target + (1 - tau) critic
  if (a.target && !a.grads_only) {
    const float omt = (float)(1.0 - a.tau), tau = (float)a.tau;
#if IL_DW_SCHED_BARRIER
    // (round 3) target <- tau target + (1 - tau) critic over the parameter arena AND its lane-ordered copies as ONE index space of 16-byte lanes, four lanes per thread
    // and trip with all eight loads requested first: the grid-stride loops below were a dependent HBM round trip per trip (the target was last touched an update ago),
    // 8 trips per thread with 33 tail blocks. Elementwise: the bits do not depend on who computes which lane.
    if ((a.polyak_n & 3) == 0 && (!a.pk_target || (a.pk_n & 3) == 0) && !a.polyak_fused) {
      const int64_t n1 = a.polyak_n >> 2, n2 = a.pk_target ? (a.pk_n >> 2) : 0, stride = (int64_t)ntb * blockDim.x;
      for (int64_t i = (int64_t)tb * blockDim.x + threadIdx.x; i < n1 + n2; i += 4 * stride) {
        f32x4 t[4], p[4]; f32x4* dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t q = i + u * stride, qc = q < n1 + n2 ? q : i;   // out of range: re-read lane i, never stored
          const bool second = qc >= n1;
          dst[u] = reinterpret_cast<f32x4*>(second ? a.pk_target : a.target) + (second ? qc - n1 : qc);
          t[u] = *dst[u]; p[u] = *(reinterpret_cast<const f32x4*>(second ? a.pk_critic : a.polyak_src) + (second ? qc - n1 : qc));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) t[u][q] = __fadd_rn(__fmul_rn(t[u][q], tau), __fmul_rn(omt, p[u][q]));
          if (i + u * stride < n1 + n2) {
#if IL_DW_STORE_MODE == 2 && IL_POLYAK_WT
            const int64_t q2 = i + u * stride;   // written through like p / m / v: the target network is not read again before the next update's forward
            if (q2 >= n1) wstore4(a.pk_target, (q2 - n1) * 4, t[u]); else wstore4(a.target, q2 * 4, t[u]);
#else
            *dst[u] = t[u];
#endif
          }
        }
      }
      return;
    }
#endif
    for (int64_t i = ((int64_t)tb * blockDim.x + threadIdx.x) * 4; i < a.polyak_n; i += (int64_t)ntb * blockDim.x * 4) {
      if (a.polyak_fused) {   // the H x H layers were stepped by the critic launch's blocks: whole lanes inside them are skipped, lanes at their edges go element by element
        // (the twin critic's layout from what the tail knows: polyak_n = 2 strides, stride = H IN + H | H H | H | H | 1 rounded up to a multiple of 4 floats)
        const int64_t fz_stride = a.polyak_n >> 1, fz_hh = (int64_t)a.hidden * a.hidden;
        const int64_t fz_w2_off = ((fz_stride - fz_hh - 3 * a.hidden - 1) / a.hidden) * a.hidden + a.hidden;
        int inr = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t r0 = i + q - fz_w2_off, r1 = r0 - fz_stride;
          inr |= ((r0 >= 0 && r0 < fz_hh) || (r1 >= 0 && r1 < fz_hh)) ? (1 << q) : 0;
        }
        if (inr == 15) continue;
        if (inr != 0) {
          for (int q = 0; q < 4; ++q)
            if (i + q < a.polyak_n && !((inr >> q) & 1)) a.target[i + q] = __fadd_rn(__fmul_rn(a.target[i + q], tau), __fmul_rn(omt, a.polyak_src[i + q]));
          continue;
        }
      }
      if (i + 3 < a.polyak_n) {
        f32x4 t = *reinterpret_cast<f32x4*>(a.target + i); const f32x4 p = *reinterpret_cast<const f32x4*>(a.polyak_src + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = __fadd_rn(__fmul_rn(t[q], tau), __fmul_rn(omt, p[q]));
        *reinterpret_cast<f32x4*>(a.target + i) = t;
      } else {
        for (int64_t q = i; q < a.polyak_n; ++q) a.target[q] = __fadd_rn(__fmul_rn(a.target[q], tau), __fmul_rn(omt, a.polyak_src[q]));
      }
    }
    if (a.pk_target)
      for (int64_t i = ((int64_t)tb * blockDim.x + threadIdx.x) * 4; i < a.pk_n; i += (int64_t)ntb * blockDim.x * 4) {
        f32x4 t = *reinterpret_cast<f32x4*>(a.pk_target + i); const f32x4 p = *reinterpret_cast<const f32x4*>(a.pk_critic + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = __fadd_rn(__fmul_rn(t[q], tau), __fmul_rn(omt, p[q]));
        *reinterpret_cast<f32x4*>(a.pk_target + i) = t;
      }
  }
}
template <int U, bool SKIPBIG = false, bool PEER = false>   // SKIPBIG: the H x H layers are done by dw_block64 workgroups of the same launch (population path)
__device__ __forceinline__ void dw_adam_body(const DwArgs& a, const int bid, const int nblocks, const DwPeer* pp = nullptr) {   // bid / nblocks: this learner's block index / count
  const int wave_in_block = threadIdx.x >> 6;
  if (bid >= a.n_dw_blocks) {  // ---- tail blocks
    const int tb = bid - a.n_dw_blocks;
    if (tb == 0) dw_tail_alpha<PEER>(a, pp);
    dw_tail_polyak(a, tb, nblocks - a.n_dw_blocks);
    return;
  }
  // ---- job decode (wave-uniform)
  const int jpb = a.jobs_per_block > 0 ? a.jobs_per_block : 4;
  if (wave_in_block >= jpb) return;
  const int IN = a.in_dim, H = a.hidden, OUT = a.out_dim;
  const int nt_h = H / 16, kt_in = (IN + 15) / 16, nt_out = (OUT + 15) / 16;
  const int j1 = nt_h * kt_in, j2 = SKIPBIG ? 0 : nt_h * nt_h, j3 = nt_out * nt_h, jb = 2 * nt_h + nt_out;
  const int per_net = j1 + j2 + j3 + jb;
  int job = bid * jpb + wave_in_block;
  if (job >= per_net * a.n_nets) return;
  const int net = job / per_net; job -= net * per_net;
  adam_consts ac = {};
  if (!a.grads_only) ac = load_adam_consts(a.opt);
  const int64_t pbase = (int64_t)net * a.net_stride;
  const int64_t oW1 = pbase, ob1 = oW1 + (int64_t)H * IN, oW2 = ob1 + H, ob2 = oW2 + (int64_t)H * H, oW3 = ob2 + H, ob3 = oW3 + (int64_t)OUT * H;
  const float* x0 = a.x0 + net * a.x0_net_stride;
  const float* h1 = a.h1 + net * a.h_net_stride; const float* h2 = a.h2 + net * a.h_net_stride;
  const float* dz1 = a.dz1 + net * a.h_net_stride; const float* dz2 = a.dz2 + net * a.h_net_stride;
  const float* dz3 = a.dz3 + net * a.dz3_net_stride;
  // the big layer first: its tiles are the long pole, the small jobs fill in behind them
  if (job < j2) {
    dw_tile<true, U>(a, ac, dz2, H, h1, 0, H, (job / nt_h) * 16, (job % nt_h) * 16, oW2, a.pk_f ? a.pk_f + (size_t)net * H * H : nullptr, a.pk_b ? a.pk_b + (size_t)net * H * H : nullptr);
    return;
  }
  job -= j2;
  if (job < j1) {
    if (a.x0_transposed) dw_tile<true, U>(a, ac, dz1, H, x0, 0, IN, (job / kt_in) * 16, (job % kt_in) * 16, oW1);
    else dw_tile<false, U>(a, ac, dz1, H, x0, a.ld_x0, IN, (job / kt_in) * 16, (job % kt_in) * 16, oW1);
    return;
  }
  job -= j1;
  if (job < j3) { dw_tile<true, U>(a, ac, dz3, OUT, h2, 0, H, (job / nt_h) * 16, (job % nt_h) * 16, oW3); return; }
  job -= j3;
  if (job < nt_h) { dw_bias(a, ac, dz1, H, job * 16, ob1); return; }
  job -= nt_h;
  if (job < nt_h) { dw_bias(a, ac, dz2, H, job * 16, ob2); return; }
  job -= nt_h;
  dw_bias(a, ac, dz3, OUT, job * 16, ob3);
}

// ---------------------------------------------------------------------------------------------
// Population launch: one WORKGROUP = a 64(n) x 64(k) block of an H x H layer's dW (+ AdamW), operands staged through LDS.
// dw_tile gives every wave its own 16 x 16 tile with operands straight from global memory: a serial chain of (8 loads -> wait -> 16 MFMAs) per wave, which at
// population scale runs at MfmaUtil 13 % whatever the operand traffic is (see k_dw_adam_pop). Here the 256 threads fetch 64-row chunks of the two [feature][B]
// panels with all their loads in flight, park them in LDS ([64][64 + 4] floats each: the +4 makes the 16 lanes of a row group hit 16 different 16-byte bank
// groups), and each wave runs a 2 x 2 arrangement of 16 x 16 tiles out of it; the next chunk's loads are issued before the current chunk's 64 MFMAs per wave.
// Every tile keeps dw_tile's two accumulators and its MFMA order (row groups ascending; k-steps 0, 2 -> acc0 and 1, 3 -> acc1), so the gradients, and with them
// the learners, stay bit-identical to the single-learner kernel.
// ---------------------------------------------------------------------------------------------
#ifndef IL_POP_XCD_DW
#define IL_POP_XCD_DW 1
#endif
#define IL_FLAG_POP_FUSE_POLYAK 0x8000u   // internal (set by il_sac_update_population unless IL_POP_FUSE_POLYAK=0): see DwArgs.fuse_polyak
#define DWB 64            // block edge (features) and batch rows per chunk
#ifndef IL_POP_DW_EARLY_PMV
#define IL_POP_DW_EARLY_PMV 1   // 0: p / m / v requested at the start of the AdamW epilogue (round 4; A/B builds)
#endif
#define DWB_LD (DWB + 4)
__device__ __forceinline__ void dw_block64(const DwArgs& a, const adam_consts& ac, const float* __restrict__ dzT, const float* __restrict__ xT, int H, int n0, int k0, int64_t poff,
                                           float* __restrict__ pkf, float* __restrict__ pkb, float* smem, const il_sac* dsrc = nullptr) {
  float* Zs = smem; float* Xs = smem + DWB * DWB_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int ti = wave >> 1, tq = wave & 1;   // this wave's 32 x 32 quadrant: tiles (2 ti + i, 2 tq + q)
  const int B = a.batch;
  f32x4 acc[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) { acc[i][q][0] = zero4(); acc[i][q][1] = zero4(); }
  // staging: thread t moves 16-byte lane (feature f = t / 16 + 16 u, rows 4 (t % 16) .. +3) of both panels, u = 0..3
  const int sf = tid >> 4, sr = (tid & 15) * 4;
  f32x4 zr[4], xr[4];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { zr[u] = gload4(dzT + (size_t)(n0 + sf + 16 * u) * B + r0 + sr); xr[u] = gload4(xT + (size_t)(k0 + sf + 16 * u) * B + r0 + sr); }
  };
  const auto stage = [&] {
    __syncthreads();   // the previous chunk's readers are done
#pragma unroll
    for (int u = 0; u < 4; ++u) { *reinterpret_cast<f32x4*>(Zs + (sf + 16 * u) * DWB_LD + sr) = zr[u]; *reinterpret_cast<f32x4*>(Xs + (sf + 16 * u) * DWB_LD + sr) = xr[u]; }
    __syncthreads();
  };
  const auto products = [&] {
#pragma unroll
    for (int u = 0; u < DWB / 16; ++u) {   // 16-row groups in ascending order, like dw_tile
      f32x4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { av[i] = *reinterpret_cast<const f32x4*>(Zs + (32 * ti + 16 * i + j) * DWB_LD + 16 * u + 4 * g); bv[i] = *reinterpret_cast<const f32x4*>(Xs + (32 * tq + 16 * i + j) * DWB_LD + 16 * u + 4 * g); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc[i][q][0] = mfma16(av[i][0], bv[q][0], acc[i][q][0]);
          acc[i][q][1] = mfma16(av[i][1], bv[q][1], acc[i][q][1]);
          acc[i][q][0] = mfma16(av[i][2], bv[q][2], acc[i][q][0]);
          acc[i][q][1] = mfma16(av[i][3], bv[q][3], acc[i][q][1]);
        }
    }
  };
  fetch(0);
  for (int r0 = 0; r0 + DWB < B; r0 += DWB) { stage(); fetch(r0 + DWB); products(); }
  // (round 5) The last chunk, written out: once it is parked in LDS the staging registers are free, and the block's p / m / v lanes - HBM, last touched an update ago -
  // are requested HERE, in front of the chunk's 64 MFMAs per wave, instead of behind the gradient block's trip through LDS: their round trip (5.9 of a block
  // workgroup's 16.1 us, profiles/r04_population_timeline.txt) runs under the products. Same loads, same arithmetic: same bits.
  stage();
  f32x4 pv[4], mv[4], vv[4];
  if (!a.grads_only && IL_POP_DW_EARLY_PMV) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t o = poff + (int64_t)(n0 + sf + 16 * u) * H + k0 + sr;
      pv[u] = gload4(a.params + o); mv[u] = gload4(a.opt.m + o); vv[u] = gload4(a.opt.v + o);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  products();
  // Epilogue through LDS: the accumulator layout (a lane owns 4 rows of ONE column) would stream p / m / v as 64-byte pieces of 16 different rows per instruction;
  // parked in LDS the block is re-read row-wise, so that every wave instruction moves whole 256-byte row segments (the AdamW traffic is the floor of this kernel:
  // 24 B per parameter). The updated parameters go back to LDS once more for the column-wise lane order of the PB copy.
  float* Gs = Zs;   // [64][DWB_LD] gradient block, then the updated parameters
  __syncthreads();
  IL_TL(a.log_alpha ? 11 : 10, 1);   // products done
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 t = acc[i][q][0] + acc[i][q][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) Gs[(32 * ti + 16 * i + 4 * g + r) * DWB_LD + 32 * tq + 16 * q + j] = t[r];
    }
  __syncthreads();
  // (round 4) The four lanes' p / m / v are requested together, ahead of the first store: params / m / v are not `restrict` against each other, so in the one-loop form
  // each trip's loads waited for the previous trip's stores - four HBM round trips in a row, 9.2 of a block workgroup's 19.3 us (profiles/r04_population_timeline.txt).
  if (a.grads_only) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = sf + 16 * u;
      *reinterpret_cast<f32x4*>(a.grads + poff + (int64_t)(n0 + rr) * H + k0 + sr) = *reinterpret_cast<const f32x4*>(Gs + rr * DWB_LD + sr);
    }
    return;
  }
  if (!IL_POP_DW_EARLY_PMV) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t o = poff + (int64_t)(n0 + sf + 16 * u) * H + k0 + sr;
      pv[u] = gload4(a.params + o); mv[u] = gload4(a.opt.m + o); vv[u] = gload4(a.opt.v + o);
    }
  }
  // (round 5, IL_POP_FUSE_POLYAK) target <- tau target + (1 - tau) critic for this block, from the NEW parameters in registers: the tail of the actor launch read the
  // target, the critic and both lane-ordered copies again (36 B per H x H parameter); here 4 B are read and 8 written. Same two multiplies and one add per element.
  f32x4 tv[4];
  float* tgt = nullptr; float* pkt = nullptr; float tau = 0.f, omt = 0.f;
  if (a.fuse_polyak && dsrc) {
    // the target's pointers and tau are read from the descriptor HERE (an opaque pointer: the compiler cannot merge these loads with the kernel's first read of the
    // descriptor), not carried in scalar registers through the products: the kernel sits at its 128-VGPR budget and three more live pointers spilled to scratch
    const il_sac* dp = dsrc;
    asm volatile("" : "+s"(dp));
    const SacWs ws = sac_ws(dp->state_dim, dp->action_dim, dp->hidden, dp->batch);
    tgt = dp->target; tau = (float)dp->polyak; omt = (float)(1.0 - dp->polyak);
    pkt = pkf ? dp->workspace + ws.pk_tf + (pkf - a.pk_f) : nullptr;   // the same network's slab of the target's forward-order copy
#pragma unroll
    for (int u = 0; u < 4; ++u) tv[u] = gload4(tgt + poff + (int64_t)(n0 + sf + 16 * u) * H + k0 + sr);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int rr = sf + 16 * u, n = n0 + rr, k = k0 + sr;
    const int64_t o = poff + (int64_t)n * H + k;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + rr * DWB_LD + sr);
#pragma unroll
    for (int c = 0; c < 4; ++c) { float pp = pv[u][c], mm = mv[u][c], v2 = vv[u][c]; adam_update(pp, gv[c], mm, v2, ac); pv[u][c] = pp; mv[u][c] = mm; vv[u][c] = v2; }
    *reinterpret_cast<f32x4*>(a.params + o) = pv[u]; *reinterpret_cast<f32x4*>(a.opt.m + o) = mv[u]; *reinterpret_cast<f32x4*>(a.opt.v + o) = vv[u];
    if (pkf) {
      *reinterpret_cast<f32x4*>(pkf + packed_fwd_index(n, k, H)) = pv[u];   // k .. k+3 of row n: one 16-byte lane of PF
      *reinterpret_cast<f32x4*>(Gs + rr * DWB_LD + sr) = pv[u];
    }
  }
  if (a.fuse_polyak && dsrc) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + sf + 16 * u, k = k0 + sr;
#pragma unroll
      for (int q = 0; q < 4; ++q) tv[u][q] = __fadd_rn(__fmul_rn(tv[u][q], tau), __fmul_rn(omt, pv[u][q]));
      *reinterpret_cast<f32x4*>(tgt + poff + (int64_t)n * H + k) = tv[u];
      if (pkt) *reinterpret_cast<f32x4*>(pkt + packed_fwd_index(n, k, H)) = tv[u];
    }
  }
  if (!pkf) return;
  __syncthreads();
  {  // PB: rows n .. n+3 of column k are one 16-byte lane; thread t takes column t % 64 and row quads (t / 64) + 4 u
    const int kc = tid & 63;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rq = (tid >> 6) + 4 * u;   // row quad 0..15
      f32x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = Gs[(4 * rq + r) * DWB_LD + kc];
      *reinterpret_cast<f32x4*>(pkb + packed_bwd_index(n0 + 4 * rq, k0 + kc, H)) = w;
    }
  }
}
static inline int dw_block64_count(int H, int nets) { return (H / DWB) * (H / DWB) * nets; }

// One network's optimiser step as uniform block jobs: [0, nbh^2) the H x H layer (bias 2 with the k0 = 0 blocks), then nbh x kin blocks of layer 1 (bias 1), then
// nout x nbh blocks of layer 3 (bias 3). Returns false when `job` is past the network's list.
__host__ __device__ static inline int dw_block_jobs(int IN, int H, int OUT) { const int nbh = H / DWS; return nbh * nbh + nbh * ((IN + DWS - 1) / DWS) + ((OUT + DWS - 1) / DWS) * nbh; }
// the gate of the single-learner kernel's block jobs (dw_block.hpp): overlapped launches wait for the launch that produces dZ / the activations while p / m / v are on
// their way; an update whose hand-off expired - in an earlier launch of this update, or in that wait - never reaches the weights ([IL_SYNC_POISON])
struct DwOvGate {
  __device__ __forceinline__ bool operator()(const DwArgs& a) const { dw_ov_wait<true>(a); return a.sync && __syncthreads_or(sync_poisoned(reinterpret_cast<const long long*>(a.sync)) ? 1 : 0); }
};
template <bool PEER = false, class Gate = DwNoGate>
__device__ __forceinline__ void dw_block_job(const DwArgs& a, int net, int job, float* smem, const DwPeer* pp = nullptr, int pjob = -1) {
  const int IN = a.in_dim, H = a.hidden, OUT = a.out_dim, nbh = H / DWS, kin = (IN + DWS - 1) / DWS;
  const int64_t pbase = (int64_t)net * a.net_stride;
  const int64_t oW1 = pbase, ob1 = oW1 + (int64_t)H * IN, oW2 = ob1 + H, ob2 = oW2 + (int64_t)H * H, oW3 = ob2 + H, ob3 = oW3 + (int64_t)OUT * H;
  const float* h1 = a.h1 + net * a.h_net_stride; const float* h2 = a.h2 + net * a.h_net_stride;
  const float* dz1 = a.dz1 + net * a.h_net_stride; const float* dz2 = a.dz2 + net * a.h_net_stride;
  if (job < nbh * nbh) {
    int nb = job / nbh, kb = job % nbh;
    if (IL_DW_XCD_BLOCKS && nbh == 8) {
      // Workgroup b runs on XCD b % 8 (and a network's job list starts at a multiple of 8), and every XCD has a private L2: with (n, k) = (job / 8, job % 8) the eight
      // blocks that share a dZ panel sit on eight different XCDs and every XCD pulls EVERY dZ panel through the fabric. Here XCD x owns the 2 x 4 rectangle of blocks
      // n in {2 (x / 2), + 1}, k in {4 (x % 2) .. + 3}: 2 + 4 panels per XCD instead of 8 + 1. A re-labelling of the jobs: same tiles, same bits.
      const int x = job & 7, slot = job >> 3;
      nb = 2 * (x >> 1) + (slot >> 2); kb = 4 * (x & 1) + (slot & 3);
    }
    dw_block32<PEER, Gate>(a, dz2, H, h1, H, nb * DWS, kb * DWS, oW2, ob2, a.pk_f ? a.pk_f + (size_t)net * H * H : nullptr, a.pk_b ? a.pk_b + (size_t)net * H * H : nullptr, smem, pp, pjob);
    return;
  }
  job -= nbh * nbh;
  if (job < nbh * kin) { dw_block32<PEER, Gate>(a, dz1, H, a.x0 + net * a.x0_net_stride, IN, (job / kin) * DWS, (job % kin) * DWS, oW1, ob1, nullptr, nullptr, smem, pp, pjob); return; }
  job -= nbh * kin;
  dw_block32<PEER, Gate>(a, a.dz3 + net * a.dz3_net_stride, OUT, h2, H, (job / nbh) * DWS, (job % nbh) * DWS, oW3, ob3, nullptr, nullptr, smem, pp, pjob);
}
static inline int dw_block32_count(int H, int nets) { return (H / DWS) * (H / DWS) * nets; }
static inline bool dw_block32_fits(int H, int B) { return H % DWS == 0 && B % DWS_ROWS == 0; }

template <bool PEER>
__device__ __forceinline__ void dw_adam_kernel(const DwArgs& a, const DwPeer* pp, float* smem) {
  IL_TL(a.log_alpha ? 2 : 1, 0);
  const int st_kid = a.log_alpha ? IL_ST_DW_ACTOR : IL_ST_DW_CRITIC;
  IL_ST_BEGIN(st_kid);   // [1] critic launch, [2] actor launch (the one with the alpha / polyak tail)
  if (a.n_big_blocks > 0) {   // [0, n_big_blocks): every layer's dW + bias as 32 x 32 block jobs through LDS (x0 must be feature-major); then the tail blocks
    const int bx = (int)blockIdx.x;
    if (bx < a.n_big_blocks) {
      const int per_net = dw_block_jobs(a.in_dim, a.hidden, a.out_dim);
      dw_block_job<PEER>(a, bx / per_net, bx % per_net, smem, pp, bx);
      IL_TL_END(a.log_alpha ? 2 : 1);
      IL_ST_END(st_kid);
      return;
    }
    DwArgs r = a;
    r.n_dw_blocks = 0;   // nothing but the tail is left
    dw_adam_body<IL_DW_U, true, PEER>(r, bx - a.n_big_blocks, (int)gridDim.x - a.n_big_blocks, pp);
    IL_TL_END(a.log_alpha ? 2 : 1);
    IL_ST_END(st_kid);
    return;
  }
  dw_adam_body<IL_DW_U>(a, (int)blockIdx.x, (int)gridDim.x);
  IL_TL_END(a.log_alpha ? 2 : 1);
  IL_ST_END(st_kid);
}
// (round 6) Two kernels instead of one: the wave-per-tile jobs (dw_tile with 16 operand lanes in flight) cost 256 VGPRs + AGPRs, i.e. ONE workgroup per CU and none beside a
// pair-mode workgroup - which did not matter while an optimiser launch had the chip to itself, but an overlapped launch (il_sac_update_gather_overlap) is resident next to
// the forward / policy launch of the other stream. k_dw_adam is the block form alone (32 x 32 block jobs + the tail: what every single-GPU update of a block-shaped
// network launches); k_dw_adam_tiles is the general body (shapes without the block form, behavioural cloning). Same device functions, same bits.
__global__ __launch_bounds__(256) void k_dw_adam(DwArgs a) {   // a.n_big_blocks > 0
  __shared__ __attribute__((aligned(16))) float smem[2 * DWS * DWS_LD];
  IL_TL(a.log_alpha ? 2 : 1, 0);
  const int st_kid = a.log_alpha ? IL_ST_DW_ACTOR : IL_ST_DW_CRITIC;
  IL_ST_BEGIN(st_kid);
  const int bx = (int)blockIdx.x;
  if (bx < a.n_big_blocks) {
    const int per_net = dw_block_jobs(a.in_dim, a.hidden, a.out_dim);
    dw_block_job<false, DwOvGate>(a, bx / per_net, bx % per_net, smem, nullptr, bx);
  } else {
    const int tb = bx - a.n_big_blocks, ntb = (int)gridDim.x - a.n_big_blocks;
    // overlapped launches: the target step needs nothing of the launch this one waits for (the critics were stepped by this stream's previous launch, the targets' last
    // readers are gone) - it runs while that launch is still busy; block 0 then waits and closes the update (Adam(log alpha), Philox counter, [IL_SYNC_MAIN_EPOCH])
    const bool poisoned = a.sync && __syncthreads_or(sync_poisoned(reinterpret_cast<const long long*>(a.sync)) ? 1 : 0);   // (the tail still closes the update: counters and signals keep the pipeline moving until the host has seen the flag)
    if (a.ov_stage) {
      if (!poisoned) dw_tail_polyak(a, tb, ntb);
      if (tb == 0) { dw_ov_wait<true>(a); dw_tail_alpha<false>(a, nullptr, poisoned || sync_poisoned(reinterpret_cast<const long long*>(a.sync))); }
    } else {
      if (tb == 0) dw_tail_alpha<false>(a, nullptr, poisoned);
      if (!poisoned) dw_tail_polyak(a, tb, ntb);
    }
  }
  IL_TL_END(a.log_alpha ? 2 : 1);
  IL_ST_END(st_kid);
  if (a.ov_stage) ov_done(reinterpret_cast<long long*>(a.sync), a.ov_stage - 1);
}
__global__ __launch_bounds__(256) void k_dw_adam_tiles(DwArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2 * DWS * DWS_LD];
  dw_adam_kernel<false>(a, nullptr, smem);
}
static inline void launch_dw_adam(const DwArgs& a, int grid, hipStream_t st) {
  if (a.n_big_blocks > 0) k_dw_adam<<<grid, 256, 0, st>>>(a); else k_dw_adam_tiles<<<grid, 256, 0, st>>>(a);
}
// the same launch with the gradient exchange of a data-parallel run inside its block jobs (il_sac_update_gather_peer; n_big_blocks > 0 is checked by the caller)
__global__ __launch_bounds__(256) void k_dw_adam_peer(DwArgs a, DwPeer p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * DWS * DWS_LD];
  dw_adam_kernel<true>(a, &p, smem);
}

static int repack_blocks(int H) { return ceil_div(H * H / 16, 256); }
__host__ __device__ static inline int dw_blocks(int IN, int H, int OUT, int nets, int skip_big = 0, int jobs_per_block = 4) {
  const int nt_h = H / 16, nt_out = (OUT + 15) / 16;
  const int per_net = nt_h * ((IN + 15) / 16) + (skip_big ? 0 : nt_h * nt_h) + nt_out * nt_h + 2 * nt_h + nt_out;
  return (per_net * nets + jobs_per_block - 1) / jobs_per_block;
}

// generic elementwise Adam over a flat arena (data-parallel path and stand-alone use)
__global__ __launch_bounds__(256) void k_adam_flat(float* __restrict__ p, const float* __restrict__ g, il_adam opt, int64_t n) {
  const adam_consts ac = load_adam_consts(opt);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pp = p[i], mm = opt.m[i], vv = opt.v[i];
    adam_update(pp, g[i], mm, vv, ac);
    p[i] = pp; opt.m[i] = mm; opt.v[i] = vv;
  }
}
__global__ void k_tick(il_adam opt) { if (threadIdx.x == 0 && blockIdx.x == 0) adam_tick(opt); }
__global__ __launch_bounds__(256) void k_polyak(float* __restrict__ t, const float* __restrict__ p, int64_t n, double tau_) {
  const float omt = (float)(1.0 - tau_), tau = (float)tau_;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    t[i] = __fadd_rn(__fmul_rn(t[i], tau), __fmul_rn(omt, p[i]));
}

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
static int check_sac(const il_sac* d, const il_batch* b) {
  IL_CHECK_ARG(d && b, "il_sac: null descriptor");
  IL_CHECK_ARG(d->hidden % 64 == 0 && d->hidden >= 64 && d->hidden <= 256, "il_sac: hidden=%d must be a multiple of 64 in [64,256]", d->hidden);
  IL_CHECK_ARG(d->batch % IL_TILE_R == 0 && d->batch > 0, "il_sac: batch=%d must be a positive multiple of %d", d->batch, IL_TILE_R);
  IL_CHECK_ARG(b->n == d->batch, "il_sac: batch rows %d != descriptor batch %d", b->n, d->batch);
  IL_NO_GATHER(b, "il_sac");
  IL_CHECK_ARG(d->action_dim >= 1 && 2 * d->action_dim <= 16, "il_sac: action_dim=%d unsupported (2A <= 16)", d->action_dim);
  IL_CHECK_ARG(d->state_dim >= 1 && d->state_dim + d->action_dim <= 508, "il_sac: state_dim too large");
  const SacWs ws = sac_ws(d->state_dim, d->action_dim, d->hidden, d->batch);
  if (!d->workspace || d->workspace_floats < ws.total) return il_set_error(IL_ERR_WORKSPACE, "il_sac: workspace too small (%lld < %lld floats)", (long long)d->workspace_floats, (long long)ws.total);
  return IL_OK;
}

extern "C" int64_t il_mlp_numel(int32_t in_dim, int32_t hidden, int32_t out_dim) { return mlp_numel(in_dim, hidden, out_dim); }
extern "C" int64_t il_mlp_stride(int32_t in_dim, int32_t hidden, int32_t out_dim) { return net_stride(in_dim, hidden, out_dim); }
extern "C" int64_t il_sac_workspace_floats(int32_t S, int32_t A, int32_t H, int32_t B) { return sac_ws(S, A, H, B).total; }

__host__ __device__ static inline int dw_block_jobs_n(int IN, int H, int OUT) { const int nbh = H / 32; return nbh * nbh + nbh * ((IN + 31) / 32) + ((OUT + 31) / 32) * nbh; }   // = dw_block_jobs (DWS = 32)
__host__ __device__ static inline bool dw_block32_on() {   // host: IL_DW_BLOCK32=0|1 overrides the build's default (developer A/B switch; same bits either way)
#ifdef __HIP_DEVICE_COMPILE__
  return IL_DW_BLOCK32 != 0;   // (device-side callers - the population launch - lay out their own grid and ignore n_big_blocks)
#else
  static const int on = [] { const char* e = getenv("IL_DW_BLOCK32"); return e ? (e[0] != '0' ? 1 : 0) : (IL_DW_BLOCK32 != 0 ? 1 : 0); }();
  return on != 0;
#endif
}
__host__ __device__ static DwArgs critic_dw_args(const il_sac* d, uint32_t flags) {
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, IN = S + A;
  const SacWs ws = sac_ws(S, A, H, B);
  DwArgs a = {};
  a.params = d->critic; a.grads = d->critic_grad; a.opt = d->critic_opt; a.grads_only = (flags & IL_FLAG_GRADS_ONLY) ? 1 : 0;
  a.n_nets = 2; a.net_stride = net_stride(IN, H, 1); a.in_dim = IN; a.hidden = H; a.out_dim = 1; a.batch = B; a.sync = d->sync;   // (sync: the poison check of k_dw_adam; the tail that signals through it belongs to the actor's launch)
  a.x0 = d->workspace + ws.c_x0; a.ld_x0 = 0; a.x0_transposed = 1; a.x0_net_stride = 0;
  a.h1 = d->workspace + ws.c_h1; a.h2 = d->workspace + ws.c_h2; a.dz1 = d->workspace + ws.c_dz1; a.dz2 = d->workspace + ws.c_dz2; a.h_net_stride = (int64_t)B * H;
  a.dz3 = d->workspace + ws.c_dz3; a.dz3_net_stride = B;
  a.pk_f = d->workspace + ws.pk_cf; a.pk_b = d->workspace + ws.pk_cb;
  a.n_big_blocks = (dw_block32_on() && H % 32 == 0 && B % 128 == 0) ? 2 * dw_block_jobs_n(IN, H, 1) : 0;   // every layer + the biases as 32 x 32 block jobs (dw_block32)
  a.jobs_per_block = 4;
  a.n_dw_blocks = a.n_big_blocks > 0 ? a.n_big_blocks : dw_blocks(IN, H, 1, 2);
  if ((flags & IL_FLAG_POP_FUSE_POLYAK) && !a.grads_only && d->target) a.fuse_polyak = 1;   // (the target's pointers are fetched from `desc` where they are used: dw_block64)
  return a;
}

// k_policy_critic helpers need all (2 + helpers) * nt workgroups resident together; IL_PC_SPLIT=0 keeps the pair's second arriver doing the tail.
static int device_cu_count() {
  static const int n = [] { int dev = 0, cu = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0; return cu; }();
  return n;
}
// IL_CHAIN_XCD_NETS=1: single-learner k_sac_chain / k_policy_critic launches place each network's tile workgroups on ONE XCD (chain_decode_xcd; grid = 8 * nt)
static bool chain_xcd_nets(int nt) { static const int on = [] { const char* e = getenv("IL_CHAIN_XCD_NETS"); return e && e[0] == '1' ? 1 : 0; }(); return on != 0 && 8 * nt <= device_cu_count(); }
static int pc_helpers(int nt) {
  static const int on = [] { const char* e = getenv("IL_PC_SPLIT"); return e && e[0] == '0' ? 0 : 1; }();
  return (on && (2 + IL_PC_HELPERS) * nt <= device_cu_count()) ? IL_PC_HELPERS : 0;
}
// Pair mode (k_sac_chain_pair / k_policy_critic_pair): H = 256, first layers of at most four 16-wide k-blocks, every workgroup of the launch resident. IL_PAIR=0 keeps
// the 16-wave workgroups (developer A/B switch; same bits either way).
static bool pair_env() { static const int on = [] { const char* e = getenv("IL_PAIR"); return e && e[0] == '0' ? 0 : 1; }(); return on != 0; }
static bool pair_shape_ok(const il_sac* d) {
  const auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  return d->hidden == 256 && round_up16(d->state_dim + d->action_dim) <= 64 && al(d->actor) && al(d->critic) && al(d->target);
}
// Pair-mode workgroups ask for (nearly) a whole CU's LDS: their tile carve + W1s needs up to 121 KB, and a 512-thread workgroup with ~170 VGPRs would otherwise leave room
// on its CU for a discriminator workgroup of the other stream - measured (round 4, first build): k_gail_grad's slowest workgroup 14.3 -> 17.7 us, its AdamW step and with it
// the relabel 4 us later. One workgroup per CU, as the 16-wave workgroups had by their register footprint.
static size_t pair_lds_bytes() { static const size_t n = [] { const char* e = getenv("IL_PAIR_LDS_KB"); const int kb = e ? atoi(e) : 160; return (size_t)(kb < 124 ? 124 : (kb > 160 ? 160 : kb)) * 1024; }(); return n; }   // (developer A/B: 124 KB still lets a small workgroup of another launch share the CU)
#define IL_PAIR_LDS_BYTES pair_lds_bytes()
static int pair_lds_ready() {
  static const int rc = [] {
    if (hipFuncSetAttribute((const void*)k_sac_chain_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IL_PAIR_LDS_BYTES) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)k_policy_critic_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IL_PAIR_LDS_BYTES) != hipSuccess) return 1;
    return 0;
  }();
  if (rc) (void)hipGetLastError();
  return rc == 0;
}
static bool chain_pair_ok(const il_sac* d, int relabel, int G) { return pair_env() && pair_shape_ok(d) && chain_pair_workgroups(d->batch / IL_TILE_R, relabel, G) <= device_cu_count() && pair_lds_ready(); }
static bool policy_critic_pair_ok(const il_sac* d) {
  const int nt = d->batch / IL_TILE_R, hp = pc_helpers(nt);
  return pair_env() && pair_shape_ok(d) && hp > 0 && (4 + hp) * nt <= device_cu_count() && pair_lds_ready();
}
static void launch_policy_critic(const il_sac* d, const il_batch* b, float* out_logp, float* out_q, size_t lds, hipStream_t st, int overlap = 0) {
  const int H = d->hidden, nt = d->batch / IL_TILE_R;
  IL_TRACE("k_policy_critic", st);
  const int hp = pc_helpers(nt);
  if (policy_critic_pair_ok(d)) { k_policy_critic_pair<<<(4 + hp) * nt, 512, IL_PAIR_LDS_BYTES, st>>>(*d, *b, out_logp, out_q, hp, overlap); return; }
  const bool px = hp > 0 && hp <= 6 && chain_xcd_nets(nt);
  k_policy_critic<<<px ? 8 * nt : (2 + hp) * nt, tile_threads(H), lds, st>>>(*d, *b, out_logp, out_q, nullptr, nullptr, px ? (hp | IL_PC_XCD_NETS) : hp);
}
extern "C" int il_sac_critic_step(const il_sac* d, const il_batch* b, const float* eps_next, uint32_t flags, il_stream_t stream_) {
  if (int rc = check_sac(d, b)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, nt = B / IL_TILE_R;
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
  { IL_TRACE("k_actor_fwd", st); k_actor_fwd<<<nt, tile_threads(H), lds, st>>>(*d, *b, eps_next, nullptr, 1, nullptr, nullptr); }
  { IL_TRACE("k_critic_fwd", st); k_critic_fwd<<<4 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr); }
  { IL_TRACE("k_critic_bwd", st); k_critic_bwd<<<2 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr); }
  DwArgs a = critic_dw_args(d, flags);
  { IL_TRACE("k_dw_adam_critic", st); launch_dw_adam(a, a.n_dw_blocks, st); }
  IL_CHECK_LAUNCH("il_sac_critic_step");
  return IL_OK;
}

__host__ __device__ static DwArgs actor_dw_args(const il_sac* d, const il_batch* b, uint32_t flags) {
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch;
  const SacWs ws = sac_ws(S, A, H, B);
  DwArgs a = {};
  a.params = d->actor; a.grads = d->actor_grad; a.opt = d->actor_opt; a.grads_only = (flags & IL_FLAG_GRADS_ONLY) ? 1 : 0;
  a.n_nets = 1; a.net_stride = 0; a.in_dim = S; a.hidden = H; a.out_dim = 2 * A; a.batch = B;
  a.x0 = d->workspace + ws.a_x0; a.ld_x0 = 0; a.x0_transposed = 1; a.x0_net_stride = 0;   // (round 3) s^T from the workspace; `b` keeps the row-major states for callers that build their own arguments
  a.h1 = d->workspace + ws.a_h1; a.h2 = d->workspace + ws.a_h2; a.dz1 = d->workspace + ws.a_dz1; a.dz2 = d->workspace + ws.a_dz2; a.h_net_stride = 0;
  a.dz3 = d->workspace + ws.a_dz3; a.dz3_net_stride = 0;
  a.pk_f = d->workspace + ws.pk_af; a.pk_b = d->workspace + ws.pk_ab;
  a.n_big_blocks = (dw_block32_on() && H % 32 == 0 && B % 128 == 0) ? dw_block_jobs_n(S, H, 2 * A) : 0;
  a.jobs_per_block = 4;
  a.n_dw_blocks = a.n_big_blocks > 0 ? a.n_big_blocks : dw_blocks(S, H, 2 * A, 1);
  a.log_alpha = d->log_alpha; a.alpha_grad = d->alpha_grad; a.alpha_opt = d->alpha_opt; a.alpha_part = d->workspace + ws.alpha_part; a.n_alpha_part = B / IL_TILE_R;
  a.target = d->target; a.polyak_src = d->critic; a.polyak_n = 2 * net_stride(S + A, H, 1); a.tau = d->polyak; a.noise_counter = d->noise_counter; a.sync = d->sync;
  if ((flags & IL_FLAG_POP_FUSE_POLYAK) && !a.grads_only && d->target) {   // the H x H layers and their copies were stepped by the critic launch's blocks (critic_dw_args)
    a.polyak_fused = 1;
    return a;
  }
  a.pk_target = d->workspace + ws.pk_tf; a.pk_critic = d->workspace + ws.pk_cf; a.pk_n = 2 * (int64_t)H * H;   // the FORWARD-order copies of both target critics only: targets are never back-propagated, so their PB copies (pk_tb) have no reader (round 2: 1.5 MB of polyak traffic per update removed)
  return a;
}

extern "C" int il_sac_actor_step(const il_sac* d, const il_batch* b, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream_) {
  if (int rc = check_sac(d, b)) return rc;
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, nt = B / IL_TILE_R;
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 3), 256, 0, st>>>(*d, 0x07u, nullptr); }  // actor + critics (the critic may have been stepped by il_adam_step)
  { IL_TRACE("k_actor_fwd", st); k_actor_fwd<<<nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, eps_cur, 2, nullptr, nullptr); }
  launch_policy_critic(d, b, out_logp, out_q, lds, st);
  DwArgs a = actor_dw_args(d, b, flags);
  const int tail = (flags & IL_FLAG_GRADS_ONLY) ? 1 : IL_TAIL_BLOCKS;
  { IL_TRACE("k_dw_adam_actor", st); launch_dw_adam(a, a.n_dw_blocks + tail, st); }
  IL_CHECK_LAUNCH("il_sac_actor_step");
  return IL_OK;
}

// k_sac_chain needs its 6 * nt workgroups resident together; IL_SAC_CHAIN=0 keeps the three separate launches (developer A/B switch).
static bool chain_enabled() { static const int on = [] { const char* e = getenv("IL_SAC_CHAIN"); return e && e[0] == '0' ? 0 : 1; }(); return on != 0; }
extern "C" int il_sac_update(const il_sac* d, const il_batch* b, const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags,
                             il_stream_t stream_) {
  if (int rc = check_sac(d, b)) return rc;
  if (flags & IL_FLAG_GRADS_ONLY) return il_set_error(IL_ERR_UNSUPPORTED, "il_sac_update: IL_FLAG_GRADS_ONLY needs the split critic/actor entry points");
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, nt = B / IL_TILE_R;
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  const bool whole = !(flags & (IL_FLAG_SAC_SKIP_FORWARD | IL_FLAG_SAC_FORWARD_ONLY));
  if (whole && chain_enabled() && 6 * nt <= device_cu_count()) {   // forward + critic loss chained per tile in one co-resident launch (k_sac_chain)
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
    ChainRelabel cr = {}; cr.xcd_nets = chain_xcd_nets(nt) ? 1 : 0;
    if (chain_pair_ok(d, 0, 0)) { IL_TRACE("k_sac_chain", st); k_sac_chain_pair<<<chain_pair_workgroups(nt, 0, 0), 512, IL_PAIR_LDS_BYTES, st>>>(*d, *b, eps_next, eps_cur, nullptr, nullptr, cr); }
    else { IL_TRACE("k_sac_chain", st); k_sac_chain<<<(cr.xcd_nets ? 8 : 6) * nt, tile_threads(H), lds, st>>>(*d, *b, eps_next, eps_cur, nullptr, nullptr, cr); }
    flags |= IL_FLAG_SAC_SKIP_FORWARD | 0x80000000u;
  }
  if ((flags & IL_FLAG_SAC_FORWARD_ONLY) && !(flags & IL_FLAG_SAC_SKIP_FORWARD) && chain_enabled() && 6 * nt <= device_cu_count()) {
    // the four forward passes as one launch chained per tile (the target critics wait for their tile's actor(s') inside it), no critic backward
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
    ChainRelabel fo = {}; fo.fwd_only = 1;
    fo.xcd_nets = chain_xcd_nets(nt) ? 1 : 0;
    { IL_TRACE("k_sac_chain", st); k_sac_chain<<<(fo.xcd_nets ? 8 : 6) * nt, tile_threads(H), lds, st>>>(*d, *b, eps_next, eps_cur, nullptr, nullptr, fo); }
    IL_CHECK_LAUNCH("il_sac_update");
    return IL_OK;
  }
  if (!(flags & IL_FLAG_SAC_SKIP_FORWARD)) {
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
    // the actor is unchanged until the last kernel of the update: both of its forward passes share one launch; neither this
    // nor the critic/target forward reads the rewards, so a caller may overlap the reward relabel with them (IL_FLAG_SAC_FORWARD_ONLY)
    { IL_TRACE("k_actor_fwd", st); k_actor_fwd<<<2 * nt, tile_threads(H), lds, st>>>(*d, *b, eps_next, eps_cur, 0, nullptr, nullptr); }
    { IL_TRACE("k_critic_fwd", st); k_critic_fwd<<<4 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr); }
  }
  if (!(flags & IL_FLAG_SAC_FORWARD_ONLY)) {
    if (!(flags & 0x80000000u)) { IL_TRACE("k_critic_bwd", st); k_critic_bwd<<<2 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr); }
    DwArgs ca = critic_dw_args(d, flags);
    { IL_TRACE("k_dw_adam_critic", st); launch_dw_adam(ca, ca.n_dw_blocks, st); }
    launch_policy_critic(d, b, out_logp, out_q, lds, st);
    DwArgs aa = actor_dw_args(d, b, flags);
    { IL_TRACE("k_dw_adam_actor", st); launch_dw_adam(aa, aa.n_dw_blocks + IL_TAIL_BLOCKS, st); }
  }
  IL_CHECK_LAUNCH("il_sac_update");
  return IL_OK;
}

// gather workgroups appended to k_sac_chain by il_sac_update_gather: one 16-byte lane per thread (this is also what the caller writes to [IL_SYNC_GATHER_WGS])
// in-launch waits of k_sac_chain / k_policy_critic that gave up since the workspace was created (must be 0); synchronous, for tests and post-mortems
extern "C" int il_sac_handoff_timeouts(const il_sac* d, uint32_t* out_host) {
  IL_CHECK_ARG(d && d->workspace && out_host, "il_sac_handoff_timeouts: bad arguments");
  const SacWs ws = sac_ws(d->state_dim, d->action_dim, d->hidden, d->batch);
  const hipError_t e = hipMemcpy(out_host, reinterpret_cast<const unsigned*>(d->workspace + ws.chain_ctr) + (d->batch / IL_TILE_R) * IL_CTR_STRIDE + 1, sizeof(uint32_t), hipMemcpyDeviceToHost);
  return e == hipSuccess ? IL_OK : il_set_error(IL_ERR_HIP, "il_sac_handoff_timeouts: %s", hipGetErrorString(e));
}

extern "C" int32_t il_sac_chain_gather_workgroups(int32_t batch, int32_t row_floats, int32_t hidden) {
  return (int32_t)(((int64_t)batch * (row_floats / 4) + tile_threads(hidden) - 1) / tile_threads(hidden));
}

extern "C" int32_t il_gail_step_workgroups(const il_disc* d) {
  if (!d) return 0;
  return (int32_t)((disc_layout(d->state_dim + (d->state_only ? 0 : d->action_dim), d->hidden, d->spectral_norm).P + 255) / 256);
}

// data-parallel with the exchange inside the optimiser launches (il_sac_update_gather_peer): bucket sizes and job counts a caller lays its peer regions out for
static int64_t round_up4(int64_t n) { return (n + 3) & ~(int64_t)3; }
extern "C" int64_t il_sac_peer_bucket_floats(const il_sac* d, int32_t which) {   // 0: the twin critic's arena; 1: the actor's arena with log alpha's gradient in the slot behind it, padded
  if (!d) return 0;                                                              // to 16 bytes - the layout of parallel.GradBuckets, so that an exchange LAUNCH can serve the same region
  return which == 0 ? 2 * net_stride(d->state_dim + d->action_dim, d->hidden, 1) : round_up4(mlp_numel(d->state_dim, d->hidden, 2 * d->action_dim) + 1);
}
extern "C" int32_t il_sac_peer_jobs(const il_sac* d, int32_t which) {   // arrival lines of the bucket (0: this shape's optimiser launches have no block form: use the exchange launches)
  if (!d || !(dw_block32_on() && d->hidden % 32 == 0 && d->batch % 128 == 0)) return 0;
  return which == 0 ? 2 * dw_block_jobs_n(d->state_dim + d->action_dim, d->hidden, 1) : dw_block_jobs_n(d->state_dim, d->hidden, 2 * d->action_dim) + 1;
}
static int check_peer_jobs(const il_peer_bucket* x, int64_t n, int32_t jobs, const char* what) {
  IL_CHECK_ARG(x->world >= 1 && x->world <= IL_PEER_MAX_RANKS && x->rank >= 0 && x->rank < x->world && x->epoch && x->status && (x->window_offset & 255) == 0, "il_sac_update_gather_peer: bad %s descriptor", what);
  IL_CHECK_ARG(jobs > 0, "il_sac_update_gather_peer: this shape's optimiser launches have no block form (il_sac_peer_jobs == 0): use the exchange launches");
  IL_CHECK_ARG(x->n == n && x->n_jobs >= jobs, "il_sac_update_gather_peer: the %s bucket must hold %lld floats and %d arrival lines (got %lld, %d)", what, (long long)n, jobs, (long long)x->n, x->n_jobs);
  for (int r = 0; r < x->world; ++r) IL_CHECK_ARG(x->windows[r], "il_sac_update_gather_peer: window of rank %d is not mapped", r);
  return IL_OK;
}
static int sac_update_gather_impl(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const il_disc* relabel, float* rewards_out,
                                  const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, const il_peer_bucket* peer_critic,
                                  const il_peer_bucket* peer_actor, il_stream_t stream_, il_stream_t stream_b_ = nullptr) {   // stream_b_: il_sac_update_gather_overlap
  if (int rc = check_sac(d, rows)) return rc;
  if (peer_critic || peer_actor) {
    IL_CHECK_ARG(peer_critic && peer_actor && !(flags & IL_FLAG_GRADS_ONLY), "il_sac_update_gather_peer: both buckets, and no IL_FLAG_GRADS_ONLY (the optimiser steps run inside)");
    if (int rc = check_peer_jobs(peer_critic, il_sac_peer_bucket_floats(d, 0), il_sac_peer_jobs(d, 0), "critic")) return rc;
    if (int rc = check_peer_jobs(peer_actor, il_sac_peer_bucket_floats(d, 1), il_sac_peer_jobs(d, 1), "actor")) return rc;
  }
  ChainRelabel rl = {};
  if (relabel) {
    IL_CHECK_ARG(d->sync && relabel->sync == d->sync, "il_sac_update_gather: the inline relabel needs the il_sync counters shared with the discriminator");
    IL_CHECK_ARG(!rewards, "il_sac_update_gather: pass either `rewards` or `relabel`");
    if (relabel->state_only || relabel->state_dim != d->state_dim || relabel->action_dim != d->action_dim)
      return il_set_error(IL_ERR_UNSUPPORTED, "il_sac_update_gather: inline relabel needs a discriminator on the critic's (s, a) input");
    const size_t spare = (size_t)(tile_threads(d->hidden) / 64) * 256 + 256 - 32;   // the workgroup's LDS behind q16 / dz3 / rewards16 (tile_lds_bytes)
    if (reward_lds_floats(d->state_dim + d->action_dim, relabel->hidden) > spare)
      return il_set_error(IL_ERR_UNSUPPORTED, "il_sac_update_gather: discriminator too large for the inline relabel (%zu > %zu LDS floats)", reward_lds_floats(d->state_dim + d->action_dim, relabel->hidden), spare);
    rl.dd = *relabel; rl.on = 1; rl.n_reduce = il_gail_step_workgroups(relabel); rl.out = rewards_out;
  }
  if (flags & IL_FLAG_SAC_WAIT_INDICES) { IL_CHECK_ARG(d->sync, "il_sac_update_gather: IL_FLAG_SAC_WAIT_INDICES needs the il_sync counters"); rl.wait_indices = 1; }
  rl.local_rewards = (!rewards && !relabel) ? 1 : 0;
  { static const int early = [] { const char* e = getenv("IL_EARLY_DRAW"); return e && e[0] == '0' ? 0 : 1; }(); rl.early_draw = early; }
  if (flags & IL_FLAG_SAC_STAGED_ROWS) {   // `ring` = the dense slab the resident sampler staged this update's rows into (il_gail_disc_step_draw_staged): no index trip
    IL_CHECK_ARG(ring && !ring->gather && ring->n == d->batch && (flags & IL_FLAG_SAC_WAIT_INDICES), "il_sac_update_gather: IL_FLAG_SAC_STAGED_ROWS takes a dense batch of %d rows and waits for [IL_SYNC_INDICES]", d->batch);
  } else
  IL_CHECK_ARG(ring && ring->gather && ring->gather_capacity > 0 && ring->n == d->batch, "il_sac_update_gather: `ring` must carry the %d drawn indices (il_batch.gather)", d->batch);
  IL_CHECK_ARG(rows->states && ring->states && rows->ld_states == ring->ld_states && ring->ld_states % 4 == 0, "il_sac_update_gather: rows / ring must be packed rows of the same width");
  IL_CHECK_ARG(!(flags & (IL_FLAG_SAC_SKIP_FORWARD | IL_FLAG_SAC_FORWARD_ONLY)), "il_sac_update_gather: whole updates (or, with IL_FLAG_GRADS_ONLY, everything up to the critic gradients)");
  IL_CHECK_ARG(!(flags & IL_FLAG_GRADS_ONLY) || d->critic_grad, "il_sac_update_gather: IL_FLAG_GRADS_ONLY needs the critic_grad arena");
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, nt = B / IL_TILE_R;
  const int G = il_sac_chain_gather_workgroups(B, ring->ld_states, H);
  if (6 * nt + G > device_cu_count()) return il_set_error(IL_ERR_UNSUPPORTED, "il_sac_update_gather: %d workgroups cannot be co-resident on %d CUs (gather first, then il_sac_update)", 6 * nt + G, device_cu_count());
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  if (stream_b_) {   // the four launches alternate over two streams and hand over through [IL_SYNC_OV_EPOCH] (include/il_hip.h il_sac_update_gather_overlap)
    hipStream_t sb = (hipStream_t)stream_b_;
    IL_CHECK_ARG(d->sync && sb != st, "il_sac_update_gather_overlap: needs the il_sync counters and two different streams");
    IL_CHECK_ARG(!(flags & IL_FLAG_GRADS_ONLY) && !peer_critic && !peer_actor, "il_sac_update_gather_overlap: whole single-GPU updates only");
    IL_CHECK_ARG(flags & IL_FLAG_SAC_PREPARED, "il_sac_update_gather_overlap: the lane-ordered weight copies must be in step (IL_FLAG_SAC_PREPARED): run one il_sac_update_gather first");
    DwArgs ca = critic_dw_args(d, flags), aa = actor_dw_args(d, rows, flags);
    if (!chain_pair_ok(d, rl.on, G) || !policy_critic_pair_ok(d) || ca.n_big_blocks <= 0 || aa.n_big_blocks <= 0)
      return il_set_error(IL_ERR_UNSUPPORTED, "il_sac_update_gather_overlap: the pair-mode shape (hidden 256, round_up16(S + A) <= 64, batch %% 128 == 0, every launch co-resident) only");
    const int hp = pc_helpers(nt), g_chain = chain_pair_workgroups(nt, rl.on, G), g_dwc = ca.n_dw_blocks, g_pc = (4 + hp) * nt, g_dwa = aa.n_dw_blocks + IL_TAIL_BLOCKS;
    if (g_chain > IL_OV_MAX_GRID || g_dwc > IL_OV_MAX_GRID || g_pc > IL_OV_MAX_GRID || g_dwa > IL_OV_MAX_GRID) return il_set_error(IL_ERR_UNSUPPORTED, "il_sac_update_gather_overlap: a launch of more than %d workgroups", IL_OV_MAX_GRID);
    rl.overlap = g_dwa;
    ca.sync = d->sync; ca.ov_stage = 1 + IL_OV_DWC; ca.ov_grid = g_chain; aa.ov_stage = 1 + IL_OV_DWA; aa.ov_grid = g_pc;
    { IL_TRACE("k_sac_chain", st); k_sac_chain_pair<<<chain_pair_workgroups(nt, rl.on, G), 512, IL_PAIR_LDS_BYTES, st>>>(*d, *ring, eps_next, eps_cur, rewards, const_cast<float*>(rows->states), rl); }
    { IL_TRACE("k_dw_adam_critic", sb); launch_dw_adam(ca, ca.n_dw_blocks, sb); }
    launch_policy_critic(d, rows, out_logp, out_q, lds, st, g_dwc);
    { IL_TRACE("k_dw_adam_actor", sb); launch_dw_adam(aa, aa.n_dw_blocks + IL_TAIL_BLOCKS, sb); }
    IL_CHECK_LAUNCH("il_sac_update_gather_overlap");
    return IL_OK;
  }
  if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
  if (chain_pair_ok(d, rl.on, G)) { IL_TRACE("k_sac_chain", st); k_sac_chain_pair<<<chain_pair_workgroups(nt, rl.on, G), 512, IL_PAIR_LDS_BYTES, st>>>(*d, *ring, eps_next, eps_cur, rewards, const_cast<float*>(rows->states), rl); }
  else {
    if (chain_xcd_nets(nt) && G <= 2 * nt) { rl.xcd_nets = 1; rl.gather_wgs = G; }
    IL_TRACE("k_sac_chain", st); k_sac_chain<<<rl.xcd_nets ? 8 * nt : 6 * nt + G, tile_threads(H), lds, st>>>(*d, *ring, eps_next, eps_cur, rewards, const_cast<float*>(rows->states), rl);
  }
  DwArgs ca = critic_dw_args(d, flags);
  if (peer_critic) { DwPeer cp = {*peer_critic, 0}; IL_TRACE("k_dw_adam_critic", st); k_dw_adam_peer<<<ca.n_dw_blocks, 256, 0, st>>>(ca, cp); }
  else { IL_TRACE("k_dw_adam_critic", st); launch_dw_adam(ca, ca.n_dw_blocks, st); }
  if (flags & IL_FLAG_GRADS_ONLY) {   // data-parallel: stop at the critic gradients (critic_grad); the caller all-reduces them and continues with il_sac_dp_phase(rows, 2) and (rows, 3)
    IL_CHECK_LAUNCH("il_sac_update_gather");
    return IL_OK;
  }
  launch_policy_critic(d, rows, out_logp, out_q, lds, st);
  DwArgs aa = actor_dw_args(d, rows, flags);
  if (peer_actor) {
    DwPeer ap = {*peer_actor, mlp_numel(d->state_dim, d->hidden, 2 * d->action_dim)};
    IL_TRACE("k_dw_adam_actor", st); k_dw_adam_peer<<<aa.n_dw_blocks + IL_TAIL_BLOCKS, 256, 0, st>>>(aa, ap);
  } else { IL_TRACE("k_dw_adam_actor", st); launch_dw_adam(aa, aa.n_dw_blocks + IL_TAIL_BLOCKS, st); }
  IL_CHECK_LAUNCH("il_sac_update_gather");
  return IL_OK;
}
extern "C" int il_sac_update_gather(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const il_disc* relabel, float* rewards_out,
                                    const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream_) {
  return sac_update_gather_impl(d, rows, ring, rewards, relabel, rewards_out, eps_next, eps_cur, out_logp, out_q, flags, nullptr, nullptr, stream_);
}
extern "C" int il_sac_update_gather_overlap(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const il_disc* relabel, float* rewards_out,
                                            const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream_a, il_stream_t stream_b) {
  IL_CHECK_ARG(stream_b, "il_sac_update_gather_overlap: stream_b must be a stream of its own");
  return sac_update_gather_impl(d, rows, ring, rewards, relabel, rewards_out, eps_next, eps_cur, out_logp, out_q, flags, nullptr, nullptr, stream_a, stream_b);
}
__global__ void k_overlap_enter(long long* sy) {   // <<<4, IL_OV_MAX_GRID>>>: block = stage, thread = workgroup flag
  const long long n = sync_read(sy, IL_SYNC_MAIN_EPOCH);
  const int st = blockIdx.x;
  if (threadIdx.x == 0) { sy[IL_SYNC_OV_EPOCH + st * IL_SYNC_STRIDE] = n; sy[IL_SYNC_OV_TICKET + st * IL_SYNC_STRIDE] = 0; }
  sy[IL_SYNC_OV_FLAGS + ((long long)st * IL_OV_MAX_GRID + threadIdx.x) * IL_SYNC_STRIDE] = n;
}
extern "C" int il_sac_overlap_enter(const il_sac* d, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->sync, "il_sac_overlap_enter: needs the il_sync counters");
  k_overlap_enter<<<4, IL_OV_MAX_GRID, 0, (hipStream_t)stream_>>>(reinterpret_cast<long long*>(d->sync));
  IL_CHECK_LAUNCH("il_sac_overlap_enter");
  return IL_OK;
}
extern "C" int il_sac_update_gather_peer(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const il_disc* relabel, float* rewards_out,
                                         const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, const il_peer_bucket* peer_critic,
                                         const il_peer_bucket* peer_actor, il_stream_t stream_) {
  IL_CHECK_ARG(peer_critic && peer_actor, "il_sac_update_gather_peer: null peer descriptor");
  return sac_update_gather_impl(d, rows, ring, rewards, relabel, rewards_out, eps_next, eps_cur, out_logp, out_q, flags, peer_critic, peer_actor, stream_);
}

// ---------------------------------------------------------------------------------------------
// Population axis: N independent learners with identical shapes advanced by the SAME launches. descs_dev / batches_dev
// are device arrays of n_learners descriptors (each learner has its own arenas, optimiser state, workspace, noise counter and batch);
// gridDim.y (z for k_repack) selects the learner. Kernels are latency-bound at B = 256 (16-64 workgroups): a population fills the chip.
// ---------------------------------------------------------------------------------------------
#ifndef IL_POP_SMALL_BLOCKS
#define IL_POP_SMALL_BLOCKS 1
#endif
__host__ __device__ static inline bool pop_small_blocks(int H, int B) { return IL_POP_SMALL_BLOCKS && H % DWS == 0 && B % DWS_ROWS == 0; }
// workgroups per learner of a population dW launch behind its nb64 dw_block64 workgroups (without the tail)
static inline int pop_dw_small_grid(int IN, int H, int OUT, int nets, int B, bool b64) {
  if (b64 && pop_small_blocks(H, B)) return (dw_block_jobs(IN, H, OUT) - (H / DWS) * (H / DWS)) * nets + (H / 16 * nets + 3) / 4;
  return dw_blocks(IN, H, OUT, nets, b64);
}
// (round 5, experiment) The dW launches' own decode of the linear workgroup id. pop_ids hands out, per group of 8 learners, the group's 64 x 64 blocks (16 us each), then its 32 x 32
// block jobs (7 - 10 us), bias jobs and tail, and only then the next group's blocks: the launch is slot-bound (3 - 4 workgroups per CU, profiles/r05_pop_dw_timeline.txt),
// so the LAST group's block workgroups start at 37 of a 52 us launch and its end is 15 us of a chip that drains. Here: the blocks of ALL full groups first, then
// everything else - the launch ends on jobs half as long. Learner l stays on XCD l % 8 (both phases start at a multiple of 8); learners behind the last full group of 8
// keep the natural decode. No job of these launches waits for another one: pure re-labelling, same bits. MEASURED NEUTRAL (three interleaved same-box A/Bs,
// profiles/r05_pop_dw_ab.txt: +0.8 %, +0.1 %; with 4 waves per SIMD -1 %): with every slot holding a block workgroup the MFMA pipes saturate (products 12 -> 15 - 20 us)
// and the small jobs' memory latency no longer hides under them. Kept as an A/B switch, OFF: IL_POP_DW_BIG_FIRST=1 selects it.
#ifndef IL_POP_DW_BIG_FIRST
#define IL_POP_DW_BIG_FIRST 0
#endif
__device__ __forceinline__ void pop_dw_ids(int& bx, int& by, int nb64) {
#if IL_POP_DW_BIG_FIRST && IL_POP_XCD
  const int nx = gridDim.x, lf = ((int)gridDim.y >> 3) * 8, g = by * nx + bx;
  if (nb64 > 0 && nb64 < nx && g < lf * nx) {
    const int nbig = lf * nb64;
    if (g < nbig) { const int q = g >> 3; by = (q / nb64) * 8 + (g & 7); bx = q % nb64; }
    else { const int r = g - nbig, q = r >> 3, nr = nx - nb64; by = (q / nr) * 8 + (r & 7); bx = nb64 + q % nr; }
    return;
  }
  if (g >= lf * nx) return;
#endif
  pop_ids(bx, by);
}
// (round 5) Register budget of 4 waves per SIMD: the default allocation (121 VGPRs + 32 AGPRs for the accumulators) leaves 3 workgroups per CU; capped at 128 the kernel
// keeps everything in VGPRs without a spill and a CU holds 4 (LDS: 4 x 34 KB). The launch is slot-bound outside its 64 x 64 blocks' products
// (profiles/r05_pop_dw_timeline.txt): +1.0 - 1.4 % on the population line in two interleaved same-box A/Bs (profiles/r05_pop_dw_ab.txt). IL_POP_DW_WAVES=0: no cap.
#ifndef IL_POP_DW_WAVES
#define IL_POP_DW_WAVES 4
#endif
#if IL_POP_DW_WAVES > 0
#define IL_POP_DW_ATTR __attribute__((amdgpu_waves_per_eu(IL_POP_DW_WAVES, IL_POP_DW_WAVES)))
#else
#define IL_POP_DW_ATTR
#endif
__global__ __launch_bounds__(256) IL_POP_DW_ATTR void k_dw_adam_pop(const il_sac* __restrict__ dL, const il_batch* __restrict__ bL, int kind, uint32_t flags, int nb64) {
  __shared__ __attribute__((aligned(16))) float smem[2 * DWB * DWB_LD];
  int bx = blockIdx.x, by = blockIdx.y;
#if IL_POP_XCD_DW
  pop_dw_ids(bx, by, nb64);   // a learner's blocks on one XCD: the four blocks that share a [64 features][B] operand panel find it in that L2
#endif
  il_sac d = dL[by]; il_batch b = bL[by];
  globalize(d); globalize(b);
  DwArgs a = kind ? actor_dw_args(&d, &b, flags) : critic_dw_args(&d, flags);
  IL_TL(kind ? 11 : 10, 0);
  IL_TLV(kind ? 11 : 10, 2, (unsigned long long)(unsigned)bx | ((unsigned long long)(unsigned)by << 32));   // which job of which learner (profiles/tools/pop_dw_timeline.py)
  IL_TLV(kind ? 11 : 10, 3, ((unsigned long long)il_st_xcc() << 16) | ((il_st_hwid() >> 8) & 0xffu));   // and where it runs
  IL_TLC(kind ? 11 : 10, 4);   // shader-clock counter at the start (slot 5: at the end) - the clock this workgroup ran at = delta / the 100 MHz counter's delta
  // Measured at 32 learners (round 2, one box; critic launch with dw_tile for every layer: 103 us = 23 TFLOP/s, MfmaUtil 12.7 %, 1.8 waves per SIMD resident on average):
  // the products alone took 79 us, the Adam epilogue alone 46 us. What did NOT move it: 32 x 32 blocks of dW per wave straight from global memory (half the operand
  // bytes per MFMA): 100 us; every learner confined to XCD l % 8 (`s_getreg XCC_ID` confirms workgroup g runs on XCD g % 8): 103 us; 16 instead of 8 operand loads in
  // flight per lane: 113 us. Neither L2 -> CU nor fabric bandwidth: a serial (8 loads -> wait -> 16 MFMAs) chain per wave. Hence dw_block64 for the H x H layers.
  if (nb64 > 0) {
    const int H = a.hidden, nbh = H / DWB, per_net = nbh * nbh;
    if (bx < nb64) {
      const int net = bx / per_net, blk = bx - net * per_net;
      adam_consts ac = {};
      if (!a.grads_only) ac = load_adam_consts(a.opt);
      const int64_t oW2 = (int64_t)net * a.net_stride + (int64_t)H * a.in_dim + H;
      dw_block64(a, ac, a.dz2 + net * a.h_net_stride, a.h1 + net * a.h_net_stride, H, (blk / nbh) * DWB, (blk % nbh) * DWB, oW2,
                 a.pk_f ? a.pk_f + (size_t)net * H * H : nullptr, a.pk_b ? a.pk_b + (size_t)net * H * H : nullptr, smem, dL + by);
      IL_TLC(kind ? 11 : 10, 5); IL_TL_END(kind ? 11 : 10);
      return;
    }
    if (pop_small_blocks(a.hidden, a.batch)) {
      // (round 3) layers 1 and 3 with their biases as 32 x 32 LDS block jobs (dw_block32: whole-line staging instead of the wave-per-tile jobs' half-line gathers, which were
      // 72 % of this launch's line requests), bias 2 as wave jobs (dw_block64 does not fold it), then the tail. Grid: pop_dw_grid().
      const int nb32 = H / DWS, small = dw_block_jobs(a.in_dim, H, a.out_dim) - nb32 * nb32, sb = bx - nb64;
      if (sb < small * a.n_nets) { dw_block_job(a, sb / small, nb32 * nb32 + sb % small, smem); IL_TLC(kind ? 11 : 10, 5); IL_TL_END(kind ? 11 : 10); return; }
      const int bias_blocks = (H / 16 * a.n_nets + 3) / 4, bb = sb - small * a.n_nets;
      if (bb < bias_blocks) {
        const int job = bb * 4 + (int)(threadIdx.x >> 6);
        if (job < H / 16 * a.n_nets) {
          const int net = job / (H / 16), jn = job % (H / 16);
          adam_consts ac = {};
          if (!a.grads_only) ac = load_adam_consts(a.opt);
          dw_bias(a, ac, a.dz2 + net * a.h_net_stride, H, jn * 16, (int64_t)net * a.net_stride + (int64_t)H * a.in_dim + H + (int64_t)H * H);
        }
        IL_TLC(kind ? 11 : 10, 5); IL_TL_END(kind ? 11 : 10);
        return;
      }
      a.n_dw_blocks = 0; a.n_big_blocks = 0; a.jobs_per_block = 4;   // the tail blocks (actor launch)
      dw_adam_body<4, true>(a, bb - bias_blocks, (int)gridDim.x - nb64 - small * a.n_nets - bias_blocks);
      IL_TLC(kind ? 11 : 10, 5); IL_TL_END(kind ? 11 : 10);
      return;
    }
    a.n_dw_blocks = dw_blocks(a.in_dim, a.hidden, a.out_dim, a.n_nets, 1); a.n_big_blocks = 0; a.jobs_per_block = 4;
    dw_adam_body<4, true>(a, bx - nb64, (int)gridDim.x - nb64);
    IL_TLC(kind ? 11 : 10, 5); IL_TL_END(kind ? 11 : 10);
    return;
  }
  a.n_dw_blocks = dw_blocks(a.in_dim, a.hidden, a.out_dim, a.n_nets, 0); a.n_big_blocks = 0; a.jobs_per_block = 4;
  dw_adam_body<4>(a, bx, (int)gridDim.x);
}

extern "C" int il_sac_update_population(const il_sac* descs_dev, const il_batch* batches_dev, int32_t n_learners, const il_sac* shape_host, uint32_t flags, il_stream_t stream_) {
  IL_CHECK_ARG(descs_dev && batches_dev && shape_host && n_learners >= 1 && n_learners <= 65535, "il_sac_update_population: bad arguments");
  IL_CHECK_ARG(!(flags & IL_FLAG_GRADS_ONLY), "il_sac_update_population: IL_FLAG_GRADS_ONLY is not supported on the population path");
  const il_sac* d = shape_host;
  IL_CHECK_ARG(d->hidden % 64 == 0 && d->hidden >= 64 && d->hidden <= 256 && d->batch % IL_TILE_R == 0 && d->batch > 0 && 2 * d->action_dim <= 16, "il_sac_update_population: unsupported shape");
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, nt = B / IL_TILE_R, L = n_learners;
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  il_sac z = *d; il_batch zb = {};
  // Tile kernels of the population launch run with HALF the waves of the single-learner launch (each wave then owns two 16-column tiles): two workgroups share a CU
  // and their load / barrier / epilogue phases overlap each other's MFMA phases (measured at 32 learners: 60.2k -> 64.5k aggregate updates/s; a quarter: 60.0k).
  // The results are bit-identical to the full-width launch (every cross-wave reduction is ordered by block index, not by wave). IL_POP_TILE_THREADS overrides.
  static const int pop_threads_env = [] { const char* e = getenv("IL_POP_TILE_THREADS"); return e ? atoi(e) : 0; }();
  const int tt_default = tile_threads(H) >= 512 ? tile_threads(H) / 2 : tile_threads(H);
  const int tt = (pop_threads_env >= 256 && pop_threads_env <= tile_threads(H) && pop_threads_env % 64 == 0) ? pop_threads_env : tt_default;
  // Measured at 32 learners (round 2): chained 61.6k aggregate updates/s vs 70.7k with the three separate launches (k_sac_chain_pop 215 us against 46 + 71 + 35 us): the
  // workgroups that wait for their tile's producers hold CU slots the oversubscribed launch needs. Hence OFF by default; IL_POP_CHAIN=1 switches it on.
  // (round 3) the `_pop` builds of the tile kernels (half weight panels, <= 80 VGPRs: three workgroups per CU instead of two) whenever the launch is at most 512 threads wide;
  // IL_POP_THREE=0 keeps the 128-VGPR builds (bit-identical either way)
  static const int pop_three = [] { const char* e = getenv("IL_POP_THREE"); return e && e[0] == '0' ? 0 : 1; }();
  const bool pop3 = pop_three && tt <= 512;
  static const int pop_chain = [] { const char* e = getenv("IL_POP_CHAIN"); return e && e[0] == '1' ? 1 : 0; }();
  const bool whole = !(flags & (IL_FLAG_SAC_SKIP_FORWARD | IL_FLAG_SAC_FORWARD_ONLY));
  if (whole && pop_chain) {   // forward + critic loss chained per tile inside one launch (k_sac_chain_pop); IL_POP_CHAIN=0: the three separate launches
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5, L), 256, 0, st>>>(z, 0x1Fu, descs_dev); }
    { IL_TRACE("k_sac_chain", st); k_sac_chain_pop<<<dim3(6 * nt, L), tt, lds, st>>>(descs_dev, batches_dev); }
    flags |= IL_FLAG_SAC_SKIP_FORWARD | 0x80000000u;
  }
  if (!(flags & IL_FLAG_SAC_SKIP_FORWARD)) {
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5, L), 256, 0, st>>>(z, 0x1Fu, descs_dev); }
    if (pop3) {
      { IL_TRACE("k_actor_fwd", st); k_actor_fwd_pop<<<dim3(2 * nt, L), tt, lds, st>>>(z, zb, nullptr, nullptr, 0, descs_dev, batches_dev); }
      { IL_TRACE("k_critic_fwd", st); k_critic_fwd_pop<<<dim3(4 * nt, L), tt, lds, st>>>(z, zb, descs_dev, batches_dev); }
    } else {
      { IL_TRACE("k_actor_fwd", st); k_actor_fwd<<<dim3(2 * nt, L), tt, lds, st>>>(z, zb, nullptr, nullptr, 0, descs_dev, batches_dev); }
      { IL_TRACE("k_critic_fwd", st); k_critic_fwd<<<dim3(4 * nt, L), tt, lds, st>>>(z, zb, descs_dev, batches_dev); }
    }
  }
  if (!(flags & IL_FLAG_SAC_FORWARD_ONLY)) {
    if (!(flags & 0x80000000u)) {
      IL_TRACE("k_critic_bwd", st);
      if (pop3) k_critic_bwd_pop<<<dim3(2 * nt, L), tt, lds, st>>>(z, zb, descs_dev, batches_dev);
      else k_critic_bwd<<<dim3(2 * nt, L), tt, lds, st>>>(z, zb, descs_dev, batches_dev);
    }
    static const int lds_dw = [] { const char* e = getenv("IL_POP_DW_LDS"); return e && e[0] == '0' ? 0 : 1; }();
    const bool b64 = lds_dw && H % DWB == 0 && B % DWB == 0;   // H x H layers as 64 x 64 blocks staged through LDS (dw_block64); IL_POP_DW_LDS=0: dw_tile for every layer
    const int nbc = b64 ? dw_block64_count(H, 2) : 0, nba = b64 ? dw_block64_count(H, 1) : 0;
    // (round 5) the target step of the critics' H x H layers inside their 64 x 64 optimiser blocks (DwArgs.fuse_polyak) - only where those blocks exist; both dW launches
    // must see the same setting (the actor launch's tail skips exactly what the critic launch's blocks stepped). IL_POP_FUSE_POLYAK=0: the whole step in the tail - same bits.
    static const uint32_t fuse = [] { const char* e = getenv("IL_POP_FUSE_POLYAK"); return e && e[0] == '0' ? 0u : IL_FLAG_POP_FUSE_POLYAK; }();
    if (b64) flags |= fuse;
    { IL_TRACE("k_dw_adam_critic", st); k_dw_adam_pop<<<dim3(nbc + pop_dw_small_grid(S + A, H, 1, 2, B, b64), L), 256, 0, st>>>(descs_dev, batches_dev, 0, flags, nbc); }
    {
      IL_TRACE("k_policy_critic", st);
      // Round 4: the policy backward is a launch of its own behind the critics (uniform workgroups in both) instead of the tail of each tile's second arriver, which made
      // half the workgroups of the launch 1.7 x as long as the rest: +11 % aggregate on one box (profiles/r04_pop_split_ab.txt). IL_POP_SPLIT_TAIL=0: the one-launch form.
      static const int split_tail = [] { const char* e = getenv("IL_POP_SPLIT_TAIL"); return e && e[0] == '0' ? 0 : 1; }();
      if (pop3) k_policy_critic_pop<<<dim3(2 * nt, L), tt, lds, st>>>(z, zb, nullptr, nullptr, descs_dev, batches_dev, split_tail ? IL_PC_NO_TAIL : 0);
      else k_policy_critic<<<dim3(2 * nt, L), tt, lds, st>>>(z, zb, nullptr, nullptr, descs_dev, batches_dev, split_tail ? IL_PC_NO_TAIL : 0);
      if (split_tail) {
        if (pop3) k_actor_bwd_pop<<<dim3(nt, L), tt, lds, st>>>(z, zb, descs_dev, batches_dev);
        else k_actor_bwd<<<dim3(nt, L), tt, lds, st>>>(z, zb, descs_dev, batches_dev);
      }
    }
    { IL_TRACE("k_dw_adam_actor", st); k_dw_adam_pop<<<dim3(nba + pop_dw_small_grid(S, H, 2 * A, 1, B, b64) + 33, L), 256, 0, st>>>(descs_dev, batches_dev, 1, flags, nba); }
  }
  IL_CHECK_LAUNCH("il_sac_update_population");
  return IL_OK;
}

// The lane-ordered weight copies only depend on the parameters: a caller may build them early (e.g. next to the replay sampling on
// another stream) and pass IL_FLAG_SAC_PREPARED to il_sac_update.
extern "C" int il_sac_prepare(const il_sac* d, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->workspace && d->hidden % 64 == 0 && d->hidden >= 64 && d->hidden <= 256, "il_sac_prepare: bad descriptor");
  { IL_TRACE("k_repack", stream_); k_repack<<<dim3(repack_blocks(d->hidden), 5), 256, 0, (hipStream_t)stream_>>>(*d, 0x1Fu, nullptr); }
  IL_CHECK_LAUNCH("il_sac_prepare");
  return IL_OK;
}

extern "C" int il_adam_step(float* p, const float* g, const il_adam* opt, int64_t n, uint32_t flags, il_stream_t stream_) {
  IL_CHECK_ARG(p && g && opt && opt->m && opt->v && opt->step && n > 0, "il_adam_step: bad arguments");
  hipStream_t st = (hipStream_t)stream_;
  if (flags & IL_FLAG_TICK) k_tick<<<1, 64, 0, st>>>(*opt);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  { IL_TRACE("k_adam_flat", st); k_adam_flat<<<blocks, 256, 0, st>>>(p, g, *opt, n); }
  IL_CHECK_LAUNCH("il_adam_step");
  return IL_OK;
}

extern "C" int il_polyak(float* target, const float* param, int64_t n, double tau, il_stream_t stream_) {
  IL_CHECK_ARG(target && param && n > 0, "il_polyak: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  { IL_TRACE("k_polyak", (hipStream_t)stream_); k_polyak<<<blocks, 256, 0, (hipStream_t)stream_>>>(target, param, n, tau); }
  IL_CHECK_LAUNCH("il_polyak");
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// Data-parallel schedule of one update: the fused path's kernels with the AdamW steps split off so that the caller can all-reduce the
// gradient arenas in between.  phase 0: re-order weights, actor forward (s' and s), critic/target forward (reward-independent)
//                              phase 1: critic loss + backward, critic gradients -> critic_grad          [all-reduce critic_grad]
//                              phase 2: AdamW(critic) (+ lane-ordered copies), policy loss through the updated critic, actor/alpha
//                                       gradients -> actor_grad / alpha_grad                              [all-reduce actor_grad|alpha_grad]
//                              phase 3: AdamW(actor), Adam(log_alpha), polyak -- one launch
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply_critic(il_sac d) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  const SacWs ws = sac_ws(S, A, H, d.batch);
  const int64_t ns = net_stride(IN, H, 1), HH = (int64_t)H * H, oW2 = (int64_t)H * IN + H;
  const adam_consts ac = load_adam_consts(d.critic_opt);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < 2 * ns; e += (int64_t)gridDim.x * blockDim.x) {
    float pp = d.critic[e], mm = d.critic_opt.m[e], vv = d.critic_opt.v[e];
    adam_update(pp, d.critic_grad[e], mm, vv, ac);
    d.critic[e] = pp; d.critic_opt.m[e] = mm; d.critic_opt.v[e] = vv;
    const int k = e >= ns; const int64_t o = e - k * ns - oW2;
    if (o >= 0 && o < HH) {  // an element of a hidden-layer matrix: keep its two lane-ordered copies in step
      const int n = (int)(o / H), kk = (int)(o - (int64_t)n * H);
      d.workspace[ws.pk_cf + k * HH + packed_fwd_index(n, kk, H)] = pp;
      d.workspace[ws.pk_cb + k * HH + packed_bwd_index(n, kk, H)] = pp;
    }
  }
}

__global__ __launch_bounds__(256) void k_apply_actor_tail(il_sac d, int n_actor_blocks) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  if ((int)blockIdx.x < n_actor_blocks) {
    const int64_t Pa = mlp_numel(S, H, 2 * A);
    const adam_consts ac = load_adam_consts(d.actor_opt);
    const SacWs wsa = sac_ws(S, A, H, d.batch);
    const int64_t oW2 = (int64_t)H * S + H, HH = (int64_t)H * H;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < Pa; e += (int64_t)n_actor_blocks * blockDim.x) {
      float pp = d.actor[e], mm = d.actor_opt.m[e], vv = d.actor_opt.v[e];
      adam_update(pp, d.actor_grad[e], mm, vv, ac);
      d.actor[e] = pp; d.actor_opt.m[e] = mm; d.actor_opt.v[e] = vv;
      const int64_t o = e - oW2;
      if (o >= 0 && o < HH) {
        const int n = (int)(o / H), kk = (int)(o - (int64_t)n * H);
        d.workspace[wsa.pk_af + packed_fwd_index(n, kk, H)] = pp;
        d.workspace[wsa.pk_ab + packed_bwd_index(n, kk, H)] = pp;
      }
    }
    return;
  }
  const int tb = (int)blockIdx.x - n_actor_blocks, ntb = (int)gridDim.x - n_actor_blocks;
  if (tb == 0 && threadIdx.x == 0) {
    const adam_consts ac = load_adam_consts(d.alpha_opt);
    float pp = d.log_alpha[0], mm = d.alpha_opt.m[0], vv = d.alpha_opt.v[0];
    adam_update(pp, d.alpha_grad[0], mm, vv, ac);
    d.log_alpha[0] = pp; d.alpha_opt.m[0] = mm; d.alpha_opt.v[0] = vv;
  }
  const int64_t n = 2 * net_stride(IN, H, 1);
  const float omt = (float)(1.0 - d.polyak), tau = (float)d.polyak;
  for (int64_t i = (int64_t)tb * blockDim.x + threadIdx.x; i < n; i += (int64_t)ntb * blockDim.x) d.target[i] = __fadd_rn(__fmul_rn(d.target[i], tau), __fmul_rn(omt, d.critic[i]));
  const SacWs ws = sac_ws(S, A, H, d.batch);
  float* pt = d.workspace + ws.pk_tf; const float* pc = d.workspace + ws.pk_cf;
  for (int64_t i = (int64_t)tb * blockDim.x + threadIdx.x; i < 2 * (int64_t)H * H; i += (int64_t)ntb * blockDim.x) pt[i] = __fadd_rn(__fmul_rn(pt[i], tau), __fmul_rn(omt, pc[i]));   // PF copies only (see actor_dw_args)
}

// ---------------------------------------------------------------------------------------------
// The apply launches of phases 2 and 3 with the gradient exchange built in (il_sac_dp_phase_peer): workgroup c exchanges chunk c of the bucket with the other ranks
// (peer_chunk_allreduce: the body of k_peer_allreduce) and steps the parameters of that chunk with the means it holds in registers - one launch and one pass over the
// gradient arena less per sync point than il_peer_allreduce_mean + il_sac_dp_phase. Same means (rank-ordered sum / W), same AdamW operations: bit-identical to that sequence.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_peer_apply_critic(il_sac d, il_peer_bucket x) {
  f32x4 mean[IL_PEER_Q];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int cnt = peer_chunk_allreduce(x, d.critic_grad, c, mean);
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  const SacWs ws = sac_ws(S, A, H, d.batch);
  const int64_t ns = net_stride(IN, H, 1), HH = (int64_t)H * H, oW2 = (int64_t)H * IN + H;
  const adam_consts ac = load_adam_consts(d.critic_opt);
#pragma unroll
  for (int j = 0; j < IL_PEER_Q; ++j) {
    const int b0 = 4 * (tid + 256 * j);
    if (b0 >= cnt) continue;
    const int64_t e0 = (int64_t)c * IL_PEER_CHUNK_FLOATS + b0;
    const bool whole = b0 + 3 < cnt;   // the arenas are 16-byte aligned and a chunk starts at a multiple of 2,048 floats: whole 16-byte lanes except at the very end
    f32x4 p4 = zero4(), m4 = zero4(), v4 = zero4();
    if (whole) { p4 = gload4(d.critic + e0); m4 = gload4(d.critic_opt.m + e0); v4 = gload4(d.critic_opt.v + e0); }
    else for (int q = 0; q < 4; ++q) if (b0 + q < cnt) { p4[q] = d.critic[e0 + q]; m4[q] = d.critic_opt.m[e0 + q]; v4[q] = d.critic_opt.v[e0 + q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { float pp = p4[q], mm = m4[q], vv = v4[q]; adam_update(pp, mean[j][q], mm, vv, ac); p4[q] = pp; m4[q] = mm; v4[q] = vv; }
    if (whole) { *reinterpret_cast<f32x4*>(d.critic + e0) = p4; *reinterpret_cast<f32x4*>(d.critic_opt.m + e0) = m4; *reinterpret_cast<f32x4*>(d.critic_opt.v + e0) = v4; }
    else for (int q = 0; q < 4; ++q) if (b0 + q < cnt) { d.critic[e0 + q] = p4[q]; d.critic_opt.m[e0 + q] = m4[q]; d.critic_opt.v[e0 + q] = v4[q]; }
    for (int q = 0; q < 4; ++q) {
      if (b0 + q >= cnt) break;
      const int64_t e = e0 + q;
      const int k = e >= ns; const int64_t o = e - k * ns - oW2;
      if (o >= 0 && o < HH) {  // an element of a hidden-layer matrix: keep its two lane-ordered copies in step
        const int n = (int)(o / H), kk = (int)(o - (int64_t)n * H);
        d.workspace[ws.pk_cf + k * HH + packed_fwd_index(n, kk, H)] = p4[q];
        d.workspace[ws.pk_cb + k * HH + packed_bwd_index(n, kk, H)] = p4[q];
      }
    }
  }
}

// blocks [0, n_chunks): the chunks of the actor bucket (actor gradient | log-alpha gradient | padding); the blocks behind them: polyak, as in k_apply_actor_tail
__global__ __launch_bounds__(256) void k_peer_apply_actor_tail(il_sac d, int n_chunks, il_peer_bucket x) {
  const int S = d.state_dim, A = d.action_dim, H = d.hidden, IN = S + A;
  if ((int)blockIdx.x < n_chunks) {
    f32x4 mean[IL_PEER_Q];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int cnt = peer_chunk_allreduce(x, d.actor_grad, c, mean);   // the bucket starts at the actor gradient; the log-alpha gradient sits at offset alpha_grad - actor_grad
    const int64_t Pa = mlp_numel(S, H, 2 * A), alpha_at = d.alpha_grad - d.actor_grad;
    const adam_consts ac = load_adam_consts(d.actor_opt);
    const SacWs wsa = sac_ws(S, A, H, d.batch);
    const int64_t oW2 = (int64_t)H * S + H, HH = (int64_t)H * H;
#pragma unroll
    for (int j = 0; j < IL_PEER_Q; ++j) {
      const int b0 = 4 * (tid + 256 * j);
      for (int q = 0; q < 4; ++q) {
        if (b0 + q >= cnt) break;
        const int64_t e = (int64_t)c * IL_PEER_CHUNK_FLOATS + b0 + q;
        if (e < Pa) {
          float pp = d.actor[e], mm = d.actor_opt.m[e], vv = d.actor_opt.v[e];
          adam_update(pp, mean[j][q], mm, vv, ac);
          d.actor[e] = pp; d.actor_opt.m[e] = mm; d.actor_opt.v[e] = vv;
          const int64_t o = e - oW2;
          if (o >= 0 && o < HH) {
            const int n = (int)(o / H), kk = (int)(o - (int64_t)n * H);
            d.workspace[wsa.pk_af + packed_fwd_index(n, kk, H)] = pp;
            d.workspace[wsa.pk_ab + packed_bwd_index(n, kk, H)] = pp;
          }
        } else if (e == alpha_at) {
          const adam_consts aa = load_adam_consts(d.alpha_opt);
          float pp = d.log_alpha[0], mm = d.alpha_opt.m[0], vv = d.alpha_opt.v[0];
          adam_update(pp, mean[j][q], mm, vv, aa);
          d.log_alpha[0] = pp; d.alpha_opt.m[0] = mm; d.alpha_opt.v[0] = vv;
        }
      }
    }
    return;
  }
  const int tb = (int)blockIdx.x - n_chunks, ntb = (int)gridDim.x - n_chunks;
  const int64_t n = 2 * net_stride(IN, H, 1);
  const float omt = (float)(1.0 - d.polyak), tau = (float)d.polyak;
  for (int64_t i = (int64_t)tb * blockDim.x + threadIdx.x; i < n; i += (int64_t)ntb * blockDim.x) d.target[i] = __fadd_rn(__fmul_rn(d.target[i], tau), __fmul_rn(omt, d.critic[i]));
  const SacWs ws = sac_ws(S, A, H, d.batch);
  float* pt = d.workspace + ws.pk_tf; const float* pc = d.workspace + ws.pk_cf;
  for (int64_t i = (int64_t)tb * blockDim.x + threadIdx.x; i < 2 * (int64_t)H * H; i += (int64_t)ntb * blockDim.x) pt[i] = __fadd_rn(__fmul_rn(pt[i], tau), __fmul_rn(omt, pc[i]));   // PF copies only (see actor_dw_args)
}

extern "C" int il_sac_dp_phase(const il_sac* d, const il_batch* b, int32_t phase, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream_) {
  if (int rc = check_sac(d, b)) return rc;
  IL_CHECK_ARG(phase >= 0 && phase <= 3, "il_sac_dp_phase: phase must be 0..3");
  IL_CHECK_ARG(d->actor_grad && d->critic_grad && d->alpha_grad, "il_sac_dp_phase: gradient arenas missing");
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden, B = d->batch, nt = B / IL_TILE_R;
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  if (phase == 0 && chain_enabled() && 6 * nt <= device_cu_count()) {
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
    ChainRelabel fo = {}; fo.fwd_only = 1;
    fo.xcd_nets = chain_xcd_nets(nt) ? 1 : 0;
    { IL_TRACE("k_sac_chain", st); k_sac_chain<<<(fo.xcd_nets ? 8 : 6) * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr, nullptr, nullptr, fo); }
  } else if (phase == 0) {
    if (!(flags & IL_FLAG_SAC_PREPARED)) { IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 5), 256, 0, st>>>(*d, 0x1Fu, nullptr); }
    { IL_TRACE("k_actor_fwd", st); k_actor_fwd<<<2 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr, 0, nullptr, nullptr); }
    { IL_TRACE("k_critic_fwd", st); k_critic_fwd<<<4 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr); }
  } else if (phase == 1) {
    { IL_TRACE("k_critic_bwd", st); k_critic_bwd<<<2 * nt, tile_threads(H), lds, st>>>(*d, *b, nullptr, nullptr); }
    DwArgs ca = critic_dw_args(d, IL_FLAG_GRADS_ONLY);
    { IL_TRACE("k_dw_adam_critic", st); launch_dw_adam(ca, ca.n_dw_blocks, st); }
  } else if (phase == 2) {
    const int64_t n = 2 * net_stride(S + A, H, 1);
    { IL_TRACE("k_apply_critic", st); k_apply_critic<<<(int)((n + 255) / 256), 256, 0, st>>>(*d); }
    launch_policy_critic(d, b, out_logp, out_q, lds, st);
    DwArgs aa = actor_dw_args(d, b, IL_FLAG_GRADS_ONLY);
    { IL_TRACE("k_dw_adam_actor", st); launch_dw_adam(aa, aa.n_dw_blocks + 1, st); }
  } else {
    const int na = (int)((mlp_numel(S, H, 2 * A) + 255) / 256);
    { IL_TRACE("k_apply_actor_tail", st); k_apply_actor_tail<<<na + 64, 256, 0, st>>>(*d, na); }
  }
  IL_CHECK_LAUNCH("il_sac_dp_phase");
  return IL_OK;
}

extern "C" int il_sac_dp_phase_peer(const il_sac* d, const il_batch* b, int32_t phase, float* out_logp, float* out_q, uint32_t flags, const il_peer_bucket* x, il_stream_t stream_) {
  if (int rc = check_sac(d, b)) return rc;
  IL_CHECK_ARG(phase == 2 || phase == 3, "il_sac_dp_phase_peer: phases 2 and 3 begin with an apply step (got %d)", phase);
  IL_CHECK_ARG(d->actor_grad && d->critic_grad && d->alpha_grad, "il_sac_dp_phase_peer: gradient arenas missing");
  IL_CHECK_ARG(x && x->world >= 1 && x->world <= IL_PEER_MAX_RANKS && x->rank >= 0 && x->rank < x->world && x->epoch && x->status, "il_sac_dp_phase_peer: bad peer descriptor");
  IL_CHECK_ARG(x->n_jobs == 0, "il_sac_dp_phase_peer: the bucket is laid out for the exchange INSIDE the optimiser launches (%d arrival lines per job): its apply kernels index epochs and arrival lines by chunk", x->n_jobs);
  for (int r = 0; r < x->world; ++r) IL_CHECK_ARG(x->windows[r], "il_sac_dp_phase_peer: window of rank %d is not mapped", r);
  hipStream_t st = (hipStream_t)stream_;
  const int S = d->state_dim, A = d->action_dim, H = d->hidden;
  const size_t lds = tile_lds_bytes(round_up16(S + A), H);
  if (phase == 2) {
    IL_CHECK_ARG(x->n == 2 * net_stride(S + A, H, 1), "il_sac_dp_phase_peer: phase 2 takes the critic bucket (%lld floats, got %lld)", (long long)(2 * net_stride(S + A, H, 1)), (long long)x->n);
    { IL_TRACE("k_peer_apply_critic", st); k_peer_apply_critic<<<(unsigned)peer_chunks(x->n), 256, 0, st>>>(*d, *x); }
    launch_policy_critic(d, b, out_logp, out_q, lds, st);
    DwArgs aa = actor_dw_args(d, b, IL_FLAG_GRADS_ONLY);
    { IL_TRACE("k_dw_adam_actor", st); launch_dw_adam(aa, aa.n_dw_blocks + 1, st); }
  } else {
    const int64_t Pa = mlp_numel(S, H, 2 * A), alpha_at = d->alpha_grad - d->actor_grad;
    IL_CHECK_ARG(alpha_at >= Pa && alpha_at < x->n, "il_sac_dp_phase_peer: phase 3 takes the actor bucket: actor_grad | alpha_grad must be one allocation of %lld floats (GradBuckets)", (long long)x->n);
    const int nc = (int)peer_chunks(x->n);
    { IL_TRACE("k_peer_apply_actor_tail", st); k_peer_apply_actor_tail<<<nc + 64, 256, 0, st>>>(*d, nc, *x); }
  }
  (void)flags;
  IL_CHECK_LAUNCH("il_sac_dp_phase_peer");
  return IL_OK;
}

extern "C" int il_sac_apply_critic_grads(const il_sac* d, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->critic_grad, "il_sac_apply_critic_grads: critic_grad arena missing");
  return il_adam_step(d->critic, d->critic_grad, &d->critic_opt, 2 * net_stride(d->state_dim + d->action_dim, d->hidden, 1), 0, stream_);
}

__global__ void k_alpha_adam(float* log_alpha, const float* g, il_adam opt) {
  if (threadIdx.x || blockIdx.x) return;
  const adam_consts ac = load_adam_consts(opt);
  float pp = log_alpha[0], mm = opt.m[0], vv = opt.v[0];
  adam_update(pp, g[0], mm, vv, ac);
  log_alpha[0] = pp; opt.m[0] = mm; opt.v[0] = vv;
}

extern "C" int il_sac_apply_actor_grads(const il_sac* d, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->actor_grad && d->alpha_grad, "il_sac_apply_actor_grads: grad arenas missing");
  if (int rc = il_adam_step(d->actor, d->actor_grad, &d->actor_opt, mlp_numel(d->state_dim, d->hidden, 2 * d->action_dim), 0, stream_)) return rc;
  { IL_TRACE("k_alpha_adam", (hipStream_t)stream_); k_alpha_adam<<<1, 64, 0, (hipStream_t)stream_>>>(d->log_alpha, d->alpha_grad, d->alpha_opt); }
  return il_polyak(d->target, d->critic, 2 * net_stride(d->state_dim + d->action_dim, d->hidden, 1), d->polyak, stream_);
}

// ---------------------------------------------------------------------------------------------
// Behavioural cloning (training.py:57-64, models.py:97-99): forward, log-prob of the clamped expert action through atanh,
// backward to the pre-activations -- one tile kernel (activations stay in LDS between forward and backward) -- then the
// shared k_dw_adam.  grid = nt
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_bc_tile(const float* __restrict__ actor, il_adam opt, int S, int A, int H, il_batch b, float* __restrict__ W,
                                                 float* __restrict__ loss_part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int B = b.n, tile = blockIdx.x, row0 = tile * IL_TILE_R, tid = threadIdx.x;
  const int Sp = round_up16(S), ldx = Sp + 4, ldh = H + 4, ldz = 20;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* part = H2s + IL_TILE_R * ldh; float* Os = part + (blockDim.x >> 6) * 256;
  float* red = Os + 256;
  float* DZ3s = part;  // reused after the head
  const SacWs ws = sac_ws(S, A, H, B);
  const MlpView net = mlp_view(actor, S, H, 2 * A);
  load_rows_cat(Xs, ldx, Sp, b.states, b.ld_states, S, nullptr, 0, 0, row0, IL_TILE_R);
  __syncthreads();
  const int lane = tid & 63, j = lane & 15, g = lane >> 4;
  tile_fwd(Xs, ldx, Sp, net.W1, S, S, H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = net.b1[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H1s[(4 * g + r) * ldh + col] = hv[r]; }
    *reinterpret_cast<f32x4*>(W + ws.a_h1 + (size_t)col * B + row0 + 4 * g) = hv;
  });
  __syncthreads();
  tile_fwd_packed(H1s, ldh, H, W + ws.pk_af, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = net.b2[col];
    f32x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { hv[r] = fmaxf(acc[r] + bb, 0.f); H2s[(4 * g + r) * ldh + col] = hv[r]; }
    *reinterpret_cast<f32x4*>(W + ws.a_h2 + (size_t)col * B + row0 + 4 * g) = hv;
  });
  __syncthreads();
  tile_fwd_small(H2s, ldh, H, net.W3, H, 2 * A, net.b3, Os, part);
  // head
  float lp_term = 0.f, dmean = 0.f, dls = 0.f; int hr = 0, hc = 0;
  const bool head = tid < IL_TILE_R * A;
  if (head) {
    hr = tid / A; hc = tid - hr * A; const int row = row0 + hr;
    const float a = fminf(fmaxf(b.actions[(size_t)row * b.ld_actions + hc], -1.f + 1e-6f), 1.f - 1e-6f);
    const float x = atanhf(a);
    const float mean = Os[hr * 16 + hc], lsr = Os[hr * 16 + A + hc];
    const float sd = expf(fminf(fmaxf(lsr, -20.f), 2.f));
    const float df = x - mean, var = sd * sd;
    const float nlp = -(df * df) / (2.f * var) - logf(sd) - LOG_SQRT_2PI;
    const float ladj = 2.f * (LOG_2 - x - softplus_f(-2.f * x));
    lp_term = nlp - ladj;
    const float up = -b.weights[(size_t)row * b.ld_weights] / (float)B;
    dmean = up * df / var;
    const float dsd = up * (df * df / (var * sd) - 1.f / sd);
    dls = (lsr >= -20.f && lsr <= 2.f) ? dsd * sd : 0.f;
  }
  __syncthreads();  // Os/part fully consumed before DZ3s (aliasing part) is written
  for (int i = tid; i < IL_TILE_R * ldz; i += blockDim.x) DZ3s[i] = 0.f;
  __syncthreads();
  if (head) { DZ3s[hr * ldz + hc] = dmean; DZ3s[hr * ldz + A + hc] = dls; }
  const float lsum = block_sum(head ? -b.weights[(size_t)(row0 + hr) * b.ld_weights] * lp_term : 0.f, red);
  if (tid == 0) { if (loss_part) loss_part[tile] = lsum; if (tile == 0) adam_tick(opt); }
  for (int i = tid; i < IL_TILE_R * 16; i += blockDim.x) W[ws.a_dz3 + (size_t)(i >> 4) * B + row0 + (i & 15)] = DZ3s[(i & 15) * ldz + (i >> 4)];
  float* DZ2s = H1s;  // h1 lives in the workspace copy from here on
  __syncthreads();
  tile_bwd_dx(DZ3s, ldz, 16, 2 * A, net.W3, H, H, [&](int kb, f32x4 acc) {
    const size_t off = (size_t)(kb + j) * B + row0 + 4 * g;
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = H2s[(4 * g + r) * ldh + kb + j] > 0.f ? acc[r] : 0.f; DZ2s[(4 * g + r) * ldh + kb + j] = o[r]; }
    *reinterpret_cast<f32x4*>(W + ws.a_dz2 + off) = o;
  });
  __syncthreads();
  tile_bwd_packed(DZ2s, ldh, H, W + ws.pk_ab, [&](int kb, f32x4 acc) {
    const size_t off = (size_t)(kb + j) * B + row0 + 4 * g;
    const f32x4 hv = *reinterpret_cast<const f32x4*>(W + ws.a_h1 + off);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = hv[r] > 0.f ? acc[r] : 0.f;
    *reinterpret_cast<f32x4*>(W + ws.a_dz1 + off) = o;
  });
}

extern "C" int il_bc_step(float* actor, float* actor_grad, const il_adam* opt, int32_t S, int32_t A, int32_t H, const il_batch* b, float* workspace,
                          int64_t workspace_floats, float* out_loss_partials, uint32_t flags, il_stream_t stream_) {
  IL_NO_GATHER(b, "il_bc_step");
  IL_CHECK_ARG(actor && opt && b && workspace, "il_bc_step: null argument");
  IL_CHECK_ARG(H % 64 == 0 && H >= 64 && H <= 256, "il_bc_step: hidden=%d must be a multiple of 64 in [64,256]", H);
  IL_CHECK_ARG(b->n > 0 && b->n % IL_TILE_R == 0, "il_bc_step: batch=%d must be a positive multiple of %d", b->n, IL_TILE_R);
  IL_CHECK_ARG(A >= 1 && 2 * A <= 16, "il_bc_step: action_dim=%d unsupported (2A <= 16)", A);
  IL_CHECK_ARG(!(flags & IL_FLAG_GRADS_ONLY) || actor_grad, "il_bc_step: IL_FLAG_GRADS_ONLY needs actor_grad");
  const SacWs ws = sac_ws(S, A, H, b->n);
  if (workspace_floats < ws.total) return il_set_error(IL_ERR_WORKSPACE, "il_bc_step: workspace too small (%lld < %lld floats)", (long long)workspace_floats, (long long)ws.total);
  hipStream_t st = (hipStream_t)stream_;
  const int nt = b->n / IL_TILE_R;
  {
    il_sac tmp = {};  // k_repack only needs the dims, the actor arena and the workspace
    tmp.state_dim = S; tmp.action_dim = A; tmp.hidden = H; tmp.batch = b->n; tmp.actor = actor; tmp.workspace = workspace;
    IL_TRACE("k_repack", st); k_repack<<<dim3(repack_blocks(H), 1), 256, 0, st>>>(tmp, 0x1u, nullptr);
  }
  { IL_TRACE("k_bc_tile", st); k_bc_tile<<<nt, tile_threads(H), tile_lds_bytes(round_up16(S + A), H), st>>>(actor, *opt, S, A, H, *b, workspace, out_loss_partials); }
  DwArgs a = {};
  a.params = actor; a.grads = actor_grad; a.opt = *opt; a.grads_only = (flags & IL_FLAG_GRADS_ONLY) ? 1 : 0;
  a.n_nets = 1; a.in_dim = S; a.hidden = H; a.out_dim = 2 * A; a.batch = b->n;
  a.x0 = b->states; a.ld_x0 = b->ld_states; a.x0_transposed = 0;
  a.h1 = workspace + ws.a_h1; a.h2 = workspace + ws.a_h2; a.dz1 = workspace + ws.a_dz1; a.dz2 = workspace + ws.a_dz2;
  a.dz3 = workspace + ws.a_dz3;
  a.n_dw_blocks = dw_blocks(S, H, 2 * A, 1);
  { IL_TRACE("k_dw_adam_bc", st); launch_dw_adam(a, a.n_dw_blocks, st); }
  IL_CHECK_LAUNCH("il_bc_step");
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// Acting (train.py:152 `actor(state).sample()`, models.py:101-102 greedy): n states, any n >= 1.  grid = ceil(n/16)
// ---------------------------------------------------------------------------------------------
struct ActTile { float* Os; float* part; };

// actor MLP on one 16-row tile of states -> Os[r*16 + c] = (mean | raw log-std) of row r; shared by k_act and k_act_step
__device__ __forceinline__ ActTile actor_tile(float* smem, const float* __restrict__ actor, int S, int A, int H, const float* __restrict__ states, int ld, int row0, int nrows) {
  const int tid = threadIdx.x;
  const int Sp = round_up16(S), ldx = Sp + 4, ldh = H + 4;
  float* Xs = smem; float* H1s = Xs + IL_TILE_R * ldx; float* H2s = H1s + IL_TILE_R * ldh; float* part = H2s + IL_TILE_R * ldh; float* Os = part + (blockDim.x >> 6) * 256;
  const MlpView net = mlp_view(actor, S, H, 2 * A);
  load_rows_cat(Xs, ldx, Sp, states, ld, S, nullptr, 0, 0, row0, nrows);
  __syncthreads();
  const int lane = tid & 63, j = lane & 15, g = lane >> 4;
  tile_fwd(Xs, ldx, Sp, net.W1, S, S, H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = net.b1[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) H1s[(4 * g + r) * ldh + col] = fmaxf(acc[r] + bb, 0.f);
  });
  __syncthreads();
  tile_fwd(H1s, ldh, H, net.W2, H, H, H, [&](int c0, f32x4 acc) {
    const int col = c0 + j; const float bb = net.b2[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) H2s[(4 * g + r) * ldh + col] = fmaxf(acc[r] + bb, 0.f);
  });
  __syncthreads();
  tile_fwd_small(H2s, ldh, H, net.W3, H, 2 * A, net.b3, Os, part);
  return ActTile{Os, part};
}

__global__ __launch_bounds__(1024) void k_act(const float* __restrict__ actor, int S, int A, int H, const float* __restrict__ states, int ld, int n, const float* __restrict__ eps,
                                             uint64_t seed, uint32_t offset, int greedy, float* __restrict__ out_a, float* __restrict__ out_logp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int row0 = blockIdx.x * IL_TILE_R, tid = threadIdx.x, nrows = min(IL_TILE_R, n - row0);
  const ActTile t = actor_tile(smem, actor, S, A, H, states, ld, row0, nrows);
  float* nl = t.part; float* la = t.part + 256;
  if (tid < IL_TILE_R * A) {
    const int r = tid / A, c = tid - r * A, row = row0 + r;
    const float mean = t.Os[r * 16 + c], lsr = t.Os[r * 16 + A + c];
    float x, a = tanhf(mean), nlp = 0.f, ladj = 0.f;
    if (!greedy) {
      const float e = eps ? (r < nrows ? eps[(size_t)row * A + c] : 0.f) : philox_normal(seed, offset, IL_STREAM_ACT, (uint32_t)(row * A + c));
      head_sample(mean, lsr, e, x, a, nlp, ladj);
    }
    nl[r * 16 + c] = nlp; la[r * 16 + c] = ladj;
    if (r < nrows) out_a[(size_t)row * A + c] = a;
  }
  __syncthreads();
  if (out_logp && tid < nrows) {
    float sn = 0.f, sl = 0.f;
    for (int c = 0; c < A; ++c) { sn += nl[tid * 16 + c]; sl += la[tid * 16 + c]; }
    out_logp[row0 + tid] = (0.f - sl) + sn;
  }
}

// SoftActor.log_prob(state, action) (models.py:97-99: action clamped to +-(1 - 1e-6), tanh-Gaussian density) for n rows: the log pi(a|s) that
// subtract_log_policy feeds to the discriminator (models.py:144). grid = ceil(n/16)
__global__ __launch_bounds__(1024) void k_actor_logp(const float* __restrict__ actor, int S, int A, int H, const float* __restrict__ states, int ld_s, const float* __restrict__ actions,
                                                    int ld_a, int n, float* __restrict__ out_logp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int row0 = blockIdx.x * IL_TILE_R, tid = threadIdx.x, nrows = min(IL_TILE_R, n - row0);
  const ActTile t = actor_tile(smem, actor, S, A, H, states, ld_s, row0, nrows);
  float* nl = t.part; float* la = t.part + 256;
  if (tid < IL_TILE_R * A) {
    const int r = tid / A, c = tid - r * A, row = row0 + min(r, nrows - 1);
    const float mean = t.Os[r * 16 + c], sd = expf(fminf(fmaxf(t.Os[r * 16 + A + c], -20.f), 2.f));
    const float a = fminf(fmaxf(actions[(size_t)row * ld_a + c], -1.f + 1e-6f), 1.f - 1e-6f);
    const float x = atanhf(a), dx = x - mean;
    nl[r * 16 + c] = -(dx * dx) / (2.f * (sd * sd)) - logf(sd) - LOG_SQRT_2PI;
    la[r * 16 + c] = 2.f * (LOG_2 - x - softplus_f(-2.f * x));
  }
  __syncthreads();
  if (tid < nrows) {
    float sn = 0.f, sl = 0.f;
    for (int c = 0; c < A; ++c) { sn += nl[tid * 16 + c]; sl += la[tid * 16 + c]; }
    out_logp[row0 + tid] = (0.f - sl) + sn;
  }
}

extern "C" int il_actor_log_prob(const float* actor, int32_t S, int32_t A, int32_t H, const float* states, int32_t ld_states, const float* actions, int32_t ld_actions, int32_t n,
                                 float* out_logp, il_stream_t stream_) {
  IL_CHECK_ARG(actor && states && actions && out_logp && n > 0, "il_actor_log_prob: null argument");
  IL_CHECK_ARG(H % 64 == 0 && H >= 64 && H <= 256 && A >= 1 && 2 * A <= 16, "il_actor_log_prob: unsupported dims (hidden=%d, action_dim=%d)", H, A);
  { IL_TRACE("k_actor_logp", stream_);
    k_actor_logp<<<ceil_div(n, IL_TILE_R), tile_threads(H), tile_lds_bytes(round_up16(S + A), H), (hipStream_t)stream_>>>(actor, S, A, H, states, ld_states, actions, ld_actions, n, out_logp); }
  IL_CHECK_LAUNCH("il_actor_log_prob");
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// One environment step of the acting worker (train.py:151-168) as ONE launch: append the pending transition to the ring
// (memory.py:40-44), optionally wrap it for absorbing states (memory.py:65-68), sample the action for the next observation
// (models.py:90-94) and hand it to the host through a pinned, device-mapped mailbox. The ring cursor lives on the device.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_act_step(const float* __restrict__ actor, int S, int A, int H, float* mail, float* __restrict__ carry, float* __restrict__ ring,
                                                  long long* __restrict__ ring_state, int row, uint64_t seed, uint32_t offset, const int* __restrict__ version, long long mirror_stride) {
  if (version) actor += (size_t)version[0] * mirror_stride;   // published parameter snapshot (il_act_publish): never the arena an update is rewriting
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int Sp4 = (S + 3) & ~3, Ap4 = (A + 3) & ~3;
  const float* m_next = mail + IL_MAIL_HEADER; const float* m_obs = m_next + Sp4; float* m_act = mail + IL_MAIL_HEADER + 2 * Sp4; float* m_echo = m_act + Ap4;
  // mail[0] is the commit word (sequence << 6 | IL_ACT_* flags), the LAST thing the host writes: one 4-byte store publishes the post.
  // carry[S+A] remembers the commit word of the last appended transition, so a launch that runs again without a new post (a replayed
  // graph, or a launch still queued when the host posts the next step) appends each transition exactly once.
  const float commit = mail[0];
  const unsigned word = (unsigned)commit, flags = word & 63u;
  float* consumed = carry + S + A;
  const long long cursor = ring_state[0], cap = ring_state[2];
  const bool pending = (flags & IL_ACT_PENDING) && __float_as_uint(consumed[0]) != word, wrap = pending && (flags & IL_ACT_WRAP_ABSORBING);
  const int o_next = S + A, o_rew = 2 * S + A;
  if (pending && tid < row) {
    const int c = tid;
    float v = 0.f;
    if (c < o_next) v = (flags & IL_ACT_CARRY_FROM_MAILBOX) ? (c < S ? m_obs[c] : m_act[c - S]) : carry[c];   // state | action of the transition
    else if (c < o_rew) v = wrap ? (c == o_rew - 1 ? 1.f : 0.f) : m_next[c - o_next];    // next_state, or the absorbing state (memory.py:67)
    else if (c == o_rew) v = mail[2];                                                    // reward
    else if (c == o_rew + 1) v = wrap ? 0.f : mail[3];                                   // terminal (cleared by the wrap)
    else if (c == o_rew + 2) v = mail[4];                                                // timeout
    else if (c == o_rew + 3) v = 1.f;                                                    // weight
    else if (c == o_rew + 4) v = mail[5];                                                // step
    ring[cursor * row + c] = v;
    if (wrap) {  // absorbing -> absorbing row (memory.py:68)
      float w = 0.f;
      if (c < S) w = (c == S - 1) ? 1.f : 0.f;
      else if (c >= o_next && c < o_rew) w = (c == o_rew - 1) ? 1.f : 0.f;
      else if (c == o_rew + 3) w = 1.f;
      else if (c == o_rew + 4) w = mail[5];
      ring[((cursor + 1) % cap) * row + c] = w;
    }
  }
  if (!(flags & IL_ACT_NO_ACTION)) {  // block-uniform
    const int greedy = flags & IL_ACT_GREEDY;
    const ActTile t = actor_tile(smem, actor, S, A, H, m_obs, Sp4, 0, 1);   // barriers inside: every carry[] read above precedes the writes below
    if (tid < A) {
      const float mean = t.Os[tid], lsr = t.Os[A + tid];
      float x, a = tanhf(mean), nlp, ladj;
      if (!greedy) head_sample(mean, lsr, philox_normal(seed, offset, IL_STREAM_ACT, (uint32_t)tid), x, a, nlp, ladj);
      m_act[tid] = a; carry[S + tid] = a;
    }
    if (tid >= 64 && tid < 64 + S) carry[tid - 64] = m_obs[tid - 64];
  }
  __threadfence_system();
  __syncthreads();   // every thread has read consumed[0], the cursor and the mailbox by now
  if (tid == 0) {
    if (pending) {
      // the cursor moves only here, behind the barrier: an append-only launch (IL_ACT_NO_ACTION) has no other barrier between the waves' loads of ring_state[0] above and
      // this store, and with rows wider than one wave (Ant: 240 floats) a wave that loaded late would have written its columns into the next row
      const long long adv = wrap ? 2 : 1, nc = cursor + adv;
      ring_state[0] = nc % cap;
      if (nc >= cap) ring_state[1] = 1;
      consumed[0] = __uint_as_float(word);
    }
    __hip_atomic_store(m_echo, commit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

extern "C" int il_actor_act(const float* actor, int32_t S, int32_t A, int32_t H, const float* states, int32_t ld_states, int32_t n, const float* eps,
                            uint64_t noise_seed, uint32_t noise_offset, int32_t greedy, float* out_action, float* out_logp, il_stream_t stream_) {
  IL_CHECK_ARG(actor && states && out_action && n > 0, "il_actor_act: null argument");
  IL_CHECK_ARG(H % 64 == 0 && H >= 64 && H <= 256 && A >= 1 && 2 * A <= 16, "il_actor_act: unsupported dims (hidden=%d, action_dim=%d)", H, A);
  {
    IL_TRACE("k_act", stream_);
    k_act<<<ceil_div(n, IL_TILE_R), tile_threads(H), tile_lds_bytes(round_up16(S + A), H), (hipStream_t)stream_>>>(actor, S, A, H, states, ld_states, n, eps, noise_seed, noise_offset, greedy,
                                                                                                       out_action, out_logp);
  }
  IL_CHECK_LAUNCH("il_actor_act");
  return IL_OK;
}

// Parameter snapshot for an acting worker that runs next to the updates (its own stream): copy the actor arena into slot (version+1)%3
// of `mirror` and then advance `version` (last workgroup to finish), so a concurrent k_act_step always reads a complete snapshot.
__global__ __launch_bounds__(256) void k_act_publish(const float* __restrict__ actor, long long n, float* __restrict__ mirror, long long stride, int* __restrict__ version,
                                                    unsigned* __restrict__ done) {
  const int next = (version[0] + 1) % 3;
  float* dst = mirror + (size_t)next * stride;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = actor[i];
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(done, 1u) == gridDim.x - 1) {
    *done = 0u;
    __threadfence();
    __hip_atomic_store(version, next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int il_act_publish(const float* actor, int64_t n, float* mirror, int64_t mirror_stride, int32_t* version_and_counter, il_stream_t stream_) {
  IL_CHECK_ARG(actor && mirror && version_and_counter && n > 0 && mirror_stride >= n, "il_act_publish: bad arguments");
  const int blocks = (int)((n + 1023) / 1024 < 64 ? (n + 1023) / 1024 : 64);
  { IL_TRACE("k_act_publish", stream_); k_act_publish<<<blocks, 256, 0, (hipStream_t)stream_>>>(actor, n, mirror, mirror_stride, version_and_counter, (unsigned*)(version_and_counter + 1)); }
  IL_CHECK_LAUNCH("il_act_publish");
  return IL_OK;
}

extern "C" int32_t il_act_mailbox_floats(int32_t S, int32_t A) { return (IL_MAIL_HEADER + 2 * ((S + 3) & ~3) + ((A + 3) & ~3) + 1 + 15) & ~15; }

extern "C" int il_act_step(const float* actor, int32_t S, int32_t A, int32_t H, float* mailbox, float* carry, float* ring, int64_t* ring_state, uint64_t noise_seed,
                           uint32_t noise_offset, const int32_t* mirror_version, int64_t mirror_stride, il_stream_t stream_) {
  IL_CHECK_ARG(actor && mailbox && carry && ring && ring_state, "il_act_step: null argument");
  IL_CHECK_ARG(H % 64 == 0 && H >= 64 && H <= 256 && A >= 1 && 2 * A <= 16, "il_act_step: unsupported dims (hidden=%d, action_dim=%d)", H, A);
  const int row = il_ring_row_floats(S, A);
  IL_CHECK_ARG(row <= tile_threads(H) && 64 + S <= tile_threads(H), "il_act_step: ring row of %d floats / state_dim %d exceed the %d-thread workgroup", row, S, tile_threads(H));
  {
    IL_TRACE("k_act_step", stream_);
    k_act_step<<<1, tile_threads(H), tile_lds_bytes(round_up16(S + A), H), (hipStream_t)stream_>>>(actor, S, A, H, mailbox, carry, ring, (long long*)ring_state, row, noise_seed, noise_offset,
                                                                                                    mirror_version, mirror_stride);
  }
  IL_CHECK_LAUNCH("il_act_step");
  return IL_OK;
}

IL_STAMP_READER(il_debug_stamps_sac)
IL_TL_READER(il_debug_timeline_sac)
IL_ST_READER(il_stamps_sac)
