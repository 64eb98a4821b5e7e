// GAIL discriminator kernels (reference training.py:85-134, models.py:152-180, torch _SpectralNorm).
//
// k_gail_grad   tiles of 16 rows. Every workgroup redoes the three spectral-norm power iterations (they depend only on
//               W, u, v: 2 mat-vecs of a [H x D] matrix) and then, for its rows, the three discriminator passes
//               (policy, expert, gradient-penalty mix) with closed-form backward:
//                 BCE:  dz = w (sigmoid(z) - y)/B (+ entropy bonus),  dh = dz w2^ [h>0]
//                 GP :  q = [h>0] w2^,  g = W1^^T q,  c = 2 lambda w/B:  dW1^ += c q g^T,  dw2^ += c [h>0] (W1^ g)
//               and the spectral-norm chain rule per pass  dW = G^/sigma - <G^, W>/sigma^2 u v^T, with <G^, W> evaluated
//               through the forward values (<G1^,W1> = sigma1 sum dh (h - b1), ...) so G^ is never materialised.
//               Each workgroup writes one partial gradient slab; no atomics => deterministic.
// k_gail_reduce one workgroup: slab sum -> grad, AdamW, spectral-norm buffers update.
// k_gail_reward eval-mode forward + AIRL / GAIL / FAIRL reward head.
#include "il_common.hpp"
#include "mt_device.hpp"
#include "mlp_tile.hpp"
#include "disc_reward.hpp"
#include "peer_device.hpp"
IL_ST_TABLE

struct DiscWs { int64_t slabs, sn_new, pu, total; };   // pu: [2][nt] per-tile sums of w softplus(z) of the policy / expert call (PUGAIL with a finite nonnegative_margin)
__host__ __device__ inline DiscWs disc_ws(int D, int H, int B) {
  DiscWs w; const int64_t P = (int64_t)H * D + 2 * H + 1; const int nt = (B + IL_TILE_R - 1) / IL_TILE_R;
  w.slabs = 0; w.sn_new = (3 * nt * P + 3) & ~(int64_t)3; w.pu = (w.sn_new + 2 * H + D + 1 + 3) & ~(int64_t)3; w.total = w.pu + 2 * nt + 4;
  return w;
}
extern "C" int64_t il_disc_workspace_floats(int32_t D, int32_t H, int32_t B) { return disc_ws(D, H, B).total; }

// ---------------------------------------------------------------------------------------------
// Spectral norm, executed by ONE wave with wave-level reductions only (no block barriers): the power iterations of the three
// discriminator calls depend on nothing but W, u, v, so wave 0 runs all of them up front while waves 1.. stage the batch rows.
// W1s: LDS copy of W1 with row stride D+1 (conflict-free for both W v and W^T u).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Spectral norm of call `c` (0-based): the reference runs c+1 chained power iterations u <- n(W v), v <- n(W^T u) before that call.
// v only ever sees M = W^T W:  v_{i+1} = n(M v_i)  (the 1/||W v|| factor cancels in the normalisation), so the chain runs on the
// small [D x D] Gram matrix (built once per workgroup by all threads) and only the LAST u is formed: u_c = n(W v_c'), v_c' = v after
// c iterations, sigma_c = u_c . (W v_{c+1}).  Layer 2 (W2 is [1 x H]) is a fixed point after its first iteration: one is enough.
// Same numbers as the reference's sequence up to fp32 rounding of the re-associated products.
// ---------------------------------------------------------------------------------------------
// One 16 x 16 tile of C = A . B with BOTH operands in LDS, arbitrary strides: A(i, k) = Ap[i*sai + k*sak], B(k, j) = Bp[k*sbk + j*sbj],
// K % 4 == 0 (fp32 MFMA 16x16x4: lane (j, g) supplies A(j, k0+g) and B(k0+g, j); holds C(4g+reg, j)). The discriminator's products
// (16 rows x H x D, a few dozen MFMAs each) were LDS-latency-bound as scalar FMA loops; here four k-steps of operands are in flight.
__device__ __forceinline__ f32x4 mfma_lds_tile(const float* Ap, int sai, int sak, const float* Bp, int sbk, int sbj, int K) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const float* ap = Ap + j * sai + g * sak;
  const float* bp = Bp + g * sbk + j * sbj;
  f32x4 acc0 = zero4(), acc1 = zero4();
  int k0 = 0;
  for (; k0 + 16 <= K; k0 += 16) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = ap[(k0 + 4 * u) * sak]; b[u] = bp[(k0 + 4 * u) * sbk]; }
    acc0 = mfma16(a[0], b[0], acc0); acc1 = mfma16(a[1], b[1], acc1); acc0 = mfma16(a[2], b[2], acc0); acc1 = mfma16(a[3], b[3], acc1);
  }
  for (; k0 < K; k0 += 4) acc0 = mfma16(ap[k0 * sak], bp[k0 * sbk], acc0);
  return acc0 + acc1;
}

// M = W1^T W1 on MFMA: tiles (a0, b0) of [Dp x Dp], reduction over the H rows; columns beyond Dp read padding / neighbours and are dropped
__device__ __forceinline__ void sn_gram(const float* W1s, float* Ms, int D, int Dp, int H, int ldw) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int nt = (Dp + 15) / 16;
  for (int t = wave; t < nt * nt; t += nw) {
    const int a0 = (t / nt) * 16, b0 = (t % nt) * 16;
    const f32x4 cc = mfma_lds_tile(W1s + a0, 1, ldw, W1s + b0, ldw, 1, H);
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int a = a0 + 4 * g + r, b = b0 + j; if (a < Dp && b < Dp) Ms[a * Dp + b] = cc[r]; }
  }
}
// one wave; n_iter >= 1 chained iterations starting from (v1, v2); outputs u1, v1, v2, sig = {sigma1, sigma2, u2}
__device__ __forceinline__ void sn_chain(const float* W1s, const float* W2s, const float* Ms, int D, int H, float* u1, float* v1, float* v2, float* tmp, int n_iter, float* sig) {
  const int lane = threadIdx.x & 63, Dp = (D + 3) & ~3, ldw = Dp + 4;
  for (int it = 0; it < n_iter; ++it) {
    if (it == n_iter - 1) {  // the u of the last iteration: u = n(W v)
      float ss = 0.f;
      for (int n = lane; n < H; n += 64) { const float s = dot4(W1s + n * ldw, v1, Dp); u1[n] = s; ss += s * s; }
      const float inv = wave_norm_scale(ss);
      for (int n = lane; n < H; n += 64) u1[n] *= inv;
    }
    float ss = 0.f;
    for (int k = lane; k < Dp; k += 64) { const float s = k < D ? dot4(Ms + k * Dp, v1, Dp) : 0.f; tmp[k] = s; ss += s * s; }
    const float inv = wave_norm_scale(ss);
    WAVE_SYNC();
    for (int k = lane; k < Dp; k += 64) v1[k] = tmp[k] * inv;
    WAVE_SYNC();
  }
  // layer 2: p = W2 . v2 ; u2 = p/|p| ; v2 = n(W2^T u2)
  float p = 0.f;
  for (int n = lane; n < H; n += 64) p += W2s[n] * v2[n];
  p = wave_sum(p);
  const float uu = p / fmaxf(fabsf(p), 1e-12f);
  float ss = 0.f;
  for (int n = lane; n < H; n += 64) { const float s = W2s[n] * uu; tmp[n] = s; ss += s * s; }
  const float inv2 = wave_norm_scale(ss);
  WAVE_SYNC();
  for (int n = lane; n < H; n += 64) v2[n] = tmp[n] * inv2;
  WAVE_SYNC();
  float a = 0.f, b = 0.f;
  for (int n = lane; n < H; n += 64) { a += u1[n] * dot4(W1s + n * ldw, v1, Dp); b += W2s[n] * v2[n]; }
  a = wave_sum(a); b = wave_sum(b);
  if (lane == 0) { sig[0] = a; sig[1] = uu * b; sig[2] = uu; }
  WAVE_SYNC();
}

// LDS of one discriminator workgroup. The arrays k_gail_reward needs come first (disc_reward_lds_floats: its launches ask for that prefix only - 34 KB at HalfCheetah dims, four
// workgroups per CU); the whole set is 73 KB, two workgroups of k_gail_grad per CU (round 3: three slots of X / wt / u,v of which only the first was used and a third
// [16][H] array made it 96 KB - ONE four-wave workgroup per CU on the population launches, every dependent phase exposed). `ts` IS `hs`: the gradient-penalty call
// replaces h by [h > 0] t' element by element (same lane reads and writes), and nothing reads h after that.
struct DiscLds {
  float *W1s, *b1s, *W2s, *Xb, *wtb, *snb, *hs, *dhs, *ts, *cg, *zs, *dzs, *red, *Ms, *tmp;
  int D, H, Dp;  // Dp = D rounded up to 4 (rows of X / cg / v1 are zero-padded so dot products run on 16-byte lanes)
  __device__ __forceinline__ float* X(int) const { return Xb; }
  __device__ __forceinline__ float* wt(int) const { return wtb; }
  __device__ __forceinline__ float* u1(int) const { return snb; }
  __device__ __forceinline__ float* v1(int) const { return snb + H; }
  __device__ __forceinline__ float* v2(int) const { return snb + H + Dp; }
  __device__ __forceinline__ float* sc(int) const { return snb + 2 * H + Dp; }
};
__host__ __device__ inline size_t disc_reward_lds_floats(int D, int H) {
  const int Dp = (D + 3) & ~3;
  return (size_t)H * (Dp + 4) + 2 * H + (size_t)IL_TILE_R * Dp + IL_TILE_R + (size_t)(2 * H + Dp + 4);
}
__host__ __device__ inline size_t disc_lds_floats(int D, int H) {
  const int Dp = (D + 3) & ~3;
  return disc_reward_lds_floats(D, H) + 2 * (size_t)IL_TILE_R * H + (size_t)IL_TILE_R * Dp + 2 * IL_TILE_R + 64 + (size_t)Dp * Dp + H + Dp;
}
__device__ __forceinline__ DiscLds carve(float* s, int D, int H) {
  DiscLds l; float* p = s;
  const int Dp = (D + 3) & ~3;
  l.D = D; l.H = H; l.Dp = Dp;
  l.W1s = p; p += H * (Dp + 4); l.b1s = p; p += H; l.W2s = p; p += H;
  l.Xb = p; p += IL_TILE_R * Dp; l.wtb = p; p += IL_TILE_R;
  l.snb = p; p += 2 * H + Dp + 4;
  l.hs = p; p += IL_TILE_R * H; l.dhs = p; p += IL_TILE_R * H; l.ts = l.hs; l.cg = p; p += IL_TILE_R * Dp;
  l.zs = p; p += IL_TILE_R; l.dzs = p; p += IL_TILE_R; l.red = p; p += 64;
  l.Ms = p; p += Dp * Dp; l.tmp = p;
  return l;
}

// stage W1 (padded rows), b1, W2 into LDS
__device__ __forceinline__ void stage_weights(const DiscLds& L, const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2, int D, int H) {
  const int Dp = L.Dp, ldw = Dp + 4, tid = threadIdx.x, bd = blockDim.x;
  if (disc_w1_flat_ok(W1, D, H)) {   // (round 4) 16-byte lanes of the flat array, six per thread in flight (disc_reward.hpp)
    const int nvec = (H * D) >> 2;
    const unsigned mdd = fastdiv_magic(D);
    DiscW1<6> ws;
    disc_w1_issue(ws, W1, nvec, 0);
    const float vb1 = gload(b1 + min(tid, H - 1)), vw2 = gload(W2 + min(tid, H - 1));
    disc_w1_commit(ws, L.W1s, D, ldw, nvec, 0, mdd);
    for (int base = 6 * bd; base < nvec; base += 6 * bd) { disc_w1_issue(ws, W1, nvec, base); disc_w1_commit(ws, L.W1s, D, ldw, nvec, base, mdd); }
    disc_w1_zero_padding(L.W1s, D, Dp, H, ldw);
    if (tid < H) { L.b1s[tid] = vb1; L.W2s[tid] = vw2; }
    for (int i = bd + tid; i < H; i += bd) { L.b1s[i] = b1[i]; L.W2s[i] = W2[i]; }
    return;
  }
  for (int i = tid; i < H * Dp; i += bd) { const int n = i / Dp, k = i - n * Dp; L.W1s[n * ldw + k] = k < D ? W1[(size_t)n * D + k] : 0.f; }
  for (int i = tid; i < H; i += bd) { L.b1s[i] = b1[i]; L.W2s[i] = W2[i]; }
}

// grid = (tiles, passes): one workgroup = 16 rows of ONE discriminator call (0 policy, 1 expert, 2 gradient-penalty mix), so the
// three calls run side by side; wave 0 chains the power iterations up to its call (pass + 1 of them) while waves 1.. stage rows.
// Loss variants (training.py:97-113): BCE and PUGAIL (nonnegative_margin = inf) are calls {policy, expert}; Mixup is ONE call on convex combinations
// with per-row soft labels. All three are  d loss / d logit = w (c_sig * sigmoid(z) - c_lab) / B  with different constants. `x` carries the optional
// inputs: the Mixup draws and the log pi(a|s) offsets of subtract_log_policy (logit z = f - log pi; computed without a graph, so a pure shift).
// The index draw of the update riding in this launch (il_gail_disc_step_draw): gridDim.x carries ONE extra column of workgroups, of which (nt, 0) is the sampler.
// The discriminator workgroups are resident early - weights staged, Gram matrix and power iterations done - and wait for [IL_SYNC_INDICES]; a sampler launched BEHIND
// this kernel in the same stream could never satisfy that wait, and one launched AHEAD of it would put this kernel's preparation back on the update's critical path.
struct GailSampler { uint32_t* state; const int64_t* rs_a; int32_t* idx_a; const int64_t* rs_b; int32_t* idx_b; int n; MtStage stage; };

// MULTI: several tiles per workgroup (k_gail_grad_pop); the one-tile instantiation is the single learner's kernel, whose registers and schedule the loop must not touch
// (as a run-time trip count it cost the headline 2 %: profiles/r04_pop_ab.txt).
template <bool MULTI>
__device__ __forceinline__ void gail_grad_body(il_disc d, il_batch pol, il_batch exp, const float* __restrict__ eps_gp, il_gail_extra x, const il_disc* __restrict__ dL,
                                               const il_batch* __restrict__ polL, const il_batch* __restrict__ expL, GailSampler sa, int pu_value_pass, int tpw_, float* smem) {
  const int tpw = MULTI ? tpw_ : 1;
  IL_TL(0, 0);
  const int has_sampler = sa.state != nullptr;
  if (has_sampler && (int)blockIdx.x == (int)gridDim.x - 1) {
    if (blockIdx.y == 0) {
      MtStage stg = sa.stage; stg.ring = as_global(stg.ring); stg.rows = as_global(stg.rows);
      mt_sample_update(*reinterpret_cast<MtShared*>(smem), as_global(sa.state), sa.n, as_global(sa.rs_a), as_global(sa.idx_a), as_global(sa.rs_b), as_global(sa.idx_b),
                       reinterpret_cast<long long*>(as_global(d.sync)), 1, stg);
    }
    return;
  }
  if (dL) { d = dL[blockIdx.z]; pol = polL[blockIdx.z]; exp = expL[blockIdx.z]; }  // population axis
  globalize(d); globalize(pol); globalize(exp); globalize(x);
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, B = d.batch, Dp = (D + 3) & ~3, ldw = Dp + 4;
  // tpw (population launch): this workgroup runs `tpw` consecutive tiles of its call one after the other - the weights in LDS, the Gram matrix and the power iterations
  // (12 of a one-tile workgroup's 21 us there) are paid once per workgroup instead of once per tile. Every tile still leaves its own slab: same bits.
  const int pass = blockIdx.y, npass = gridDim.y, nt = (B + IL_TILE_R - 1) / IL_TILE_R, tid = threadIdx.x, tile0 = (int)blockIdx.x * tpw;
  int tile = tile0, row0 = tile * IL_TILE_R, nrows = min(IL_TILE_R, B - row0);
  const int kind = d.loss_function == IL_LOSS_MIXUP ? (pass == 0 ? 3 : 2) : pass;   // 0 policy, 1 expert, 2 gradient-penalty mix, 3 mixup mix
  const DiscLayout lay = disc_layout(D, H, d.spectral_norm);
  const DiscWs wsl = disc_ws(D, H, B);
  if (pu_value_pass && kind != 0 && kind != 1) return;   // the value pass only needs the logits of the policy and the expert call
  const float b2 = d.params[lay.ob2];
  DiscLds L = carve(smem, D, H);
  bool stamp = tile == 0 && pass == 2;
  IL_STAMP(stamp, 0);
  stage_weights(L, d.params + lay.oW1, d.params + lay.ob1, d.params + lay.oW2, D, H);
  IL_STAMP(stamp, 1);
  if (d.spectral_norm) {
    for (int i = tid; i < H; i += blockDim.x) { L.u1(0)[i] = d.u1[i]; L.v2(0)[i] = d.v2[i]; }
    for (int i = tid; i < Dp; i += blockDim.x) L.v1(0)[i] = i < D ? d.v1[i] : 0.f;
    if (tid == 0) L.sc(0)[2] = d.u2[0];
  } else if (tid == 0) { L.sc(0)[0] = 1.f; L.sc(0)[1] = 1.f; L.sc(0)[2] = 0.f; }
  if (tid == 0 && tile0 == 0 && pass == 0 && !pu_value_pass) adam_tick(d.opt);
  __syncthreads();
  IL_STAMP(stamp, 2);
  if (!d.sync) IL_TL(0, 1);   // (slots 1 / 2 on the hand-off path: before / after the wait for the index draw)
  if (d.spectral_norm) { sn_gram(L.W1s, L.Ms, D, Dp, H, ldw); __syncthreads(); }
  IL_STAMP(stamp, 3);
  IL_TL(0, 3);
  float* X = L.X(0);
  // rows (and mixing weights) of this call, staged by threads [t0, blockDim.x)
  uint32_t ctr = 0u;   // Philox counter of this update (read below, once it is known to be this update's)
  auto mix_eps = [&](int row) -> float {   // U(0,1) of the gradient penalty (training.py:118) or the Beta(alpha, alpha) draw of Mixup (:106; on chip only alpha = 1)
    const float* given = kind == 2 ? eps_gp : x.eps_mix;
    return given ? given[row] : philox_uniform(d.noise_seed, ctr, kind == 2 ? IL_STREAM_GP : IL_STREAM_MIX, (uint32_t)row);
  };
  auto stage_rows = [&](int t0) {
    const int nthr = blockDim.x - t0;
    for (int i = tid - t0; i >= 0 && i < IL_TILE_R * Dp; i += nthr) {
      const int r = i / Dp, k = i - r * Dp; float xv = 0.f;
      if (r < nrows && k < D) {
        const int row = row0 + r;
        float xp = 0.f, xe = 0.f;
        if (kind != 1) { const size_t pr = brow(pol, row); xp = k < S ? pol.states[pr * pol.ld_states + k] : pol.actions[pr * pol.ld_actions + k - S]; }
        if (kind != 0) { const size_t er = brow(exp, row); xe = k < S ? exp.states[er * exp.ld_states + k] : exp.actions[er * exp.ld_actions + k - S]; }
        if (kind >= 2) { const float e = mix_eps(row); xv = e * xe + (1.f - e) * xp; }
        else xv = kind == 0 ? xp : xe;
      }
      X[i] = xv;
    }
    if (tid >= t0 && tid < t0 + IL_TILE_R) {
      const int r = tid - t0, row = row0 + r; float w = 0.f;
      if (r < nrows) {
        const float wp = kind != 1 ? pol.weights[brow(pol, row) * pol.ld_weights] : 0.f;
        const float we = kind != 0 ? exp.weights[brow(exp, row) * exp.ld_weights] : 0.f;
        if (kind >= 2) { const float e = mix_eps(row); w = e * we + (1.f - e) * wp; }
        else w = kind == 0 ? wp : we;
      }
      L.wt(0)[r] = w;
    }
  };
  if (d.sync) {
    // Launched without a stream dependency on the gather: everything that only needs the parameters (weights in LDS, Gram matrix, the power
    // iterations) has run or runs now; then wait for THIS update's rows (every gather workgroup has signalled) and stage them with all threads.
    if (tid < 64 && d.spectral_norm) sn_chain(L.W1s, L.W2s, L.Ms, D, H, L.u1(0), L.v1(0), L.v2(0), L.tmp, pass + 1, L.sc(0));
    long long* sy = reinterpret_cast<long long*>(d.sync);
    IL_TL(0, 1);
    if (pol.gather && exp.gather) {
      // rows come straight from the rings: only the draw has to be done (with the in-launch sampler a second wait follows, and the acquire - by every wave - sits behind that one)
      if (has_sampler) sync_wait_only(sy, IL_SYNC_INDICES, sync_read(sy, IL_SYNC_SIDE_EPOCH) + 1);
      else sync_wait(sy, IL_SYNC_INDICES, sync_read(sy, IL_SYNC_SIDE_EPOCH) + 1);
      // (round 5) the draw may now be AHEAD of the previous update's end: this step reads the Philox counter that update's actor step advances (below), so it still starts
      // behind [IL_SYNC_MAIN_EPOCH] (= the number of discriminator steps closed so far) - 4 us earlier than when the draw itself waited for it
      if (has_sampler) sync_wait(sy, IL_SYNC_MAIN_EPOCH, sync_read(sy, IL_SYNC_SIDE_EPOCH));
    } else {
      sync_wait(sy, IL_SYNC_ROWS, (sync_read(sy, IL_SYNC_SIDE_EPOCH) + 1) * sy[IL_SYNC_GATHER_WGS]);   // gathered batches: every gather workgroup of this update has signalled
    }
    IL_TL(0, 2);
    ctr = d.noise_counter ? *d.noise_counter : 0u;   // after the wait: the previous update's actor step (which bumps it) precedes this update's gather
    // (round 4) The rows of this call with every load of a round in flight: stage_rows' element loop makes, per element, an index load and then a dependent row load -
    // four serialised fabric / HBM round trips for the two elements a thread owns at HalfCheetah dims - and draws the mixing coefficient of a row once per ELEMENT
    // (a Philox call each). Here: the coefficients once per row (16 threads, into zs, which nothing reads before the loss), the indices of up to four elements per thread
    // requested together, then their rows. Same values, same X: same bits.
    float* eps_s = L.zs;
    if (kind >= 2 && tid < IL_TILE_R) eps_s[tid] = tid < nrows ? mix_eps(row0 + tid) : 0.f;
    const unsigned mdp = fastdiv_magic(Dp);
    for (int base = 0; base < IL_TILE_R * Dp; base += 4 * (int)blockDim.x) {
      int64_t pr[4], er[4]; int rr[4], kk[4]; bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + q * (int)blockDim.x + tid, ic = min(i, IL_TILE_R * Dp - 1);
        rr[q] = fastdiv(ic, mdp); kk[q] = ic - rr[q] * Dp; ok[q] = i < IL_TILE_R * Dp && rr[q] < nrows && kk[q] < D;
        const int row = row0 + min(rr[q], nrows - 1);
        pr[q] = (kind != 1 && pol.gather) ? (int64_t)gload(pol.gather + row) : (int64_t)row;
        er[q] = (kind != 0 && exp.gather) ? (int64_t)gload(exp.gather + row) : (int64_t)row;
      }
      int64_t wpr = row0 + min(tid, nrows - 1), wer = wpr;
      if (base == 0 && tid < IL_TILE_R) { if (kind != 1 && pol.gather) wpr = gload(pol.gather + wpr); if (kind != 0 && exp.gather) wer = gload(exp.gather + wer); }
      float xp[4], xe[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = min(kk[q], D - 1);
        int64_t a = pr[q], b = er[q];
        if (pol.gather) a = a < 0 ? 0 : (a >= pol.gather_capacity ? pol.gather_capacity - 1 : a);
        if (exp.gather) b = b < 0 ? 0 : (b >= exp.gather_capacity ? exp.gather_capacity - 1 : b);
        xp[q] = kind != 1 ? gload(k < S ? pol.states + (size_t)a * pol.ld_states + k : pol.actions + (size_t)a * pol.ld_actions + (k - S)) : 0.f;
        xe[q] = kind != 0 ? gload(k < S ? exp.states + (size_t)b * exp.ld_states + k : exp.actions + (size_t)b * exp.ld_actions + (k - S)) : 0.f;
      }
      float wp = 0.f, we = 0.f;
      if (base == 0 && tid < IL_TILE_R) {
        if (pol.gather) wpr = wpr < 0 ? 0 : (wpr >= pol.gather_capacity ? pol.gather_capacity - 1 : wpr);
        if (exp.gather) wer = wer < 0 ? 0 : (wer >= exp.gather_capacity ? exp.gather_capacity - 1 : wer);
        if (kind != 1) wp = gload(pol.weights + (size_t)wpr * pol.ld_weights);
        if (kind != 0) we = gload(exp.weights + (size_t)wer * exp.ld_weights);
      }
      if (base == 0) __syncthreads();   // the rows' mixing coefficients are in LDS
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + q * (int)blockDim.x + tid;
        if (i < IL_TILE_R * Dp) {
          float xv = 0.f;
          if (ok[q]) { if (kind >= 2) { const float e = eps_s[rr[q]]; xv = e * xe[q] + (1.f - e) * xp[q]; } else xv = kind == 0 ? xp[q] : xe[q]; }
          X[i] = xv;
        }
      }
      if (base == 0 && tid < IL_TILE_R) {
        float w = 0.f;
        if (tid < nrows) { if (kind >= 2) { const float e = eps_s[tid]; w = e * we + (1.f - e) * wp; } else w = kind == 0 ? wp : we; }
        L.wt(0)[tid] = w;
      }
    }
  } else {
    ctr = d.noise_counter ? *d.noise_counter : 0u;
    if (tid < 64) {  // ---- wave 0: the power iterations of calls 0..pass on the Gram matrix
      if (d.spectral_norm) sn_chain(L.W1s, L.W2s, L.Ms, D, H, L.u1(0), L.v1(0), L.v2(0), L.tmp, pass + 1, L.sc(0));
      IL_STAMP(stamp, 5);
    } else {         // ---- waves 1..3: rows of this call
      stage_rows(64);
    }
  }
  __syncthreads();
  for (int ti = 0;; ++ti) {   // the tiles of this workgroup (one, except on the population launch)
  float* slab = d.workspace + wsl.slabs + ((size_t)pass * nt + tile) * lay.P;
  stamp = tile == 0 && pass == 2;
  IL_STAMP(stamp, 6);
  IL_TL(0, 4);

  const int r = tid >> 4, sub = tid & 15;
  const int wave = tid >> 6, nw = blockDim.x >> 6, lane = tid & 63, jj = lane & 15, gg = lane >> 4;
  const float fB = (float)B;
  const bool valid = r < nrows;
  const float s1 = L.sc(0)[0], s2 = L.sc(0)[1], u2 = L.sc(0)[2];
  const float* u1 = L.u1(0); const float* v1 = L.v1(0); const float* v2 = L.v2(0);
  float* zw = L.dhs;  // [waves][16] per-wave partial logits (dhs is written after the barriers below)
  // ---- forward on MFMA: hs[r][n] = (X . W1^T)[r][n] / s1 + b1[n]  (one wave per 16 hidden units), per-row partial logits
  {
    float zp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = wave; t < H / 16; t += nw) {
      const int n = 16 * t + jj;
      const f32x4 c = mfma_lds_tile(X, Dp, 1, L.W1s + 16 * t * ldw, 1, ldw, Dp);
      const float bb = L.b1s[n], w2 = L.W2s[n] / s2;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float h = c[q] / s1 + bb; L.hs[(4 * gg + q) * H + n] = h; zp[q] += w2 * fmaxf(h, 0.f); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float t = group16_sum(zp[q]); if (jj == 0) zw[wave * 16 + 4 * gg + q] = t; }
  }
  __syncthreads();
  float f = b2;   // the network's logit
  for (int w = 0; w < nw; ++w) f += zw[w * 16 + r];
  __syncthreads();  // zw (aliasing dhs) is free again
  float ip1, ip2;
  if (kind != 2) {
    const float w = L.wt(0)[r];
    const int row = row0 + min(r, nrows - 1);
    const float* off = kind == 0 ? x.logit_offset_policy : (kind == 1 ? x.logit_offset_expert : (kind == 3 ? x.logit_offset_mix : nullptr));
    const float z = off ? f - off[row] : f;   // subtract_log_policy (models.py:175)
    const bool pu = d.loss_function == IL_LOSS_PUGAIL;
    if (pu_value_pass) {   // training.py:100-102 with a finite margin: V = prior mean(w_e bce(z_e, 0)) - mean(w_p bce(z_p, 0)) decides whether the clamped term has a gradient.
      // This launch only leaves the per-tile sums of w softplus(z) (= w bce(z, 0)); the real launch that follows reads them all and decides (every workgroup the same way).
      const float part = block_sum(sub == 0 && valid ? w * softplus_f(z) : 0.f, L.red);
      if (tid == 0) d.workspace[wsl.pu + (size_t)kind * nt + tile] = part;
      return;
    }
    float pu_on = 1.f;   // 1: the clamp passes the gradient (always, with nonnegative_margin = inf)
    if (pu && d.pu_clamped) {
      float se = 0.f, sp = 0.f;
      for (int t = 0; t < nt; ++t) { sp += d.workspace[wsl.pu + t]; se += d.workspace[wsl.pu + nt + t]; }
      const float V = d.pos_class_prior * (se / fB) - sp / fB;
      pu_on = V >= -d.nonnegative_margin ? 1.f : 0.f;   // torch.clamp(min = -margin): gradient where the input is not below the bound
    }
    // d loss / d z = w (c_sig sigmoid(z) - c_lab) / B: BCE {1, label}; PUGAIL policy {-1, 0}, expert {2 prior, prior} (clamped away: policy {0, 0}, expert {prior, prior}); Mixup {1, eps}
    const float c_sig = pu ? (kind == 1 ? (1.f + pu_on) * d.pos_class_prior : -pu_on) : 1.f;
    const float c_lab = kind == 3 ? mix_eps(row) : (kind == 1 ? (pu ? d.pos_class_prior : 1.f) : 0.f);
    const float p = sigmoid_f(z);
    float dz = valid ? w * (c_sig * p - c_lab) / fB : 0.f;
    if (d.entropy_bonus > 0.f && valid) dz += d.entropy_bonus * w * z * p * (1.f - p) / fB;
    if (sub == 0) { L.dzs[r] = dz; L.zs[r] = z; }
    float a1 = 0.f;
    for (int n = sub; n < H; n += 16) {
      const float h = L.hs[r * H + n];
      const float dh = h > 0.f ? dz * (L.W2s[n] / s2) : 0.f;
      L.dhs[r * H + n] = dh;
      a1 += dh * (h - L.b1s[n]);
    }
    ip1 = s1 * block_sum(a1, L.red);
    ip2 = s2 * block_sum(sub == 0 ? dz * (f - b2) : 0.f, L.red);
  } else {
    // q = [h>0] w2^ -> dhs ; g = q . W1^ (MFMA) ; cg = c g ; t' = cg . W1^^T (MFMA)
    for (int n = sub; n < H; n += 16) L.dhs[r * H + n] = L.hs[r * H + n] > 0.f ? (L.W2s[n] / s2) : 0.f;
    if (sub == 0) L.dzs[r] = valid ? 2.f * d.grad_penalty * L.wt(0)[r] / fB : 0.f;   // c_r
    __syncthreads();
    for (int t = wave; t < (Dp + 15) / 16; t += nw) {
      const int k = 16 * t + jj;
      const f32x4 c = mfma_lds_tile(L.dhs, H, 1, L.W1s + 16 * t, ldw, 1, H);
      if (k < Dp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.cg[(4 * gg + q) * Dp + k] = k < D ? L.dzs[4 * gg + q] * (c[q] / s1) : 0.f;
      }
    }
    __syncthreads();
    float S_ip = 0.f;
    for (int t = wave; t < H / 16; t += nw) {
      const int n = 16 * t + jj;
      const f32x4 c = mfma_lds_tile(L.cg, Dp, 1, L.W1s + 16 * t * ldw, 1, ldw, Dp);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 4 * gg + q; const float tp = c[q] / s1;
        L.ts[row * H + n] = L.hs[row * H + n] > 0.f ? tp : 0.f;
        S_ip += L.dhs[row * H + n] * tp;
      }
    }
    const float Ssum = block_sum(S_ip, L.red);
    ip1 = s1 * Ssum; ip2 = s2 * Ssum;
  }
  __syncthreads();
  IL_STAMP(stamp, 7);
  IL_TL(0, 5);
  // ---- this call's gradient slab:  G1^[n][k] = sum_r left[r][n] right[r][k]  on MFMA (reduction over the tile's 16 rows)
  const float* left = L.dhs;                         // [16][H]: dh (BCE) or q (GP)
  const float* right = kind != 2 ? X : L.cg;         // [16][Dp]: x (BCE) or c*g (GP)
  const float k1 = d.spectral_norm ? ip1 / (s1 * s1) : 0.f, k2 = d.spectral_norm ? ip2 / (s2 * s2) : 0.f;
  {
    const int nkt = (D + 15) / 16;
    for (int t = wave; t < (H / 16) * nkt; t += nw) {
      const int n0 = (t / nkt) * 16, k = (t % nkt) * 16 + jj;
      const f32x4 c = mfma_lds_tile(left + n0, 1, H, right + (t % nkt) * 16, Dp, 1, IL_TILE_R);
      if (k < D) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int n = n0 + 4 * gg + q; slab[lay.oW1 + (size_t)n * D + k] = c[q] / s1 - (d.spectral_norm ? k1 * u1[n] * v1[k] : 0.f); }
      }
    }
  }
  for (int n = tid; n < H; n += blockDim.x) {
    float g2 = 0.f, gb = 0.f;
    for (int rr = 0; rr < IL_TILE_R; ++rr) {
      if (kind != 2) { g2 += L.dzs[rr] * fmaxf(L.hs[rr * H + n], 0.f); gb += L.dhs[rr * H + n]; }
      else g2 += L.ts[rr * H + n];
    }
    slab[lay.oW2 + n] = g2 / s2 - (d.spectral_norm ? k2 * u2 * v2[n] : 0.f);
    slab[lay.ob1 + n] = gb;   // zero for the gradient-penalty call (no bias gradient)
  }
  if (tid == 0) {
    float gb2 = 0.f;
    if (kind != 2) for (int rr = 0; rr < IL_TILE_R; ++rr) gb2 += L.dzs[rr];
    slab[lay.ob2] = gb2;
  }
  IL_STAMP(stamp, 8);
  IL_TL(0, 6);
  if (!MULTI || ti + 1 >= tpw || tile + 1 >= nt) break;
  ++tile; row0 += IL_TILE_R; nrows = min(IL_TILE_R, B - row0);
  __syncthreads();   // every reader of the previous tile's rows, activations and loss terms is done
  stage_rows(0);
  __syncthreads();
  }
  if (tile0 == 0 && pass == npass - 1 && d.spectral_norm) {  // final u, v of this update: the last call's iteration
    float* o = d.workspace + wsl.sn_new;
    for (int i = tid; i < H; i += blockDim.x) { o[i] = L.u1(0)[i]; o[H + D + 1 + i] = L.v2(0)[i]; }
    for (int i = tid; i < D; i += blockDim.x) o[H + i] = L.v1(0)[i];
    if (tid == 0) o[H + D] = L.sc(0)[2];
  }
  IL_TL_END(0);
}

__global__ __launch_bounds__(256) void k_gail_grad(il_disc d, il_batch pol, il_batch exp, const float* __restrict__ eps_gp, il_gail_extra x, const il_disc* __restrict__ dL,
                                                   const il_batch* __restrict__ polL, const il_batch* __restrict__ expL, GailSampler sa, int pu_value_pass) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  IL_ST_BEGIN(IL_ST_GAIL_GRAD);
  gail_grad_body<false>(d, pol, exp, eps_gp, x, dL, polL, expL, sa, pu_value_pass, 1, smem);
  IL_ST_END(IL_ST_GAIL_GRAD);
}
__global__ __launch_bounds__(256) void k_gail_grad_pop(il_disc d, const il_disc* __restrict__ dL, const il_batch* __restrict__ polL, const il_batch* __restrict__ expL, int tpw) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gail_grad_body<true>(d, il_batch{}, il_batch{}, nullptr, il_gail_extra{}, dL, polL, expL, GailSampler{}, 0, tpw, smem);
}

__host__ __device__ inline int gail_calls(const il_disc& d) { return (d.loss_function == IL_LOSS_MIXUP ? 1 : 2) + (d.grad_penalty > 0.f ? 1 : 0); }

// grid = ceil(P / 256): one gradient element per thread, slabs summed in tile order (deterministic)
// close_epoch (il_gail_disc_step with IL_FLAG_GAIL_CLOSE_EPOCH): no relabel kernel follows on this stream - the stepped parameters are consumed by the
// critic-loss workgroups of k_sac_chain - so each workgroup reports [IL_SYNC_PARAMS] and the last one closes the side branch's epoch.
__global__ __launch_bounds__(256) void k_gail_reduce(il_disc d, int apply, const il_disc* __restrict__ dL, int close_epoch, il_peer_bucket peer) {
  IL_TL(1, 0);
  if (!dL) IL_ST_BEGIN(IL_ST_GAIL_REDUCE);
  if (dL) d = dL[blockIdx.y];
  globalize(d);
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, B = d.batch;
  const DiscLayout lay = disc_layout(D, H, d.spectral_norm);
  const DiscWs wsl = disc_ws(D, H, B);
  const int nt = ((B + IL_TILE_R - 1) / IL_TILE_R) * gail_calls(d);  // one slab per (call, tile)
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float pp = 0.f, mm = 0.f, vv = 0.f, g = 0.f;
  if (e < lay.P) {
    const float* sl = d.workspace + wsl.slabs + e;
    if (apply) { pp = d.params[e]; mm = d.opt.m[e]; vv = d.opt.v[e]; }   // independent of the slab sum: in flight while it runs
    int t = 0;
    for (; t + 24 <= nt; t += 24) {   // the slabs were written by other XCDs: every load is a full-latency miss, so keep 24 in flight (2 rounds at B = 256)
      float v[24];
#pragma unroll
      for (int u = 0; u < 24; ++u) v[u] = sl[(size_t)(t + u) * lay.P];
#pragma unroll
      for (int u = 0; u < 24; ++u) g += v[u];
    }
    for (; t + 8 <= nt; t += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = sl[(size_t)(t + u) * lay.P];
#pragma unroll
      for (int u = 0; u < 8; ++u) g += v[u];
    }
    for (; t < nt; ++t) g += sl[(size_t)t * lay.P];
  }
  if (peer.world > 0) {   // data-parallel: the gradient becomes its mean over the ranks inside this launch (peer_device.hpp; one arrival line per workgroup)
    const PeerJob pj = peer_job_begin(peer, (int)blockIdx.x);
    if (e < lay.P) peer_job_push1(peer, pj, e, g);
    peer_job_exchange(peer, pj, (int)blockIdx.x);
    if (e < lay.P) g = peer_job_mean1(peer, pj, e);
    peer_job_end(peer, pj, (int)blockIdx.x);
  }
  // an update whose hand-off expired (this launch's producers waited for rows / indices that never came) never reaches the weights: [IL_SYNC_POISON]. The step still signals.
  const bool poisoned = sync_poisoned(reinterpret_cast<const long long*>(d.sync));
  if (e < lay.P) {
    d.grad[e] = g;
    if (apply && !poisoned) {
      const adam_consts ac = load_adam_consts(d.opt);
      adam_update(pp, g, mm, vv, ac);
      // close_epoch: the stepped parameters are consumed by a launch of ANOTHER stream that is already resident (the inline relabel of k_sac_chain*): written THROUGH
      // (sc0 sc1: acknowledged once they are in memory) like every other in-launch hand-off of the schedule, and read below the caches there (disc_reward_tile<.., COH>)
      if (close_epoch) wstore1(d.params, e, pp); else d.params[e] = pp;
      d.opt.m[e] = mm; d.opt.v[e] = vv;
    }
  }
  if (blockIdx.x == 0 && d.spectral_norm && !poisoned) {
    const float* o = d.workspace + wsl.sn_new;
    if (close_epoch) {
      for (int i = threadIdx.x; i < H; i += blockDim.x) { wstore1(d.u1, i, o[i]); wstore1(d.v2, i, o[H + D + 1 + i]); }
      for (int i = threadIdx.x; i < D; i += blockDim.x) wstore1(d.v1, i, o[H + i]);
      if (threadIdx.x == 0) wstore1(d.u2, 0, o[H + D]);
    } else {
      for (int i = threadIdx.x; i < H; i += blockDim.x) { d.u1[i] = o[i]; d.v2[i] = o[H + D + 1 + i]; }
      for (int i = threadIdx.x; i < D; i += blockDim.x) d.v1[i] = o[H + i];
      if (threadIdx.x == 0) d.u2[0] = o[H + D];
    }
  }
  if (close_epoch && d.sync) {
    long long* sy = reinterpret_cast<long long*>(d.sync);
    sync_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      const long long done = __hip_atomic_fetch_add(sy + IL_SYNC_PARAMS, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (done % (long long)gridDim.x == 0) __hip_atomic_fetch_add(sy + IL_SYNC_SIDE_EPOCH, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  IL_TL(1, 7);
  if (!dL) IL_ST_END(IL_ST_GAIL_REDUCE);
}

__global__ __launch_bounds__(256) void k_gail_reward(il_disc d, il_batch b, float* __restrict__ out_r, float* __restrict__ out_logit, const float* __restrict__ logit_offset, const il_disc* __restrict__ dL,
                                                     const il_batch* __restrict__ bL, float* const* __restrict__ outL) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (dL) { d = dL[blockIdx.y]; b = bL[blockIdx.y]; out_r = outL[blockIdx.y]; out_logit = nullptr; }
  globalize(d); globalize(b); out_r = as_global(out_r);
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, Dp = (D + 3) & ~3;
  const int row0 = blockIdx.x * IL_TILE_R, tid = threadIdx.x, nrows = min(IL_TILE_R, b.n - row0);
  DiscLds L = carve(smem, D, H);
  IL_TL(2, 0);
  if (d.sync && !b.gather) {   // gathered rows: the discriminator step before this kernel may have run off the index draw alone, so their arrival is checked here
    long long* sy = reinterpret_cast<long long*>(d.sync);
    sync_wait(sy, IL_SYNC_ROWS, (sync_read(sy, IL_SYNC_SIDE_EPOCH) + 1) * sy[IL_SYNC_GATHER_WGS]);
  }
  {   // the tile's rows, four elements per thread and round: their indices requested together, then their rows (an index -> row chain per element made 2 x 2 serial round trips)
    const unsigned mdp = fastdiv_magic(Dp);
    for (int base = 0; base < IL_TILE_R * Dp; base += 4 * (int)blockDim.x) {
      size_t sr[4]; int kk[4]; bool ok[4]; float xv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = base + q * (int)blockDim.x + tid, ic = min(i, IL_TILE_R * Dp - 1), r = fastdiv(ic, mdp);
        kk[q] = ic - r * Dp; ok[q] = i < IL_TILE_R * Dp && r < nrows && kk[q] < D;
        sr[q] = brow(b, row0 + min(r, nrows - 1));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int k = min(kk[q], D - 1); xv[q] = gload(k < S ? b.states + sr[q] * b.ld_states + k : b.actions + sr[q] * b.ld_actions + (k - S)); }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = base + q * (int)blockDim.x + tid; if (i < IL_TILE_R * Dp) L.X(0)[i] = ok[q] ? xv[q] : 0.f; }
    }
  }
  IL_TL(2, 1);
  const RewardLds R = {L.W1s, L.b1s, L.W2s, L.u1(0), L.v1(0), L.v2(0), L.sc(0)};
  disc_reward_tile<6>(d, R, L.X(0), Dp, nrows, logit_offset, row0, [&](int r, float reward, float logit) {
    if (d.sync) wstore1(out_r, row0 + r, reward);   // consumed by the critic-loss workgroups of a resident launch of the other stream ([IL_SYNC_REWARDS]): written through, like the parameters in k_gail_reduce
    else out_r[row0 + r] = reward;
    if (out_logit) out_logit[row0 + r] = logit;
  });
  IL_TL_END(2);
  if (d.sync) {   // rewards of this tile are in place; the workgroup that completes the relabel closes the side branch's epoch
    long long* sy = reinterpret_cast<long long*>(d.sync);
    sync_drain_stores();
    __syncthreads();
    if (tid == 0) {
      const long long done = __hip_atomic_fetch_add(sy + IL_SYNC_REWARDS, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (done % (long long)gridDim.x == 0) __hip_atomic_fetch_add(sy + IL_SYNC_SIDE_EPOCH, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static int ensure_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return IL_OK;
  if (bytes > 160 * 1024) return il_set_error(IL_ERR_UNSUPPORTED, "kernel needs %zu bytes of LDS (> 160 KiB per CU)", bytes);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return IL_OK;
}

// Dynamic LDS of the single-learner k_gail_grad launches: at least 81 KB, so that its workgroups keep a CU each as they did with the round-3 layout (they are resident
// early and wait for the index draw; two of them per CU would share the issue slots of the preparation every one of them runs). The population launch asks for what
// the layout needs (two workgroups per CU).
static size_t disc_lds_single(int D, int H) { const size_t b = disc_lds_floats(D, H) * sizeof(float); return b < 81 * 1024 ? 81 * 1024 : b; }

static int check_disc(const il_disc* d) {
  IL_CHECK_ARG(d && d->params && d->grad && d->workspace, "il_disc: null descriptor field");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  IL_CHECK_ARG(d->hidden >= 16 && d->hidden <= 512 && d->hidden % 16 == 0 && D >= 1 && D <= 512, "il_disc: dims out of range (D=%d, hidden=%d: hidden must be a multiple of 16)", D, d->hidden);
  IL_CHECK_ARG(disc_lds_floats(D, d->hidden) * sizeof(float) <= 160 * 1024, "il_disc: D=%d hidden=%d needs more than 160 KiB of LDS", D, d->hidden);
  IL_CHECK_ARG(d->reward_function >= 0 && d->reward_function <= 2, "il_disc: reward_function must be 0 (AIRL), 1 (GAIL) or 2 (FAIRL)");
  IL_CHECK_ARG(d->loss_function >= 0 && d->loss_function <= 2, "il_disc: loss_function must be 0 (BCE), 1 (PUGAIL) or 2 (Mixup)");
  if (d->spectral_norm) IL_CHECK_ARG(d->u1 && d->v1 && d->u2 && d->v2, "il_disc: spectral-norm buffers missing");
  if (d->workspace_floats < disc_ws(D, d->hidden, d->batch).total) return il_set_error(IL_ERR_WORKSPACE, "il_disc: workspace too small");
  return IL_OK;
}

extern "C" int il_gail_disc_step(const il_disc* d, const il_batch* pol, const il_batch* exp, const float* eps_gp, const il_gail_extra* extra, uint32_t flags, il_stream_t stream_) {
  if (int rc = check_disc(d)) return rc;
  IL_CHECK_ARG(pol && exp && pol->n == d->batch && exp->n == d->batch, "il_gail_disc_step: policy/expert batches must both have %d rows", d->batch);
  il_gail_extra x = {};
  if (extra) x = *extra;
  IL_CHECK_ARG(d->loss_function != IL_LOSS_MIXUP || (!x.logit_offset_policy && !x.logit_offset_expert), "il_gail_disc_step: with Mixup the log-policy offset belongs to the mixed batch (logit_offset_mix)");
  hipStream_t st = (hipStream_t)stream_;
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const int nt = ceil_div(d->batch, IL_TILE_R);
  const size_t lds = disc_lds_single(D, d->hidden);
  if (int rc = ensure_lds((const void*)k_gail_grad, lds)) return rc;
  if (d->loss_function == IL_LOSS_PUGAIL && d->pu_clamped) {   // finite nonnegative_margin: a value pass (logits only) ahead of the gradient pass, which reads the clamp decision
    IL_CHECK_ARG(!d->sync, "il_gail_disc_step: PUGAIL with a finite nonnegative_margin runs on one stream (no il_sync hand-off)");
    IL_CHECK_ARG(d->nonnegative_margin >= 0.f, "il_gail_disc_step: nonnegative_margin must be >= 0");
    { IL_TRACE("k_gail_grad", st); k_gail_grad<<<dim3(nt, 2), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, nullptr, nullptr, nullptr, GailSampler{}, 1); }
  }
  { IL_TRACE("k_gail_grad", st); k_gail_grad<<<dim3(nt, gail_calls(*d)), 256, lds, st>>>(*d, *pol, *exp, eps_gp, x, nullptr, nullptr, nullptr, GailSampler{}, 0); }
  const int64_t P = disc_layout(D, d->hidden, d->spectral_norm).P;
  // (1 KB of LDS it never touches: a workgroup with no LDS can be placed on a CU whose LDS a pair-mode workgroup of the SAC branch holds entirely - sac.hip
  // IL_PAIR_LDS_BYTES - and its loads then queue behind that workgroup's weight stream: measured 4.95 -> 9.4 us for this launch, the relabel 4 us later)
  { IL_TRACE("k_gail_reduce", st); k_gail_reduce<<<(int)((P + 255) / 256), 256, 1024, st>>>(*d, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1, nullptr, (flags & IL_FLAG_GAIL_CLOSE_EPOCH) ? 1 : 0, il_peer_bucket{}); }
  IL_CHECK_LAUNCH("il_gail_disc_step");
  return IL_OK;
}

static int gail_disc_step_draw_impl(const il_disc* d, const il_batch* pol, const il_batch* exp, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                                    const int64_t* ring_state_b, int32_t* idx_b, uint32_t flags, const il_peer_bucket* peer, il_stream_t stream_, float* stage_rows = nullptr) {
  if (int rc = check_disc(d)) return rc;
  IL_CHECK_ARG(pol && exp && pol->n == d->batch && exp->n == d->batch && pol->gather && exp->gather, "il_gail_disc_step_draw: both batches must be rings read through il_batch.gather");
  IL_CHECK_ARG(d->sync && mt_state_dev && ring_state_a && idx_a && ring_state_b && idx_b, "il_gail_disc_step_draw: the il_sync counters, the generator state and both rings' states / index arrays are required");
  IL_CHECK_ARG(pol->gather == idx_a && exp->gather == idx_b, "il_gail_disc_step_draw: the batches must gather through the index arrays this call draws");
  hipStream_t st = (hipStream_t)stream_;
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const int nt = ceil_div(d->batch, IL_TILE_R);
  const size_t lds = disc_lds_single(D, d->hidden);
  IL_CHECK_ARG(lds >= sizeof(MtShared), "il_gail_disc_step_draw: discriminator too small to host the sampler's state in its workgroup LDS");
  if (int rc = ensure_lds((const void*)k_gail_grad, lds)) return rc;
  GailSampler sa = {mt_state_dev, ring_state_a, idx_a, ring_state_b, idx_b, d->batch, MtStage{}};
  if (stage_rows) {   // the policy batch IS the agent ring read through idx_a: its packed rows start at `states`
    IL_CHECK_ARG(pol->ld_states % 4 == 0 && (reinterpret_cast<uintptr_t>(pol->states) & 15) == 0 && (reinterpret_cast<uintptr_t>(stage_rows) & 15) == 0, "il_gail_disc_step_draw_staged: packed rows of whole 16-byte lanes");
    sa.stage = MtStage{pol->states, stage_rows, (long long)pol->gather_capacity, pol->ld_states / 4};
  }
  { IL_TRACE("k_gail_grad", st); k_gail_grad<<<dim3(nt + 1, gail_calls(*d)), 256, lds, st>>>(*d, *pol, *exp, nullptr, il_gail_extra{}, nullptr, nullptr, nullptr, sa, 0); }
  const int64_t P = disc_layout(D, d->hidden, d->spectral_norm).P;
  il_peer_bucket px = {};
  if (peer) {
    IL_CHECK_ARG(!(flags & IL_FLAG_GRADS_ONLY), "il_gail_disc_step_draw_peer: the AdamW step runs inside (no IL_FLAG_GRADS_ONLY)");
    IL_CHECK_ARG(peer->world >= 1 && peer->world <= IL_PEER_MAX_RANKS && peer->rank >= 0 && peer->rank < peer->world && peer->epoch && peer->status && (peer->window_offset & 255) == 0, "il_gail_disc_step_draw_peer: bad peer descriptor");
    IL_CHECK_ARG(peer->n == P && peer->n_jobs >= (int)((P + 255) / 256), "il_gail_disc_step_draw_peer: the bucket must hold %lld floats and %d arrival lines (got %lld, %d)", (long long)P, (int)((P + 255) / 256), (long long)peer->n, peer->n_jobs);
    for (int r = 0; r < peer->world; ++r) IL_CHECK_ARG(peer->windows[r], "il_gail_disc_step_draw_peer: window of rank %d is not mapped", r);
    px = *peer;
  }
  { IL_TRACE("k_gail_reduce", st); k_gail_reduce<<<(int)((P + 255) / 256), 256, 1024, st>>>(*d, (flags & IL_FLAG_GRADS_ONLY) ? 0 : 1, nullptr, (flags & IL_FLAG_GAIL_CLOSE_EPOCH) ? 1 : 0, px); }
  IL_CHECK_LAUNCH("il_gail_disc_step_draw");
  return IL_OK;
}
extern "C" int il_gail_disc_step_draw(const il_disc* d, const il_batch* pol, const il_batch* exp, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                                      const int64_t* ring_state_b, int32_t* idx_b, uint32_t flags, il_stream_t stream_) {
  return gail_disc_step_draw_impl(d, pol, exp, mt_state_dev, ring_state_a, idx_a, ring_state_b, idx_b, flags, nullptr, stream_);
}
extern "C" int il_gail_disc_step_draw_staged(const il_disc* d, const il_batch* pol, const il_batch* exp, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                                             const int64_t* ring_state_b, int32_t* idx_b, float* stage_rows, uint32_t flags, il_stream_t stream_) {
  IL_CHECK_ARG(stage_rows, "il_gail_disc_step_draw_staged: null staging slab");
  return gail_disc_step_draw_impl(d, pol, exp, mt_state_dev, ring_state_a, idx_a, ring_state_b, idx_b, flags, nullptr, stream_, stage_rows);
}
extern "C" int il_gail_disc_step_draw_peer(const il_disc* d, const il_batch* pol, const il_batch* exp, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                                           const int64_t* ring_state_b, int32_t* idx_b, uint32_t flags, const il_peer_bucket* peer, il_stream_t stream_) {
  IL_CHECK_ARG(peer, "il_gail_disc_step_draw_peer: null peer descriptor");
  return gail_disc_step_draw_impl(d, pol, exp, mt_state_dev, ring_state_a, idx_a, ring_state_b, idx_b, flags, peer, stream_);
}

// population axis: discriminator step + reward relabel of n_learners independent discriminators (same shapes) in three launches
extern "C" int il_gail_step_population(const il_disc* descs_dev, const il_batch* policy_dev, const il_batch* expert_dev, float* const* rewards_out_dev, int32_t n_learners,
                                       const il_disc* shape_host, il_stream_t stream_) {
  if (int rc = check_disc(shape_host)) return rc;
  IL_CHECK_ARG(descs_dev && policy_dev && expert_dev && rewards_out_dev && n_learners >= 1 && n_learners <= 65535, "il_gail_step_population: bad arguments");
  IL_CHECK_ARG(shape_host->loss_function != IL_LOSS_MIXUP, "il_gail_step_population: loss_function Mixup is not available on the population path");
  const il_disc* d = shape_host;
  hipStream_t st = (hipStream_t)stream_;
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim), nt = ceil_div(d->batch, IL_TILE_R), L = n_learners;
  static const int compact = [] { const char* e = getenv("IL_POP_DISC_LDS"); return e && e[0] == '0' ? 0 : 1; }();   // IL_POP_DISC_LDS=0: one workgroup per CU, as in round 3 (A/B)
  const size_t need = disc_lds_floats(D, d->hidden) * sizeof(float);
  const size_t lds = compact || need > (size_t)96 * 1024 ? need : (size_t)96 * 1024, lds_r = compact ? disc_reward_lds_floats(D, d->hidden) * sizeof(float) : lds;
  if (int rc = ensure_lds((const void*)k_gail_grad_pop, lds)) return rc;
  if (int rc = ensure_lds((const void*)k_gail_reward, lds_r)) return rc;
  const int64_t P = disc_layout(D, d->hidden, d->spectral_norm).P;
  il_batch zb = {};
  // tiles per workgroup of the gradient launch (k_gail_grad `tpw`): IL_POP_DISC_TPW, default 4
  static const int tpw_env = [] { const char* e = getenv("IL_POP_DISC_TPW"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
  const int tpw = tpw_env < nt ? tpw_env : nt;
  { IL_TRACE("k_gail_grad", st); k_gail_grad_pop<<<dim3(ceil_div(nt, tpw), gail_calls(*d), L), 256, lds, st>>>(*d, descs_dev, policy_dev, expert_dev, tpw); }
  { IL_TRACE("k_gail_reduce", st); k_gail_reduce<<<dim3((int)((P + 255) / 256), L), 256, 0, st>>>(*d, 1, descs_dev, 0, il_peer_bucket{}); }
  { IL_TRACE("k_gail_reward", st); k_gail_reward<<<dim3(nt, L), 256, lds_r, st>>>(*d, zb, nullptr, nullptr, nullptr, descs_dev, policy_dev, rewards_out_dev); }
  IL_CHECK_LAUNCH("il_gail_step_population");
  return IL_OK;
}

__global__ __launch_bounds__(256) void k_disc_adam(il_disc d, int64_t P) {
  const adam_consts ac = load_adam_consts(d.opt);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (int64_t)gridDim.x * blockDim.x) {
    float pp = d.params[e], mm = d.opt.m[e], vv = d.opt.v[e];
    adam_update(pp, d.grad[e], mm, vv, ac);
    if (d.sync) wstore1(d.params, e, pp); else d.params[e] = pp;   // (written through where the inline relabel of a resident launch consumes them: k_gail_reduce)
    d.opt.m[e] = mm; d.opt.v[e] = vv;
  }
  if (d.sync) {   // data-parallel schedule with the device-side hand-off: this is the discriminator branch's last kernel (cf. k_gail_reduce with close_epoch): the inline relabel of
    long long* sy = reinterpret_cast<long long*>(d.sync);   // il_sac_update_gather waits for [IL_SYNC_PARAMS]; the last workgroup closes the branch's epoch
    sync_drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      const long long done = __hip_atomic_fetch_add(sy + IL_SYNC_PARAMS, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (done % (long long)gridDim.x == 0) __hip_atomic_fetch_add(sy + IL_SYNC_SIDE_EPOCH, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
extern "C" int il_gail_apply_grads(const il_disc* d, il_stream_t stream_) {
  if (int rc = check_disc(d)) return rc;
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const int64_t P = disc_layout(D, d->hidden, d->spectral_norm).P;
  { IL_TRACE("k_disc_adam", (hipStream_t)stream_); k_disc_adam<<<(int)((P + 255) / 256), 256, 0, (hipStream_t)stream_>>>(*d, P); }
  IL_CHECK_LAUNCH("il_gail_apply_grads");
  return IL_OK;
}

extern "C" int il_gail_reward(const il_disc* d, const il_batch* b, float* out_rewards, float* out_logits, const float* logit_offset, il_stream_t stream_) {
  if (int rc = check_disc(d)) return rc;
  IL_CHECK_ARG(b && out_rewards && b->n > 0, "il_gail_reward: bad arguments");
  const int D = d->state_dim + (d->state_only ? 0 : d->action_dim);
  const size_t lds = disc_reward_lds_floats(D, d->hidden) * sizeof(float);
  if (int rc = ensure_lds((const void*)k_gail_reward, lds)) return rc;
  { IL_TRACE("k_gail_reward", stream_); k_gail_reward<<<ceil_div(b->n, IL_TILE_R), 256, lds, (hipStream_t)stream_>>>(*d, *b, out_rewards, out_logits, logit_offset, nullptr, nullptr, nullptr); }
  IL_CHECK_LAUNCH("il_gail_reward");
  return IL_OK;
}

IL_STAMP_READER(il_debug_stamps_gail)
IL_TL_READER(il_debug_timeline_gail)
IL_ST_READER(il_stamps_gail)
