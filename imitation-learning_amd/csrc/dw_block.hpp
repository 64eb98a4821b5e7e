// The block form of the weight-gradient + AdamW step (moved out of sac.hip in round 6 so that the general-shape tile engine of general.hip can launch the same block jobs):
// one workgroup = a 32 x 32 block of ANY layer's dW with the layer's bias folded into the blocks of its first k-column, operands staged through LDS, the optimiser (and the
// lane-ordered copies of an H x H layer) in the epilogue. See the comment above dw_block32.
#pragma once
#include "il_common.hpp"
#include "mlp_tile.hpp"
#include "peer_device.hpp"

// How the optimiser epilogues store p / m / v and the lane-ordered copies (IL_DW_STORE_MODE): 0 = plain stores (the lines sit dirty in this XCD's L2 until the
// end-of-kernel write-back, which is on the critical path of the following launch boundary), 1 = `nt` (streaming) stores, 2 = `sc0 sc1` write-through stores (the data
// leaves for memory while the kernel still runs; nothing of it is left to flush). m and v are not read again before the next update, p only by other XCDs.
#ifndef IL_POLYAK_WT
#define IL_POLYAK_WT 1
#endif
#ifndef IL_DW_STORE_MODE
#define IL_DW_STORE_MODE 2
#endif
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dw_store4(float* base, int64_t off, const f32x4& v) {
#if IL_DW_STORE_MODE == 1
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + off));
#elif IL_DW_STORE_MODE == 2
  wstore4(base, off, v);
#else
  *reinterpret_cast<f32x4*>(base + off) = v;
#endif
}
struct DwArgs {
  float* params; float* grads; il_adam opt; int grads_only;
  int n_nets; int64_t net_stride;
  int in_dim, hidden, out_dim, batch;
  const float* x0; int ld_x0; int x0_transposed; int64_t x0_net_stride;   // layer-1 input: [in][B] (transposed) or row-major [B][ld_x0]
  const float* h1; const float* h2; const float* dz1; const float* dz2; int64_t h_net_stride;   // [H][B]
  const float* dz3; int64_t dz3_net_stride;                                // [out][B]
  float* pk_f; float* pk_b;                                                 // lane-ordered copies of W2 kept in step with the AdamW update (NULL: none)
  int n_dw_blocks;
  int jobs_per_block;   // wave-per-tile jobs per 256-thread workgroup (0 = 4): with 32 KB of half-line operand gathers per job, four jobs on one CU take 3.8 us of texture-address time - the single learner has CUs to spare and runs two
  int n_big_blocks;   // single learner: leading workgroups that each own a 32 x 32 block of an H x H layer's dW (dw_block32); 0 = wave-per-tile jobs for every layer
  // tail
  float* log_alpha; float* alpha_grad; il_adam alpha_opt; const float* alpha_part; int n_alpha_part;
  float* target; const float* polyak_src; int64_t polyak_n; double tau; uint32_t* noise_counter; int64_t* sync;
  float* pk_target; const float* pk_critic; int64_t pk_n;   // lane-ordered copies of the target / critic hidden layers (polyak is elementwise, so it commutes with the re-ordering)
  // (round 5, population; +2.2 % on the population line, profiles/r05_pop_dw_ab.txt) the target step of the critics' H x H layers folded into their optimiser pass: `fuse_polyak` (critic launch: dw_block64 also
  // writes target / pk_target from the new parameters in its registers); `polyak_fused` (actor launch's tail: skips the two networks' H x H ranges
  // and the lane-ordered copies, and steps the rest of the arena element by element at the ranges' edges)
  int fuse_polyak, polyak_fused;
  // (round 6) overlapped launches (il_sac_update_gather_overlap): 0 = off; otherwise 1 + this launch's stage (IL_OV_DWC / IL_OV_DWA). The launch is resident while the
  // launch that produces its dZ / activations (stage - 1, on the other stream) still runs: block jobs request their p / m / v lanes, the tail's target step runs, and only
  // then they wait for [IL_SYNC_OV_EPOCH + stage - 1] > own epoch. `sync` is set for both launches in this mode.
  int ov_stage, ov_grid;   // ov_grid: the grid size of the launch this one waits for
};

// data-parallel: the gradient exchange of an optimiser step rides in its block jobs (peer_device.hpp "the exchange INSIDE the kernel that produces the gradients"): job = the
// block job's index (the log-alpha step: n_big_blocks). A kernel argument of its own, and a kernel of its own (k_dw_adam_peer: the functions below are templated on PEER):
// the descriptor's window array is indexed at run time, which sends whatever struct holds it to scratch memory - inside DwArgs that cost the single-GPU k_dw_adam
// 592 bytes of scratch per lane and 4 us per launch (measured, round 3).
struct DwPeer { il_peer_bucket x; int64_t alpha_at; };

// ---------------------------------------------------------------------------------------------
// Single learner (round 3): one WORKGROUP = a 32(n) x 32(k) block of an H x H layer's dW (+ AdamW), operands staged through LDS.
// What bounded the wave-per-tile form was not latency but the texture-address rate: lane (j, g) of a tile's operand load reads 16 bytes of feature j, so ONE wave
// instruction touches 16 half-used 128-byte lines (14 B/clk/CU measured for this pattern, mlp_tile.hpp) and a CU with four tiles pulls 4 x 32 KB through it: 9.4k
// clocks = 3.9 us - the duration of the launch, whatever the schedule of the loads (round 3 A/B: all operand lanes ahead of the MFMAs, 16 lanes per operand in flight:
// no change; the same jobs at 16 per CU inside k_sac_chain: 4x longer). Here the 256 threads of a workgroup fetch the two [32 features][B] panels with every wave
// instruction covering whole lines (64 lanes = 2 features x 512 contiguous bytes), 16 KB per tile instead of 32 KB, park them in LDS in two 128-row chunks (all
// loads of both chunks are requested up front, with the block's p / m / v lanes), and each wave runs ONE 16 x 16 tile out of LDS with dw_tile's accumulators and MFMA
// order (row groups ascending; k-steps 0, 2 -> acc0 and 1, 3 -> acc1): same bits. The epilogue goes through LDS like dw_block64's: AdamW row-wise on 16-byte lanes
// (128-byte row segments per 8 threads instead of 4-byte pieces of 16 rows), the updated block once more for the column-wise lane order of the PB copy.
// ---------------------------------------------------------------------------------------------
#define DWS 32
#define DWS_ROWS 128
#define DWS_LD (DWS_ROWS + 4)
#define DWS_GLD (DWS + 4)
// General form (round 3, second step): ANY layer's dW - dZ [Nvalid][B] x X [Kvalid][B], both feature-major - as 32 x 32 blocks, partial blocks included (features past
// the matrix clamp their address; their rows / columns are masked in the epilogue), and the layer's BIAS gradient folded into the blocks of the first k-column: the dZ
// panel is in LDS anyway, 128 threads sum it with dw_bias's exact order (lane (f, g): rows 16 i + 4 g .. + 3 for ascending i as one 4-vector, (s0 + s1) + (s2 + s3),
// then the four g as (g0 + g1) + (g2 + g3)), so the bias keeps its bits too. With it a network's whole optimiser step is nbh^2 + 2 nbh uniform workgroups (H = 256: 80)
// and no wave-per-tile job is left: those were 72 % of the launch's line requests (32 KB of half-line gathers each).
// `boff` >= 0: parameter offset of the bias of this dZ (only looked at by blocks with k0 == 0).
// Gate: called by every thread once the block's p / m / v lanes are requested and before its operands are; true = leave without storing anything (sac.hip's overlapped
// launches wait there for the launch that produces dZ, and a poisoned learner stops there)
struct DwNoGate { __device__ __forceinline__ bool operator()(const DwArgs&) const { return false; } };
template <bool PEER = false, class Gate = DwNoGate>
__device__ __forceinline__ void dw_block32(const DwArgs& a, const float* __restrict__ dzT, int Nvalid, const float* __restrict__ xT, int Kvalid, int n0, int k0, int64_t poff,
                                           int64_t boff, float* __restrict__ pkf, float* __restrict__ pkb, float* smem, const DwPeer* pp = nullptr, int pjob = -1) {
  float* Zs = smem; float* Xs = smem + DWS * DWS_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int ti = wave >> 1, tq = wave & 1;   // this wave's tile of the block
  const int B = a.batch;
  // staging map: thread t moves the 16-byte lane (feature t / 32 + 8 u, rows 4 (t % 32) .. +3) of a 128-row chunk of both panels, u = 0..3
  const int sf = tid >> 5, sr = (tid & 31) * 4;
  // epilogue map: thread t owns the 16-byte lane (row t / 8, columns 4 (t % 8) .. +3) of the block
  const int er = tid >> 3, ec = (tid & 7) * 4;
  const int en = n0 + er, ek = k0 + ec;
  const bool full = n0 + DWS <= Nvalid && k0 + DWS <= Kvalid && (Kvalid & 3) == 0 && (poff & 3) == 0;   // whole block inside the matrix, rows 16-byte aligned: 16-byte lanes
  // (round 4) A first-layer block (all Kvalid <= 32 columns of 32 whole rows) is not `full` - its rows are narrower than the block, and 18 floats wide they are not even
  // 16-byte aligned - and took the element-wise path below: twelve dword loads and twelve plain dword stores per thread, 0.9 us behind the H x H blocks at the end of BOTH
  // optimiser launches (profiles/tools/dw_stragglers.py). But its parameters are ONE contiguous run of 32 Kvalid floats: `flat` treats it as 8 Kvalid 16-byte lanes (one per
  // thread), gathers each lane's four gradients out of the LDS block, and stores write-through like a full block. Same gradients, same AdamW: same bits.
  // Last-layer blocks (OUT < 32 rows of H columns) are whole 16-byte lanes too, only fewer rows: `rowg` keeps the lane form with a row guard.
  const bool rowg = !full && k0 + DWS <= Kvalid && (Kvalid & 3) == 0 && (poff & 3) == 0 && n0 < Nvalid;
  const bool flat = !full && !rowg && k0 == 0 && Kvalid <= DWS && n0 + DWS <= Nvalid && ((poff + (int64_t)n0 * Kvalid) & 3) == 0 && ((DWS * Kvalid) & 3) == 0;
  const bool flat_on = flat && tid < DWS * Kvalid / 4;
  const int64_t fo = poff + (int64_t)n0 * Kvalid + 4 * (int64_t)min(tid, DWS * Kvalid / 4 - 1);
  const int64_t eo = flat ? fo : poff + (int64_t)min(en, Nvalid - 1) * Kvalid + min(ek, Kvalid - 1);
  f32x4 pv = zero4(), mv = zero4(), vv = zero4();
  if (!a.grads_only) {   // HBM (last touched an update ago): requested before anything else
    if (full || flat || rowg) { pv = gload4(a.params + eo); mv = gload4(a.opt.m + eo); vv = gload4(a.opt.v + eo); }   // (rowg: rows past the matrix clamp to its last row and are not stored)
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) { const int64_t o = poff + (int64_t)min(en, Nvalid - 1) * Kvalid + min(ek + c, Kvalid - 1); pv[c] = gload(a.params + o); mv[c] = gload(a.opt.m + o); vv[c] = gload(a.opt.v + o); }
    }
  }
  const bool do_bias = boff >= 0 && k0 == 0;
  const int bf = tid >> 2, bg = tid & 3;   // bias: threads 0..127 = (feature, row group of 4)
  f32x4 bs4 = zero4();
  const bool bias_owner = do_bias && tid < 128 && bg == 0 && n0 + bf < Nvalid;
  float bpp = 0.f, bmm = 0.f, bvv = 0.f;
  if (bias_owner && !a.grads_only) { const int64_t o = boff + n0 + bf; bpp = gload(a.params + o); bmm = gload(a.opt.m + o); bvv = gload(a.opt.v + o); }   // the bias's Adam operands: with the block's, up front
  if (Gate()(a)) return;
  f32x4 acc0 = zero4(), acc1 = zero4();
  for (int r0 = 0; r0 < B; r0 += 2 * DWS_ROWS) {   // two chunks per trip, all their loads in flight together (B = 256: one trip)
    f32x4 zr[2][4], xr[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = min(r0 + c * DWS_ROWS, B - DWS_ROWS) + sr;   // (B % 256 == 128: the second chunk of the last trip re-reads the first and is not used)
        zr[c][u] = gload4(dzT + (size_t)min(n0 + sf + 8 * u, Nvalid - 1) * B + rr); xr[c][u] = gload4(xT + (size_t)min(k0 + sf + 8 * u, Kvalid - 1) * B + rr);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (r0 + c * DWS_ROWS >= B) break;
      __syncthreads();   // the previous chunk's readers are done
#pragma unroll
      for (int u = 0; u < 4; ++u) { *reinterpret_cast<f32x4*>(Zs + (sf + 8 * u) * DWS_LD + sr) = zr[c][u]; *reinterpret_cast<f32x4*>(Xs + (sf + 8 * u) * DWS_LD + sr) = xr[c][u]; }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < DWS_ROWS / 16; ++u) {   // 16-row groups in ascending order, like dw_tile
        const f32x4 av = *reinterpret_cast<const f32x4*>(Zs + (16 * ti + j) * DWS_LD + 16 * u + 4 * g), bv = *reinterpret_cast<const f32x4*>(Xs + (16 * tq + j) * DWS_LD + 16 * u + 4 * g);
        acc0 = mfma16(av[0], bv[0], acc0);
        acc1 = mfma16(av[1], bv[1], acc1);
        acc0 = mfma16(av[2], bv[2], acc0);
        acc1 = mfma16(av[3], bv[3], acc1);
      }
      if (do_bias && tid < 128) {
#pragma unroll
        for (int u = 0; u < DWS_ROWS / 16; ++u) bs4 += *reinterpret_cast<const f32x4*>(Zs + bf * DWS_LD + 16 * u + 4 * bg);
      }
    }
  }
  float* Gs = Zs;   // [32][DWS_GLD] gradient block, then the updated parameters
  __syncthreads();
  IL_TL(a.log_alpha ? 2 : 1, 1);   // products done
  {
    const f32x4 t = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) Gs[(16 * ti + 4 * g + r) * DWS_GLD + 16 * tq + j] = t[r];
  }
  adam_consts ac = {};
  if (!a.grads_only) ac = load_adam_consts(a.opt);
  float bsum = 0.f;
  if (do_bias && tid < 128) {   // (all 64 lanes of waves 0, 1 take part in the shuffles)
    bsum = (bs4[0] + bs4[1]) + (bs4[2] + bs4[3]);
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
  }
  __syncthreads();
  f32x4 gv = *reinterpret_cast<const f32x4*>(Gs + er * DWS_GLD + ec);
  if (flat) {   // this thread's lane of the contiguous run: elements 4 tid .. + 3 = (row e / Kvalid, column e % Kvalid) of the block
    const unsigned mk = fastdiv_magic(Kvalid);
    const int e0 = 4 * min(tid, DWS * Kvalid / 4 - 1);
    int row = fastdiv(e0, mk), col = e0 - row * Kvalid;
#pragma unroll
    for (int c = 0; c < 4; ++c) { gv[c] = Gs[row * DWS_GLD + col]; if (++col == Kvalid) { col = 0; ++row; } }
  }
  if (PEER) {   // data-parallel: this block's gradients (and bias gradients) become their mean over the ranks before the optimiser sees them
    const il_peer_bucket& x = pp->x;
    const PeerJob pj = peer_job_begin(x, pjob);
    if (full || flat_on || (rowg && en < Nvalid)) peer_job_push4(x, pj, eo, gv);
    else if (flat || rowg) { }
    else if (en < Nvalid) {
#pragma unroll
      for (int c = 0; c < 4; ++c) if (ek + c < Kvalid) peer_job_push1(x, pj, poff + (int64_t)en * Kvalid + ek + c, gv[c]);
    }
    if (bias_owner) peer_job_push1(x, pj, boff + n0 + bf, bsum);
    peer_job_exchange(x, pj, pjob);
    if (full || flat_on || (rowg && en < Nvalid)) gv = peer_job_mean4(x, pj, eo);
    else if (flat || rowg) { }
    else if (en < Nvalid) {
#pragma unroll
      for (int c = 0; c < 4; ++c) if (ek + c < Kvalid) gv[c] = peer_job_mean1(x, pj, poff + (int64_t)en * Kvalid + ek + c);
    }
    if (bias_owner) bsum = peer_job_mean1(x, pj, boff + n0 + bf);
    peer_job_end(x, pj, pjob);
  }
  if (bias_owner) {
    const int64_t o = boff + n0 + bf;
    if (a.grads_only) a.grads[o] = bsum;
    else { adam_update(bpp, bsum, bmm, bvv, ac); a.params[o] = bpp; a.opt.m[o] = bmm; a.opt.v[o] = bvv; }
  }
  if (flat || rowg) {
    if (flat ? !flat_on : en >= Nvalid) return;
    if (a.grads_only) { *reinterpret_cast<f32x4*>(a.grads + eo) = gv; return; }
#pragma unroll
    for (int c = 0; c < 4; ++c) { float pp_ = pv[c], mm = mv[c], v2 = vv[c]; adam_update(pp_, gv[c], mm, v2, ac); pv[c] = pp_; mv[c] = mm; vv[c] = v2; }
    dw_store4(a.params, eo, pv); dw_store4(a.opt.m, eo, mv); dw_store4(a.opt.v, eo, vv);
    return;
  }
  if (!full) {   // partial block (last layer; first layers wider than a block): element-wise with guards
    if (en < Nvalid) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (ek + c >= Kvalid) continue;
        const int64_t o = poff + (int64_t)en * Kvalid + ek + c;
        if (a.grads_only) { a.grads[o] = gv[c]; continue; }
        float pp = pv[c], mm = mv[c], v2 = vv[c];
        adam_update(pp, gv[c], mm, v2, ac);
        a.params[o] = pp; a.opt.m[o] = mm; a.opt.v[o] = v2;
      }
    }
    return;
  }
  if (a.grads_only) { *reinterpret_cast<f32x4*>(a.grads + eo) = gv; return; }
#pragma unroll
  for (int c = 0; c < 4; ++c) { float pp = pv[c], mm = mv[c], v2 = vv[c]; adam_update(pp, gv[c], mm, v2, ac); pv[c] = pp; mv[c] = mm; vv[c] = v2; }
  dw_store4(a.params, eo, pv); dw_store4(a.opt.m, eo, mv); dw_store4(a.opt.v, eo, vv);
  IL_TL(a.log_alpha ? 2 : 1, 2);   // AdamW stores issued
  if (!pkf) return;
  dw_store4(pkf, (int64_t)packed_fwd_index(en, ek, Kvalid), pv);   // k .. k+3 of row n: one 16-byte lane of PF
  *reinterpret_cast<f32x4*>(Gs + er * DWS_GLD + ec) = pv;
  __syncthreads();
  {  // PB: rows n .. n+3 of column k are one 16-byte lane; thread t takes column t % 32 and row quad t / 32
    const int kc = tid & 31, rq = tid >> 5;
    f32x4 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = Gs[(4 * rq + r) * DWS_GLD + kc];
    dw_store4(pkb, (int64_t)packed_bwd_index(n0 + 4 * rq, k0 + kc, Kvalid), w);
  }
}
