// Pieces of the GAIL discriminator shared by gail.hip and sac.hip: parameter layout, the one-wave spectral-norm step, and the eval-mode forward +
// reward head of one 16-row tile (models.py:152-180). k_gail_reward and the critic-loss workgroups of k_sac_chain (inline relabel) run the SAME
// function with the same thread mapping (the first 256 threads of the workgroup), so the rewards are bit-identical on both paths.
#pragma once
#include "il_common.hpp"

struct DiscLayout { int64_t oW1, ob1, oW2, ob2, P; };
__host__ __device__ inline DiscLayout disc_layout(int D, int H, int sn) {
  DiscLayout l;
  if (sn) { l.ob1 = 0; l.oW1 = H; l.ob2 = H + (int64_t)H * D; l.oW2 = l.ob2 + 1; }
  else { l.oW1 = 0; l.ob1 = (int64_t)H * D; l.oW2 = l.ob1 + H; l.ob2 = l.oW2 + H; }
  l.P = (int64_t)H * D + 2 * H + 1;
  return l;
}
// LDS dot products with several loads in flight (a plain `for k: s += a[k]*b[k]` waits ~100 cycles per LDS read)
__device__ __forceinline__ float dot4(const float* a, const float* b, int n4) {  // both 16-B aligned, n4 % 4 == 0
  f32x4 s0 = zero4(), s1 = zero4();
  int k = 0;
  for (; k + 8 <= n4; k += 8) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(a + k), b0 = *reinterpret_cast<const f32x4*>(b + k);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(a + k + 4), b1 = *reinterpret_cast<const f32x4*>(b + k + 4);
    s0 += a0 * b0; s1 += a1 * b1;
  }
  if (k < n4) s0 += *reinterpret_cast<const f32x4*>(a + k) * *reinterpret_cast<const f32x4*>(b + k);
  s0 += s1;
  return (s0[0] + s0[1]) + (s0[2] + s0[3]);
}
__device__ __forceinline__ float dot_strided(const float* a, const float* b, int bstride, int n) {  // a contiguous (16-B aligned), b[i*bstride], n % 4 == 0
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int i = 0; i < n; i += 4) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + i);
    s0 += av[0] * b[(i + 0) * bstride]; s1 += av[1] * b[(i + 1) * bstride]; s2 += av[2] * b[(i + 2) * bstride]; s3 += av[3] * b[(i + 3) * bstride];
  }
  return (s0 + s1) + (s2 + s3);
}

// wave-synchronous LDS hand-off between lanes of ONE wave: LDS ops of a wave execute in order, this only pins the compiler
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__device__ __forceinline__ float wave_norm_scale(float ss) { return 1.f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f); }

// one wave: (u1,v1,u2,v2) <- one power iteration (if iterate), then sigmas. in/out vectors live in LDS.
__device__ __forceinline__ void sn_wave(const float* W1s, const float* W2s, int D, int H, float* u1, float* v1, float* u2, float* v2, bool iterate, float* sig) {
  const int lane = threadIdx.x & 63, Dp = (D + 3) & ~3, ldw = Dp + 4;
  if (iterate) {
    float ss = 0.f;
    for (int n = lane; n < H; n += 64) { const float s = dot4(W1s + n * ldw, v1, Dp); u1[n] = s; ss += s * s; }
    float inv = wave_norm_scale(ss);
    for (int n = lane; n < H; n += 64) u1[n] *= inv;
    WAVE_SYNC();
    ss = 0.f;
    for (int k = lane; k < D; k += 64) { const float s = dot_strided(u1, W1s + k, ldw, H); v1[k] = s; ss += s * s; }
    inv = wave_norm_scale(ss);
    for (int k = lane; k < D; k += 64) v1[k] *= inv;
    WAVE_SYNC();
    float p = 0.f;
    for (int n = lane; n < H; n += 64) p += W2s[n] * v2[n];
    p = wave_sum(p);
    const float uu = p / fmaxf(fabsf(p), 1e-12f);
    ss = 0.f;
    for (int n = lane; n < H; n += 64) { const float s = W2s[n] * uu; v2[n] = s; ss += s * s; }
    inv = wave_norm_scale(ss);
    for (int n = lane; n < H; n += 64) v2[n] *= inv;
    if (lane == 0) u2[0] = uu;
    WAVE_SYNC();
  }
  float a = 0.f, b = 0.f;
  for (int n = lane; n < H; n += 64) { a += u1[n] * dot4(W1s + n * ldw, v1, Dp); b += W2s[n] * v2[n]; }
  a = wave_sum(a); b = wave_sum(b);
  if (lane == 0) { sig[0] = a; sig[1] = u2[0] * b; }
  WAVE_SYNC();
}


// W1 [H][D] (contiguous in the flat parameter vector) -> LDS rows of Dp + 4 floats, as 16-byte lanes of the FLAT array with every load of a round requested before the first
// LDS store (the rows are D floats long - 23 at HalfCheetah dims - so a row-wise copy is a dword per load, and a load -> store loop makes one fabric round trip per
// element a thread owns: 24 of them in a 256-thread workgroup, 7.8 us of a 21 us k_gail_grad workgroup on the population path, profiles/r04_population_disc_timeline.txt).
// NV = 16-byte lanes per thread per round. The caller zeroes the padding columns [D, Dp). Same values in the same places: bit-identical to the element loop.
template <int NV> struct DiscW1 { f32x4 v[NV]; };
__device__ __forceinline__ bool disc_w1_flat_ok(const float* W1, int D, int H) { return ((reinterpret_cast<uintptr_t>(W1) & 15) == 0) && ((H * D) & 3) == 0; }
// COH (the reward relabel inside the critic-loss launch, sac.hip): the parameters were stepped by a kernel of ANOTHER stream while this launch was resident, and the pre-step
// lines may sit in this XCD's L2 (that kernel's own gradient launch read them there): every parameter load goes below the caches (sc0 sc1, like the polls) instead of relying
// on an invalidate (il_common.hpp sync_acquire_all; profiles/r06_soak_under_load.md).
// (base: wave-uniform - it becomes the buffer resource; off: this lane's element)
template <bool COH> __device__ __forceinline__ float disc_pload(const float* base, int64_t off) { return COH ? sload1(base, off) : gload(base + off); }
template <bool COH> __device__ __forceinline__ f32x4 disc_pload4(const float* base, int64_t off) {
  if (!COH) return gload4(base + off);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7ffffff0, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4), 0, 17));   // sc0 | sc1
}
template <int NV, bool COH = false>
__device__ __forceinline__ void disc_w1_issue(DiscW1<NV>& s, const float* __restrict__ W1, int nvec, int base) {
#pragma unroll
  for (int q = 0; q < NV; ++q) s.v[q] = disc_pload4<COH>(W1, 4 * (int64_t)min(base + q * (int)blockDim.x + (int)threadIdx.x, nvec - 1));
}
template <int NV>
__device__ __forceinline__ void disc_w1_commit(const DiscW1<NV>& s, float* W1s, int D, int ldw, int nvec, int base, unsigned magic_d) {   // (H * D < 2^16: the rows fit the LDS)
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int i = base + q * (int)blockDim.x + (int)threadIdx.x;
    if (i < nvec) {
      int n = fastdiv(4 * i, magic_d), k = 4 * i - n * D;
#pragma unroll
      for (int c = 0; c < 4; ++c) { W1s[n * ldw + k] = s.v[q][c]; if (++k == D) { k = 0; ++n; } }
    }
  }
}
__device__ __forceinline__ void disc_w1_zero_padding(float* W1s, int D, int Dp, int H, int ldw) {
  const int pad = Dp - D;
  for (int i = threadIdx.x; i < H * pad; i += blockDim.x) { const int n = i / pad, k = D + (i - n * pad); W1s[n * ldw + k] = 0.f; }
}

// LDS of one reward tile: W1 (rows padded to Dp + 4), b1, W2, u1, v1, v2, {sigma1, sigma2, u2, -}
__host__ __device__ inline size_t reward_lds_floats(int D, int H) { const int Dp = (D + 3) & ~3; return (size_t)H * (Dp + 4) + 2 * H + 2 * H + Dp + 8; }
struct RewardLds { float *W1s, *b1s, *W2s, *u1, *v1, *v2, *sc; };
__device__ __forceinline__ RewardLds reward_carve(float* p, int D, int H) {
  const int Dp = (D + 3) & ~3;
  RewardLds l; l.W1s = p; p += H * (Dp + 4); l.b1s = p; p += H; l.W2s = p; p += H; l.u1 = p; p += H; l.v1 = p; p += Dp; l.v2 = p; p += H; l.sc = p;
  return l;
}
// Every thread of the workgroup calls (barriers inside). X: the tile's rows in LDS, row stride ldX (16-byte aligned, zero-padded to Dp columns).
// Thread tid < 256 with (tid & 15) == 0 and row r = tid >> 4 < nrows gets reward / logit of row r through `emit(r, reward, logit)`.
// NV: 16-byte lanes of W1 a thread requests per round (ceil(H D / 4 / blockDim.x) at the shape the caller is tuned for; any shape works, in more rounds).
template <int NV = 6, bool COH = false, class Emit>
__device__ __forceinline__ void disc_reward_tile(const il_disc& d, const RewardLds& L, const float* X, int ldX, int nrows, const float* __restrict__ logit_offset, int row0, Emit emit) {
  const int S = d.state_dim, A = d.state_only ? 0 : d.action_dim, D = S + A, H = d.hidden, Dp = (D + 3) & ~3, ldw = Dp + 4, tid = threadIdx.x;
  const DiscLayout lay = disc_layout(D, H, d.spectral_norm);
  const float b2 = disc_pload<COH>(d.params, lay.ob2);
  {
    // (round 4) EVERY parameter this thread stages is requested before its first LDS store: the parameters were rewritten by the AdamW launch an instant ago (each line a
    // fabric / HBM round trip), and the load -> store loops this replaces made one such trip after the other - three for W1 in a 512-thread workgroup, then b1 / W2, then
    // u / v: 4.5 us of the relabel's 6 (profiles/r04_update_timeline.md).
    const float* W1 = d.params + lay.oW1; const float* b1 = d.params + lay.ob1; const float* W2 = d.params + lay.oW2;
    const int bd = blockDim.x, sn = d.spectral_norm, th = min(tid, H - 1), tk = min(tid, D - 1);
    const bool flat = disc_w1_flat_ok(W1, D, H);
    const int nvec = (H * D) >> 2;
    const unsigned md = fastdiv_magic(Dp), mdd = fastdiv_magic(D);   // (H * Dp < 2^16 for every shape whose tile fits the LDS)
    DiscW1<NV> ws; float v[4];
    if (flat) disc_w1_issue<NV, COH>(ws, W1, nvec, 0);
    else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = min(q * bd + tid, H * Dp - 1), n = fastdiv(i, md), k = min(i - n * Dp, D - 1); v[q] = disc_pload<COH>(W1, (size_t)n * D + k); }
    }
    const float vb1 = disc_pload<COH>(b1, th), vw2 = disc_pload<COH>(W2, th);
    float vu1 = 0.f, vv2 = 0.f, vv1 = 0.f, vu2 = 0.f;
    if (sn) { vu1 = disc_pload<COH>(d.u1, th); vv2 = disc_pload<COH>(d.v2, th); vv1 = disc_pload<COH>(d.v1, tk); vu2 = disc_pload<COH>(d.u2, 0); }
    if (flat) {
      disc_w1_commit(ws, L.W1s, D, ldw, nvec, 0, mdd);
      for (int base = NV * bd; base < nvec; base += NV * bd) { disc_w1_issue<NV, COH>(ws, W1, nvec, base); disc_w1_commit(ws, L.W1s, D, ldw, nvec, base, mdd); }   // (shapes beyond NV lanes per thread)
      disc_w1_zero_padding(L.W1s, D, Dp, H, ldw);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = q * bd + tid; if (i < H * Dp) { const int n = fastdiv(i, md), k = i - n * Dp; L.W1s[n * ldw + k] = k < D ? v[q] : 0.f; } }
      for (int i = 4 * bd + tid; i < H * Dp; i += bd) { const int n = i / Dp, k = i - n * Dp; L.W1s[n * ldw + k] = k < D ? disc_pload<COH>(W1, (size_t)n * D + k) : 0.f; }   // (shapes beyond four elements per thread)
    }
    if (tid < H) { L.b1s[tid] = vb1; L.W2s[tid] = vw2; }
    for (int i = bd + tid; i < H; i += bd) { L.b1s[i] = disc_pload<COH>(b1, i); L.W2s[i] = disc_pload<COH>(W2, i); }
    if (sn) {
      if (tid < H) { L.u1[tid] = vu1; L.v2[tid] = vv2; }
      for (int i = bd + tid; i < H; i += bd) { L.u1[i] = disc_pload<COH>(d.u1, i); L.v2[i] = disc_pload<COH>(d.v2, i); }
      if (tid < Dp) L.v1[tid] = tid < D ? vv1 : 0.f;
      for (int i = bd + tid; i < Dp; i += bd) L.v1[i] = i < D ? disc_pload<COH>(d.v1, i) : 0.f;
      if (tid == 0) L.sc[2] = vu2;
    } else if (tid == 0) { L.sc[0] = 1.f; L.sc[1] = 1.f; }
  }
  __syncthreads();
  if (d.spectral_norm && tid < 64) sn_wave(L.W1s, L.W2s, D, H, L.u1, L.v1, &L.sc[2], L.v2, false, L.sc);  // eval mode: sigma only
  __syncthreads();
  if (tid < 256) {
    const float s1 = L.sc[0], s2 = L.sc[1];
    const int r = tid >> 4, sub = tid & 15;
    float zp = 0.f;
    for (int n = sub; n < H; n += 16) zp += (L.W2s[n] / s2) * fmaxf(dot4(L.W1s + n * ldw, X + r * ldX, Dp) / s1 + L.b1s[n], 0.f);
    zp = group16_sum(zp);
    if (sub == 0 && r < nrows) {
      const float f = zp + b2, z = logit_offset ? f - logit_offset[row0 + r] : f, Dp_ = sigmoid_f(z);
      float h = d.reward_function == 1 ? -log1pf(-Dp_ + 1e-6f) : logf(Dp_ + 1e-6f) - log1pf(-Dp_ + 1e-6f);
      if (d.reward_function == 2) h = expf(h) * -h;
      emit(r, h, z);
    }
  }
}
