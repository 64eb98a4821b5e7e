// PWIL greedy Wasserstein-coupling reward (reference models.py:216-249) for gfx950.
//
// One workgroup of 1024 threads per environment step. The reference deletes consumed rows (three O(N D) copies per
// deletion); here a consumed atom is marked by a negative weight, each thread keeps the running minimum of the atoms it
// owns (strided ownership), and one greedy iteration is a 1024-way (distance, index) arg-min -- ties resolve to the lowest
// index, which is what argmin on the reference's order-preserving shrunk tensor returns -- followed by a rescan by the
// single owner of the consumed atom. Cost / remaining weight are accumulated in double like the reference's Python floats.
#include <float.h>
#include <limits.h>

#include "il_common.hpp"

__global__ __launch_bounds__(1024) void k_pwil_reset(il_pwil d) {
  const float w = (float)(1.0 / (double)d.n_atoms);
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

extern "C" int il_pwil_reset(const il_pwil* d, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->weights && d->n_atoms > 0, "il_pwil_reset: bad arguments");
  { IL_TRACE("k_pwil_reset", (hipStream_t)stream_); k_pwil_reset<<<ceil_div(d->n_atoms, 1024) < 256 ? ceil_div(d->n_atoms, 1024) : 256, 1024, 0, (hipStream_t)stream_>>>(*d); }
  IL_CHECK_LAUNCH("il_pwil_reset");
  return IL_OK;
}

extern "C" int il_pwil_reward(const il_pwil* d, const float* state, const float* action, float* out_reward, il_stream_t stream_) {
  IL_CHECK_ARG(d && d->atoms && d->weights && d->dists && d->scale && d->offset && state && out_reward, "il_pwil_reward: bad arguments");
  IL_CHECK_ARG(d->dim >= 1 && d->dim <= 512 && d->n_atoms > 0, "il_pwil_reward: dim=%d out of range [1,512]", d->dim);
  IL_CHECK_ARG(d->state_dim == d->dim || action, "il_pwil_reward: action pointer missing");
  { IL_TRACE("k_pwil_reward", (hipStream_t)stream_); k_pwil_reward<<<1, 1024, 0, (hipStream_t)stream_>>>(*d, state, action, out_reward); }
  IL_CHECK_LAUNCH("il_pwil_reward");
  return IL_OK;
}
