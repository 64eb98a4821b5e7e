// C-ABI plumbing: error text, version, device info.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "il_common.hpp"

static thread_local char g_err[512] = "";

int il_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* il_last_error(void) { return g_err; }
extern "C" int il_abi_version(void) { return IL_ABI_VERSION; }

extern "C" int il_device_info(char* name_host, int name_len, int* cu_count_host) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (name_host && name_len > 0) { strncpy(name_host, p.gcnArchName, (size_t)name_len - 1); name_host[name_len - 1] = 0; }
  if (cu_count_host) *cu_count_host = p.multiProcessorCount;
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// per-kernel event tracing
// ---------------------------------------------------------------------------------------------
#define IL_TRACE_MAX 8192
static int g_trace_on = 0, g_trace_n = 0, g_trace_pool = 0;
static const char* g_trace_name[IL_TRACE_MAX];
static hipEvent_t g_trace_ev[IL_TRACE_MAX][2];

il_trace_scope::il_trace_scope(const char* name, hipStream_t s) : st(s), slot(-1) {
  if (!g_trace_on || g_trace_n >= IL_TRACE_MAX) return;
  slot = g_trace_n++;
  if (slot >= g_trace_pool) { (void)hipEventCreate(&g_trace_ev[slot][0]); (void)hipEventCreate(&g_trace_ev[slot][1]); g_trace_pool = slot + 1; }
  g_trace_name[slot] = name;
  (void)hipEventRecord(g_trace_ev[slot][0], st);
}
il_trace_scope::~il_trace_scope() { if (slot >= 0) (void)hipEventRecord(g_trace_ev[slot][1], st); }

extern "C" int il_trace_enable(int on) { g_trace_on = on; g_trace_n = 0; return IL_OK; }

// Synchronises the device and writes "name count total_ms\n" lines (aggregated by kernel name) into buf.
extern "C" int il_trace_report(char* buf_host, int len) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "il_trace_report: %s", hipGetErrorString(e));
  const char* names[64]; double tot[64]; int cnt[64]; int nn = 0;
  for (int i = 0; i < g_trace_n; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_trace_ev[i][0], g_trace_ev[i][1]) != hipSuccess) continue;
    int k = 0;
    for (; k < nn; ++k) if (!strcmp(names[k], g_trace_name[i])) break;
    if (k == nn) { if (nn == 64) continue; names[nn] = g_trace_name[i]; tot[nn] = 0; cnt[nn] = 0; ++nn; }
    tot[k] += ms; cnt[k] += 1;
  }
  int off = 0;
  if (buf_host && len > 0) buf_host[0] = 0;
  for (int k = 0; k < nn && buf_host; ++k) { const int w = snprintf(buf_host + off, (size_t)(len - off), "%s %d %.6f\n", names[k], cnt[k], tot[k]); if (w < 0 || w >= len - off) break; off += w; }
  g_trace_n = 0;
  return IL_OK;
}


// ---------------------------------------------------------------------------------------------
// Probe for the device-side hand-off (il_sync): one waiter and one setter kernel with the same bounded wait the update kernels use.
// A caller enqueues the waiter on its side stream FIRST and the setter on its main stream (inside the same kind of two-stream graph
// it will run updates with); if the runtime executes the two streams one after the other, the waiter times out and
// sync[IL_SYNC_TIMEOUTS] counts it - the caller then keeps plain stream dependencies.
// ---------------------------------------------------------------------------------------------
__global__ void k_sync_probe(long long* sync, int setter) {
  if (setter) { sync_signal(sync + IL_SYNC_PROBE_FLAG); return; }
  const long long e = sync[IL_SYNC_PROBE_EPOCH];
  sync_wait(sync, IL_SYNC_PROBE_FLAG, e + 1, 20000);   // ~10 ms: the probe is enqueued back to back, and a serialising runtime should be detected quickly
  if (threadIdx.x == 0) sync[IL_SYNC_PROBE_EPOCH] = e + 1;
}
extern "C" void il_sync_layout(int32_t* out) { out[0] = IL_SYNC_SLOTS; out[1] = IL_SYNC_TIMEOUTS; out[2] = IL_SYNC_GATHER_WGS; out[3] = IL_SYNC_STRIDE; out[4] = IL_SYNC_SPIN; out[5] = IL_SYNC_HOST_FLAG; }
// out[0 .. n): il_sync_layout's six, then [6] IL_SYNC_POISON, [7] IL_SYNC_OV_EPOCH, [8] IL_SYNC_OV_TICKET, [9] IL_SYNC_MAIN_EPOCH
extern "C" void il_sync_layout_ex(int32_t* out, int32_t n) {
  const int32_t v[10] = {IL_SYNC_SLOTS, IL_SYNC_TIMEOUTS, IL_SYNC_GATHER_WGS, IL_SYNC_STRIDE, IL_SYNC_SPIN, IL_SYNC_HOST_FLAG, IL_SYNC_POISON, IL_SYNC_OV_EPOCH, IL_SYNC_OV_TICKET, IL_SYNC_MAIN_EPOCH};
  for (int i = 0; i < n && i < 10; ++i) out[i] = v[i];
}
__global__ void k_sync_clear_poison(long long* sync) { if (threadIdx.x == 0 && blockIdx.x == 0) { sync[IL_SYNC_POISON] = 0; sync[IL_SYNC_TIMEOUTS] = 0; } }
extern "C" int il_sync_clear_poison(int64_t* sync, il_stream_t stream) {
  IL_CHECK_ARG(sync, "il_sync_clear_poison: null counters");
  k_sync_clear_poison<<<1, 64, 0, (hipStream_t)stream>>>((long long*)sync);
  IL_CHECK_LAUNCH("il_sync_clear_poison");
  return IL_OK;
}
extern "C" int il_sync_probe(int64_t* sync, int32_t setter, il_stream_t stream) {
  IL_CHECK_ARG(sync, "il_sync_probe: null counters");
  k_sync_probe<<<1, 64, 0, (hipStream_t)stream>>>((long long*)sync, setter);
  IL_CHECK_LAUNCH("il_sync_probe");
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// il_noise_fill: the update kernels' on-chip noise as a function of (key, update counter, stream id, index) -- the same device functions
// (philox_normal / philox_uniform, il_common.hpp) they evaluate when their eps pointer is NULL. A caller that knows the counter range a
// captured run covered can therefore record exactly what those updates consumed (the parity tests replay it through the CPU oracle).
// ---------------------------------------------------------------------------------------------
__global__ void k_noise_fill(uint64_t seed, uint32_t ctr, uint32_t stream_id, long long n, int normal, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = normal ? philox_normal(seed, ctr, stream_id, (uint32_t)i) : philox_uniform(seed, ctr, stream_id, (uint32_t)i);
}
extern "C" int il_noise_fill(uint64_t noise_seed, uint32_t ctr, uint32_t stream_id, int64_t n, float* out, il_stream_t stream) {
  IL_CHECK_ARG(out && n >= 0 && n < (1LL << 32), "il_noise_fill: bad arguments");
  const bool normal = stream_id == IL_STREAM_EPS_NEXT || stream_id == IL_STREAM_EPS_CUR || stream_id == IL_STREAM_ACT;
  IL_CHECK_ARG(normal || stream_id == IL_STREAM_GP || stream_id == IL_STREAM_MIX || stream_id == 5u || stream_id == 6u || stream_id == 11u, "il_noise_fill: unknown stream id %u", stream_id);   // 5, 6, 11: dropout uniforms (dril.hip)
  if (n == 0) return IL_OK;
  k_noise_fill<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(noise_seed, ctr, stream_id, (long long)n, normal ? 1 : 0, out);
  IL_CHECK_LAUNCH("il_noise_fill");
  return IL_OK;
}

// il_noise_fill_beta: n Beta(alpha, alpha) draws of the Mixup stream at the update counter *ctr_dev (read on the DEVICE, so the launch can sit in a captured update: the
// counter is the discriminator's noise_counter, advanced once per update by the actor step). `out` is then passed as il_gail_extra.eps_mix (training.py:105-107).
__global__ void k_noise_fill_beta(uint64_t seed, const uint32_t* __restrict__ ctr_dev, float alpha, long long n, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_beta(seed, ctr_dev ? *ctr_dev : 0u, IL_STREAM_MIX, (uint32_t)i, alpha);
}
extern "C" int il_noise_fill_beta(uint64_t noise_seed, const uint32_t* ctr_dev, float alpha, int64_t n, float* out, il_stream_t stream) {
  IL_CHECK_ARG(out && n >= 0 && n < (1LL << 32) && alpha > 0.f, "il_noise_fill_beta: bad arguments (alpha must be > 0: train.py:46)");
  if (n == 0) return IL_OK;
  k_noise_fill_beta<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(noise_seed, ctr_dev, alpha, (long long)n, out);
  IL_CHECK_LAUNCH("il_noise_fill_beta");
  return IL_OK;
}

// ---------------------------------------------------------------------------------------------
// Launch stamps of the headline schedule (il_common.hpp "Always-on launch stamps"): per kernel id of the LAST launch that ran, in ticks of the 100 MHz device-wide
// counter: out_host[kid][0..4] = {min begin, max begin, min end, max end, workgroups stamped}; all 0 when the kernel has not run since il_kernel_stamps_clear().
// Synchronises the device. bench.py derives `roofline.kernels` from these after its timed graph replays.
// ---------------------------------------------------------------------------------------------
extern "C" int il_stamps_sac(unsigned long long*); extern "C" int il_stamps_gail(unsigned long long*); extern "C" int il_stamps_gmmil(unsigned long long*);
extern "C" int il_stamps_sac_clear(); extern "C" int il_stamps_gail_clear(); extern "C" int il_stamps_gmmil_clear();
extern "C" int32_t il_kernel_stamp_ids(void) { return IL_ST_K; }
extern "C" int il_kernel_stamps(uint64_t* out_host) {
  IL_CHECK_ARG(out_host, "il_kernel_stamps: null argument");
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "il_kernel_stamps: %s", hipGetErrorString(e));
  typedef unsigned long long table_t[IL_ST_K][IL_ST_WGS][4];
  static table_t tb;
  memset(out_host, 0, sizeof(uint64_t) * IL_ST_K * 5);
  struct Src { int (*read)(unsigned long long*); int first, last; };
  const Src srcs[3] = {{il_stamps_gail, IL_ST_GAIL_GRAD, IL_ST_GAIL_REDUCE}, {il_stamps_sac, IL_ST_CHAIN, IL_ST_DW_ACTOR}, {il_stamps_gmmil, IL_ST_GMMIL, IL_ST_GMMIL}};
  for (const Src& sc : srcs) {
    if (sc.read(&tb[0][0][0]) != 0) return il_set_error(IL_ERR_HIP, "il_kernel_stamps: reading a stamp table failed");
    for (int k = sc.first; k <= sc.last; ++k) {
      uint64_t* o = out_host + 5 * k;
      for (int w = 0; w < IL_ST_WGS; ++w) {
        const uint64_t b = tb[k][w][0], en = tb[k][w][1];
        if (!b || !en) continue;
        if (!o[4]) { o[0] = o[1] = b; o[2] = o[3] = en; }
        if (b < o[0]) o[0] = b;
        if (b > o[1]) o[1] = b;
        if (en < o[2]) o[2] = en;
        if (en > o[3]) o[3] = en;
        o[4] += 1;
      }
    }
  }
  return IL_OK;
}
// the raw rows of one kernel id: out_host [IL_ST_WGS][4] = {begin, end, placement (XCC_ID << 16 | SE / SH / CU byte of HW_ID), 0} per workgroup of the last launch (zeros: not stamped)
extern "C" int32_t il_kernel_stamp_workgroups(void) { return IL_ST_WGS; }
extern "C" int il_kernel_stamp_rows(int32_t kid, uint64_t* out_host) {
  IL_CHECK_ARG(out_host && kid >= 0 && kid < IL_ST_K, "il_kernel_stamp_rows: bad arguments");
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "il_kernel_stamp_rows: %s", hipGetErrorString(e));
  typedef unsigned long long table_t[IL_ST_K][IL_ST_WGS][4];
  static table_t tb;
  int (*read)(unsigned long long*) = kid <= IL_ST_GAIL_REDUCE ? il_stamps_gail : (kid <= IL_ST_DW_ACTOR ? il_stamps_sac : il_stamps_gmmil);
  if (read(&tb[0][0][0]) != 0) return il_set_error(IL_ERR_HIP, "il_kernel_stamp_rows: reading a stamp table failed");
  memcpy(out_host, &tb[kid][0][0], sizeof(unsigned long long) * IL_ST_WGS * 4);
  return IL_OK;
}
extern "C" int il_kernel_stamps_clear(void) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "il_kernel_stamps_clear: %s", hipGetErrorString(e));
  return (il_stamps_gail_clear() == 0 && il_stamps_sac_clear() == 0 && il_stamps_gmmil_clear() == 0) ? IL_OK : il_set_error(IL_ERR_HIP, "il_kernel_stamps_clear: clearing a stamp table failed");
}

// A stream whose kernels may only run on the CUs of `mask` (hipExtStreamCreateWithCUMask; bit i = CU i in the runtime's enumeration: on a multi-XCD part the bits are
// dealt round-robin to the XCDs). Experiment of round 5: the discriminator branch on CUs of its own instead of the whole-CU LDS requests that keep its
// workgroups off the pair-mode workgroups' CUs. Not used by default.
extern "C" int il_stream_create_cu_mask(const uint32_t* mask_host, int32_t words, void** stream_out) {
  IL_CHECK_ARG(mask_host && words > 0 && stream_out, "il_stream_create_cu_mask: bad arguments");
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask_host);
  if (e != hipSuccess) return il_set_error(IL_ERR_HIP, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *stream_out = (void*)s;
  return IL_OK;
}
extern "C" int il_stream_destroy(void* stream) {
  if (!stream) return IL_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? IL_OK : il_set_error(IL_ERR_HIP, "hipStreamDestroy: %s", hipGetErrorString(e));
}

// sizeof() of the descriptor structs as this library was compiled: a binding checks its own struct definitions against these
// (0 il_batch, 1 il_adam, 2 il_sac, 3 il_disc, 4 il_pwil, 5 il_sample_args, 6 il_red, 7 il_dril, 8 il_disc_shaped, 9 il_disc_deep, 10 il_peer_bucket).
extern "C" int32_t il_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(il_batch); case 1: return (int32_t)sizeof(il_adam); case 2: return (int32_t)sizeof(il_sac); case 3: return (int32_t)sizeof(il_disc);
    case 4: return (int32_t)sizeof(il_pwil); case 5: return (int32_t)sizeof(il_sample_args); case 6: return (int32_t)sizeof(il_red); case 7: return (int32_t)sizeof(il_dril); case 8: return (int32_t)sizeof(il_disc_shaped); case 9: return (int32_t)sizeof(il_disc_deep); case 10: return (int32_t)sizeof(il_peer_bucket); case 11: return (int32_t)sizeof(il_disc_shaped_deep);
    default: return -1;
  }
}
