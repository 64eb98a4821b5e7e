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

