// A launcher thread for recorded update launches (round 6; include/il_hip.h "il_launcher").
//
// One SAC+GAIL update is two library calls (six kernel launches, ~20 us of hipLaunchKernel work on the host) - plus an ActingWorker's append before and parameter snapshot
// after it. Issued from the thread that also steps the environment (train.py:151-203: act -> env.step -> append -> update, one after the other), those 20 us sit between two
// environment steps: profiles/r06_acting.json, 11.0-11.5k env-steps/s against 17.4k updates/s. Here the recorded calls (UpdatePlan.record_direct: entry point + arguments,
// descriptors by reference) are re-issued by a thread of this library: il_launcher_submit() returns at once and the caller goes on to post the next observation, launch its
// act kernel and step the environment while the update's launches go out - in the recorded order, one pass per submit, passes in submission order.
//
// A recorded call is (function pointer, up to 16 integer / pointer arguments): every launch entry point of this ABI takes pointers, sizes, flags and a stream, none takes a
// struct by value; entry points with floating-point arguments (il_gmmil_reward's bandwidths) are refused by the recorder on the Python side. The thread makes the caller's
// device current, spins briefly for work and then sleeps on a condition variable; the first non-zero status of a recorded call is kept (with its il_last_error text) and
// returned by every later submit / wait.
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "il_common.hpp"

namespace {
typedef int (*il_call16)(uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
struct LaunchCall { il_call16 fn; uint64_t a[16]; };
struct Launcher {
  std::vector<LaunchCall> calls;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::atomic<long long> submitted{0}, done{0};
  std::atomic<int> status{0}, stop{0};
  int device = 0, depth = 4;
  char error[512] = "";

  void run() {
    (void)hipSetDevice(device);
    long long mine = 0;
    for (;;) {
      int spins = 0;
      while (submitted.load(std::memory_order_acquire) == mine && !stop.load(std::memory_order_acquire)) {
        if (++spins < 20000) { __builtin_ia32_pause(); continue; }   // ~100 us of spinning: an update per environment step arrives well inside it
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return submitted.load(std::memory_order_acquire) != mine || stop.load(std::memory_order_acquire); });
      }
      if (stop.load(std::memory_order_acquire) && submitted.load(std::memory_order_acquire) == mine) return;
      if (!status.load(std::memory_order_relaxed)) {
        for (const LaunchCall& c : calls) {
          const uint64_t* a = c.a;
          const int rc = c.fn(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
          if (rc != 0) {   // sticky: later passes are skipped, the caller sees it at its next submit / wait
            snprintf(error, sizeof(error), "%s", il_last_error());
            status.store(rc, std::memory_order_release);
            break;
          }
        }
      }
      ++mine;
      { std::lock_guard<std::mutex> lk(mu); done.store(mine, std::memory_order_release); }
      cv_done.notify_all();
    }
  }
};
}  // namespace

extern "C" int il_launcher_create(void** out) {
  IL_CHECK_ARG(out, "il_launcher_create: null argument");
  Launcher* l = new Launcher();
  (void)hipGetDevice(&l->device);
  l->worker = std::thread([l] { l->run(); });
  *out = l;
  return IL_OK;
}

extern "C" int il_launcher_destroy(void* h) {
  if (!h) return IL_OK;
  Launcher* l = static_cast<Launcher*>(h);
  { std::lock_guard<std::mutex> lk(l->mu); l->stop.store(1, std::memory_order_release); }
  l->cv_work.notify_all();
  if (l->worker.joinable()) l->worker.join();
  delete l;
  return IL_OK;
}

// every submitted pass has been issued (NOT: has run - follow with a stream / device synchronisation for that); returns the sticky status of the recorded calls
extern "C" int il_launcher_wait(void* h) {
  IL_CHECK_ARG(h, "il_launcher_wait: null argument");
  Launcher* l = static_cast<Launcher*>(h);
  const long long want = l->submitted.load(std::memory_order_acquire);
  int spins = 0;
  while (l->done.load(std::memory_order_acquire) < want) {
    if (++spins < 20000) { __builtin_ia32_pause(); continue; }
    std::unique_lock<std::mutex> lk(l->mu);
    l->cv_done.wait(lk, [&] { return l->done.load(std::memory_order_acquire) >= want; });
  }
  const int rc = l->status.load(std::memory_order_acquire);
  return rc ? il_set_error(rc, "il_launcher: a recorded call failed: %s", l->error) : IL_OK;
}

// forget the recorded calls (and a sticky status); waits for the passes in flight first
extern "C" int il_launcher_clear(void* h) {
  IL_CHECK_ARG(h, "il_launcher_clear: null argument");
  (void)il_launcher_wait(h);
  Launcher* l = static_cast<Launcher*>(h);
  l->calls.clear();
  l->status.store(0, std::memory_order_release);
  l->error[0] = 0;
  return IL_OK;
}

// append one call to the recorded pass: fn = an entry point of this ABI that takes only pointers / integers (<= 16 of them), args_host = their values as 64-bit words.
// Descriptors passed by reference stay the caller's: they must outlive the launcher's use of them (UpdatePlan keeps them), and later changes of their fields apply.
extern "C" int il_launcher_add(void* h, void* fn, const uint64_t* args_host, int32_t nargs) {
  IL_CHECK_ARG(h && fn && (args_host || nargs == 0) && nargs >= 0 && nargs <= 16, "il_launcher_add: bad arguments (at most 16 integer / pointer arguments)");
  Launcher* l = static_cast<Launcher*>(h);
  IL_CHECK_ARG(l->submitted.load(std::memory_order_acquire) == l->done.load(std::memory_order_acquire), "il_launcher_add: passes are in flight (il_launcher_wait first)");
  LaunchCall c = {};
  c.fn = reinterpret_cast<il_call16>(fn);
  for (int i = 0; i < nargs; ++i) c.a[i] = args_host[i];
  l->calls.push_back(c);
  return IL_OK;
}

// one pass over the recorded calls, issued by the launcher thread; returns at once (blocks only while `depth` passes - 4 - are waiting to be issued: the host cannot run
// further ahead of its own launches). Returns the sticky status of earlier passes.
extern "C" int il_launcher_submit(void* h) {
  IL_CHECK_ARG(h, "il_launcher_submit: null argument");
  Launcher* l = static_cast<Launcher*>(h);
  const int rc = l->status.load(std::memory_order_acquire);
  if (rc) return il_set_error(rc, "il_launcher: a recorded call failed: %s", l->error);
  while (l->submitted.load(std::memory_order_relaxed) - l->done.load(std::memory_order_acquire) >= l->depth) __builtin_ia32_pause();
  { std::lock_guard<std::mutex> lk(l->mu); l->submitted.fetch_add(1, std::memory_order_release); }
  l->cv_work.notify_one();
  return IL_OK;
}

extern "C" int64_t il_launcher_pending(void* h) {
  if (!h) return 0;
  Launcher* l = static_cast<Launcher*>(h);
  return (int64_t)(l->submitted.load(std::memory_order_acquire) - l->done.load(std::memory_order_acquire));
}
