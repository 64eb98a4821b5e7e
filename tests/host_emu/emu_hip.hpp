// TEST INFRASTRUCTURE: runs the HIP kernels of imitation-learning_amd/csrc on the HOST, from their own source text, so that `pytest -m "not gpu"` can execute them
// against the reference fixtures where there is no GPU. tests/host_emu/build.py rewrites `k<<<grid, block, lds, stream>>>(args)` into EMU_LAUNCH, the dynamic
// `extern __shared__` declaration into a pointer, drops address-space qualifiers and empty asm pins, and compiles the result with g++ against this header in the
// place of <hip/hip_runtime.h>.
//
// Execution model: everything on ONE OS thread. A workgroup's threads are fibers; a wave's 64 lanes take turns until each sits at a __syncthreads() or has returned,
// then the next wave runs; a barrier releases when every live fiber has arrived. Wave-level instructions - __shfl*, DPP, readlane, ballot, the wave barrier,
// v_mfma_f32_16x16x4_f32 - are built on one exchange: every live lane of the wave publishes its operand under a per-lane sequence number and yields until all of them
// have. A launch is a job in its stream's queue: jobs of one stream run in order, the jobs at the heads of different streams side by side, workgroups dispatched in
// blockIdx order (x fastest). Several workgroups can be RESIDENT at once: a lane that polls a device-side counter (every polling loop of the kernels sleeps between two
// polls: s_sleep) suspends its workgroup, and the others - later workgroups of the same launch, workgroups of the launch on the other stream - run until the counter
// moves; that is what co-residency gives the device-side hand-offs on the GPU. Launches on the null stream run to completion at once. Atomics are plain
// read-modify-writes, fences and s_waitcnt nothing: program order on one thread is stronger than any of them.
// IL_EMU_SCHEDULE=reverse|random:<seed> perturbs every choice the model leaves open (lane order, wave order between two barriers, the dispatch order of a launch's
// workgroups, which stream's launch takes the next turn): a result that changes with it is a missing barrier, a race between workgroups or a hand-off that only works in
// one order.
// What this can show: indexing, tile / slab / counter layouts, iteration orders, the hand-off protocols' logic, the arithmetic (MFMA = an fmaf chain over k, as on the
// device) - everything a parity test compares. What it cannot: performance, the memory model (a missing fence or a non-atomic flag is invisible here), code that leans
// on wave lockstep without a wave barrier. Sums follow the device's association order (the DPP steps and readlanes are emulated lane for lane); libm is glibc's, so
// comparisons use the parity tests' tolerances, not bit equality.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <deque>
#include <functional>
#include <map>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
using std::max;
using std::min;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }

// device-only builtins the emulated sources mention: scheduling hints and memory-order fences mean nothing to one OS thread running one workgroup at a time
#define __builtin_amdgcn_sched_barrier(x) do { } while (0)
#define __builtin_amdgcn_fence(...) do { } while (0)
#define __HIP_MEMORY_SCOPE_AGENT 0
template <class T> inline T __hip_atomic_fetch_add(T* p, T v, int, int) { const T old = *p; *p = old + v; return old; }
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T> inline void __hip_atomic_store(T* p, T v, int, int) { *p = v; }
template <class T> inline T __hip_atomic_exchange(T* p, T v, int, int) { const T old = *p; *p = v; return old; }
template <class V> inline V emu_elementwise_fma(V a, V b, V c) { V r = c; for (unsigned i = 0; i < sizeof(V) / sizeof(float); ++i) r[i] = fmaf(a[i], b[i], c[i]); return r; }
#define __builtin_elementwise_fma emu_elementwise_fma

namespace emu {
enum { RUNNABLE = 0, AT_BARRIER = 1, DONE = 2 };
enum { MAX_THREADS = 1024, STACK_BYTES = 256 * 1024, SHFL_RING = 64 };
#if defined(__x86_64__)
// A fiber switch that saves the callee-saved registers and swaps stack pointers (swapcontext also saves the signal mask: two system calls per switch, and the MFMA /
// DPP emulation switches ~128 times per wave-level instruction).
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.weak emu_switch
.type emu_switch, @function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");
struct Fiber { void* sp; int state; };
inline void* sched_sp = nullptr;
#define EMU_FAST_SWITCH 1
#else
struct Fiber { ucontext_t ctx; int state; };
inline ucontext_t sched_ctx;
#define EMU_FAST_SWITCH 0
#endif
// ---- workgroups, launches, streams
// A launch is a job in its stream's queue; jobs of one stream run in order, jobs of different streams side by side. A workgroup (Block) owns its fibers, stacks, LDS and
// exchange buffers, so that several can be resident at once: a lane that polls a device-side counter (every polling loop of the kernels sleeps: s_sleep) suspends its
// WORKGROUP and lets the others - of the same launch, or of the launch at the head of another stream - run; that is what co-residency gives the device hand-offs on
// the GPU. Launches on the null stream run to completion at once (everything else drained first), which is all the per-function tests need.
struct Block {
  dim3 idx, gdim, bdim;
  int n = 0, cap = 0;
  Fiber* fibers = nullptr; char* stacks = nullptr;
  void* lds = nullptr;
  uint64_t* shfl_val = nullptr; unsigned* shfl_tag = nullptr; unsigned* shfl_seq = nullptr;
  unsigned char* coll_val = nullptr; unsigned* coll_tag = nullptr; unsigned* coll_seq = nullptr;   // wave-level exchange: payload slots [ring][lane], per (ring slot, wave) {arrivals, exchange number}, per-lane exchange counter
  unsigned* wave_done = nullptr;   // per wave: lanes that have returned from the kernel
  unsigned* or_seq = nullptr; int or_val[2] = {0, 0};   // __syncthreads_or: per-thread call counter, two alternating accumulators
  std::vector<std::pair<const void*, void*>> statics;   // static __shared__ variables of the kernel: per workgroup (build.py turns the declarations into lookups)
  const std::function<void()>* body = nullptr;
  bool yield_requested = false;
};
inline Block* blk = nullptr;   // the workgroup whose fiber is running
inline int cur = 0;
inline void* dyn_smem = nullptr;
inline int block_threads = 0;
enum { COLL_RING = 8, COLL_BYTES = 16 };
inline int schedule_mode = -1;          // 0 forward, 1 reverse, 2 random (IL_EMU_SCHEDULE, read at the first launch)
inline unsigned long long rng_state = 1;
inline unsigned next_random() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng_state >> 33); }
inline void read_schedule() {
  if (schedule_mode >= 0) return;
  const char* e = getenv("IL_EMU_SCHEDULE");
  schedule_mode = !e || !strcmp(e, "forward") ? 0 : (!strcmp(e, "reverse") ? 1 : 2);
  if (schedule_mode == 2) { const char* c = strchr(e, ':'); rng_state = c ? strtoull(c + 1, nullptr, 10) * 2 + 1 : 1; }
}

#if EMU_FAST_SWITCH
inline void to_scheduler() { emu_switch(&blk->fibers[cur].sp, sched_sp); }
inline void to_fiber(int i) { emu_switch(&sched_sp, blk->fibers[i].sp); }
#else
inline void to_scheduler() { swapcontext(&blk->fibers[cur].ctx, &sched_ctx); }
inline void to_fiber(int i) { swapcontext(&sched_ctx, &blk->fibers[i].ctx); }
#endif
inline void fiber_entry() {
  (*blk->body)();
  blk->fibers[cur].state = DONE;
  ++blk->wave_done[cur >> 6];
  to_scheduler();
  abort();   // a finished fiber is never resumed
}
inline void make_fiber(Block* b, int i) {
  char* base = b->stacks + (size_t)i * STACK_BYTES;
#if EMU_FAST_SWITCH
  void** sp = (void**)(((uintptr_t)base + STACK_BYTES) & ~(uintptr_t)15);
  *--sp = nullptr;                       // where a return address would sit: fiber_entry starts with rsp = 8 mod 16 like any called function
  *--sp = (void*)&fiber_entry;           // popped by emu_switch's `ret`
  for (int r = 0; r < 6; ++r) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
  b->fibers[i].sp = sp;
#else
  getcontext(&b->fibers[i].ctx);
  b->fibers[i].ctx.uc_stack.ss_sp = base;
  b->fibers[i].ctx.uc_stack.ss_size = STACK_BYTES;
  b->fibers[i].ctx.uc_link = nullptr;
  makecontext(&b->fibers[i].ctx, fiber_entry, 0);
#endif
}
inline int linear_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }

inline std::vector<Block*> free_blocks;   // finished workgroups keep their stacks (page faults on fresh stacks would dominate small kernels)
inline Block* acquire_block(int n) {
  Block* b = nullptr;
  for (size_t i = 0; i < free_blocks.size(); ++i) if (free_blocks[i]->cap >= n) { b = free_blocks[i]; free_blocks.erase(free_blocks.begin() + i); break; }
  if (!b) {
    b = new Block();
    b->cap = n;
    b->stacks = (char*)mmap(nullptr, (size_t)n * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);   // committed page by page as touched
    if (b->stacks == (char*)MAP_FAILED) { perror("emu: mmap of fiber stacks"); abort(); }
    b->fibers = new Fiber[n];
    b->shfl_val = new uint64_t[(size_t)SHFL_RING * n]; b->shfl_tag = new unsigned[(size_t)SHFL_RING * n]; b->shfl_seq = new unsigned[n];
    b->coll_val = new unsigned char[(size_t)COLL_RING * n * COLL_BYTES]; b->coll_tag = new unsigned[(size_t)COLL_RING * n]; b->coll_seq = new unsigned[n]; b->wave_done = new unsigned[(n + 63) / 64]; b->or_seq = new unsigned[n];
  }
  return b;
}
inline void start_block(Block* b, dim3 idx, dim3 gdim, dim3 bdim, size_t lds_bytes, const std::function<void()>* body) {
  b->idx = idx; b->gdim = gdim; b->bdim = bdim; b->n = (int)(bdim.x * bdim.y * bdim.z); b->body = body; b->yield_requested = false;
  b->lds = malloc(lds_bytes ? lds_bytes : 16);   // exactly the launch's LDS (16-byte aligned like every malloc): an AddressSanitizer build sees the first byte past it
  memset(b->lds, 0xcd, lds_bytes);               // LDS is not zero on the GPU either: poison it so that a read of an unwritten word shows
  for (int i = 0; i < b->n; ++i) { make_fiber(b, i); b->fibers[i].state = RUNNABLE; b->shfl_seq[i] = 0; b->coll_seq[i] = 0; b->or_seq[i] = 0; }
  b->or_val[0] = b->or_val[1] = 0;
  memset(b->shfl_tag, 0xff, sizeof(unsigned) * SHFL_RING * b->cap);
  memset(b->coll_tag, 0xff, sizeof(unsigned) * COLL_RING * b->cap);
  memset(b->wave_done, 0, sizeof(unsigned) * ((b->cap + 63) / 64));
}
inline void finish_block(Block* b) {
  free(b->lds); b->lds = nullptr;
  for (auto& kv : b->statics) free(kv.second);
  b->statics.clear();
  free_blocks.push_back(b);
}
// a static __shared__ variable of the running workgroup (key: the address of a string literal naming the declaration)
inline void* block_static(const void* key, size_t bytes) {
  for (auto& kv : blk->statics) if (kv.first == key) return kv.second;
  void* p = malloc(bytes ? bytes : 1);
  memset(p, 0xcd, bytes);
  blk->statics.emplace_back(key, p);
  return p;
}

// Runs the workgroup until every fiber has returned (true) or one of them asked to let other workgroups run (false): one wave at a time, its lanes taking turns until
// each sits at a barrier or has returned (lanes that yield inside a wave-level exchange stay runnable); a barrier releases when every live fiber has arrived.
// IL_EMU_SCHEDULE = reverse | random:<seed> changes the order of the waves between two barriers and of the lanes within a wave: a kernel whose result depends on that
// order is missing a barrier (an LDS race the forward order happens to hide).
inline bool run_slice(Block* b) {
  blk = b; blockIdx = b->idx; gridDim = b->gdim; blockDim = b->bdim; dyn_smem = b->lds; block_threads = b->n;
  const int n = b->n, nwaves = (n + 63) / 64;
  int wave_order[MAX_THREADS / 64], lane_order[64];
  for (;;) {
    for (int w = 0; w < nwaves; ++w) wave_order[w] = schedule_mode == 1 ? nwaves - 1 - w : w;
    for (int l = 0; l < 64; ++l) lane_order[l] = schedule_mode == 1 ? 63 - l : l;
    if (schedule_mode == 2) {
      for (int w = nwaves - 1; w > 0; --w) std::swap(wave_order[w], wave_order[next_random() % (unsigned)(w + 1)]);
      for (int l = 63; l > 0; --l) std::swap(lane_order[l], lane_order[next_random() % (unsigned)(l + 1)]);
    }
    for (int wi = 0; wi < nwaves; ++wi) {
      const int w0 = 64 * wave_order[wi];
      for (bool any = true; any;) {
        any = false;
        for (int li = 0; li < 64; ++li) {
          const int i = w0 + lane_order[li];
          if (i >= n || b->fibers[i].state != RUNNABLE) continue;
          any = true;
          cur = i;
          threadIdx = dim3(i % b->bdim.x, (i / b->bdim.x) % b->bdim.y, i / (b->bdim.x * b->bdim.y));
          to_fiber(i);
          if (b->yield_requested) { b->yield_requested = false; return false; }
        }
      }
    }
    int waiting = 0;
    for (int i = 0; i < n; ++i) waiting += b->fibers[i].state == AT_BARRIER;
    if (!waiting) return true;                                            // every fiber has returned
    for (int i = 0; i < n; ++i) if (b->fibers[i].state == AT_BARRIER) b->fibers[i].state = RUNNABLE;   // barrier: all live fibers arrived
  }
}
// called by a lane inside a polling loop (s_sleep): the workgroup steps aside
inline void poll_yield() { blk->yield_requested = true; to_scheduler(); }

// Wave-level exchange: every LIVE lane of the calling lane's wave publishes `bytes` of payload; returns once all of them have, with `all` pointing at the 64 payload
// slots (a lane that has already returned from the kernel counts as arrived; its slot is stale). The building block of the emulated MFMA / DPP / readlane / ballot.
inline const unsigned char* wave_exchange(const void* mine, size_t bytes) {
  Block* b = blk;
  const int t = linear_tid(), w0 = t & ~63, w1 = std::min(w0 + 64, b->n), cap = b->cap;
  const unsigned k = b->coll_seq[t]++;
  const unsigned slot = k % COLL_RING;
  memcpy(b->coll_val + ((size_t)slot * cap + t) * COLL_BYTES, mine, bytes);
  // arrivals are counted per (ring slot, wave): the lanes of a wave go through the same exchanges in the same order, so the first lane to reach exchange k re-arms the
  // slot's counter; a lane that has returned from the kernel counts as arrived for good (wave_done)
  const int w = t >> 6, nw = (cap + 63) >> 6;
  unsigned* cnt = b->coll_tag + ((size_t)slot * nw + w) * 2;
  if (cnt[1] != k) { cnt[1] = k; cnt[0] = 0; }
  ++cnt[0];
  while ((int)(cnt[0] + b->wave_done[w]) < w1 - w0 && cnt[1] == k) to_scheduler();
  return b->coll_val + ((size_t)slot * cap + w0) * COLL_BYTES;
}
inline bool lane_live(int t) { return blk->fibers[t].state != DONE; }

struct Job {
  dim3 grid, block; size_t lds = 0; std::function<void()> body;
  unsigned next = 0, total = 0, done = 0;
  std::vector<Block*> resident;
  std::vector<unsigned> order;   // dispatch order of the workgroups when the schedule is perturbed (empty: blockIdx order)
  uintptr_t stream = 0; uint64_t seq = 0;
  bool is_wait = false; uintptr_t wait_stream = 0; uint64_t wait_seq = 0;   // a marker: the stream goes on once `wait_stream` has finished its job number `wait_seq`
};
struct StreamQ { std::deque<Job*> q; uint64_t enqueued = 0, finished = 0; };
inline std::map<uintptr_t, StreamQ> streams;
inline bool draining = false;

inline dim3 nth_block(const Job* j, unsigned i) { return dim3(i % j->grid.x, (i / j->grid.x) % j->grid.y, i / (j->grid.x * j->grid.y)); }
// one turn of a launch: every resident (suspended) workgroup gets a slice, then one more workgroup is dispatched; true if anything finished or started
inline bool step(Job* j) {
  bool progress = false;
  for (size_t r = 0; r < j->resident.size();) {
    if (run_slice(j->resident[r])) { finish_block(j->resident[r]); j->resident.erase(j->resident.begin() + r); ++j->done; progress = true; } else ++r;
  }
  if (j->next < j->total) {
    // IL_EMU_SCHEDULE also perturbs the DISPATCH order of a launch's workgroups (reverse / a random permutation). The device dispatches in blockIdx order, and the
    // kernels' hand-offs rely on that only for progress (a waiter is never dispatched before ALL CUs are taken by workgroups that wait for it); with unbounded
    // residency any order makes progress here, and a result that depends on it is a race between workgroups of one launch (they run concurrently on the device).
    if (schedule_mode != 0 && j->order.empty()) {
      j->order.resize(j->total);
      for (unsigned i = 0; i < j->total; ++i) j->order[i] = schedule_mode == 1 ? j->total - 1 - i : i;
      if (schedule_mode == 2) for (unsigned i = j->total; i > 1; --i) std::swap(j->order[i - 1], j->order[next_random() % i]);
    }
    const unsigned which = j->order.empty() ? j->next : j->order[j->next];
    ++j->next;
    Block* b = acquire_block((int)(j->block.x * j->block.y * j->block.z));
    start_block(b, nth_block(j, which), j->grid, j->block, j->lds, &j->body);
    progress = true;
    if (run_slice(b)) { finish_block(b); ++j->done; } else j->resident.push_back(b);
  }
  return progress;
}
// runs every queued launch of every stream to completion
inline void drain() {
  if (draining) return;
  draining = true;
  read_schedule();
  for (;;) {
    bool any = false;
    std::vector<uintptr_t> order;
    for (auto& kv : streams) if (!kv.second.q.empty()) order.push_back(kv.first);
    if (order.empty()) break;
    if (schedule_mode == 1) std::reverse(order.begin(), order.end());
    if (schedule_mode == 2) for (size_t i = order.size(); i > 1; --i) std::swap(order[i - 1], order[next_random() % (unsigned)i]);
    for (uintptr_t sid : order) {
      StreamQ& s = streams[sid];
      Job* j = s.q.front();
      if (j->is_wait) {
        if (streams[j->wait_stream].finished >= j->wait_seq) { s.q.pop_front(); s.finished = j->seq; delete j; any = true; }
        continue;
      }
      any |= step(j);
      if (j->done == j->total) { s.q.pop_front(); s.finished = j->seq; delete j; any = true; }
    }
    (void)any;   // no progress in a round = every workgroup is polling: their own bounds end that (a wait that nothing will satisfy expires, as on the device)
  }
  draining = false;
}
inline bool capture_active();
inline bool capture_member(uintptr_t s);
inline void capture_wait(uintptr_t waiter, uintptr_t on);
inline void stream_wait(uintptr_t waiter, uintptr_t on) {   // hipStreamWaitEvent(waiter, event recorded on `on` now)
  if (waiter == on) return;
  if (capture_active() && (capture_member(on) || capture_member(waiter))) { capture_wait(waiter, on); return; }
  if (streams[on].enqueued == streams[on].finished) return;
  Job* j = new Job(); j->is_wait = true; j->wait_stream = on; j->wait_seq = streams[on].enqueued; j->stream = waiter; j->seq = ++streams[waiter].enqueued;
  streams[waiter].q.push_back(j);
}

// ---- stream capture (hipGraph): while a graph is being captured, launches and stream waits on the streams that belong to the capture are RECORDED - kernel arguments by
// value, as hipGraph does - instead of queued; a launch of the graph queues copies of them: the origin stream's onto the launching stream, every forked stream's onto a
// stream of its own that exists for this graph only, with the recorded waits between them.
struct GraphEntry { uintptr_t stream; Job job; uintptr_t wait_on = 0; };
struct Graph { uintptr_t origin = 0; std::vector<GraphEntry> entries; std::vector<uintptr_t> members; };
inline std::map<uint64_t, Graph> graphs;
inline uint64_t capturing = 0;
inline bool in_capture(uintptr_t s) { if (!capturing) return false; for (uintptr_t m : graphs[capturing].members) if (m == s) return true; return false; }
inline bool capture_active() { return capturing != 0; }
inline bool capture_member(uintptr_t s) { return in_capture(s); }
inline void capture_wait(uintptr_t waiter, uintptr_t on) {
  Graph& g = graphs[capturing];
  if (!in_capture(on)) return;                       // waiting for work outside the capture: already ordered before the capture began (capture_begin drained)
  if (!in_capture(waiter)) g.members.push_back(waiter);   // a stream that waits for a capturing stream joins the capture (a fork)
  GraphEntry e; e.stream = waiter; e.wait_on = on; g.entries.push_back(e);
}
inline void capture_begin(uint64_t id, uintptr_t stream) { drain(); Graph& g = graphs[id]; g = Graph(); g.origin = stream; g.members.push_back(stream); capturing = id; }
inline void capture_end(uint64_t) { capturing = 0; }
inline void graph_launch(uint64_t id, uintptr_t on) {
  Graph& g = graphs[id];
  auto map_stream = [&](uintptr_t s) { return s == g.origin ? on : (uintptr_t)(0x4000000000000000ull | (id << 24) | (s & 0xffffff)); };
  for (uintptr_t m : g.members) if (m != g.origin) stream_wait(map_stream(m), on);   // a forked branch starts after what the launching stream has queued so far
  for (GraphEntry& e : g.entries) {
    const uintptr_t sid = map_stream(e.stream);
    if (e.wait_on) { stream_wait(sid, map_stream(e.wait_on)); continue; }
    Job* j = new Job(e.job);
    j->stream = sid; j->seq = ++streams[sid].enqueued;
    streams[sid].q.push_back(j);
  }
  for (uintptr_t m : g.members) if (m != g.origin) stream_wait(on, map_stream(m));   // ... and the launching stream goes on when all of them are done
  if (!on) drain();
}

template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream, Args... args) {
  const int n = (int)(block.x * block.y * block.z);
  if (n > MAX_THREADS) { fprintf(stderr, "emu: %d threads per block\n", n); abort(); }
  Job* j = new Job();
  j->grid = grid; j->block = block; j->lds = lds_bytes; j->total = grid.x * grid.y * grid.z; j->stream = (uintptr_t)stream;
  j->body = [=]() { kernel(args...); };
  if (in_capture(j->stream)) { GraphEntry e; e.stream = j->stream; e.job = *j; graphs[capturing].entries.push_back(e); delete j; return; }
  StreamQ& s = streams[j->stream];
  j->seq = ++s.enqueued;
  s.q.push_back(j);
  if (!stream) drain();   // the null stream: synchronous with everything
}
}  // namespace emu


// IL_EMU_LOG_LAUNCHES=<file>: one kernel name per launch (which kernels did a test reach?)
inline void emu_log_launch(const char* name) { static const char* f = getenv("IL_EMU_LOG_LAUNCHES"); if (f) { FILE* o = fopen(f, "a"); if (o) { fprintf(o, "%s\n", name); fclose(o); } } }
#define EMU_LAUNCH(kernel, grid, block, lds, stream, ...) (emu_log_launch(#kernel), emu::launch(kernel, dim3(grid), dim3(block), (size_t)(lds), (hipStream_t)(stream), __VA_ARGS__))

inline void __syncthreads() {
  emu::blk->fibers[emu::cur].state = emu::AT_BARRIER;
  emu::to_scheduler();
}
// every thread of the workgroup calls; non-zero if any thread's predicate is. Two alternating accumulators: a slot is cleared (by thread 0, behind the second barrier) before
// any thread can reach the call after next, which is the next user of that slot.
inline int __syncthreads_or(int pred) {
  emu::Block* b = emu::blk;
  const int t = emu::linear_tid(), slot = (int)(b->or_seq[t]++ & 1u);
  if (pred) b->or_val[slot] = 1;
  __syncthreads();
  const int r = b->or_val[slot];
  __syncthreads();
  if (t == 0) b->or_val[slot] = 0;
  return r;
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) <= 8, "emulated __shfl: at most 8 bytes");
  emu::Block* b = emu::blk;
  const int t = emu::linear_tid(), cap = b->cap;
  const unsigned k = b->shfl_seq[t]++;
  const size_t row = (size_t)(k % emu::SHFL_RING) * cap;
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  b->shfl_val[row + t] = raw; b->shfl_tag[row + t] = k;
  const int wave0 = t & ~63, s = wave0 + ((t - wave0) / width) * width + (src % width);
  while (b->shfl_tag[row + s] != k) emu::to_scheduler();   // the source lane has not reached this shuffle yet
  emu::to_scheduler();                                     // let every lane publish before any lane runs ahead and recycles the ring
  T out; memcpy(&out, &b->shfl_val[row + s], sizeof(T));
  return out;
}

// ---- wave-level builtins on the exchange above (all live lanes of the wave must reach the call, as on the device)
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
inline int __builtin_amdgcn_readlane(int v, int lane) {
  const unsigned char* all = emu::wave_exchange(&v, 4);
  int out; memcpy(&out, all + (size_t)lane * emu::COLL_BYTES, 4); return out;
}
inline int __builtin_amdgcn_readfirstlane(int v) {
  const int t = emu::linear_tid(), w0 = t & ~63;
  const unsigned char* all = emu::wave_exchange(&v, 4);
  for (int l = 0; l < 64; ++l) if (w0 + l < emu::block_threads && emu::lane_live(w0 + l)) { int out; memcpy(&out, all + (size_t)l * emu::COLL_BYTES, 4); return out; }
  return v;
}
// DPP: the source lane of `ctrl` within the lane's row of 16 (row_shr / row_ror / row_shl / quad_perm are what the kernels use); invalid source + !bound_ctrl -> `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = emu::linear_tid() & 63, row = lane & ~15, i = lane & 15;
  const unsigned char* all = emu::wave_exchange(&src, 4);
  int from = -1;
  if (ctrl >= 0x121 && ctrl <= 0x12f) from = row + ((i - (ctrl & 15)) & 15);                 // row_ror:n
  else if (ctrl >= 0x111 && ctrl <= 0x11f) from = i - (ctrl & 15) >= 0 ? row + i - (ctrl & 15) : -1;   // row_shr:n
  else if (ctrl >= 0x101 && ctrl <= 0x10f) from = i + (ctrl & 15) <= 15 ? row + i + (ctrl & 15) : -1;  // row_shl:n
  else if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);            // quad_perm
  else { fprintf(stderr, "emu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
  if (from < 0) return bound_ctrl ? 0 : old;
  int out; memcpy(&out, all + (size_t)from * emu::COLL_BYTES, 4); return out;
}
typedef float emu_f32x4 __attribute__((vector_size(16)));
// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] B[4x16]; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds D[4 (l >> 4) + reg][l & 15]; an fmaf chain over k
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  const int lane = emu::linear_tid() & 63, col = lane & 15, rg = lane >> 4;
  float ab[2] = {a, b};
  const unsigned char* all = emu::wave_exchange(ab, 8);
  auto A = [&](int i, int k) { float v; memcpy(&v, all + (size_t)(i + 16 * k) * emu::COLL_BYTES, 4); return v; };
  auto B = [&](int k, int j) { float v; memcpy(&v, all + (size_t)(j + 16 * k) * emu::COLL_BYTES + 4, 4); return v; };
  for (int r = 0; r < 4; ++r) {
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(A(4 * rg + r, k), B(k, col), acc);
    c[r] = acc;
  }
  return c;
}
inline unsigned long long __ballot(int pred) {
  const int t = emu::linear_tid(), w0 = t & ~63;
  const int p = pred != 0;
  const unsigned char* all = emu::wave_exchange(&p, 4);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) if (w0 + l < emu::block_threads && emu::lane_live(w0 + l)) { int v; memcpy(&v, all + (size_t)l * emu::COLL_BYTES, 4); if (v) m |= 1ull << l; }
  return m;
}
inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred); }
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (emu::linear_tid() & (width - 1)) ^ mask, width); }
template <class T> inline T __shfl_up(T v, unsigned delta, int width = 64) { const int l = emu::linear_tid() & (width - 1); const T o = __shfl(v, l >= (int)delta ? l - (int)delta : l, width); return o; }
template <class T> inline T __shfl_down(T v, unsigned delta, int width = 64) { const int l = emu::linear_tid() & (width - 1); return __shfl(v, l + (int)delta < width ? l + (int)delta : l, width); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }   // v_exp_f32 (the device's is within 1 ulp and flushes denormal results)
inline void __threadfence() {}
inline void __threadfence_system() {}
inline void __threadfence_block() {}
#define __builtin_amdgcn_s_sleep(x) emu::poll_yield()   // every polling loop of the kernels sleeps between two polls: the place where the workgroup steps aside
inline void emu_wave_barrier() { const int z = 0; (void)emu::wave_exchange(&z, 4); }   // lanes of a wave run one after the other here: a wave barrier has to be a real rendezvous
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_s_memtime() 0ull
inline unsigned long long emu_realtime() { static unsigned long long t = 1000; return ++t; }   // a monotonic stand-in for the 100 MHz device counter (launch stamps: begin < end)
#define __builtin_amdgcn_s_memrealtime() emu_realtime()
#define __builtin_amdgcn_s_barrier() __syncthreads()
namespace emu { inline unsigned xcc_id() { static const char* e = getenv("IL_EMU_XCC"); const unsigned l = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); return e ? (l * 2654435761u >> 7) & 7u : l & 7u; } }   // IL_EMU_XCC=scatter: a placement under which partners do not share an XCD
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __HIP_MEMORY_SCOPE_SYSTEM 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_WAVEFRONT 3
template <class T> inline T atomicAdd(T* p, T v) { const T old = *p; *p = old + v; return old; }
template <class T> inline T atomicMax(T* p, T v) { const T old = *p; if (v > old) *p = v; return old; }
template <class T> inline T atomicMin(T* p, T v) { const T old = *p; if (v < old) *p = v; return old; }
template <class T> inline T atomicExch(T* p, T v) { const T old = *p; *p = v; return old; }
// raw buffer resources: base pointer + byte offsets (the cache-policy bits mean nothing here)
struct __amdgpu_buffer_rsrc_t { char* base; };
template <class T> inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(T* base, int, int, int) { return {(char*)base}; }
typedef unsigned emu_u32x4 __attribute__((vector_size(16)));
inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.base + voff + soff, &v, 16); }
inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.base + voff + soff, &v, 4); }
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { emu_u32x4 v; memcpy(&v, r.base + voff + soff, 16); return v; }
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) { unsigned v; memcpy(&v, r.base + voff + soff, 4); return v; }
// the rest of the runtime API the emulated sources mention
typedef void* hipEvent_t;
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };
enum { hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3, hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }   // (csrc/launcher.hip's thread makes the caller's device current; the launcher itself is not driven on the emulator: its fibers are not thread-safe)
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "host-emulation"); p->multiProcessorCount = 256; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { emu::drain(); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { emu::drain(); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { emu::drain(); memcpy(d, s, n); return hipSuccess; }
#define HIP_SYMBOL(x) x
template <class T> inline hipError_t hipMemcpyFromSymbol(void* d, const T& sym, size_t n) { emu::drain(); memcpy(d, &sym, n); return hipSuccess; }
template <class T> inline hipError_t hipMemcpyToSymbol(T& sym, const void* s, size_t n) { emu::drain(); memcpy(&sym, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t*) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// device allocations and "inter-process" windows: one address space, so a window's IPC handle is its address
enum { hipDeviceMallocUncached = 3, hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
inline hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { *p = aligned_alloc(256, (bytes + 255) / 256 * 256); return *p ? hipSuccess : 2; }
template <class T> inline hipError_t hipMalloc(T** p, size_t bytes) { return hipExtMallocWithFlags((void**)p, bytes, 0); }
inline hipError_t hipMemset(void* p, int v, size_t bytes) { emu::drain(); memset(p, v, bytes); return hipSuccess; }
inline hipError_t hipFree(void* p) { emu::drain(); free(p); return hipSuccess; }
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof(*h)); memcpy(h->reserved, &p, sizeof(p)); return hipSuccess; }
inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, h.reserved, sizeof(*p)); return hipSuccess; }
inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { static uintptr_t next = 0x9000; next += 0x10; *s = (hipStream_t)next; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { static uintptr_t next = 0x7000; next += 0x10; *s = (hipStream_t)next; return hipSuccess; }
