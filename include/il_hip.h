/* il_hip.h -- C ABI of libil_hip.so: the MI355X (gfx950) off-policy update hot path of
 * Kaixhin/imitation-learning, rebuilt as hand-written HIP kernels.
 *
 * The reference has no FFI layer; its de-facto operator API is the set of Python callables
 * train.py imports (reference train.py:13-18).  Each entry point below names the reference
 * function it replaces (file:line under the reference repo).  The Python package mirrors the
 * reference signatures on top of this ABI through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 on success, a non-zero IL_ERR_* otherwise; il_last_error() gives text;
 *   - all pointers are DEVICE pointers unless the parameter name ends in _host;
 *   - nothing allocates, frees or synchronises: work is enqueued on `stream` (a hipStream_t) and is
 *     hipGraph-capturable; buffers are owned by the caller (torch tensors in the Python host layer);
 *   - all floating point is IEEE fp32 (parity target rtol 1e-5 vs the reference CPU path);
 *   - networks are "depth 2" MLPs Linear(in,H)-ReLU-Linear(H,H)-ReLU-Linear(H,out) stored as ONE flat
 *     fp32 vector in torch parameters() order: W1[H,in], b1[H], W2[H,H], b2[H], W3[out,H], b3[out]
 *     (reference models.py:48-69), H a multiple of 64, H <= 256 (one workgroup tile keeps two [16 x H] activation slabs in LDS);
 *   - Adam step counters live on the device (`il_adam.step[0]`): the gradient-producing entry point
 *     increments it, so a captured graph replays with the right bias correction.
 */
#ifndef IL_HIP_H
#define IL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IL_ABI_VERSION 1

enum { IL_OK = 0, IL_ERR_ARG = 1, IL_ERR_UNSUPPORTED = 2, IL_ERR_HIP = 3, IL_ERR_WORKSPACE = 4 };

/* flags */
#define IL_FLAG_GRADS_ONLY 1u /* write gradients to the *_grad arenas and skip the optimiser (data-parallel: all-reduce, then il_adam_step) */
#define IL_FLAG_TICK 2u       /* il_adam_step: increment the step counter first (stand-alone use) */
#define IL_FLAG_SAC_FORWARD_ONLY 4u /* il_sac_update: only the reward-independent forward kernels (actor on s and s', critics, targets) */
#define IL_FLAG_SAC_PREPARED 16u    /* the lane-ordered weight copies in the workspace match the parameters: il_sac_prepare() ran, or the previous
                                       il_sac_update / il_sac_dp_phase / il_sac_update_population call on this descriptor left them in step (its Adam and
                                       polyak epilogues update them) and nothing else touched the parameters since. Skips the re-ordering kernel. */
#define IL_FLAG_GAIL_CLOSE_EPOCH 32u /* il_gail_disc_step with il_sync counters: no il_gail_reward follows (il_sac_update_gather relabels inline): the AdamW
                                      workgroups report [IL_SYNC_PARAMS] and the last one closes the discriminator branch's epoch */
#define IL_FLAG_SAC_SKIP_FORWARD 8u /* il_sac_update: everything after them (the caller already ran FORWARD_ONLY on this batch) */
#define IL_FLAG_SAC_STAGED_ROWS 0x400u /* il_sac_update_gather: `ring` is the dense slab of this update's rows staged by il_gail_disc_step_draw_staged (no il_batch.gather) */
#define IL_FLAG_SAC_WAIT_INDICES 64u /* il_sac_update_gather: no sampling kernel precedes this call in its stream - wait on the device for [IL_SYNC_INDICES] (il_replay_draw_resident) */

typedef void* il_stream_t; /* hipStream_t */

const char* il_last_error(void);
int il_abi_version(void);
/* device the library was built for / sees: writes arch name (e.g. "gfx950") */
int il_device_info(char* name_host, int name_len, int* cu_count_host);

/* Per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline line).
 * il_trace_enable(1) starts recording every kernel launched through this library (do not enable under graph capture);
 * il_trace_report synchronises the device and writes "kernel_name launches total_ms" lines into buf_host. */
int il_trace_enable(int on);
int il_trace_report(char* buf_host, int len);
/* Launch stamps (always on; SURVEY.md 8d measurement, no reference line: the reference has no kernels): thread 0 of every workgroup of the headline schedule's kernels
 * stores the 100 MHz device-wide counter at its start and after its last wave (two fire-and-forget stores per workgroup; each launch overwrites the previous one's).
 * il_kernel_stamps synchronises the device and writes, per kernel id (IL_STAMP_*), {min begin, max begin, min end, max end, workgroups} of the LAST launch into
 * out_host [il_kernel_stamp_ids()][5] (ticks of 10 ns; zeros: not launched since il_kernel_stamps_clear). bench.py reads them after its timed graph replays. */
enum { IL_STAMP_GAIL_GRAD = 0, IL_STAMP_GAIL_REDUCE = 1, IL_STAMP_CHAIN = 2, IL_STAMP_DW_CRITIC = 3, IL_STAMP_POLICY_CRITIC = 4, IL_STAMP_DW_ACTOR = 5, IL_STAMP_GMMIL = 6, IL_STAMP_PWIL = 7 };
int32_t il_kernel_stamp_ids(void);
int il_kernel_stamps(uint64_t* out_host);
int il_kernel_stamps_clear(void);
/* the raw rows of one kernel id: out_host [il_kernel_stamp_workgroups()][4] = {begin, end, placement, 0} per workgroup of the last launch; placement = XCC_ID << 16 | the
 * shader-engine / shader-array / CU byte of HW_ID: tests assert from it that side-stream workgroups never share a CU with a pair-mode workgroup */
/* a stream restricted to the CUs of mask_host [words x 32 bits] (hipExtStreamCreateWithCUMask); experiment only */
int il_stream_create_cu_mask(const uint32_t* mask_host, int32_t words, void** stream_out);
int il_stream_destroy(void* stream);
int32_t il_kernel_stamp_workgroups(void);
int il_kernel_stamp_rows(int32_t kernel_id, uint64_t* out_host);

/* The on-chip noise of the update kernels as a function: out[i] = draw #i of noise stream `stream_id` at update counter `ctr` under key `noise_seed`,
 * evaluated by the same device functions the kernels call when their eps pointer is NULL (Philox4x32-10 keyed by noise_seed, counter words
 * {i, ctr, stream_id, 0}; normals by Box-Muller from words 0, 1; uniforms = (word 0 >> 8) / 2^24 like torch.rand). The counter of an il_sac / il_disc
 * descriptor (`noise_counter`) starts at 0 and is advanced by 1 at the end of every il_sac_actor_step, so update #k of a learner consumed ctr = k:
 *   IL_NOISE_EPS_NEXT [B*A] = the N(0,1) draws of policy.sample() on s' (training.py:21), IL_NOISE_EPS_CUR [B*A] = rsample on s (training.py:35),
 *   IL_NOISE_GP [B] = the U(0,1) of the gradient penalty (training.py:118), IL_NOISE_MIX [B] = Mixup coefficients at alpha = 1 (training.py:106),
 *   IL_NOISE_ACT [n*A] = actor(state).sample() of il_actor_act / il_act_step (train.py:152; there `ctr` is the call's noise_offset).
 * This is how a captured run is recorded for replay through a CPU oracle (tests/test_timed_path_oracle.py).
 *   IL_NOISE_DROP_IN / _HID / _HID2 = the U(0,1) behind the DRIL dropout keep-masks (keep = u >= p) of the input [rows*S] and the hidden layers [rows*H], rows in
 *   repeat_interleave order for the 5-member ensemble; ctr = the call's noise_offset (+ *il_dril.noise_counter). */
enum { IL_NOISE_EPS_NEXT = 1, IL_NOISE_EPS_CUR = 2, IL_NOISE_GP = 3, IL_NOISE_ACT = 4, IL_NOISE_DROP_IN = 5, IL_NOISE_DROP_HID = 6, IL_NOISE_MIX = 7, IL_NOISE_DROP_HID2 = 11 };
int il_noise_fill(uint64_t noise_seed, uint32_t ctr, uint32_t stream_id, int64_t n, float* out, il_stream_t stream);
/* n Beta(alpha, alpha) draws (Mixup with mixup_alpha != 1, training.py:105-107) of the Mixup stream at the update counter *ctr_dev, read on the DEVICE: the launch can sit
 * in a captured update ahead of il_gail_disc_step, which takes `out` as il_gail_extra.eps_mix. Philox + Marsaglia-Tsang: equal in distribution to torch's CPU Beta sampler,
 * not in bits; a pure function of (noise_seed, *ctr_dev, index, alpha). */
int il_noise_fill_beta(uint64_t noise_seed, const uint32_t* ctr_dev, float alpha, int64_t n, float* out, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A batch of transitions as strided fp32 fields: row b of field f is f + b*ld_f.
 * Mirrors the `transitions` dict of reference memory.py:58-63 (keys states, actions, rewards,
 * next_states, terminals, weights, absorbing); `step`/`timeouts` are not read by any kernel.
 * ------------------------------------------------------------------------------------------ */
typedef struct il_batch {
  const float *states, *actions, *rewards, *next_states, *terminals, *weights, *absorbing;
  int32_t ld_states, ld_actions, ld_rewards, ld_next_states, ld_terminals, ld_weights, ld_absorbing;
  int32_t n; /* rows */
  /* Optional row indirection: with gather != NULL the field pointers describe a replay RING (ring row 0, ld = ring row floats) and batch row r
   * of every field is ring row gather[r], clamped to [0, gather_capacity) like il_replay_gather. Honoured by il_gail_disc_step, il_gail_reward and
   * il_sac_update_gather only (they then need an update's index DRAW but not its gather); every other entry point rejects it. */
  const int32_t* gather;
  int64_t gather_capacity;
} il_batch;

/* ------------------------------------------------------------------------------------------
 * Replay ring (reference memory.py:12-68).  HBM layout: row-major [capacity][row] fp32 with
 *   row = il_ring_row_floats(S, A) = roundup4(2S + A + 5) and fields at
 *   states @0, actions @S, next_states @S+A, rewards @2S+A, terminals +1, timeouts +2, weights +3, step +4.
 * A sampled batch is the same packed layout [n][row], so every il_batch field is a strided view.
 * ------------------------------------------------------------------------------------------ */
int32_t il_ring_row_floats(int32_t state_dim, int32_t action_dim);

/* memory.py:40-44 `append` (n_rows consecutive slots starting at `cursor`, wrapping at capacity);
 * rows_src is device-accessible memory (device or pinned host) holding n_rows packed rows. */
int il_replay_write_rows(float* ring, int64_t capacity, int32_t row_floats, int64_t cursor, const float* rows_src, int32_t n_rows,
                         il_stream_t stream);
/* memory.py:65-68 `wrap_for_absorbing_states`: rewrites row `last` (next_state <- absorbing state, terminal <- 0)
 * and writes the absorbing->absorbing row (step copied from `last`) at `cursor`. */
int il_replay_wrap_absorbing(float* ring, int64_t capacity, int32_t state_dim, int32_t action_dim, int64_t last, int64_t cursor,
                             il_stream_t stream);
/* models.py:287-290 mix_expert_agent_transitions and models.py:293-318 RewardRelabeller.resample_and_relabel on packed batch rows
 * (il_ring_row_floats floats per row): rows[0:n_expert) <- expert_rows[0:n_expert) (every field), then rewards relabelled:
 *   label 0: untouched (plain mixing);  1 (SQIL): expert 1, policy 0;
 *   2 (AdRIL): expert `reward_expert` (= 1/|expert trajectories|), policy -[round_num > ceil(row.step / update_freq)] / max(policy_trajectories, 1). */
int il_batch_mix_relabel(float* rows, const float* expert_rows, int32_t n, int32_t state_dim, int32_t action_dim, int32_t n_expert, int32_t label,
                         int32_t update_freq, int64_t round_num, float reward_expert, int64_t policy_trajectories, il_stream_t stream);
/* The same with the per-update quantities read from DEVICE memory, so that the launch can sit in a captured graph: dyn = int64[3] {n_expert (balanced AdRIL alternates
 * all-expert / all-policy batches), round_num = ceil(step / update_freq), policy_trajectories}; the caller refreshes it with a stream-ordered copy before each replay. */
int il_batch_mix_relabel_dyn(float* rows, const float* expert_rows, int32_t n, int32_t state_dim, int32_t action_dim, int32_t label, int32_t update_freq, float reward_expert,
                             const int64_t* dyn, il_stream_t stream);
/* memory.py:58-63 `sample` gather step: out[i] = ring[idx[i]] for i < n (packed rows, coalesced 16-B lanes). */
int il_replay_gather(const float* ring, int64_t capacity, int32_t row_floats, const int32_t* idx, int32_t n, float* out_rows,
                     il_stream_t stream);

/* memory.py:51-59 index draws, bit-exact with numpy's legacy MT19937 `np.random.randint(0, high)` stream plus the
 * reference's rejection of slot (idx-1) % size.  HOST function; state_host = 625 uint32 (624 words + position). */
int il_mt19937_seed(uint32_t* state_host, uint32_t seed);
int il_mt19937_sample_indices(uint32_t* state_host, int32_t n, int64_t size, int64_t idx, int32_t full, int32_t* out_host);
/* n plain `np.random.randint(0, high)` draws from the same stream (reference environments.py:113 consumes it for subsampling offsets). */
int il_mt19937_randint(uint32_t* state_host, int64_t high, int32_t n, int32_t* out_host);
/* Same stream, generated ON the device (state_dev = 625 uint32 in HBM) so a captured update needs no H2D copy.
 * ring_state_dev = {int64 idx, int64 full, int64 size}. */
int il_mt19937_sample_indices_device(uint32_t* state_dev, const int64_t* ring_state_dev, int32_t n, int32_t* out_dev, il_stream_t stream);
/* train.py:173 `memory.sample(B), expert_memory.sample(B)` in ONE launch: n draws for ring A, then n for ring B (same stream,
 * same order as the reference), then both row gathers. Ring B may be NULL (algorithm=SAC). rows_a = rows_b = NULL: the draw only
 * (the consumers then read the rings through il_batch.gather). */
int32_t il_replay_gather_workgroups(int32_t n, int32_t row_floats_a, int32_t row_floats_b); /* row_floats_b = 0 without ring B */
int il_replay_sample_device(uint32_t* state_dev, int32_t n, const int64_t* ring_state_a, const float* ring_a, int64_t capacity_a, int32_t row_floats_a,
                            int32_t* idx_a, float* rows_a, const int64_t* ring_state_b, const float* ring_b, int64_t capacity_b, int32_t row_floats_b,
                            int32_t* idx_b, float* rows_b, int64_t* sync, il_stream_t stream);

/* The index draws of an update as a RESIDENT launch (train.py:173 draws only; consumers read the rings through il_batch.gather): enqueue it on the discriminator
 * branch's stream, which has no dependency on the main stream, so the kernel is running - its generator state already in LDS - before the previous update has
 * finished. It waits on the device until [IL_SYNC_MAIN_EPOCH] equals the number of draws made so far ([IL_SYNC_INDICES]): the previous update no longer reads the
 * index arrays and every append that precedes this update in stream order has moved the ring cursor; then it draws and signals [IL_SYNC_INDICES].
 * il_sac_update_gather(IL_FLAG_SAC_WAIT_INDICES) on the main stream waits for that signal instead of for a sampling kernel ahead of it in its own stream: the draw
 * (one workgroup, ~7 us as a launch of its own) leaves the critical path of the update except for its ~2 us of arithmetic. Ring B may be NULL. */
int il_replay_draw_resident(uint32_t* state_dev, int32_t n, const int64_t* ring_state_a, int32_t* idx_a, const int64_t* ring_state_b, int32_t* idx_b, int64_t* sync,
                            il_stream_t stream);

/* Population axis: per-learner arguments of il_replay_sample_device, as a device array. */
typedef struct il_sample_args {
  uint32_t* state;                 /* this learner's MT19937 state (625 uint32, device) */
  const int64_t* ring_state_a; const float* ring_a; int64_t capacity_a; int32_t row_floats_a; int32_t* idx_a; float* rows_a;
  const int64_t* ring_state_b; const float* ring_b; int64_t capacity_b; int32_t row_floats_b; int32_t* idx_b; float* rows_b;   /* ring_b may be NULL */
} il_sample_args;
int il_replay_sample_population(const il_sample_args* args_dev, int32_t n_learners, int32_t n, int32_t max_row_floats, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser state (torch.optim.AdamW / Adam single-tensor step; reference train.py:66,84,95).
 * ------------------------------------------------------------------------------------------ */
typedef struct il_adam {
  float *m, *v;  /* exp_avg, exp_avg_sq */
  int32_t* step; /* device buffer of >= 16 int32: [0] = number of steps taken; [4..11] = fp32 constants of that step (written by the library) */
  double lr, beta1, beta2, eps, weight_decay; /* Python-float hyper-parameters (1 - beta etc. are formed in double like torch does);
                                                  weight_decay != 0 => decoupled decay p *= 1 - lr*wd */
} il_adam;

/* p <- AdamW(p, g) elementwise for n parameters using t = *opt.step (after optional IL_FLAG_TICK). */
int il_adam_step(float* p, const float* g, const il_adam* opt, int64_t n, uint32_t flags, il_stream_t stream);
/* models.py:79-81 update_target_network: target <- tau*target + (1-tau)*param */
int il_polyak(float* target, const float* param, int64_t n, double tau, il_stream_t stream);

/* Device-side hand-off for an update whose discriminator branch and SAC forward run on two streams (one hipGraph). `sync` = device
 * int64[IL_SYNC_SLOTS], zero-initialised once, shared by the il_sac / il_disc descriptors and il_replay_sample_device of ONE learner:
 *   [IL_SYNC_ROWS]       += 1 per finished gather workgroup        -> k_gail_grad waits for (side_epoch + 1) * [IL_SYNC_GATHER_WGS]
 *   [IL_SYNC_REWARDS]    += 1 per finished reward workgroup        -> k_critic_bwd waits for (main_epoch + 1) * ceil(batch / 16)
 *   [IL_SYNC_SIDE_EPOCH] += 1 when a reward relabel has finished;  [IL_SYNC_MAIN_EPOCH] += 1 at the end of the actor step
 *   [IL_SYNC_TIMEOUTS]   += 1 whenever a bounded wait gave up (must stay 0: check it on the host after the first update)
 *   [IL_SYNC_GATHER_WGS] = il_replay_gather_workgroups(n, row_floats_a, row_floats_b), written by the caller when it creates the buffer.
 *   [IL_SYNC_SPIN]       = polls before a bounded wait of this learner gives up; 0 = the built-in ~1 s (IL_SYNC_SPIN_LIMIT). Written by the caller; shares the read-only
 *                          line of [IL_SYNC_GATHER_WGS]. A data-parallel rank whose waits sit downstream of a gradient exchange raises it above the exchange's own bound.
 *   [IL_SYNC_HOST_FLAG]  = address of a host-mapped (pinned) int64, or 0: a wait that gives up ALSO stores the new [IL_SYNC_TIMEOUTS] value there (system scope), so the host
 *                          loop can notice an expired wait by reading its own memory every step - no synchronisation, no copy node in the captured update. Same read-only line.
 *   [IL_SYNC_INDICES]    += 1 per finished index draw              -> k_gail_grad on il_batch.gather batches waits for side_epoch + 1 (not for the rows)
 *   [IL_SYNC_CHAIN_DONE] += 1 per finished workgroup of the forward / critic-loss launch (k_sac_chain[_pair]: the last reader of an update's index arrays on the main
 *                          stream); [IL_SYNC_CHAIN_WGS] = that launch's grid size, published by the launch itself. The resident sampler draws the NEXT update's indices as
 *                          soon as [IL_SYNC_CHAIN_DONE] >= draws so far x [IL_SYNC_CHAIN_WGS] (round 5: ~30 us before that update's first kernel needs them) instead of
 *                          waiting for [IL_SYNC_MAIN_EPOCH]; the discriminator workgroups, which read the Philox counter the actor step advances, still wait for the epoch.
 *   [IL_SYNC_PARAMS]     += 1 per finished AdamW(discriminator) workgroup (IL_FLAG_GAIL_CLOSE_EPOCH) -> the inline relabel of il_sac_update_gather waits
 *                          for (main_epoch + 1) * il_gail_step_workgroups()
 *   [IL_SYNC_OV_TICKET + stage] += 1 (release, no returned value) by every retiring workgroup of that stage's launch (il_sac_update_gather_overlap only). The next stage's
 *                          launch sits on the OTHER stream and is resident before this one ends. Its workgroup 0 - the only poller of the ticket line - waits for
 *                          ticket >= the producer's grid behind its independent prologue, CLOSES the stage ([IL_SYNC_OV_EPOCH + stage] = launch number, ticket = 0) and
 *                          stores the launch number into [IL_SYNC_OV_FLAGS + stage, w] for every workgroup w of its own launch; workgroup w polls that line of its own
 *                          (160 workgroups polling ONE line were measured to slow every kernel on the chip down: the line's memory channel queues).
 *                          il_sac_overlap_enter() sets the four epochs and every flag to [IL_SYNC_MAIN_EPOCH]: a stage's own epoch is also the number of finished updates.
 *   [IL_SYNC_POISON]     != 0 once a bounded wait of this learner has given up: the optimiser epilogues (k_dw_adam, k_gail_reduce) of every later launch skip their
 *                          stores - an update whose hand-off expired never reaches the weights - until the host clears the word (il_sync_clear_poison). Same read-only line
 *                          as [IL_SYNC_SPIN].
 * With it the two branches need no stream dependency between the gather and the critic loss (fork at the start of the update, join at
 * its end). NULL everywhere = plain stream ordering (the caller serialises or uses events). */
/* il_sync_probe: [IL_SYNC_PROBE_FLAG], [IL_SYNC_PROBE_EPOCH] of the same buffer; enqueue setter = 0 on the side stream first, then setter = 1 on the main stream. */
int il_sync_probe(int64_t* sync, int32_t setter, il_stream_t stream);
/* Every counter sits on its own 128-byte line (IL_SYNC_STRIDE int64 apart): agent-scope atomics and polls execute at the memory side, where accesses to one line
 * serialise - ~150 polling workgroups on [IL_SYNC_INDICES] must not queue in front of the signals on [IL_SYNC_PARAMS]. il_sync_layout() reports the indices to a
 * host that does not compile this header. */
#ifndef IL_SYNC_STRIDE
#define IL_SYNC_STRIDE 16
#endif
enum { IL_SYNC_ROWS = 0, IL_SYNC_REWARDS = 1 * IL_SYNC_STRIDE, IL_SYNC_SIDE_EPOCH = 2 * IL_SYNC_STRIDE, IL_SYNC_MAIN_EPOCH = 3 * IL_SYNC_STRIDE, IL_SYNC_TIMEOUTS = 4 * IL_SYNC_STRIDE,
       IL_SYNC_GATHER_WGS = 5 * IL_SYNC_STRIDE, IL_SYNC_PROBE_FLAG = 6 * IL_SYNC_STRIDE, IL_SYNC_PROBE_EPOCH = 7 * IL_SYNC_STRIDE, IL_SYNC_INDICES = 8 * IL_SYNC_STRIDE,
       IL_SYNC_PARAMS = 9 * IL_SYNC_STRIDE, IL_SYNC_CHAIN_DONE = 10 * IL_SYNC_STRIDE, IL_SYNC_CHAIN_WGS = 11 * IL_SYNC_STRIDE, IL_SYNC_SPIN = 5 * IL_SYNC_STRIDE + 1,
       IL_SYNC_HOST_FLAG = 5 * IL_SYNC_STRIDE + 2, IL_SYNC_POISON = 5 * IL_SYNC_STRIDE + 3,
       IL_SYNC_OV_EPOCH = 12 * IL_SYNC_STRIDE,  /* + stage * IL_SYNC_STRIDE, stage = IL_OV_CHAIN .. IL_OV_DWA (lines 12 .. 15) */
       IL_SYNC_OV_TICKET = 16 * IL_SYNC_STRIDE, /* + stage * IL_SYNC_STRIDE (lines 16 .. 19) */
       IL_SYNC_OV_FLAGS = 20 * IL_SYNC_STRIDE,  /* + (stage * IL_OV_MAX_GRID + workgroup) * IL_SYNC_STRIDE: one line per WAITING workgroup of the stage's consumer launch */
       IL_OV_MAX_GRID = 256,
       IL_SYNC_SLOTS = (20 + 4 * 256) * IL_SYNC_STRIDE };
/* Stages of the SAC branch when its four launches alternate over two streams (il_sac_update_gather_overlap): forward / critic loss, critic optimiser,
 * policy / critic, actor optimiser + tail. */
enum { IL_OV_CHAIN = 0, IL_OV_DWC = 1, IL_OV_PC = 2, IL_OV_DWA = 3 };
/* out[0] = IL_SYNC_SLOTS (int64 elements to allocate and zero), out[1] = IL_SYNC_TIMEOUTS, out[2] = IL_SYNC_GATHER_WGS, out[3] = IL_SYNC_STRIDE, out[4] = IL_SYNC_SPIN, out[5] = IL_SYNC_HOST_FLAG */
void il_sync_layout(int32_t* out);
/* the same, open-ended: out[0 .. n) = il_sync_layout's six, then [6] IL_SYNC_POISON, [7] IL_SYNC_OV_EPOCH, [8] IL_SYNC_OV_TICKET, [9] IL_SYNC_MAIN_EPOCH */
void il_sync_layout_ex(int32_t* out, int32_t n);

/* ------------------------------------------------------------------------------------------
 * SAC (reference training.py:14-54 `sac_update`; models.py:84-141 SoftActor / TwinCritic).
 * ------------------------------------------------------------------------------------------ */
typedef struct il_sac {
  int32_t state_dim, action_dim, hidden, batch;
  float *actor;     /* [Pa]   Pa = il_mlp_numel(S, H, 2A)                         */
  float *critic;    /* [2*Ps] critic_1 | critic_2 at stride Ps = il_mlp_stride(S+A, H, 1) */
  float *target;    /* [2*Ps] target critics, same layout                           */
  float *log_alpha; /* [1]                                                        */
  float *actor_grad, *critic_grad, *alpha_grad; /* [Pa], [2*Ps], [1]: filled when IL_FLAG_GRADS_ONLY */
  il_adam actor_opt, critic_opt, alpha_opt;
  float discount, entropy_target;
  double polyak;
  float* workspace;          /* >= il_sac_workspace_floats() floats */
  int64_t workspace_floats;
  uint64_t noise_seed;       /* Philox4x32-10 key when eps pointers are NULL */
  uint32_t* noise_counter;   /* device uint32, incremented once per il_sac_actor_step */
  float *out_logp, *out_q;   /* optional [B] outputs (training.py:54) used when the call passes NULL output pointers (population path) */
  int64_t* sync;             /* il_sync counters or NULL (see below) */
  float* debug_masks;        /* NULL, or [10][batch][hidden] floats (tests only): 1 where a hidden PRE-activation was > 0 as this path computed it, for the passes that are
                              * back-propagated - [0,1] actor(s) layers 1,2; [2 + 2k, 3 + 2k] critic_k(s, a); [6 + 2k, 7 + 2k] the updated critic_k(s, a~). Two correct
                              * fp32 evaluations disagree on the sign of a pre-activation within rounding of 0; the oracle replays an update WITH these masks
                              * (oracle/nets.py), which isolates that effect from everything else (tests/test_timed_path_oracle.py) */
} il_sac;

int64_t il_mlp_numel(int32_t in_dim, int32_t hidden, int32_t out_dim);
/* distance in floats between consecutive networks of one arena (critic_1 -> critic_2): numel rounded up to 4 so that
 * every network starts on a 16-byte boundary; the pad floats are never read as parameters. */
int64_t il_mlp_stride(int32_t in_dim, int32_t hidden, int32_t out_dim);
int64_t il_sac_workspace_floats(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t batch);

/* training.py:19-31: target values, twin-critic weighted MSE, backward, AdamW(critic).
 * eps_next [B,A] = the N(0,1) draws of `policy.sample()` on s' (NULL => on-chip Philox). */
int il_sac_critic_step(const il_sac* d, const il_batch* batch, const float* eps_next, uint32_t flags, il_stream_t stream);
/* training.py:34-52: policy loss through the UPDATED critic, AdamW(actor), Adam(log_alpha), polyak.
 * eps_cur [B,A] = rsample noise on s.  out_logp/out_q [B] = the (log_probs, Q_values) the reference returns (:54). */
int il_sac_actor_step(const il_sac* d, const il_batch* batch, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags,
                      il_stream_t stream);
/* Data-parallel schedule of one update in four phases (0 forward, 1 critic gradients, 2 AdamW(critic) + actor/alpha gradients,
 * 3 AdamW(actor) + Adam(log_alpha) + polyak); the caller all-reduces critic_grad after phase 1 and actor_grad|alpha_grad after phase 2.
 * Same kernels as il_sac_update, Philox noise. */
int il_sac_dp_phase(const il_sac* d, const il_batch* batch, int32_t phase, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream);
/* DP tail after the all-reduce of actor_grad/alpha_grad: AdamW(actor) + Adam(log_alpha) + polyak (same kernels, no recompute). */
int il_sac_apply_actor_grads(const il_sac* d, il_stream_t stream);
int il_sac_apply_critic_grads(const il_sac* d, il_stream_t stream);
/* Population axis (the reference's own usage: 10-seed sweeps, Ax trials -- README.md:96-99, train_all.py:26): n_learners independent
 * learners with identical shapes advanced by the same launches. descs_dev / batches_dev are DEVICE arrays of descriptors (each learner
 * owns its arenas, optimiser state, workspace, noise counter, batch); shape_host supplies the common dimensions. Philox noise only. */
int il_sac_update_population(const il_sac* descs_dev, const il_batch* batches_dev, int32_t n_learners, const il_sac* shape_host, uint32_t flags,
                             il_stream_t stream);
/* Builds the lane-ordered copies of the hidden-layer weights in the workspace (k_repack). il_sac_update does this itself unless told
 * IL_FLAG_SAC_PREPARED; exposing it lets a caller overlap it with the replay sampling. */
int il_sac_prepare(const il_sac* d, il_stream_t stream);
/* whole training.py:14-54 in one call (critic step then actor step) */
int il_sac_update(const il_sac* d, const il_batch* batch, const float* eps_next, const float* eps_cur, float* out_logp, float* out_q,
                  uint32_t flags, il_stream_t stream);
/* il_sac_update for a batch that has been DRAWN but not gathered: `ring` is an il_batch with `gather` set (the replay ring and this update's indices),
 * `rows` describes the destination of the gather (packed rows: states at offset 0, ld_states = row floats) and is what the later kernels of the
 * update read. The forward / critic-loss launch (k_sac_chain) reads its rows straight from the ring while extra workgroups of the same launch
 * write `rows` (and signal [IL_SYNC_ROWS], il_sac_chain_gather_workgroups() times), so no gather kernel precedes the update.
 * rewards: optional dense [B] rewards that replace the ring's (train.py:194 relabelled rewards); NULL = the ring's reward field. Whole updates of a
 * single learner only; IL_ERR_UNSUPPORTED when the launch cannot be co-resident (then gather first and call il_sac_update).
 * IL_FLAG_GRADS_ONLY (data-parallel schedule): stops after the critic gradients (critic_grad arena): the caller all-reduces them and continues with
 * il_sac_dp_phase(rows, 2) / (rows, 3). */
int il_sac_update_gather(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const struct il_disc* relabel, float* rewards_out,
                         const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream);
/* relabel != NULL (needs d->sync; the discriminator must be stepped by il_gail_disc_step(..., IL_FLAG_GAIL_CLOSE_EPOCH) on another stream): the rewards
 * are predict_reward(s, a) of that discriminator (models.py:177-180, train.py:192-194), computed per 16-row tile INSIDE the critic-loss workgroups as
 * soon as [IL_SYNC_PARAMS] says its AdamW step is done - no relabel kernel, no reward hand-off; written to rewards_out [B] if given. Same code and
 * thread mapping as il_gail_reward, so the values are bit-identical. IL_ERR_UNSUPPORTED for state_only discriminators or ones too large for the
 * workgroup's spare LDS (then relabel with il_gail_reward and pass `rewards`). */
/* il_sac_update_gather with the SAC branch's four launches ALTERNATING over two streams (round 6; reference train.py:173-203, training.py:26-31,34-42,52 - the same update):
 *   stream_a: forward / critic loss (k_sac_chain_pair) -> policy / critic (k_policy_critic_pair);   stream_b: critic optimiser (k_dw_adam) -> actor optimiser + tail (k_dw_adam).
 * Same-stream order gives the two-hop dependencies; the one-hop dependency is a device counter: every launch is dispatched while its predecessor still runs, does what does
 * not depend on it (row gathers through the indices, the optimiser's p / m / v streams, the target step, the critics' forward) and then waits for
 * [IL_SYNC_OV_EPOCH + predecessor] (bounded like every device-side wait; an expired wait poisons the learner: [IL_SYNC_POISON]). No stream edge is created by this call.
 * Needs d->sync, the pair-mode shape (hidden 256, round_up16(S + A) <= 64) and the block form of the optimiser launches (batch % 128 == 0): IL_ERR_UNSUPPORTED otherwise -
 * fall back to il_sac_update_gather. Not with IL_FLAG_GRADS_ONLY. The caller (1) calls il_sac_overlap_enter() on stream_a and orders stream_b behind it (an event) before the
 * FIRST overlapped update that follows anything else on this descriptor, (2) orders stream_a behind stream_b (an event) before it reads results or issues any other il_sac_*
 * call. Consecutive overlapped updates need neither. Bit-identical to il_sac_update_gather. */
int il_sac_update_gather_overlap(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const struct il_disc* relabel, float* rewards_out,
                                 const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream_a, il_stream_t stream_b);
/* one tiny launch on `stream`: [IL_SYNC_OV_EPOCH + 0..3] = [IL_SYNC_MAIN_EPOCH], [IL_SYNC_OV_TICKET + 0..3] = 0 */
int il_sac_overlap_enter(const il_sac* d, il_stream_t stream);
/* [IL_SYNC_POISON] = 0 and [IL_SYNC_TIMEOUTS] = 0 (stream-ordered): the host has seen the expired wait and restored a consistent state (e.g. reloaded a checkpoint) */
int il_sync_clear_poison(int64_t* sync, il_stream_t stream);
int32_t il_gail_step_workgroups(const struct il_disc* d); /* AdamW workgroups of il_gail_disc_step = what [IL_SYNC_PARAMS] advances by per step */
int32_t il_sac_chain_gather_workgroups(int32_t batch, int32_t row_floats, int32_t hidden);
/* Bounded in-launch waits (tile counters of the chained forward / critic-loss launch and of the policy helpers) that gave up since the last k_repack of this workspace (= the
 * first update after creation, or any update without IL_FLAG_SAC_PREPARED): must be 0 (they cannot expire while the launch is co-resident or dispatched in block order). Synchronous copy; out_host[0]. */
int il_sac_handoff_timeouts(const il_sac* d, uint32_t* out_host);

/* training.py:57-64 behavioural_cloning_update + models.py:97-99 SoftActor.log_prob (clamp, atanh). */
int il_bc_step(float* actor, float* actor_grad, const il_adam* opt, int32_t state_dim, int32_t action_dim, int32_t hidden,
               const il_batch* batch, float* workspace, int64_t workspace_floats, float* out_loss_partials, uint32_t flags,
               il_stream_t stream); /* workspace >= il_sac_workspace_floats(S,A,H,B); out_loss_partials [B/16] (sum/B = loss) or NULL */
/* train.py:152 `actor(state).sample()` / models.py:101-102 get_greedy_action for n states: out_action [n,A].
 * eps [n,A] or NULL (Philox with noise_seed/noise_offset); greedy != 0 => tanh(mean). out_logp may be NULL. */
int il_actor_act(const float* actor, int32_t state_dim, int32_t action_dim, int32_t hidden, const float* states, int32_t ld_states,
                 int32_t n, const float* eps, uint64_t noise_seed, uint32_t noise_offset, int32_t greedy, float* out_action,
                 float* out_logp, il_stream_t stream);
/* models.py:97-99 SoftActor.log_prob(state, action) for n rows (action clamped like the reference): the log pi(a|s) of subtract_log_policy. */
int il_actor_log_prob(const float* actor, int32_t state_dim, int32_t action_dim, int32_t hidden, const float* states, int32_t ld_states,
                      const float* actions, int32_t ld_actions, int32_t n, float* out_logp, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * General actor / critic shapes (models.py:48-69 `_create_fcnn`: any depth, activation relu / tanh / sigmoid): the networks outside the fused kernels above (depth 2, ReLU,
 * hidden <= 256, 2A <= 16), composed a layer at a time from exact-fp32 MFMA tiles (csrc/general.hip). Parameters in torch `parameters()` order for `depth` hidden layers of
 * `hidden` units; twin critics at il_mlp_stride_general floats. activation: 0 relu, 1 tanh, 2 sigmoid. depth 1-8, widths <= 2048. Per-function path only (no captured plan).
 * ------------------------------------------------------------------------------------------ */
int64_t il_mlp_numel_general(int32_t in_dim, int32_t hidden, int32_t depth, int32_t out_dim);
int64_t il_mlp_stride_general(int32_t in_dim, int32_t hidden, int32_t depth, int32_t out_dim);
int64_t il_sac_workspace_floats_general(int32_t state_dim, int32_t action_dim, int32_t actor_hidden, int32_t actor_depth, int32_t critic_hidden, int32_t critic_depth, int32_t batch);
int64_t il_actor_workspace_floats_general(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t n);
/* training.py:14-54 (il_sac_update) on an il_sac whose arenas hold networks of these shapes (d->hidden = the actor's width; the actor and the critics are configured
 * separately, like reinforcement.actor / reinforcement.critic) and whose workspace has il_sac_workspace_floats_general floats.
 * flags: 0 or IL_FLAG_GRADS_ONLY (gradients into actor_grad / critic_grad / alpha_grad, no optimiser step, no target update). */
int il_sac_update_general(const il_sac* d, const il_batch* batch, int32_t actor_depth, int32_t actor_activation, int32_t critic_hidden, int32_t critic_depth, int32_t critic_activation,
                          const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, il_stream_t stream);
/* il_actor_act / il_actor_log_prob / il_bc_step (models.py:90-102, training.py:57-64) for those shapes; workspace >= il_actor_workspace_floats_general(S, A, H, depth, n).
 * il_bc_step_general: out_loss [1] = mean(w * -log pi) or NULL. */
int il_actor_act_general(const float* actor, int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t activation, const float* states, int32_t ld_states, int32_t n,
                         const float* eps, uint64_t noise_seed, uint32_t noise_offset, int32_t greedy, float* out_action, float* out_logp, float* workspace, int64_t workspace_floats,
                         il_stream_t stream);
int il_actor_log_prob_general(const float* actor, int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t activation, const float* states, int32_t ld_states,
                              const float* actions, int32_t ld_actions, int32_t n, float* out_logp, float* workspace, int64_t workspace_floats, il_stream_t stream);
int il_bc_step_general(float* actor, float* actor_grad, const il_adam* opt, int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t activation, const il_batch* batch,
                       float* workspace, int64_t workspace_floats, float* out_loss, uint32_t flags, il_stream_t stream);

/* One environment step of the acting worker (train.py:151-168) as ONE launch:
 *   [memory.append of the pending transition, memory.py:40-44] + [wrap_for_absorbing_states, memory.py:65-68] +
 *   [actor(state).sample(), models.py:90-94], with the action returned through host-pinned, device-mapped memory.
 * mailbox: il_act_mailbox_floats(S,A) floats of pinned host memory, Sp = roundup4(S), Ap = roundup4(A):
 *   host -> device  [0] commit word = sequence * 64 + IL_ACT_* flags (as a float, < 2^23), written LAST: it publishes the post
 *                   [2] reward  [3] terminal  [4] timeout  [5] step
 *                   [8, 8+S)       next_state of the pending transition
 *                   [8+Sp, 8+Sp+S) observation to act on (== next_state unless the episode ended and the env was reset)
 *   device -> host  [8+2Sp, +A)    action;   [8+2Sp+Ap] echo of the commit word, stored last with system-scope release
 *                   (the host spins on it instead of synchronising the stream).
 * carry (device, S+A+4 floats, zero-initialised): state | action of the pending transition written by the previous call, then the
 *   commit word of the last appended transition: a launch that runs again without a new post appends nothing (exactly-once).
 * ring_state (device int64[3] = cursor, full, capacity) is advanced on the device (by 2 when the wrap is requested). */
#define IL_MAIL_HEADER 8
#define IL_ACT_PENDING 1u        /* a transition (carry, mailbox) is waiting to be appended */
#define IL_ACT_WRAP_ABSORBING 2u /* episode ended by true termination with absorbing=true: rewrite + extra row */
#define IL_ACT_GREEDY 4u         /* tanh(mean) instead of a sample */
#define IL_ACT_NO_ACTION 8u      /* append only: no policy evaluation, carry left untouched */
#define IL_ACT_CARRY_FROM_MAILBOX 16u /* the pending transition's state | action come from the mailbox's observation / action slots
                                       * (written back by the host) instead of `carry`: for a worker whose act launches run ahead of its appends */
int32_t il_act_mailbox_floats(int32_t state_dim, int32_t action_dim);
/* mirror_version != NULL: `actor` is the base of 3 parameter snapshots `mirror_stride` floats apart and *mirror_version selects one
 * (see il_act_publish); NULL: `actor` is the live arena (launch on the stream the updates run on). */
int il_act_step(const float* actor, int32_t state_dim, int32_t action_dim, int32_t hidden, float* mailbox, float* carry, float* ring,
                int64_t* ring_state, uint64_t noise_seed, uint32_t noise_offset, const int32_t* mirror_version, int64_t mirror_stride,
                il_stream_t stream);
/* Publish the actor arena (n floats) into snapshot slot (version+1)%3 of `mirror` and advance the version. version_and_counter: device
 * int32[2] = {version, internal completion counter}, zero-initialised. Enqueue after every update (capturable into the update's graph). */
int il_act_publish(const float* actor, int64_t n, float* mirror, int64_t mirror_stride, int32_t* version_and_counter, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GAIL discriminator (reference training.py:85-134 adversarial_imitation_update with loss_function=BCE;
 * models.py:152-180 GAILDiscriminator depth 1 + ReLU; torch _SpectralNorm: one power iteration per call).
 * params order = discriminator.parameters(): spectral_norm ? {b1[H], W1[H,D], b2[1], W2[1,H]} : {W1, b1, W2, b2}.
 * ------------------------------------------------------------------------------------------ */
typedef struct il_disc {
  int32_t state_dim, action_dim, hidden, batch;
  int32_t spectral_norm, state_only, reward_function; /* 0 AIRL, 1 GAIL, 2 FAIRL (models.py:177-180) */
  float* params;             /* [P], P = H*D + H + H + 1, D = state_only ? S : S+A */
  float *u1, *v1, *u2, *v2;  /* spectral-norm buffers [H],[D],[1],[H] (ignored when !spectral_norm) */
  float* grad;               /* [P] summed gradient of the last step (always written) */
  il_adam opt;
  float grad_penalty, entropy_bonus;
  float* workspace;
  int64_t workspace_floats;
  uint64_t noise_seed;
  uint32_t* noise_counter;
  int64_t* sync;             /* il_sync counters or NULL (see below) */
  int32_t loss_function;     /* IL_LOSS_BCE / IL_LOSS_PUGAIL / IL_LOSS_MIXUP (training.py:97-113) */
  float pos_class_prior;     /* PUGAIL (training.py:101-102) */
  int32_t pu_clamped;        /* PUGAIL: 0 = nonnegative_margin is inf (conf/algorithm/GAIL.yaml: the clamp never binds); 1 = finite margin below: il_gail_disc_step then runs a
                                value pass (the logits of both calls) ahead of the gradient pass, whose workgroups all evaluate the clamp of training.py:102 on the batch-wide value */
  float nonnegative_margin;
} il_disc;
enum { IL_LOSS_BCE = 0, IL_LOSS_PUGAIL = 1, IL_LOSS_MIXUP = 2 };
/* optional inputs of the loss variants; every pointer may be NULL */
typedef struct il_gail_extra {
  const float* eps_mix;              /* [B] the Beta(alpha, alpha) draws of Mixup (training.py:106); NULL => U(0,1) from Philox, i.e. alpha = 1 */
  const float* logit_offset_policy;  /* [B] log pi(a|s) of the policy batch when subtract_log_policy (models.py:144,175) */
  const float* logit_offset_expert;  /* [B] same for the expert batch */
  const float* logit_offset_mix;     /* [B] Mixup with subtract_log_policy: log pi of the MIXED (state, action) rows (training.py:108: the caller mixes with the eps_mix it passes) */
} il_gail_extra;

int64_t il_disc_workspace_floats(int32_t in_dim, int32_t hidden, int32_t batch);
/* eps_gp [B] = the U(0,1) draw of training.py:118 (NULL => Philox). */
int il_gail_disc_step(const il_disc* d, const il_batch* policy, const il_batch* expert, const float* eps_gp, const il_gail_extra* extra,
                      uint32_t flags, il_stream_t stream);
/* AdamW(discriminator) from d->grad after the all-reduce of the data-parallel schedule. With il_sync counters in the descriptor this is the discriminator branch's last
 * kernel: its workgroups report [IL_SYNC_PARAMS] and the last one closes the branch's epoch, exactly like il_gail_disc_step(IL_FLAG_GAIL_CLOSE_EPOCH) does on one GPU, so
 * that il_sac_update_gather(relabel, IL_FLAG_GRADS_ONLY) on the other stream relabels inline as soon as the all-reduced step has landed. */
int il_gail_apply_grads(const il_disc* d, il_stream_t stream);
/* il_gail_disc_step with the update's index draw (il_replay_draw_resident) riding in the same launch as ONE extra workgroup: the discriminator workgroups are resident
 * early (weights staged, power iterations done) and wait for [IL_SYNC_INDICES]; the sampler workgroup waits for the previous update's end ([IL_SYNC_MAIN_EPOCH]),
 * draws the agent batch then the expert batch (train.py:173) and signals. A sampler launched behind this kernel in the same stream could never satisfy its wait;
 * one launched ahead of it would put the kernel's preparation back on the critical path. policy / expert must be rings read through il_batch.gather = idx_a / idx_b;
 * on-chip Philox noise; flags as il_gail_disc_step. The SAC branch on the other stream uses il_sac_update_gather(IL_FLAG_SAC_WAIT_INDICES). */
int il_gail_disc_step_draw(const il_disc* d, const il_batch* policy, const il_batch* expert, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                           const int64_t* ring_state_b, int32_t* idx_b, uint32_t flags, il_stream_t stream);
/* The same, and the sampler workgroup also copies the drawn AGENT rows (memory.py:58-63: the packed ring rows behind policy->states, through idx_a) into the dense slab
 * stage_rows [batch][policy->ld_states] before it signals [IL_SYNC_INDICES]. With the early draw (il_sync: [IL_SYNC_CHAIN_DONE]) that is tens of microseconds before the
 * next update starts; il_sac_update_gather(IL_FLAG_SAC_STAGED_ROWS, ring = a dense il_batch over stage_rows) then reads its rows with one global trip instead of two. */
int il_gail_disc_step_draw_staged(const il_disc* d, const il_batch* policy, const il_batch* expert, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                                  const int64_t* ring_state_b, int32_t* idx_b, float* stage_rows, uint32_t flags, il_stream_t stream);
/* Population axis: discriminator step + AIRL/GAIL/FAIRL reward relabel for n_learners discriminators; rewards_out_dev[l] -> float[batch]. */
int il_gail_step_population(const il_disc* descs_dev, const il_batch* policy_dev, const il_batch* expert_dev, float* const* rewards_out_dev,
                            int32_t n_learners, const il_disc* shape_host, il_stream_t stream);
/* models.py:177-180 predict_reward (eval mode: no power iteration). out_logits may be NULL. */
int il_gail_reward(const il_disc* d, const il_batch* batch, float* out_rewards, float* out_logits, const float* logit_offset,
                   il_stream_t stream); /* logit_offset [n] = log pi(a|s) when subtract_log_policy, else NULL */

/* ------------------------------------------------------------------------------------------
 * GAIL discriminator with reward shaping (reference models.py:152-180, reward_shaping = true):
 *   f = g(x) + (1 - terminal)(discount * h(s') - h(s)),  g = Linear(Dg, 1),  h = Linear(S, H) -> ReLU -> Linear(H, 1),  Dg = S (+ A unless state_only).
 * params in parameters() order: spectral norm  {g.bias, g.original[1,Dg], h.0.bias[H], h.0.original[H,S], h.2.bias, h.2.original[1,H]},
 * otherwise {g.weight, g.bias, h.0.weight, h.0.bias, h.2.weight, h.2.bias}; il_disc_shaped_numel floats. Buffers ug[1] vg[Dg] u1[H] v1[S] u2[1] v2[H].
 * Losses BCE / PUGAIL (any margin) / Mixup (il_gail_extra.eps_mix, or the on-chip U(0,1) stream for alpha = 1); il_gail_extra carries the subtract_log_policy offsets. Batches must carry next_states, terminals.
 * ------------------------------------------------------------------------------------------ */
typedef struct il_disc_shaped {
  int32_t state_dim, action_dim, hidden, batch;
  int32_t spectral_norm, state_only, reward_function, loss_function;
  float* params;
  float *ug, *vg, *u1, *v1, *u2, *v2;
  float* grad;
  il_adam opt;
  float grad_penalty, entropy_bonus, pos_class_prior, discount;
  float* workspace;          /* >= il_disc_shaped_workspace_floats() */
  int64_t workspace_floats;
  uint64_t noise_seed;
  uint32_t* noise_counter;
  int32_t pu_clamped;        /* PUGAIL with a finite nonnegative_margin, as in il_disc: a value pass ahead of the gradients decides whether the clamped term has one */
  float nonnegative_margin;
} il_disc_shaped;
int64_t il_disc_shaped_numel(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t state_only);
int64_t il_disc_shaped_workspace_floats(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t batch, int32_t state_only);
int il_gail_shaped_step(const il_disc_shaped* d, const il_batch* policy, const il_batch* expert, const float* eps_gp, const il_gail_extra* extra,
                        uint32_t flags, il_stream_t stream);
int il_gail_shaped_reward(const il_disc_shaped* d, const il_batch* batch, float* out_rewards, float* out_logits, const float* logit_offset,
                          il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GAIL discriminator of any `_create_fcnn` shape without reward shaping (models.py:152-162): depth 1-2, relu / tanh
 * (conf/hyperparameter_search_space/GAIL.yaml). il_disc / gail.hip is the fast path of the depth-1 ReLU shape every shipped configuration uses.
 * params in `discriminator.parameters()` order: per Linear (bias, weight) with spectral norm, (weight, bias) without; layers: hidden x depth, then H -> 1.
 * sn: the spectral-norm buffers, per layer [u (out) | v (in)] in layer order (il_disc_deep_sn_numel floats).
 * ------------------------------------------------------------------------------------------ */
typedef struct il_disc_deep {
  int32_t state_dim, action_dim, hidden, batch;
  int32_t spectral_norm, state_only, reward_function, loss_function;   /* as in il_disc */
  int32_t depth;        /* hidden layers: 1 or 2 (0 = 1) */
  int32_t activation;   /* 0 relu, 1 tanh */
  float* params;
  float* sn;
  float* grad;          /* [P] summed gradient of the last step (always written) */
  il_adam opt;
  float grad_penalty, entropy_bonus, pos_class_prior, reserved;
  float* workspace;     /* >= il_disc_deep_workspace_floats() */
  int64_t workspace_floats;
  uint64_t noise_seed;
  uint32_t* noise_counter;
  int32_t pu_clamped;        /* as in il_disc (training.py:100-102 with a finite nonnegative_margin) */
  float nonnegative_margin;
} il_disc_deep;
int64_t il_disc_deep_numel(int32_t in_dim, int32_t hidden, int32_t depth);
int64_t il_disc_deep_sn_numel(int32_t in_dim, int32_t hidden, int32_t depth);
int64_t il_disc_deep_workspace_floats(int32_t in_dim, int32_t hidden, int32_t depth, int32_t batch);
int64_t il_disc_deep_lds_bytes(int32_t in_dim, int32_t hidden, int32_t depth); /* LDS one workgroup needs; shapes beyond 160 KiB (e.g. input 120, hidden 128, depth 2) are refused */
/* adversarial_imitation_update (training.py:85-134) / predict_reward (models.py:177-180); arguments as il_gail_disc_step / il_gail_reward */
int il_gail_deep_step(const il_disc_deep* d, const il_batch* policy, const il_batch* expert, const float* eps_gp, const il_gail_extra* extra, uint32_t flags,
                      il_stream_t stream);
int il_gail_deep_reward(const il_disc_deep* d, const il_batch* batch, float* out_rewards, float* out_logits, const float* logit_offset, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GAIL discriminator with reward shaping whose potential h is any `_create_fcnn` shape (models.py:157-160 with discriminator.depth in {1, 2} and activation in
 * {relu, tanh}: conf/hyperparameter_search_space/GAIL.yaml); il_disc_shaped above is the depth-1 ReLU potential of the default configuration.
 *   f = g(x) + (1 - terminal)(discount * h(s') - h(s)),  g = Linear(Dg, 1),  h = [Linear - act] x depth - Linear(H, 1) on the state,  Dg = S (+ A unless state_only).
 * params in parameters() order: g (bias, weight) then h's Linears (bias, weight) each with spectral norm; (weight, bias) without. il_disc_shaped_deep_numel floats.
 * sn: ug[1] | vg[Dg] | per layer of h [u (out) | v (in)] (il_disc_shaped_deep_sn_numel floats). state <= 128, hidden <= 128, LDS per il_disc_shaped_deep_lds_bytes.
 * Losses, il_gail_extra, eps_gp and the batches as for il_disc_shaped.
 * ------------------------------------------------------------------------------------------ */
typedef struct il_disc_shaped_deep {
  int32_t state_dim, action_dim, hidden, batch;
  int32_t spectral_norm, state_only, reward_function, loss_function;
  int32_t depth;        /* hidden layers of h: 1 or 2 (0 = 1) */
  int32_t activation;   /* 0 relu, 1 tanh */
  float* params;
  float* sn;
  float* grad;          /* [P] summed gradient of the last step (always written) */
  il_adam opt;
  float grad_penalty, entropy_bonus, pos_class_prior, discount;
  float* workspace;     /* >= il_disc_shaped_deep_workspace_floats() */
  int64_t workspace_floats;
  uint64_t noise_seed;
  uint32_t* noise_counter;
  int32_t pu_clamped;        /* as in il_disc (training.py:100-102 with a finite nonnegative_margin) */
  float nonnegative_margin;
} il_disc_shaped_deep;
int64_t il_disc_shaped_deep_numel(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t state_only);
int64_t il_disc_shaped_deep_sn_numel(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t state_only);
int64_t il_disc_shaped_deep_workspace_floats(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t batch, int32_t state_only);
int64_t il_disc_shaped_deep_lds_bytes(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth, int32_t state_only);
/* adversarial_imitation_update (training.py:85-134) / predict_reward (models.py:177-180); arguments as il_gail_shaped_step / il_gail_shaped_reward */
int il_gail_shaped_deep_step(const il_disc_shaped_deep* d, const il_batch* policy, const il_batch* expert, const float* eps_gp, const il_gail_extra* extra, uint32_t flags,
                             il_stream_t stream);
int il_gail_shaped_deep_reward(const il_disc_shaped_deep* d, const il_batch* batch, float* out_rewards, float* out_logits, const float* logit_offset, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GMMIL (reference models.py:25-44, 183-201): d(x,y) = mean_k (x_k-y_k)^2, two RBF bandwidths.
 * ------------------------------------------------------------------------------------------ */
/* The workspace holds per-row-tile arrival counters that every call leaves at ZERO: zero-fill it once when it is created, and do not share one workspace between
 * calls of different (n_policy, n_expert, dim) - the counters sit at shape-dependent offsets (round 4: the reward is ONE launch that reads the batches directly; the
 * packing launch that used to zero them is gone). */
int64_t il_gmmil_workspace_floats(int32_t n_policy, int32_t n_expert, int32_t dim);
/* rewards[i] = sum_gamma w~_i sum_j K(x_i,e_j) w~e_j  -  w~_i sum_j K(x_i,x_j) w~_j ;  x = cat(state, action) (or state). */
int il_gmmil_reward(const il_batch* policy, const il_batch* expert, int32_t state_dim, int32_t action_dim, int32_t state_only,
                    float gamma_1, float gamma_2, float* out_rewards, float* out_similarity, float* out_self_similarity,
                    float* workspace, int64_t workspace_floats, il_stream_t stream);
/* models.py:25-28 _squared_distance matrix [na, nb] (used once for the median heuristic, models.py:193-195). */
int il_gmmil_sqdist(const il_batch* a, const il_batch* b, int32_t state_dim, int32_t action_dim, int32_t state_only, float* out,
                    float* workspace, int64_t workspace_floats, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PWIL (reference models.py:205-249): greedy Wasserstein coupling against the expert atoms.
 * atoms are already standardised (scale*(x+offset)); weights[N] is the consumable mass.
 * ------------------------------------------------------------------------------------------ */
typedef struct il_pwil {
  int32_t n_atoms, dim, state_dim, action_dim;
  const float* atoms;  /* [N, dim] */
  float* weights;      /* [N]; <0 marks a consumed (deleted) atom */
  float* dists;        /* scratch, >= il_pwil_scratch_floats(N, agent_weight) floats */
  const float *scale, *offset; /* [dim] */
  double reward_scale, reward_bandwidth, agent_weight; /* Python-float hyper-parameters; agent_weight = 1/T - 1e-6 (models.py:235) */
} il_pwil;
/* floats of il_pwil.dists: the per-chunk candidate lists of a step (or the N distances of the one-workgroup kernel) + one 16-byte slot at the very end, the arrival
 * counter of the one-launch step kernel (round 3: every workgroup selects its chunk's candidates, the last to arrive merges). il_pwil_reset zeroes that counter, and the
 * step kernel leaves it at zero: a caller that allocates the scratch itself must call il_pwil_reset before the first il_pwil_reward (models.py:216-230 does). */
int64_t il_pwil_scratch_floats(int32_t n_atoms, double agent_weight);
int il_pwil_reset(const il_pwil* d, il_stream_t stream);
/* compute_reward for one (state, action); writes the reward (double precision accumulate like the reference's Python floats) to out_reward[0]. */
int il_pwil_reward(const il_pwil* d, const float* state, const float* action, float* out_reward, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RED (reference models.py:252-284 REDDiscriminator, training.py:68-75 target_estimation_update).
 * predictor / target: Linear(D,H) -> ReLU -> Linear(H,D), D = state_dim (+ action_dim unless state_only), flat arenas in
 * torch parameters() order [W1 (H,D) | b1 (H) | W2 (D,H) | b2 (D)], il_red_numel(D,H) floats each.
 * ------------------------------------------------------------------------------------------ */
typedef struct il_red {
  int32_t state_dim, action_dim, hidden, batch, state_only;
  int32_t depth;        /* hidden layers of both networks: 1 or 2 (0 = 1) */
  float* predictor;     /* trained */
  const float* target;  /* frozen random network */
  float* grad;          /* [P] */
  il_adam opt;          /* AdamW state of the predictor (train.py:84) */
  float* workspace;     /* >= il_red_workspace_floats(D, H, batch, depth) */
  float sigma_1;        /* reward bandwidth (models.py:274-277 set_sigma or imitation.reward_bandwidth_scale) */
  int32_t activation;   /* 0 relu, 1 tanh (imitation.discriminator.activation) */
  float *out_pred, *out_target; /* filled by il_red_forward; leave NULL */
  float p_in, p;        /* input_dropout / dropout of the PREDICTOR (models.py:256; the target has none); applied in train mode only */
  uint64_t noise_seed;  /* Philox key of the keep-masks when the mask pointers are NULL */
} il_red;
/* parameters in torch order: W1[H,D] b1 (W2[H,H] b2 when depth = 2) Wo[D,H] bo */
int64_t il_red_numel(int32_t input_dim, int32_t hidden, int32_t depth);
int64_t il_red_workspace_floats(int32_t input_dim, int32_t hidden, int32_t batch, int32_t depth);
/* target_estimation_update: loss = mean_i w_i mean_c (pred_ic - target_ic)^2, AdamW on the predictor (IL_FLAG_GRADS_ONLY: d->grad only), train mode
 * (train.py:115-123 run before :147 `discriminator.eval()`): keep-masks mask_in [B,D], mask_h1 / mask_h2 [B,H] (0/1) or NULL = drawn on chip
 * (Philox counter noise_offset). out_loss [1] or NULL. */
int il_red_step(const il_red* d, const il_batch* expert, const float* mask_in, const float* mask_h1, const float* mask_h2, uint32_t noise_offset, float* out_loss,
                uint32_t flags, il_stream_t stream);
/* predict_reward (models.py:279-280, eval mode: training = 0): out_reward[i] = exp(-sigma_1 * mean_c (pred - target)^2), and / or the embeddings
 * out_pred, out_target [n, D] (what set_sigma feeds to the pairwise distance + median; set_sigma runs in TRAIN mode, train.py:128: training = 1 + masks). */
int il_red_forward(const il_red* d, const il_batch* batch, int32_t training, const float* mask_in, const float* mask_h1, const float* mask_h2, uint32_t noise_offset,
                   float* out_reward, float* out_pred, float* out_target, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DRIL (reference models.py:84-120: SoftActor built from conf/algorithm/DRIL.yaml's discriminator config = Dropout(p_in) -> Linear(S,H)
 * -> Dropout(p) -> Tanh -> Linear(H,2A), kept in train mode; training.py:57-64 behavioural_cloning_update trains it).
 * Flat arena in parameters() order [W1 (H,S) | b1 (H) | W2 (2A,H) | b2 (2A)], il_dril_numel floats.
 * Dropout keep-masks (1 = kept) may be supplied (what the parity tests do) or NULL = drawn on chip (Philox, noise_seed / noise_offset).
 * ------------------------------------------------------------------------------------------ */
typedef struct il_dril {
  int32_t state_dim, action_dim, hidden, batch;
  float p_in, p;        /* input_dropout, dropout */
  float* params;
  float* grad;          /* [P] */
  il_adam opt;
  float* workspace;     /* >= il_dril_workspace_floats(S, A, H, batch, depth) */
  uint64_t noise_seed;
  float q;              /* uncertainty threshold (models.py:110-111 set_uncertainty_threshold) */
  int32_t activation;   /* 0 tanh (conf/algorithm/DRIL.yaml), 1 relu */
  int32_t depth;        /* hidden layers: 1 or 2 (0 = 1) */
  int32_t reserved;
  const uint32_t* noise_counter; /* optional device uint32 added to every call's noise_offset when the masks are drawn on chip: a captured update (hipGraph replay) passes
                                    a constant noise_offset and lets the counter an il_sac descriptor advances once per update (il_sac.noise_counter) move the stream on */
} il_dril;
/* parameters in torch order: W1[H,S] b1 (Wh[H,H] bh when depth = 2) W2[2A,H] b2 */
int64_t il_dril_numel(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t depth);
int64_t il_dril_workspace_floats(int32_t state_dim, int32_t action_dim, int32_t hidden, int32_t batch, int32_t depth);
/* behavioural_cloning_update on the dropout policy: loss = mean_i w_i * -log pi(a_i | s_i; masks), AdamW. mask_in [B,S], mask_hidden / mask_hidden2 [B,H] or NULL. */
int il_dril_bc_step(const il_dril* d, const il_batch* expert, const float* mask_in, const float* mask_hidden, const float* mask_hidden2, uint32_t noise_offset,
                    float* out_loss, uint32_t flags, il_stream_t stream);
/* models.py:104-120: Monte-Carlo dropout uncertainty = unbiased variance over 5 masks of exp(log_prob(s, a)); masks [5n,S], [5n,H] (one per hidden layer) in
 * repeat_interleave order or NULL. out_uncertainty [n] and / or out_reward [n] = (uncertainty <= d->q ? +1 : -1). */
int il_dril_uncertainty(const il_dril* d, const il_batch* batch, const float* mask_in, const float* mask_hidden, const float* mask_hidden2, uint32_t noise_offset,
                        float* out_uncertainty, float* out_reward, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange as ONE kernel per sync point over peer-mapped windows (the reference has no multi-GPU path: this is the
 * exchange step BASELINE.json's north_star adds between training.py:31 / :50 `backward()` and `optimiser.step()`, and between training.py:132 and :133).
 * il_peer_allreduce_mean(x, bucket) == torch.distributed.all_reduce(bucket, AVG) for the ranks of `x`: every rank stores its bucket into slot [rank] of every
 * rank's receive window (one fabric crossing, each peer over its own xGMI link), waits on the device for the other ranks' arrival words and sums the slabs in
 * rank order (bit-identical results on every rank). Unlike the rest of this ABI the window functions DO allocate (fine-grained device memory) and synchronise:
 * they run once, at set-up. Set-up (imitation_learning_amd/parallel.py PeerExchange): every rank allocates one window of sum(il_peer_region_bytes) bytes,
 * the 64-byte handles are all-gathered with the host-side process group, every rank opens the other ranks' handles; buckets are told apart by
 * `window_offset`. A call may sit in a captured graph (epochs are device counters). RCCL stays the fallback (IL_PEER_EXCHANGE=0, or set-up / self-test failure).
 * ------------------------------------------------------------------------------------------ */
#define IL_PEER_MAX_RANKS 16
#define IL_PEER_HANDLE_BYTES 64      /* sizeof(hipIpcMemHandle_t) */
#define IL_PEER_CHUNK_FLOATS 2048    /* one workgroup's share of a bucket */
#define IL_PEER_FLAG_STRIDE 32       /* uint32 words per chunk in the arrival array: one 128-byte line per chunk, word r = the last epoch rank r pushed */
#define IL_PEER_WRITE_THROUGH 1      /* il_peer_bucket.flags: payload through sc0 sc1 (write-through / cache-bypassing) accesses + drained stores instead of system-scope fences;
                                        only for windows that il_peer_window_alloc reported as UNCACHED (kind 0) */
#define IL_PEER_SPIN_LIMIT (1 << 23) /* default bound of a device-side wait for the peers (polls of >= 8 sleep quanta: several seconds) */
typedef struct il_peer_bucket {
  int32_t rank, world;                 /* this process's rank among the `world` <= IL_PEER_MAX_RANKS ranks that exchange */
  int64_t n;                           /* floats in the bucket */
  int64_t window_offset;               /* byte offset of this bucket's region (il_peer_region_bytes) in EVERY rank's window; multiple of 256 */
  void* windows[IL_PEER_MAX_RANKS];    /* windows[r] = rank r's window as mapped in this process ([rank] = the own allocation) */
  uint32_t* epoch;                     /* local device uint32[ceil(n / IL_PEER_CHUNK_FLOATS)], zero-initialised: exchanges done per chunk */
  int64_t* status;                     /* local device int64[2], zero-initialised: [0] += 1 per wait that gave up (must stay 0); [1] = address of a host-mapped int64 that
                                          receives the new count whenever [0] moves (0 = none): the host notices without synchronising */
  int32_t spin_limit, flags;           /* polls before a wait gives up (0 = IL_PEER_SPIN_LIMIT); IL_PEER_WRITE_THROUGH or 0 */
  int32_t n_jobs, reserved;            /* 0: a region of il_peer_region_bytes (one arrival line per chunk: il_peer_allreduce_mean). > 0: a region of il_peer_job_region_bytes with
                                          n_jobs arrival lines, for an exchange that rides in the kernel PRODUCING the gradients (il_sac_update_gather_peer, il_gail_disc_step_draw_peer);
                                          `epoch` then has n_jobs entries */
} il_peer_bucket;
/* bytes of a bucket's region: slots float[2 parities][world][n rounded up to chunks] + arrival words; -1 on bad arguments */
int64_t il_peer_region_bytes(int32_t world, int64_t n);
/* the same with n_jobs arrival lines (one per producing workgroup) instead of one per chunk */
int64_t il_peer_job_region_bytes(int32_t world, int64_t n, int32_t n_jobs);
/* zero-filled uncached (failing that, fine-grained) device allocation on the current device + its IPC handle (IL_PEER_HANDLE_BYTES bytes, host) */
int il_peer_window_alloc(int64_t bytes, void** window_host, unsigned char* handle_host, int32_t* kind_host);   /* *kind_host: 0 uncached, 1 fine-grained */
/* maps another rank's window (a handle produced by il_peer_window_alloc in ANOTHER process) into this process */
int il_peer_window_open(const unsigned char* handle_host, void** window_host);
int il_peer_window_close(void* window);  /* a window obtained from il_peer_window_open */
int il_peer_window_free(void* window);   /* a window obtained from il_peer_window_alloc */
/* bucket[i] <- mean over ranks of bucket[i], i < x->n, in place; one launch of ceil(n / IL_PEER_CHUNK_FLOATS) workgroups. Every rank must issue the same
 * sequence of calls per bucket (like a collective). */
int il_peer_allreduce_mean(const il_peer_bucket* x, float* bucket, il_stream_t stream);

/* il_sac_dp_phase(2) / (3) with the gradient exchange of the phase's bucket carried by its apply launch: workgroup c exchanges chunk c (the body of
 * il_peer_allreduce_mean) and steps that chunk's parameters with the means it holds in registers - one launch and one pass over the gradient arena less per sync
 * point. x = the descriptor of the critic bucket (phase 2: il_sac.critic_grad, 2 * il_mlp_stride floats) or of the actor bucket (phase 3: il_sac.actor_grad with
 * il_sac.alpha_grad inside the same allocation, parallel.GradBuckets). Bit-identical to il_peer_allreduce_mean(x, bucket) followed by il_sac_dp_phase. */
int il_sac_dp_phase_peer(const il_sac* d, const il_batch* batch, int32_t phase, float* out_logp, float* out_q, uint32_t flags, const il_peer_bucket* x, il_stream_t stream);

/* Data-parallel update with NO extra launch: il_sac_update_gather whose two optimiser launches carry the gradient exchange. The workgroup that holds a 32 x 32 block of
 * a layer's dW (or a bias slice, or log alpha's gradient) pushes it into every rank's window at the parameter's own offset, waits for the same workgroup of the other
 * ranks, averages the slabs in rank order and runs its AdamW epilogue on the mean: the launch sequence of ONE GPU (train.py:171-203 per rank), replicas bit-identical to
 * each other and to il_sac_update_gather(IL_FLAG_GRADS_ONLY) + il_peer_allreduce_mean + il_sac_dp_phase. Buckets: il_sac_peer_bucket_floats(d, 0 | 1) floats,
 * il_sac_peer_jobs(d, 0 | 1) arrival lines (0: this shape has no block form - use the exchange launches), regions of il_peer_job_region_bytes. Every rank must issue the
 * same sequence of calls. */
int64_t il_sac_peer_bucket_floats(const il_sac* d, int32_t which);   /* 0 critic [2 * il_mlp_stride], 1 actor [round-up-4(Pa + 1)]: log alpha's gradient in slot Pa (parallel.GradBuckets' layout) */
int32_t il_sac_peer_jobs(const il_sac* d, int32_t which);
int il_sac_update_gather_peer(const il_sac* d, const il_batch* rows, const il_batch* ring, const float* rewards, const il_disc* relabel, float* rewards_out,
                              const float* eps_next, const float* eps_cur, float* out_logp, float* out_q, uint32_t flags, const il_peer_bucket* peer_critic,
                              const il_peer_bucket* peer_actor, il_stream_t stream);
/* il_gail_disc_step_draw whose reduce + AdamW launch carries the exchange of the discriminator's gradient (bucket: the parameter count, il_gail_step_workgroups(d) arrival lines) */
int il_gail_disc_step_draw_peer(const il_disc* d, const il_batch* policy, const il_batch* expert, uint32_t* mt_state_dev, const int64_t* ring_state_a, int32_t* idx_a,
                                const int64_t* ring_state_b, int32_t* idx_b, uint32_t flags, const il_peer_bucket* peer, il_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Launcher thread (round 6). The reference's loop (train.py:151-203) acts, steps the environment, appends and updates one after the other on one host thread; with the
 * update fused into two library calls its ~20 us of launch work were what stood between two environment steps. An il_launcher re-issues RECORDED calls - entry points of
 * this ABI with their arguments (pointers / integers only, <= 16; descriptors by reference) - from a thread of the library: il_launcher_submit returns at once, the calls go
 * out in recorded order, one pass per submit, passes in submission order. il_launcher_wait: every submitted pass has been ISSUED (synchronise the streams afterwards to wait
 * for the kernels); both return the sticky status of the recorded calls (il_last_error carries the failing call's text). The thread uses the device that was current at
 * il_launcher_create. Python: UpdatePlan.launch_async() / launcher_wait(). */
int il_launcher_create(void** out);
int il_launcher_destroy(void* launcher);
int il_launcher_clear(void* launcher);
int il_launcher_add(void* launcher, void* entry_point, const uint64_t* args_host, int32_t nargs);
int il_launcher_submit(void* launcher);
int il_launcher_wait(void* launcher);
int64_t il_launcher_pending(void* launcher);

/* sizeof() of the descriptor structs in this build (0 il_batch, 1 il_adam, 2 il_sac, 3 il_disc, 4 il_pwil, 5 il_sample_args, 6 il_red,
 * 7 il_dril, 8 il_disc_shaped, 9 il_disc_deep, 10 il_peer_bucket, 11 il_disc_shaped_deep; -1 otherwise): lets a binding verify its own struct definitions. */
int32_t il_struct_size(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* IL_HIP_H */
