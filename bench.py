#!/usr/bin/env python3
"""SAC+GAIL gradient-updates/s at batch 256, HalfCheetah dims (BASELINE.json metric) on N MI355X.

One "step" = one execution of the reference's update block (train.py:173-203, algorithm=GAIL default config):
2 replay samples (MT19937 index draws + row gathers) + adversarial_imitation_update (BCE, grad-penalty 1, spectral norm)
+ AIRL reward relabel + sac_update, on synthetic D4RL-shaped transitions resident in HBM.
N > 1 (torch.distributed.run): one rank per GPU, own replay shard, three RCCL gradient all-reduces per update (weak scaling:
per-GPU batch 256); value = N x synchronous global steps/s.

N > 1: `python bench.py --gpus N` with no rank environment starts the N ranks itself (torch.distributed.run on 127.0.0.1; fewer than N visible GPUs is an error, never an
N = 1 line); under torch.distributed.run it joins the ranks it is given. After the timed loop every rank hashes its replica state: `config.replicas_bit_identical`,
`config.exchange` (peer write-through | peer fences | rccl), `config.exchange_soak`; a peer-window run that expired a wait or left the replicas apart is re-timed on RCCL.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the launch stream; whole-update `hbm_frac` / `fp32_frac` at its top level) and,
at N=1, `cpu_baseline` = the reference's OWN code timed live on this host's cores (oracle/_ref: its three hot-path modules byte-compiled from /root/reference by
__graft_entry__.build(), run by oracle/ref_cpu_baseline.py in a subprocess at 1 / 8 / all threads, with and without memory.sample; only when oracle/_ref is absent the
committed build-container measurement profiles/cpu_reference.json stands in, and `where` says so) next to `cpu_port` = the numpy oracle port timed live;
`population` = the aggregate rate of 128 independent learners (two sub-populations of 64) advanced by the same launches; `secondary` = the other single-GPU configurations of BASELINE.json.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

S, A, H, HD, B = 18, 6, 256, 64, 256  # HalfCheetah dims incl. absorbing bit; GAIL.yaml discriminator
HBM_PEAK_GBS, FP32_PEAK_TFLOPS = 8000.0, 157.3  # /opt/skills/guides/MI355X_MICROARCH.md


class Cfg(dict):
  __getattr__ = dict.__getitem__


def synthetic_transitions(rs, n, state_shift=0.0):
  import inputs as gi
  return gi.transitions(rs, n, S, A, state_shift=state_shift, absorbing_frac=0.01, terminal_frac=0.001)


def algorithmic_model():
  Pa = H * S + H + H * H + H + 2 * A * H + 2 * A
  Pc = H * (S + A) + H + H * H + H + H + 1
  Pd = HD * (S + A) + HD + HD + 1
  mac_a, mac_c = S * H + H * H + H * 2 * A, (S + A) * H + H * H + H
  bytes_k = {
      'k_dw_adam_critic': 24 * 2 * Pc, 'k_dw_adam_actor': 24 * (Pa + 1) + 8 * 2 * Pc, 'k_gail_reduce': 24 * Pd, 'k_gather2': 2 * B * (2 * S + A + 5) * 4 + 2 * B * 4,
  }
  mac_d = (S + A) * HD + HD
  flops_k = {
      'k_gail_grad': 2 * B * mac_d * (3 + 2 * 2 + 3), 'k_gail_reward': 2 * B * mac_d,   # 3 forwards, 2 BCE backwards, closed-form GP second-order terms
      'k_actor_fwd': 2 * 2 * B * mac_a, 'k_critic_fwd': 4 * 2 * B * mac_c, 'k_critic_bwd': 2 * 2 * B * H * H, 'k_dw_adam_critic': 2 * 2 * B * mac_c,
      'k_policy_critic': 2 * 2 * B * (mac_c + H * H + H * A) + 2 * B * (2 * A * H + H * H),   # both critics on (s, a~) fwd + dQ/da, then the policy backward as the pair's tail
      'k_dw_adam_actor': 2 * B * mac_a,
  }
  flops_k['k_sac_chain'] = flops_k['k_actor_fwd'] + flops_k['k_critic_fwd'] + flops_k['k_critic_bwd']   # the three as one launch (il_sac_update, whole update)
  update_bytes = 24 * (Pa + 2 * Pc + Pd + 1) + 8 * 2 * Pc + 2 * B * (2 * S + A + 5) * 4
  return bytes_k, flops_k, update_bytes, sum(v for k, v in flops_k.items() if k != 'k_sac_chain')


def build(device, rank, seed=0, learner_id=None):
  import imitation_learning_amd as il
  torch.manual_seed(seed)
  cfg = Cfg(hidden_size=H, depth=2, activation='relu')
  actor, critic = il.SoftActor(S, A, cfg, device=device), il.TwinCritic(S, A, cfg, device=device)
  target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=device)
  ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=1.0, entropy_bonus=0.0, learning_rate=3e-5, weight_decay=10,
             discriminator=Cfg(hidden_size=HD, depth=1, activation='relu', reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
  disc = il.GAILDiscriminator(S, A, icfg, 0.97, device=device)
  do = il.AdamW(disc, lr=3e-5, weight_decay=10)
  rs = np.random.RandomState(1000 + rank)  # per-rank replay shard
  mem = il.ReplayMemory(1_000_000, S, A, True, device=device)
  n_fill = 100_000
  tr = synthetic_transitions(rs, n_fill)
  for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
    getattr(mem, k)[:n_fill] = torch.from_numpy(tr[k]).to(device)
  mem.step[:n_fill] = torch.arange(1, n_fill + 1, dtype=torch.float32, device=device)
  mem.idx, mem.full = n_fill, False
  mem._sync_ring_state()
  et = synthetic_transitions(np.random.RandomState(77), 25_000, state_shift=0.5)  # full copy of the expert buffer on every rank
  emem = il.ReplayMemory(25_000, S, A, True, transitions={**{k: torch.from_numpy(v) for k, v in et.items() if k != 'absorbing'}, 'num_trajectories': 25}, device=device)
  if learner_id is None:
    il.seed(seed + rank)
  else:
    mem.index_rng = il.IndexStream(seed + rank)   # each learner of a population draws from its own MT19937 stream
  plan = il.UpdatePlan('GAIL', actor, critic, log_alpha, target, mem, ao, co, to, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc, discriminator_optimiser=do,
                       imitation_cfg=icfg, learner_id=learner_id)
  return plan, (actor, critic, target, log_alpha, disc), (tr, et)


def secondary(device, plan, nets):
  """The other single-GPU configurations of BASELINE.json, reported next to the headline line: SAC only (configs[1]), the GAIL
  discriminator step + relabel alone (configs[2]), the GMMIL pairwise-RBF reward at B = 1024 with Ant dims (configs[3]) and the PWIL per-step reward
  against 25,000 expert atoms.  Each: captured where capturable, 300 timed repetitions, inputs resident."""
  import imitation_learning_amd as il
  from imitation_learning_amd import training as T
  out = {}

  def timed(step, reps=300, warm=30):
    for _ in range(warm): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return reps / (time.perf_counter() - t0)

  actor, critic, target, log_alpha, disc = nets
  keep = plan._keep
  sac_plan = il.UpdatePlan('SAC', actor, critic, log_alpha, target, plan.memory, keep[4], keep[5], keep[6], B, 0.97, -0.5 * A, 0.99, learner_id=9001)
  sac_plan.run()
  if sac_plan.direct_launch_ok():   # as the headline: the update's launches issued directly (two library calls per update), no hipGraph
    sac_plan.record_direct()
    out['sac_only_updates_per_s'] = round(timed(sac_plan.launch_direct, 2000, 200), 1)
    out['sac_only_launch'] = 'direct launches'
  else:
    sac_plan.capture(warmup=0)
    out['sac_only_updates_per_s'] = round(timed(sac_plan.replay, 1000, 100), 1)
    out['sac_only_launch'] = 'hipGraph replay'

  # the headline update with FOUR consecutive updates captured per replay (UpdatePlan.capture(updates=4): offline training / several updates per environment step):
  # the ~3.5 us of queue work between two graph launches is then paid once per four updates. The headline `value` replays one update per launch, like train.py does.
  was_side, was_main = plan.graph_side, plan.graph
  if was_main is not None:
    plan.capture(warmup=0, updates=4)
    out['sac_gail_updates_per_s_4_per_replay'] = round(4 * timed(plan.replay, 500, 50), 1)
    plan.graph_side, plan.graph = was_side, was_main

  g = torch.cuda.CUDAGraph()
  L = _lib_mod().lib()
  side = torch.cuda.Stream()
  dd, pb, eb, rew = plan.disc, plan.pb, plan.eb, plan.rewards
  was = plan.device_sync
  plan._set_device_sync(False)
  def disc_step():
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.il_gail_disc_step(C.byref(plan.disc), C.byref(pb), C.byref(eb), None, None, 0, st) == 0
    assert L.il_gail_reward(C.byref(plan.disc), C.byref(pb), C.c_void_p(rew.data_ptr()), None, None, st) == 0
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    disc_step(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
      disc_step()
    r_graph, r_direct = timed(g.replay, 1000, 100), timed(disc_step, 1000, 100)
    out['gail_disc_step_and_relabel_per_s'] = round(r_direct, 1)   # the two library calls issued directly: the product's default launch path on one GPU (train.py, bench.py)
    out['gail_disc_step_and_relabel_forms'] = dict(direct_calls=round(r_direct, 1), graph_replays=round(r_graph, 1))   # (rounds 1-5 reported the better of the two under the first key)
  torch.cuda.current_stream().wait_stream(side)
  plan._set_device_sync(was)
  del dd

  # GMMIL, Ant dims (S = 112 incl. absorbing bit, A = 8 -> D = 120), B = 1024 policy rows against 1024 expert rows
  rs = np.random.RandomState(5)
  Sg, Ag, Bg = 112, 8, 1024
  mk = lambda shift: (torch.from_numpy((rs.standard_normal((Bg, Sg)) + shift).astype(np.float32)).to(device), torch.from_numpy(rs.uniform(-1, 1, (Bg, Ag)).astype(np.float32)).to(device))
  (xs, xa), (es, ea) = mk(0.0), mk(0.5)
  w = torch.ones(Bg, device=device)
  gm = il.GMMILDiscriminator(Sg, Ag, Cfg(state_only=False))
  gm.predict_reward(xs, xa, es, ea, w, w)   # first call fixes the bandwidths (median heuristic)
  rate = timed(lambda: gm.predict_reward(xs, xa, es, ea, w, w))
  pair_flops = 2 * (2 * Bg * Bg) * (3 * (Sg + Ag))   # SURVEY.md 8(d)'s unit: two gammas share the distances: 2 matrices x B^2 pairs x 3 flop per pair-feature (direct form)
  mfma_flops = (2 * Bg * Bg) * (2 * 128)              # what k_gmmil_mfma executes: 2 flop per pair-feature of the centred Gram product, D padded to 128
  gst = _lib_mod().kernel_stamps().get('k_gmmil_direct')
  kus = gst['duration_us'] if gst else None
  gform = 'k_gmmil_mfma' if os.environ.get('IL_GMMIL_MFMA', '1') != '0' else 'k_gmmil_sx'
  out['gmmil_reward_B1024_ant'] = dict(calls_per_s=round(rate, 1), pair_feature_TFLOPs=round(pair_flops * rate / 1e12 / 2, 2), kernel=gform, kernel_us=kus,
                                        kernel_fp32_frac=(round(pair_flops / 2 / (kus * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4) if kus else None),
                                        kernel_mfma_frac=(round(mfma_flops / (kus * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 4) if kus and gform == 'k_gmmil_mfma' else None),
                                        note='round 6: k_gmmil_mfma (default, D <= 128): pair distances as a CENTRED Gram product |x-c|^2 + |y-c|^2 - 2 (x-c).(y-c) on v_mfma_f32_16x16x4_f32, one launch, 64 x 128 pairs per workgroup, as close to float64 as the direct-difference form whatever the data\'s offset (tests/test_gmmil_centred_form.py, test_gmmil_centred_gram_form_is_as_close_to_float64_as_the_direct_form); kernel_fp32_frac prices SURVEY 8(d)\'s 3 flop per pair-feature (the unit of earlier rounds), kernel_mfma_frac the 2 flop per pair-feature the launch executes; IL_GMMIL_MFMA=0: k_gmmil_sx (direct differences on the VALU, 18.5 us) and the forms behind it; calls_per_s is back-to-back predict_reward calls, kernel_us the launch itself (device stamps); the reference materialises [B,B,D] temporaries (0.33 s per call on its CPU path, SURVEY.md a20)')

  # BASELINE.json configs[3] as WHOLE updates: algorithm=GMMIL env=ant, batch 1024 - 2 replay samples + the pairwise-RBF reward + sac_update as one captured graph per step
  rs2 = np.random.RandomState(6)
  import inputs as gi
  cfgn = Cfg(hidden_size=H, depth=2, activation='relu')
  a2, c2 = il.SoftActor(Sg, Ag, cfgn, device=device), il.TwinCritic(Sg, Ag, cfgn, device=device)
  t2, la2 = il.create_target_network(c2), torch.zeros(1, device=device)
  o2 = (il.AdamW(a2, lr=3e-4, weight_decay=0), il.AdamW(c2, lr=3e-4, weight_decay=0), il.Adam(la2, lr=3e-4))
  def ring(n, cap, shift):
    tr = gi.transitions(rs2, n, Sg, Ag, state_shift=shift, absorbing_frac=0.01, terminal_frac=0.001)
    m = il.ReplayMemory(cap, Sg, Ag, True, device=device)
    for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
      getattr(m, k)[:n] = torch.from_numpy(tr[k]).to(device)
    m.step[:n] = torch.arange(1, n + 1, dtype=torch.float32, device=device)
    m.idx, m.full = n % cap, n == cap
    m._sync_ring_state()
    return m
  gplan = il.UpdatePlan('GMMIL', a2, c2, la2, t2, ring(100_000, 1_000_000, 0.0), *o2, Bg, 0.99, -1.0 * Ag, 0.995, expert_memory=ring(25_000, 25_000, 0.5),
                        discriminator=il.GMMILDiscriminator(Sg, Ag, Cfg(state_only=False)), learner_id=9002)
  gplan.run()
  if gplan.direct_launch_ok() and os.environ.get('IL_BENCH_LAUNCH', 'direct') != 'graph':   # (round 6) a one-stream plan is library calls only: recorded once, re-issued per update (no hipGraph: a replay costs
    gplan.record_direct()                                                                    # the replay-to-replay gap and torch.cuda.CUDAGraph's per-replay generator fill, ~11 us of 148)
    out['gmmil_ant_b1024_updates_per_s'] = round(timed(gplan.launch_direct, 300, 30), 1)
    out['gmmil_ant_b1024_launch'] = 'direct launches (UpdatePlan.launch_direct)'
    gplan.capture(warmup=0)
    out['gmmil_ant_b1024_graph_replays_per_s'] = round(timed(gplan.replay, 300, 30), 1)
  else:
    gplan.capture(warmup=0)
    out['gmmil_ant_b1024_updates_per_s'] = round(timed(gplan.replay, 300, 30), 1)
    out['gmmil_ant_b1024_launch'] = 'hipGraph replays'

  # PWIL: one (state, action) against N = 25,000 standardised expert atoms, D = 24, consumed greedily (models.py:232-249)
  emem = plan.expert_memory
  pw = il.PWILDiscriminator(S, A, Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), emem, 1000)
  s1, a1 = torch.zeros(1, S, device=device), torch.zeros(1, A, device=device)
  k = [0]
  def pwil_step():
    pw.compute_reward_async(s1, a1)
    k[0] += 1
    if k[0] % 1000 == 0: pw.reset()
  out['pwil_reward_25k_atoms_steps_per_s'] = round(timed(pwil_step, 2000, 100), 1)

  # an actor / critic shape outside the fused kernels (models.py:48-69: depth 3, tanh): WHOLE updates (2 device draws + gather + sac_update through csrc/general.hip's
  # layer-at-a-time launches) as one captured graph per update (round 5: UpdatePlan accepts these shapes on one stream), next to the per-function call on a fixed batch.
  # Reported, never part of `value`; a failure here must not cost the line.
  try:
    cfg3 = Cfg(hidden_size=H, depth=3, activation='tanh')
    ga, gc = il.SoftActor(S, A, cfg3, device=device), il.TwinCritic(S, A, cfg3, device=device)
    gt, gla = il.create_target_network(gc), torch.zeros(1, device=device)
    gao, gco, gto = il.AdamW(ga, lr=3e-4, weight_decay=0), il.AdamW(gc, lr=3e-4, weight_decay=0), il.Adam(gla, lr=3e-4)
    from imitation_learning_amd.memory import batch_views
    gb = batch_views(plan.memory.ring[:B].clone(), S, A, True)   # the first B rows of the ring: no index draw (the generator's state is the timed schedule's)
    out['sac_general_shape_depth3_tanh_updates_per_s'] = round(timed(lambda: il.sac_update(ga, gc, gla, gt, gb, gao, gco, gto, 0.97, -0.5 * A, 0.99), 300, 30), 1)
    gplan3 = il.UpdatePlan('SAC', ga, gc, gla, gt, plan.memory, gao, gco, gto, B, 0.97, -0.5 * A, 0.99, learner_id=9003)
    gplan3.run()
    if gplan3.direct_launch_ok():
      gplan3.record_direct()
      out['sac_general_shape_depth3_tanh_plan_direct_launches_updates_per_s'] = round(timed(gplan3.launch_direct, 300, 30), 1)
    gplan3.capture(warmup=0)
    out['sac_general_shape_depth3_tanh_captured_plan_updates_per_s'] = round(timed(gplan3.replay, 300, 30), 1)
  except Exception as e:   # noqa: BLE001
    out['sac_general_shape_depth3_tanh_updates_per_s'] = f'failed: {type(e).__name__}: {e}'[:200]
  return out


def _lib_mod():
  from imitation_learning_amd import _lib
  return _lib


def cpu_baseline(tr, et, budget_s=12.0):
  """The oracle port of the same update block on the host cores (index draws through numpy's legacy RNG like the reference)."""
  from oracle import gail as ogail
  from oracle import nets as onets
  from oracle import replay as oreplay
  from oracle import sac as osac
  rs = np.random.RandomState(0)
  import inputs as gi
  st = osac.SacState(S, A, H)
  st.actor[:] = gi.mlp_params(rs, S, H, 2, 2 * A, out_scale=0.3)
  st.critic[:] = np.concatenate([gi.mlp_params(rs, S + A, H, 2, 1) for _ in range(2)]); st.target[:] = st.critic
  g = gi.gail_case(1)
  ds = ogail.DiscState(S + A, HD, True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = g[k]
  n = tr['states'].shape[0]
  mem = oreplay.ReplayOracle(1_000_000, S, A, True)
  for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
    getattr(mem, k)[:n] = tr[k]
  mem.idx = n
  emem = oreplay.ReplayOracle(et['states'].shape[0], S, A, True, transitions={**et, 'num_trajectories': 25})
  np.random.seed(0)

  def draw(m, k):
    out, high, excl = [], (m.size if m.full else m.idx - 1), (m.idx - 1) % m.size
    while len(out) < k:
      v = int(np.random.randint(0, high))
      if v != excl:
        out.append(v)
    return out
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)

  def one():
    b, e = mem.gather(draw(mem, B)), emem.gather(draw(emem, B))
    ogail.gail_update(ds, cat(b), b['weights'], cat(e), e['weights'], rs.uniform(size=B).astype(np.float32), lr=3e-5, weight_decay=10, grad_penalty=1.0)
    b['rewards'] = ogail.predict_reward(ds, cat(b))
    osac.sac_update(st, b, rs.standard_normal((B, A)).astype(np.float32), rs.standard_normal((B, A)).astype(np.float32), discount=0.97, entropy_target=-0.5 * A, polyak_factor=0.99)
  for _ in range(3):
    one()
  t0, k = time.perf_counter(), 0
  while time.perf_counter() - t0 < budget_s:
    one(); k += 1
  dt = time.perf_counter() - t0
  try:
    import threadpoolctl
    threads = max([p['num_threads'] for p in threadpoolctl.threadpool_info()] + [1])
  except Exception:
    threads = os.cpu_count()
  return dict(value=round(k / dt, 2), unit='updates/s', cores=int(threads), kind='port', where='GPU box host',
              sample=f'{k} SAC+GAIL updates (numpy float32 oracle port of train.py:173-203, batch {B}, same synthetic buffers) in {dt:.1f} s; numpy BLAS pool of {int(threads)} threads '
                     f'(the `cores` field) on a host with {os.cpu_count()} logical cores; everything outside the GEMMs is single-threaded')


def cpu_reference(budget_s=20.0):
  """The reference's OWN code on its CPU path, timed LIVE on this host beside the GPU number: oracle/ref_cpu_baseline.py in a subprocess runs the
  reference's training.py / models.py / memory.py - byte-compiled from /root/reference into the git-ignored oracle/_ref/ by __graft_entry__.build() (oracle/build_ref.py),
  shipped to the GPU box like the in-tree .so - through train.py:173-203 at 1 thread and at all cores, with and without memory.sample. Only when oracle/_ref is absent does
  the committed build-container measurement (profiles/cpu_reference.json) stand in, and `where` says so."""
  import subprocess
  runner = os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py')
  try:
    r = subprocess.run([sys.executable, runner, '--budget', str(budget_s)], capture_output=True, text=True, timeout=60 + 6 * budget_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
    if r.returncode == 0:
      j = json.loads(r.stdout.strip().splitlines()[-1])
      res, n = j['results'], j['threads_all_cores']
      with_s = {1: res['one_thread_with_memory_sample'], n: res['all_cores_with_memory_sample']}
      if 'eight_threads_with_memory_sample' in res: with_s[8] = res['eight_threads_with_memory_sample']
      best = max(with_s, key=with_s.get)
      return dict(value=with_s[best], unit='updates/s', cores=best, kind='reference', by_threads_with_memory_sample=with_s,
                  where='this host, live (oracle/_ref: the reference\'s own modules byte-compiled from /root/reference)', cpu_model=j['cpu_model'], nproc=j['nproc'],
                  one_thread=res['one_thread_with_memory_sample'], all_cores=res['all_cores_with_memory_sample'], threads_all_cores=n,
                  without_memory_sample=dict(one_thread=res['one_thread_without_memory_sample'], all_cores=res['all_cores_without_memory_sample']), torch=j['torch'],
                  sample=f"{sum(j['updates_timed'].values())} updates of train.py:173-203 (algorithm=GAIL, batch {B}, same synthetic buffers) in {j['seconds']} s of CPU work: "
                         f"{j['updates_timed']}; `value` = the best of 1 / 8 / {n} threads WITH the two memory.sample calls (`cores` says which; a large OpenMP pool slows the reference's tiny aten calls down)",
                  reference_sources_sha256={m: v['source_sha256'][:16] for m, v in j['manifest']['modules'].items()})
    print(f'[bench] oracle/ref_cpu_baseline.py: rc {r.returncode}: {(r.stdout + r.stderr).strip()[-300:]}', file=sys.stderr)
  except Exception as e:
    print(f'[bench] live reference timing failed ({type(e).__name__}: {e}); using the committed measurement', file=sys.stderr)
  try:
    ref = json.load(open(os.path.join(ROOT, 'profiles', 'cpu_reference.json')))
  except Exception:
    return None
  r, n = ref['results'], ref['nproc']
  return dict(value=r[f'threads_{n}_with_memory_sample'], unit='updates/s', cores=n, kind='reference', where=ref['where'] + ' - NOT measured in this run: oracle/_ref is absent here', cpu_model=ref['cpu_model'],
              one_thread=r['threads_1_with_memory_sample'], without_memory_sample=dict(one_thread=r['threads_1_without_memory_sample'], all_cores=r[f'threads_{n}_without_memory_sample']),
              sample=f"{ref['updates_timed']} updates after {ref['warmup']} warm-up: {ref['what']}; source profiles/cpu_reference.json (committed; not re-timed in this run)")


STAMP_MODEL = {   # launch-stamp kernel name (imitation_learning_amd._lib.STAMP_KERNELS) -> key of algorithmic_model()'s tables / of profiles/pmc_latest.json
    'k_gail_grad': 'k_gail_grad', 'k_gail_reduce': 'k_gail_reduce', 'k_sac_chain_pair': 'k_sac_chain', 'k_dw_adam_critic': 'k_dw_adam_critic', 'k_policy_critic_pair': 'k_policy_critic',
    'k_dw_adam_actor': 'k_dw_adam_actor'}
PMC_NAMES = {'k_gail_grad': 'k_gail_grad', 'k_gail_reduce': 'k_gail_reduce', 'k_sac_chain_pair': 'k_sac_chain_pair', 'k_dw_adam_critic': 'k_dw_adam', 'k_policy_critic_pair': 'k_policy_critic_pair',
             'k_dw_adam_actor': 'k_dw_adam'}


def load_pmc(schedule_kernels):
  """profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed with their collection date and commit). Raises when the file does not list the
  kernels of the schedule that was just timed: a stale counter file must not decorate a new schedule's line."""
  pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
  missing = sorted({PMC_NAMES[k] for k in schedule_kernels if k in PMC_NAMES} - set(pmc['kernels']))
  if missing:
    raise RuntimeError(f'profiles/pmc_latest.json (collected {pmc.get("collected", "?")}) has no entry for {missing}: the counter passes predate the timed schedule - '
                       're-collect them (profiles/tools/r5_profile.sh pmc) or run with --no-pmc')
  return pmc


def stamp_sample(L_mod):
  """One reading of the launch stamps (the LAST launch of each headline kernel): {kernel: duration_us} + the main stream's boundaries and span."""
  st = L_mod.kernel_stamps()
  out = {k: v['duration_us'] for k, v in st.items()}
  main = [k for k in ('k_sac_chain_pair', 'k_dw_adam_critic', 'k_policy_critic_pair', 'k_dw_adam_actor') if k in st]
  if len(main) == 4:
    for a, b in zip(main[:-1], main[1:]):
      out[f'boundary:{a}->{b}'] = st[b]['begin_us'] - st[a]['end_us']   # (overlapped launches: negative - the next launch's first workgroup starts before this one's last has ended)
    out['span:main_stream'] = st[main[-1]]['end_us'] - st[main[0]]['begin_us']
    for k in main:   # overlapped launches: first workgroup past its wait for the other stream's launch -> last workgroup's end ("active"), and how long before that it was resident
      g = L_mod.kernel_stamp_gates(k)
      if g is not None:
        out[f'active:{k}'] = st[k]['end_us'] - g[0]
        out[f'resident_before_gate:{k}'] = g[0] - st[k]['begin_us']
  return out


def roofline_from_stamps(samples, ms_per_step, use_pmc=True):
  """`roofline` of the headline line from device stamps taken INSIDE the timed graph replays (include/il_hip.h il_kernel_stamps): per kernel, first workgroup's start to
  last workgroup's end of the last launch of a burst of replays; `samples` = one reading per timed repeat / burst; medians reported."""
  bytes_k, flops_k, update_bytes, update_flops = algorithmic_model()
  med = lambda xs: float(np.median(np.asarray(xs, dtype=np.float64)))
  names = [k for k in samples[0] if ':' not in k]
  overlapped = any(k.startswith('active:') for k in samples[0])
  kern, per_kernel = {}, {}
  for k in names:
    xs = [sm[k] for sm in samples if k in sm]
    kern[k] = med(xs)
    e = dict(avg_us=round(kern[k], 3), min_us=round(min(xs), 3), max_us=round(max(xs), 3), samples=len(xs), launches_per_update=1.0)
    mk = STAMP_MODEL.get(k)
    if mk in bytes_k: e['hbm_GBps'] = round(bytes_k[mk] / (kern[k] * 1e-6) / 1e9, 2)
    if mk in flops_k: e['fp32_TFLOPs'] = round(flops_k[mk] / (kern[k] * 1e-6) / 1e12, 3)
    if f'active:{k}' in samples[0]:   # overlapped launches: avg_us is residency (it contains the wait for the other stream's launch); active_us starts when the first workgroup got past that wait
      e['active_us'] = round(med([sm[f'active:{k}'] for sm in samples if f'active:{k}' in sm]), 3)
      e['resident_before_gate_us'] = round(med([sm[f'resident_before_gate:{k}'] for sm in samples if f'resident_before_gate:{k}' in sm]), 3)
      if mk in flops_k: e['fp32_TFLOPs_active'] = round(flops_k[mk] / (e['active_us'] * 1e-6) / 1e12, 3)
    per_kernel[k] = e
  side = ('k_gail_grad', 'k_gail_reduce')   # the discriminator branch runs beside the SAC branch on its own stream
  dom = max((k for k in kern if k not in side), key=lambda k: kern[k])
  mk = STAMP_MODEL[dom]
  if mk in flops_k:
    ach = flops_k[mk] / (kern[dom] * 1e-6) / 1e12
    roof = dict(bound='mfma', kernel=dom, note='fp32: MFMA f32 rate == VALU f32 rate == 157.3 TFLOP/s on gfx950', achieved=round(ach, 3), peak=FP32_PEAK_TFLOPS, unit='TFLOP/s',
                frac=round(ach / FP32_PEAK_TFLOPS, 5), traffic=None)
  else:
    ach = bytes_k.get(mk, 0) / (kern[dom] * 1e-6) / 1e9
    roof = dict(bound='hbm', kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 5), traffic=None)
  roof['duration_source'] = (f'device stamps inside the timed hipGraph replays (il_kernel_stamps: thread 0 of every workgroup stores the 100 MHz device counter at start / end; '
                             f'duration = last end - first start of the last launch of a burst), median of {len(samples)} readings; compare profiles/r06_headline_kernel_stats.md (rocprofv3 '
                             '--kernel-trace of the same command: dispatch-to-completion, i.e. + the command processor\'s launch and end-of-kernel work)')
  try:   # the committed rocprofv3 --kernel-trace of the same command (dispatch-to-completion: + the command processor's launch and end-of-kernel work, i.e. part of what the
    # stamps show as launch boundaries): the dominant kernel's duration and fraction by that clock, beside the stamps'
    prof = os.path.join(ROOT, 'profiles', 'r06_headline_kernel_stats.md')
    for line in open(prof):
      cells = [c.strip() for c in line.strip().strip('|').split('|')]
      if len(cells) >= 3 and cells[0] == dom:
        avg = float(cells[2])
        roof['rocprofv3_reference'] = dict(file='profiles/r06_headline_kernel_stats.md', kernel=dom, avg_us=avg, frac=round(flops_k[mk] / (avg * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 5) if mk in flops_k else None,
                                           note='rocprofv3 times a dispatch from the packet to the completion signal; the stamps from the first workgroup\'s first instruction to the last workgroup\'s last: the difference (~2.3 us for a 163 x 512-thread launch with 160 KB of LDS per workgroup) is what `update.launch_boundaries_us` holds')
        break
  except Exception:
    pass
  if use_pmc:
    pmc = load_pmc(names)
    roof['traffic'] = pmc['kernels'][PMC_NAMES[dom]]['traffic_bytes']
    roof['traffic_source'] = dict(file='profiles/pmc_latest.json', collected=pmc.get('collected'), commit=pmc.get('commit'), command=pmc.get('command'),
                                  how='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, 2*FETCH + WRITE (gfx950 correction of MI355X_MICROARCH.md), per launch; a counter-collecting '
                                      'profiler serialises kernels, so the passes run `bench.py --no-graph --no-overlap` (same kernels, stream-dependency schedule); not collected in this run')
    for k, e in per_kernel.items():
      if PMC_NAMES.get(k) in pmc['kernels']: e['hbm_traffic_bytes'] = pmc['kernels'][PMC_NAMES[k]]['traffic_bytes']
  upd_gbs = update_bytes / (ms_per_step * 1e-3) / 1e9
  roof['hbm_frac'] = round(upd_gbs / HBM_PEAK_GBS, 5)
  roof['fp32_frac'] = round(update_flops / (ms_per_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5)
  roof['binding_roof'] = 'fp32 (exact-fp32 MFMA = VALU rate): the update needs 3.7 us of fp32 issue but only 0.8 us of HBM time; `frac` above is the dominant kernel against that roof'
  bnd = {k.split(':', 1)[1]: round(med([sm[k] for sm in samples if k in sm]), 3) for k in samples[0] if k.startswith('boundary:')}
  if 'span:main_stream' in samples[0]:   # what is left of the update period: the last launch of an update -> the first launch of the next one
    bnd['k_dw_adam_actor->k_sac_chain_pair (next update)'] = round(ms_per_step * 1e3 - med([sm['span:main_stream'] for sm in samples if 'span:main_stream' in sm]), 3)
  if overlapped:
    roof['overlapped_launches'] = ('the four main launches alternate over two streams: a launch is RESIDENT before its predecessor ends (negative launch boundaries below) and its '
                                   'duration contains the wait; `frac` prices the dominant launch by its whole residency (what rocprofv3 reports), kernels[*].active_us by the '
                                   'time from its first workgroup passing the wait; the honest whole-update figures are fp32_frac / hbm_frac (algorithmic work / update period)')
    if 'active_us' in per_kernel.get(dom, {}) and mk in flops_k:
      roof['frac_active'] = round(flops_k[mk] / (per_kernel[dom]['active_us'] * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 5)
  roof['update'] = dict(algorithmic_bytes=update_bytes, achieved_GBps=round(upd_gbs, 2), hbm_frac=round(upd_gbs / HBM_PEAK_GBS, 5), algorithmic_flops=update_flops,
                        fp32_frac=roof['fp32_frac'], sum_main_stream_kernel_us=round(sum(kern[k] for k in kern if k not in side), 2), launch_boundaries_us=bnd,
                        main_stream_span_us=(round(med([sm['span:main_stream'] for sm in samples if 'span:main_stream' in sm]), 3) if 'span:main_stream' in samples[0] else None))
  roof['kernels'] = per_kernel
  return roof


def roofline_eager(run_fn, trace_steps, units_per_launch, ms_per_step, side_stream_disc, kernel_units=None):
  """Per-kernel average durations from HIP events recorded on the launch stream(s) (il_trace_*), eager launches of the same kernels;
  `units_per_launch` = updates one step / replay advances (1, or the number of learners on the population path); `kernel_units` = updates ONE kernel launch advances when that
  differs (a population replayed as g parallel sub-populations: learners / g per launch; the per-kernel rates are then lower bounds - a launch shares the chip with the other
  branches' launches for part of its HIP-event window)."""
  from imitation_learning_amd import _lib
  bytes_k, flops_k, update_bytes, update_flops = algorithmic_model()
  ku = units_per_launch if kernel_units is None else kernel_units
  L = _lib.lib()
  L.il_trace_enable(1)
  for _ in range(trace_steps):
    run_fn()
  buf = C.create_string_buffer(1 << 16)
  _lib.check(L.il_trace_report(buf, len(buf)))
  L.il_trace_enable(0)
  kern = {}
  for line in buf.value.decode().strip().splitlines():
    name, cnt, tot = line.split()
    kern[name] = dict(launches_per_update=int(cnt) / trace_steps, avg_us=float(tot) / int(cnt) * 1e3)
  # dominant = largest share of the critical path: the discriminator kernels run on the side stream next to the SAC forward kernels
  side = ('k_gail_grad', 'k_gail_reduce', 'k_gail_reward') if side_stream_disc else ()
  dom = max((k for k in kern if k not in side), key=lambda k: kern[k]['avg_us'] * kern[k]['launches_per_update'])
  per_kernel = {}
  for k, v in kern.items():
    e = dict(avg_us=round(v['avg_us'], 3), launches_per_update=v['launches_per_update'])
    if k in bytes_k:
      e['hbm_GBps'] = round(ku * bytes_k[k] / (v['avg_us'] * 1e-6) / 1e9, 2)
    if k in flops_k:
      e['fp32_TFLOPs'] = round(ku * flops_k[k] / (v['avg_us'] * 1e-6) / 1e12, 3)
    per_kernel[k] = e
  if dom in flops_k:
    ach = ku * flops_k[dom] / (kern[dom]['avg_us'] * 1e-6) / 1e12
    roof = dict(bound='mfma', kernel=dom, note='fp32: MFMA f32 rate == VALU f32 rate == 157.3 TFLOP/s on gfx950', achieved=round(ach, 3), peak=FP32_PEAK_TFLOPS, unit='TFLOP/s',
                frac=round(ach / FP32_PEAK_TFLOPS, 5), traffic=None)
  else:
    ach = ku * bytes_k.get(dom, 0) / (kern[dom]['avg_us'] * 1e-6) / 1e9
    roof = dict(bound='hbm', kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 5), traffic=None)
  roof['duration_source'] = 'HIP events around EAGER launches of the same kernels (il_trace_*): an upper bound on each duration (the window includes device-side waits the timed schedule overlaps)'
  upd_gbs = units_per_launch * update_bytes / (ms_per_step * 1e-3) / 1e9
  # Whole-update fractions at the top level. The contract figure is the HBM fraction; the BINDING roof of this path is fp32 compute:
  # 0.587 GFLOP at 157.3 TFLOP/s is 3.7 us per update = 268k updates/s = 22 % of the HBM roof, so hbm_frac cannot exceed 0.22 before fp32_frac reaches 1.
  roof['hbm_frac'] = round(upd_gbs / HBM_PEAK_GBS, 5)
  roof['fp32_frac'] = round(units_per_launch * update_flops / (ms_per_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5)
  roof['binding_roof'] = 'fp32 (exact-fp32 MFMA = VALU rate): the update needs 3.7 us of fp32 issue but only 0.8 us of HBM time; `frac` above is the dominant kernel against that roof'
  roof['update'] = dict(algorithmic_bytes=update_bytes, achieved_GBps=round(upd_gbs, 2), hbm_frac=round(upd_gbs / HBM_PEAK_GBS, 5), algorithmic_flops=update_flops,
                        fp32_frac=round(units_per_launch * update_flops / (ms_per_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5),
                        sum_kernel_us=round(sum(v['avg_us'] * v['launches_per_update'] for v in kern.values()), 2))
  roof['kernels'] = per_kernel
  return roof


def self_launch(args) -> int:
  """`python bench.py --gpus N` with no rank environment: start the N ranks here (python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), pass rank 0's
  JSON line through as the last line of stdout and return the launcher's exit code. Fewer than N visible GPUs is an error, not an N = 1 result. If the run with the
  peer-window gradient exchange fails outright (a rank raised or was ended by its watchdog), ONE retry with IL_PEER_EXCHANGE=0 (RCCL all-reduces); the retry is named in
  `config.exchange_fallback`."""
  import socket
  import subprocess
  n, have = args.gpus, torch.cuda.device_count()
  if have < n and os.environ.get('IL_BENCH_SHARE_GPU') != '1':
    print(f'bench.py: --gpus {n} needs {n} visible GPUs, this host shows {have}: refusing to report an N = {n} number from fewer devices '
          '(IL_BENCH_SHARE_GPU=1 lets test ranks share a device over gloo)', file=sys.stderr)
    return 2
  attempts = [({}, None)]
  os.environ.setdefault('IL_PEER_EXCHANGE', '1')   # the benchmark A/Bs both exchanges in one job (main()); the product's default is RCCL (parallel.DataParallelUpdate)
  if os.environ.get('IL_PEER_EXCHANGE', '1') != '0':
    attempts.append((dict(IL_PEER_EXCHANGE='0'), 'the run with the peer-window gradient exchange failed (launcher exit code {rc}); this is the retry with IL_PEER_EXCHANGE=0 (RCCL all-reduces)'))
  rc, note = 1, None
  for extra, why in attempts:
    sock = socket.socket(); sock.bind(('127.0.0.1', 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), **extra)
    if why is not None:
      note = why.format(rc=rc)
      print(f'[bench] {note}', file=sys.stderr)
    try:
      r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=float(os.environ.get('IL_BENCH_LAUNCH_TIMEOUT_S', 1500)))
      rc, out = r.returncode, r.stdout
    except subprocess.TimeoutExpired as e:
      rc, out = 124, (e.stdout or '') if isinstance(e.stdout, str) else ''
    lines = [l for l in out.strip().splitlines() if l.strip()]
    last = next((l for l in reversed(lines) if l.startswith('{') and '"metric"' in l), None)
    for l in lines:
      if l is not last: print(l, file=sys.stderr)
    if rc == 0 and last is not None:
      j = json.loads(last)
      j.setdefault('config', {})['launched_by'] = f'bench.py itself (torch.distributed.run, {n} ranks)'
      if note: j['config']['exchange_fallback'] = note
      print(json.dumps(j), flush=True)
      return 0
  return rc or 1


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2000)
  ap.add_argument('--warmup', type=int, default=200)
  ap.add_argument('--repeats', type=int, default=5, help='timed repeats; `value` is their median (SURVEY.md 8d)')
  ap.add_argument('--min-seconds', type=float, default=0.2, help='a timed repeat runs max(--steps, enough replays for this many seconds)')
  ap.add_argument('--stamp-bursts', type=int, default=20, help='extra untimed bursts of 50 replays after the timed repeats, each read for its launch stamps')
  ap.add_argument('--no-pmc', action='store_true', help='do not attach roofline.traffic from profiles/pmc_latest.json (what the counter-collecting passes themselves run with)')
  ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
  ap.add_argument('--launch', choices=('graph', 'direct'), default=os.environ.get('IL_BENCH_LAUNCH', 'direct'), help='how the timed single-GPU updates are issued: UpdatePlan.launch_direct (the two branches as direct launches: two library calls per update, no hipGraph; the default since round 5: 17.5k against 16.6k updates/s on one box, profiles/r05_launch_ab.txt) or two hipGraph replays per update')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--trace-steps', type=int, default=100)
  ap.add_argument('--no-overlap', action='store_true', help='one stream, no device-side hand-off: the same kernels back to back (what a counter-collecting profiler needs; il_sac_update still takes its chained launch)')
  ap.add_argument('--no-population', action='store_true')
  ap.add_argument('--no-secondary', action='store_true', help='skip the SAC-only / discriminator-only / GMMIL / PWIL rates')
  ap.add_argument('--population-learners', type=int, default=128, help='learners of the population line (round 4: 128 as two sub-populations of 64; 64 learners: --population-learners 64)')
  ap.add_argument('--population-wide', type=int, default=0, help='a second, wider population point (learners; sub-populations of 32); 0 = skip')
  ap.add_argument('--population-groups', type=int, default=2, help='sub-populations replayed as parallel graph branches (BatchedPopulationPlan(groups=))')
  ap.add_argument('--learners', type=int, default=1, help='population axis: N independent learners per GPU advanced by one graph replay (aggregate updates/s)')
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args))   # `python bench.py --gpus N` on its own: start the N ranks (torch.distributed.run), pass their JSON line through
  world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
  if world > 1: os.environ.setdefault('IL_PEER_EXCHANGE', '1')   # N > 1: time the peer-window exchange AND the RCCL all-reduces in this one job (exchange_ab), report the better valid one; RCCL is the product's default
  if world != args.gpus:
    sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...` '
             f'(or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)')
  assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback for the HIP path)'
  share = os.environ.get('IL_BENCH_SHARE_GPU') == '1'   # tests only: the ranks time-slice the visible GPU(s) and meet over gloo (RCCL refuses two ranks on one device)
  n_dev = torch.cuda.device_count()
  if world > 1 and n_dev < world and not share:
    sys.exit(f'bench.py: --gpus {world} needs {world} visible GPUs, this host shows {n_dev}: refusing to report an N = {world} number from fewer devices')
  local = local % n_dev if share else local
  torch.cuda.set_device(local)
  device = torch.device('cuda', local)
  import torch.distributed as dist
  from imitation_learning_amd import parallel
  dog = parallel.Watchdog(float(os.environ.get('IL_WATCHDOG_S', 240)), what=f'bench.py --gpus {world}') if world > 1 else None   # a rank that dies inside a collective must not hang the others
  beat = (lambda phase: dog.beat(phase)) if dog is not None else (lambda phase: None)
  backend = os.environ.get('IL_BENCH_BACKEND', 'gloo' if share else 'nccl')
  if world > 1 or os.environ.get('IL_FORCE_ALLREDUCE') == '1':
    import datetime
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29517')
    kw = dict(device_id=device) if backend == 'nccl' else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(60.0, dog.timeout_s if dog else 600.0)), **kw)
  from imitation_learning_amd import _lib
  from imitation_learning_amd.parallel import DataParallelUpdate, broadcast_parameters

  plan, nets, (tr, et) = build(device, rank, learner_id=0 if args.learners > 1 else None)
  if args.no_overlap:
    plan.overlap = False
    plan._set_device_sync(False)
  runner = plan
  if args.learners > 1:
    from imitation_learning_amd import PopulationPlan
    assert world == 1, 'the population axis is a single-GPU mode'
    plans = [plan] + [build(device, rank, seed=l, learner_id=l)[0] for l in range(1, args.learners)]
    runner = PopulationPlan(plans)
  if world > 1 or os.environ.get('IL_FORCE_DP') == '1':  # IL_FORCE_DP=1: the split grads-only / all-reduce / apply path on a single rank
    broadcast_parameters([n.flat if hasattr(n, 'flat') else n for n in nets] + [nets[4].sn])
    runner = DataParallelUpdate(plan)
  gloo_eager = world > 1 and backend != 'nccl'   # gloo collectives synchronise the host: not capturable - eager launches unless the peer-window kernels carry the exchange

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
      torch.cuda.synchronize()

  def measure():
    """warm-up (eager, then captured), the timed repeats, max over ranks. Returns (median seconds of a repeat, launch mode, timing record)."""
    beat('eager warm-up')
    for _ in range(5):
      runner.run()   # loads code objects before capture; the first one sets up communicators / peer windows (self-test + soak)
    torch.cuda.synchronize()
    if runner is not plan and getattr(runner, 'handoff', False) and not runner.agree_on_handoff():   # collective: an expired device-side wait on ANY rank sends every rank to the stream-dependency schedule
      print(f'[bench] rank {rank}: device-side hand-off left (a bounded wait expired on some rank during the eager warm-up); using stream dependencies', file=sys.stderr)
    launch, step = 'eager', runner.run
    if args.launch == 'direct' and runner is plan and not args.no_graph and getattr(plan, 'device_sync', False):
      beat('recording the direct launches')
      plan.record_direct()
      step, launch = plan.launch_direct, 'direct launches (UpdatePlan.launch_direct: two library calls per update, six kernel launches on two streams, no hipGraph)'
      if plan._direct_overlap:   # round 6: back-to-back updates, nothing enqueued between them: the next update's first launch is dispatched while this one's last still runs
        import functools
        step = functools.partial(plan.launch_direct, join=False)
        launch = ('direct launches, SAC branch alternating over two streams (il_sac_update_gather_overlap: forward / critic loss and policy / critic on one, the two optimiser '
                  'launches on another, stage epochs on the device instead of stream order; discriminator branch on a third): two library calls per update, six kernel launches, no hipGraph')
    elif args.launch == 'direct' and runner is not plan and not args.no_graph and getattr(runner, 'direct_launch_ok', None) is not None and runner.direct_launch_ok():
      beat('recording the direct launches (data parallel, exchanges inside the optimiser launches)')
      runner.record_direct()
      step, launch = runner.launch_direct, 'direct launches (DataParallelUpdate.launch_direct: the plan\'s two branches with the gradient exchanges inside the optimiser launches, two library calls per update, no hipGraph)'
    elif not args.no_graph and not (gloo_eager and getattr(runner, 'peer', None) is None):
      beat('graph capture')
      try:
        runner.capture(warmup=0)
        step, launch = runner.replay, 'hipGraph replay'
      except Exception as e:  # e.g. a collective that refuses stream capture: keep measuring, eagerly, and say so
        if world == 1 and os.environ.get('IL_FORCE_ALLREDUCE') != '1':
          raise
        torch.cuda.synchronize()
        launch = f'eager (graph capture failed: {type(e).__name__})'
        print(f'[bench] rank {rank}: graph capture failed, falling back to eager launches: {e}', file=sys.stderr)
    barrier()   # no rank enters the replays while another is still capturing (bounded device-side waits downstream of the exchange)
    beat('warm-up')
    for _ in range(args.warmup):
      step()
    barrier()

    def timed(n):
      """n steps between two barrier + synchronize brackets; seconds, max over ranks."""
      t0 = time.perf_counter()
      for _ in range(n):
        step()
      barrier()
      t = torch.tensor([time.perf_counter() - t0], device=device if backend == 'nccl' else 'cpu', dtype=torch.float64)
      if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
      return float(t.item())

    beat('timed region')
    # `--steps` replays alone can be a ~1 ms window (the driver's --steps 20): every repeat times max(--steps, enough replays for --min-seconds), sized from one pass
    # of --steps (max over ranks, so every rank computes the same count); `value` = the median of --repeats such repeats.
    probe = timed(args.steps)
    n = max(args.steps, int(np.ceil(args.min_seconds / max(probe / args.steps, 1e-9))))
    reps, stamps = [], []
    stamped = args.learners == 1
    if stamped: _lib.check(_lib.lib().il_kernel_stamps_clear())
    for _ in range(max(1, args.repeats)):
      reps.append(timed(n))
      if stamped and rank == 0: stamps.append(stamp_sample(_lib))   # the last replay's launches, read outside the timed bracket
    for _ in range(args.stamp_bursts if stamped else 0):   # more readings of the same schedule: short bursts of back-to-back replays (untimed), the last replay of each is read
      timed(min(n, 50))
      if rank == 0: stamps.append(stamp_sample(_lib))
    return float(np.median(reps)), launch, dict(timed_replays=n, repeats=len(reps), repeat_seconds=[round(r, 6) for r in reps], probe_seconds=round(probe, 6), stamps=stamps)

  def health():
    """Collective verdict on the run just timed: (ok on every rank, replicas bit-identical, digests, note)."""
    ex = runner.exchange_timeouts() if getattr(runner, 'exchange_timeouts', None) is not None else 0
    hs = plan.sync_timeouts() if getattr(plan, 'device_sync', False) else 0
    if getattr(plan, 'device_sync', False) and plan.poisoned(): hs = max(hs, 1)   # (an expired wait also stops the optimiser launches from storing: [IL_SYNC_POISON])
    ok = parallel._agree(ex == 0 and hs == 0) if world > 1 else (ex == 0 and hs == 0)
    same, digests = (parallel.replicas_bit_identical(runner.replica_state()) if runner is not plan and hasattr(runner, 'replica_state') else (True, []))
    note = f'rank {rank}: {ex} expired exchange waits, {hs} expired hand-off waits' if (ex or hs) else None
    return ok, same, digests, note

  elapsed, launch, timing = measure()
  steps_timed = timing['timed_replays']
  exchange_fallback = None
  ok, same, digests, note = health()
  # N > 1: BOTH gradient exchanges timed in this one invocation (an 8-GPU node may be available for a single run per round): first the default - the peer-window exchange,
  # inside the optimiser launches where the schedule allows - then, from rank 0's replica state, torch.distributed all-reduces (RCCL). `value` is the better of the runs
  # that passed the post-run check (no expired wait, replicas bit-identical); config.exchange names it, config.exchange_ab carries both.
  exchange_ab = None
  if world > 1 and getattr(runner, 'peer', None) is not None and ok and same and os.environ.get('IL_BENCH_EXCHANGE_AB', '1') != '0':
    first = dict(exchange=runner.exchange_name(), updates_per_s=round(world * steps_timed / elapsed, 1), ms_per_step=round(elapsed / steps_timed * 1e3, 5), launch=launch,
                 replicas_bit_identical=same, replica_digests=[d[:16] for d in digests], exchange_soak=getattr(runner.peer, 'soak_report', None), valid=True)
    peer_form = getattr(runner.peer, 'form', 0)
    beat('A/B: the collectives')
    runner.use_collectives('A/B run of bench.py: the same job re-timed with torch.distributed all-reduces')
    plan.sync[plan._sync_timeouts] = 0
    if getattr(plan, 'device_sync', False) or plan.poisoned(): plan.clear_poison()   # (an expired wait of the run just discarded must not keep this rank's optimiser launches from storing)
    runner.resync_replicas()
    elapsed2, launch2, timing2 = measure()
    ok2, same2, digests2, note2 = health()
    second = dict(exchange=runner.exchange_name(), updates_per_s=round(world * timing2['timed_replays'] / elapsed2, 1), ms_per_step=round(elapsed2 / timing2['timed_replays'] * 1e3, 5), launch=launch2,
                  replicas_bit_identical=same2, replica_digests=[d[:16] for d in digests2], valid=bool(ok2 and same2), note=note2)
    exchange_ab = dict(peer=first, collectives=second, chosen='peer' if (not second['valid'] or first['ms_per_step'] <= second['ms_per_step']) else 'collectives')
    if exchange_ab['chosen'] == 'collectives':
      elapsed, launch, ok, same, digests, note, timing = elapsed2, launch2, ok2, same2, digests2, note2, timing2
      steps_timed = timing['timed_replays']
    else:
      exchange_ab['reported_exchange'] = first['exchange']   # (the runner itself now sits on the collectives; the line below reports the run that was chosen)
  if world > 1 and getattr(runner, 'peer', None) is not None and not (ok and same):
    # One retry, in process, on the collectives: the peer-window exchange delivered late (expired waits) or wrong (replicas apart). Rank 0's state everywhere, then RCCL.
    exchange_fallback = f'peer-window exchange failed the post-run check ({"replicas differ" if not same else "expired waits"}{"; " + note if note else ""}): re-timed with torch.distributed all-reduces'
    print(f'[bench] rank {rank}: {exchange_fallback}', file=sys.stderr)
    beat('fall-back to the collectives')
    runner.use_collectives(exchange_fallback)
    plan.sync[plan._sync_timeouts] = 0
    if getattr(plan, 'device_sync', False) or plan.poisoned(): plan.clear_poison()   # (an expired wait of the run just discarded must not keep this rank's optimiser launches from storing)
    runner.resync_replicas()
    elapsed, launch, timing = measure()
    steps_timed = timing['timed_replays']
    ok, same, digests, note = health()
  finite = all(bool(torch.isfinite(n.flat if hasattr(n, 'flat') else n).all()) for n in nets)
  if not ok:
    raise RuntimeError(f'device-side waits expired during the timed run ({note}): the result is invalid (IL_PEER_EXCHANGE=0 selects RCCL all-reduces, IL_DEVICE_SYNC=0 stream dependencies)')
  if not same:
    raise RuntimeError(f'data-parallel replicas are NOT bit-identical after the timed run (digests {digests}): the gradient exchange delivered different values to different ranks')
  beat('reporting')

  single = None
  if world > 1 and os.environ.get('IL_BENCH_SINGLE', '1') != '0':
    # The N = 1 schedule (one GPU's own update plan: no split path, no exchange) timed on every rank at the same time, each on its own GPU and its own fresh learner, in this
    # same job: data-parallel efficiency = (value / n_gpus) / config.single_gpu_same_job.updates_per_s is computable from this one line.
    beat('N = 1 schedule, same job')
    from imitation_learning_amd import training as il_training
    plan1, nets1, _ = build(device, rank, seed=1000 + rank, learner_id=1000 + rank)
    plan1.capture(warmup=3)
    for _ in range(min(args.warmup, 300)): plan1.replay()
    barrier()
    n1 = min(args.steps, 2000)
    t1 = time.perf_counter()
    for _ in range(n1): plan1.replay()
    torch.cuda.synchronize()
    e1 = time.perf_counter() - t1
    assert plan1.sync_timeouts() == 0
    single = dict(updates_per_s=round(n1 / e1, 1), ms_per_step=round(e1 / n1 * 1e3, 5), steps=n1, note=f'rank 0 of {world}, all ranks running their own single-GPU plan concurrently (one per GPU)')
    barrier()
    del plan1, nets1
  if rank == 0:
    ms_per_step = elapsed / steps_timed * 1e3
    ups = world * args.learners * steps_timed / elapsed
    # ---- per-kernel durations: device stamps taken inside the timed graph replays (il_kernel_stamps). Schedules whose kernels carry no stamps (IL_PAIR=0, shapes outside
    # pair mode, --no-graph --no-overlap runs under a counter-collecting profiler) fall back to HIP events around eager launches of the same kernels.
    samples = [sm for sm in timing['stamps'] if 'k_sac_chain_pair' in sm and 'k_policy_critic_pair' in sm and 'k_dw_adam_actor' in sm]
    if samples and world == 1:
      roof = roofline_from_stamps(samples, ms_per_step, use_pmc=not args.no_pmc)
    else:
      # (traced with the index draw stream-ordered ahead of the forward / critic-loss launch: with the resident draw that launch is dispatched while the draw is still
      #  waiting for the previous update, so its HIP-event window would include that wait instead of its own work; both schedules are bit-identical)
      fused_desc, plan.peer_desc = getattr(plan, 'peer_desc', None), None   # (data-parallel runs: the per-kernel windows are taken on this rank's plain single-GPU launches - the fused exchange lives on the resident-sampler schedule only, and rank 0 alone runs this section)
      plan.stream_ordered_draw = True
      roof = roofline_eager(plan.run, args.trace_steps, 1, ms_per_step, getattr(plan, 'overlap', False))
      plan.stream_ordered_draw = False
      plan.peer_desc = fused_desc
      if samples:   # N > 1: rank 0's stamps of the data-parallel launches, beside the eager windows of its single-GPU launches
        roof['stamped_kernels_data_parallel'] = roofline_from_stamps(samples, ms_per_step, use_pmc=False)['kernels']
    roof['timing'] = {k: v for k, v in timing.items() if k != 'stamps'}

    out = dict(metric='SAC+GAIL grad-updates/sec (batch 256, HalfCheetah dims)', value=round(ups, 1), unit='updates/s', n_gpus=world, steps=args.steps, warmup=args.warmup, timed_replays=steps_timed, repeats=timing['repeats'],
               ms_per_step=round(ms_per_step, 5), higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload='algorithm=GAIL env=halfcheetah: 2 replay samples + discriminator step (BCE+GP+SN) + AIRL relabel + sac_update per step',
                           batch_per_gpu=B, global_batch=B * world, state_dim=S, action_dim=A, hidden=H, replay_capacity=1_000_000, replay_fill=100_000, expert_rows=25_000,
                           learners_per_gpu=args.learners, parallelism=f'dp{world}' + ('(split path' + (', device hand-off between the discriminator and SAC branches, own communicator per branch)' if getattr(runner, 'handoff', False) else ', stream dependencies)') if runner is not plan else ''), launch=launch, noise='on-chip Philox4x32-10', finite=finite,
                           gradient_exchange=(None if runner is plan else ('the peer-window exchange (see exchange_ab.peer)' if exchange_ab and exchange_ab['chosen'] == 'peer' else None) or ('one kernel per sync point over peer-mapped windows (il_peer_allreduce_mean: push to every rank, rank-ordered sum; ' + ('write-through payload, no fences' if getattr(runner.peer, 'form', 0) else 'system-scope fences') + ')'
                                                                             if getattr(runner, 'peer', None) is not None else f'{backend.replace("nccl", "RCCL")} all-reduce (mean) per sync point')),
                           exchange=(None if runner is plan else (exchange_ab['reported_exchange'] if exchange_ab and 'reported_exchange' in exchange_ab else runner.exchange_name())),
                           exchange_ab=exchange_ab, single_gpu_same_job=single,
                           dp_efficiency=(round(ups / world / single['updates_per_s'], 4) if single else None),
                           exchange_note=(None if runner is plan else runner.peer_note), exchange_fallback=exchange_fallback,
                           exchange_soak=(exchange_ab['peer']['exchange_soak'] if exchange_ab else (getattr(runner.peer, 'soak_report', None) if getattr(runner, 'peer', None) is not None else None)),
                           replicas_bit_identical=(same if runner is not plan else None), replica_digests=([d[:16] for d in digests] if runner is not plan else None),
                           branch_sync=('device counters (two graphs, no cross-stream edge)' if getattr(plan, 'device_sync', False) and runner is plan else 'stream dependencies'),
                           rows=('read from the rings through the drawn indices (il_batch.gather); reward relabel as a role of k_sac_chain_pair (16-wave schedule: inline in the critic-loss workgroups)' if getattr(plan, 'inline_relabel', False) and runner is plan
                                 else ('read from the rings through the drawn indices (il_batch.gather)' if getattr(plan, 'ring_mode', False) and runner is plan else 'gathered by k_gather2'))),
               roofline=roof)
    out['config']['handoffs'] = ('in-launch hand-offs between the two streams: producers write through (sc0 sc1) and drain, consumers read below the caches or acquire with every wave; soaked beside a '
                                 'busy neighbour process: 0 mismatching 50k-update runs in 108 (profiles/r06_soak_under_load.md; the round-5 forms: 27 %, at 2 % more updates/s)')
    if world == 1 and args.learners == 1 and not args.no_population:
      # population axis (the reference's own usage: 10-seed sweeps / Ax trials): independent batch-256 learners advanced by the
      # SAME launches (learner id = grid dimension). Reported next to, never instead of, the single-learner `value`.
      from imitation_learning_amd import BatchedPopulationPlan
      Lp = args.population_learners
      pop = BatchedPopulationPlan([build(device, rank, seed=100 + l, learner_id=100 + l)[0] for l in range(Lp)], groups=args.population_groups)
      for _ in range(3):
        pop.run()
      torch.cuda.synchronize()
      pop.capture()
      for _ in range(30):
        pop.replay()
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(200):
        pop.replay()
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t1) / 200
      proof = roofline_eager(pop.run, 10, Lp, dt * 1e3, False, kernel_units=Lp // max(1, args.population_groups))
      if args.population_groups > 1: proof['note'] = proof.get('note', '') + f'; {args.population_groups} sub-populations run as parallel branches: `frac` is a lower bound (a launch covers {Lp // args.population_groups} learners and shares the chip with the other branch for part of its HIP-event window), the whole-replay fp32_frac is exact'
      proof.pop('kernels', None)
      out['population'] = dict(learners=Lp, groups=args.population_groups, aggregate_updates_per_s=round(Lp / dt, 1), ms_per_replay=round(dt * 1e3, 5), roofline=proof,
                               note=f'{Lp} independent batch-256 SAC+GAIL learners (il_*_population launches: learner id = grid dimension, learner l on XCD l % 8; {args.population_groups} sub-populations as parallel graph branches), own replay ring / index stream / Philox counter each')
      del pop
      torch.cuda.empty_cache()
      if args.population_wide > Lp:
        # the same axis one step wider (round 4): `population_wide` learners as population_wide / 32 sub-populations of 32 - the launches of more branches fill each other's
        # dependent phases. Rate only (the per-kernel roofline above is the 64-learner one).
        Lw, gw = args.population_wide, max(1, args.population_wide // 32)
        popw = BatchedPopulationPlan([build(device, rank, seed=100 + l, learner_id=100 + l)[0] for l in range(Lw)], groups=gw)
        for _ in range(3):
          popw.run()
        torch.cuda.synchronize()
        popw.capture()
        for _ in range(20):
          popw.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(100):
          popw.replay()
        torch.cuda.synchronize()
        dtw = (time.perf_counter() - t1) / 100
        _, _, _, update_flops = algorithmic_model()
        out['population']['wide'] = dict(learners=Lw, groups=gw, aggregate_updates_per_s=round(Lw / dtw, 1), ms_per_replay=round(dtw * 1e3, 5),
                                         fp32_frac=round(Lw * update_flops / dtw / 1e12 / FP32_PEAK_TFLOPS, 5))
        del popw
        torch.cuda.empty_cache()
    if world == 1 and args.learners == 1 and not args.no_secondary:
      out['secondary'] = secondary(device, plan, nets)
    if world == 1 and not args.no_cpu_baseline:
      port, ref = cpu_baseline(tr, et), cpu_reference()
      out['cpu_baseline'], out['cpu_port'] = (ref, port) if ref is not None else (port, None)
      if out['cpu_port'] is None: out.pop('cpu_port')
  if dist.is_initialized():
    beat('teardown')
    dist.barrier()
    dist.destroy_process_group()
  if dog is not None: dog.stop()
  if rank == 0:
    C.CDLL(None).fflush(None)   # RCCL writes its banner through C stdio (possibly at teardown): drain it so that the JSON line is the LAST line of stdout
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
  main()
