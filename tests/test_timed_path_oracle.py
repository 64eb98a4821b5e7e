"""`-m gpu`: the TIMED path against the CPU oracle, by record and replay.

What `bench.py` measures is `UpdatePlan` at the BASELINE configuration (algorithm=GAIL, HalfCheetah dims, B = 256, ring 1e6 / fill 1e5, 25k expert
rows): device-side MT19937 index draws, rows read from the rings through the indices, on-chip Philox noise, inline reward relabel, two hipGraphs
handing over through device counters (reference train.py:171-203).  The per-function parity tests feed injected noise through a different branch of
the kernels; here the captured plan itself is run, what it consumed is RECORDED -- the index stream is bit-exact with the oracle's own MT19937, the
noise of update k is a pure function of (key, k, stream) exported by `il_noise_fill` -- and the oracle (`oracle.replay` -> `oracle.gail.gail_update`
-> `predict_reward` -> `oracle.sac.sac_update`) REPLAYS the same updates from the same initial state.  Every tensor is compared at the bounds of
tests/gpu_util.py.  Same for a 3-learner `BatchedPopulationPlan` and for the acting launch `il_act_step` (f2).
"""
import ctypes as C

import numpy as np
import pytest
import torch

import inputs as gi
from oracle import gail as ogail
from oracle import nets as onets
from oracle import philox
from oracle import replay as oreplay
from oracle import sac as osac
from oracle.mt19937 import MT19937

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
  import bench
  import imitation_learning_amd as il
  from imitation_learning_amd import _lib
  from imitation_learning_amd import training as il_training
  from gpu_util import DEV, N, Cfg, bracket, close, close_params, close_sparse, crit_from_flat

S, A, H, HD, B = 18, 6, 256, 64, 256
LR, LR_D, WD_D, DISCOUNT, POLYAK, ENT = 3e-4, 3e-5, 10.0, 0.97, 0.99, -0.5 * 6


def record_noise(seed, ctr, stream_id, n):
  """What update `ctr` drew from noise stream `stream_id` (include/il_hip.h il_noise_fill: the kernels' own device functions)."""
  out = torch.empty(n, device=DEV)
  _lib.check(_lib.lib().il_noise_fill(C.c_uint64(seed), ctr, stream_id, n, _lib.ptr(out), _lib.stream_ptr()))
  return N(out)


# ------------------------------------------------------------------------------------------------ the recorder itself
def test_noise_recorder_matches_the_independent_philox_restatement():
  """il_noise_fill against oracle/philox.py (Random123 known answers are checked on the CPU side): uniforms bit-exact (pure integer path), normals to a
  few float32 ulp of numpy's log / cos / sqrt, so the recorder cannot share a mistake with the kernels unnoticed."""
  for seed, ctr in ((0, 0), (3, 11), (0x1234_5678_9ABC_DEF0, 70000)):
    np.testing.assert_array_equal(record_noise(seed, ctr, philox.STREAM_GP, 4096), philox.uniform(seed, ctr, philox.STREAM_GP, 4096))
    np.testing.assert_array_equal(record_noise(seed, ctr, philox.STREAM_MIX, 300), philox.uniform(seed, ctr, philox.STREAM_MIX, 300))
    for stream in (philox.STREAM_EPS_NEXT, philox.STREAM_EPS_CUR, philox.STREAM_ACT):
      got, want = record_noise(seed, ctr, stream, 1 << 15), philox.normal(seed, ctr, stream, 1 << 15)
      assert np.abs(got - want).max() <= 4e-6, (seed, ctr, stream, np.abs(got - want).max())   # |x| <= 6: a few ulp of the factors, absolute because cos() crosses 0


def test_philox_streams_are_standard_normal_and_uniform():
  """Moments and a Kolmogorov-Smirnov test on 2^20 draws of each kind (the old check was 0.05 < std < 1)."""
  from scipy import stats
  n = 1 << 20
  z = record_noise(12345, 7, philox.STREAM_EPS_CUR, n).astype(np.float64)
  assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 4 * np.sqrt(2 / n)
  assert abs(stats.skew(z)) < 4 * np.sqrt(6 / n) and abs(stats.kurtosis(z)) < 4 * np.sqrt(24 / n)
  assert stats.kstest(z, 'norm').pvalue > 1e-3
  u = record_noise(12345, 7, philox.STREAM_GP, n).astype(np.float64)
  assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 4 / np.sqrt(12 * n) and stats.kstest(u, 'uniform').pvalue > 1e-3
  # different counters / streams / keys are different, uncorrelated sequences
  for other in (record_noise(12345, 8, philox.STREAM_EPS_CUR, n), record_noise(12345, 7, philox.STREAM_EPS_NEXT, n), record_noise(12346, 7, philox.STREAM_EPS_CUR, n)):
    assert abs(np.corrcoef(z, other)[0, 1]) < 5 / np.sqrt(n)


# ------------------------------------------------------------------------------------------------ oracle twin of a learner
class OracleLearner:
  """The oracle's copy of one bench learner: same initial parameters / buffers, its own MT19937 stream."""

  def __init__(self, nets, plan, tr, et, index_seed):
    actor, critic, target, log_alpha, disc = nets
    self.st = osac.SacState(S, A, H)
    self.st.actor[:], self.st.critic[:], self.st.target[:] = N(actor.flat), crit_from_flat(critic, critic.flat), crit_from_flat(critic, target.flat)
    self.st.log_alpha[:] = N(log_alpha)
    self.ds = ogail.DiscState(S + A, HD, True)
    self.ds.unpack_into(N(disc.flat))
    v = disc.views()
    for k in ('u1', 'v1', 'u2', 'v2'):
      getattr(self.ds, k)[...] = N(v[k])
    n = tr['states'].shape[0]
    self.mem = oreplay.ReplayOracle(plan.memory.size, S, A, True)
    for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
      getattr(self.mem, k)[:n] = tr[k]
    self.mem.step[:n] = np.arange(1, n + 1, dtype=np.float32)
    self.mem.idx = n
    self.emem = oreplay.ReplayOracle(et['states'].shape[0], S, A, True, transitions={**et, 'num_trajectories': 25})
    assert (self.mem.idx, self.mem.full) == (plan.memory.idx, plan.memory.full) and (self.emem.idx, self.emem.full) == (plan.expert_memory.idx, plan.expert_memory.full)
    self.gen = MT19937(index_seed)
    self.key = int(plan.sac.noise_seed)
    assert int(plan.disc.noise_seed) == self.key and plan.sac.noise_counter == plan.disc.noise_counter, 'one Philox key / counter per learner'

  def update(self, k, masks=None):
    """Update #k (0-based) of train.py:173-203, consuming the recorded draws of counter k. Returns (idx, eidx, rewards, logp, q).
    masks: the HIP path's ReLU decisions of this update (relu_masks()): the SAC step then back-propagates through the units the device back-propagated through."""
    cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
    idx, eidx = self.mem.sample_idx(self.gen, B), self.emem.sample_idx(self.gen, B)   # train.py:173: agent batch first, then the expert batch, one stream
    b, e = self.mem.gather(idx), self.emem.gather(eidx)
    eps_gp = record_noise(self.key, k, philox.STREAM_GP, B)
    eps_next, eps_cur = record_noise(self.key, k, philox.STREAM_EPS_NEXT, B * A).reshape(B, A), record_noise(self.key, k, philox.STREAM_EPS_CUR, B * A).reshape(B, A)
    ogail.gail_update(self.ds, cat(b), b['weights'], cat(e), e['weights'], eps_gp, lr=LR_D, weight_decay=WD_D, grad_penalty=1.0)        # train.py:178-180
    b['rewards'] = ogail.predict_reward(self.ds, cat(b), 'AIRL')                                                                      # train.py:192-194
    self.last_x = cat(b)   # the rows the rewards were predicted on (the bracket below re-evaluates them on the HIP path's own discriminator state)
    logp, q = osac.sac_update(self.st, b, eps_next, eps_cur, discount=DISCOUNT, entropy_target=ENT, polyak_factor=POLYAK, lr=LR, masks=masks)       # train.py:203
    return np.array(idx), np.array(eidx), b['rewards'], logp, q


def relu_masks(plan):
  """il_sac.debug_masks of the update that has just run, in oracle.sac.sac_update's layout."""
  m = N(plan.relu_masks) > 0.5
  return dict(actor=[m[0], m[1]], critic=[[m[2], m[3]], [m[4], m[5]]], pcritic=[[m[6], m[7]], [m[8], m[9]]])


def compare_learner_masked(o, nets, plan, k, tag='masked oracle: '):
  """The SAC tensors of a learner against the oracle that replayed every update WITH the HIP path's ReLU decisions: what remains is fp32 re-association, so
  the tight bound (rtol 1e-5 + 1e-5 of the scale) must hold for all but 1e-5 of the elements of EVERY tensor - no allowance for moved rows, twin critics included.
  This is the gate; the unmasked comparison below keeps its documented allowance (two correct evaluations do differ there)."""
  actor, critic, target, log_alpha, disc = nets
  ao, co, to = plan._keep[4], plan._keep[5], plan._keep[6]
  s, f = 1e-5 * k, 1e-5
  # (the actor: 73,740 elements - 1e-5 of them is less than ONE element. Adam turns ulp-level gradient noise on an element whose gradient is ~eps into a fraction of a step
  #  (close_params); seeds 4 and 5 leave 2 and 1 such elements 7e-6 outside the bound after 12 updates. Up to four elements; a moved ROW is 256.)
  fa = max(f, 4.0 / actor.flat.numel())
  close_params(N(actor.flat), o.st.actor, f'{tag}actor after {k}', LR, k, outlier_frac=fa); close_params(crit_from_flat(critic, critic.flat), o.st.critic, f'{tag}critic after {k}', LR, k, outlier_frac=f)
  close_params(crit_from_flat(critic, target.flat), o.st.target, f'{tag}target after {k}', LR, k, outlier_frac=f)
  close(N(log_alpha), o.st.log_alpha, f'{tag}log_alpha after {k}', atol_scale=s)
  close_sparse(N(ao.exp_avg), o.st.actor_m, f'{tag}actor exp_avg', atol_scale=s, outlier_frac=f); close_sparse(N(ao.exp_avg_sq), o.st.actor_v, f'{tag}actor exp_avg_sq', atol_scale=s, outlier_frac=f)
  close_sparse(crit_from_flat(critic, co.exp_avg), o.st.critic_m, f'{tag}critic exp_avg', atol_scale=s, outlier_frac=f); close_sparse(crit_from_flat(critic, co.exp_avg_sq), o.st.critic_v, f'{tag}critic exp_avg_sq', atol_scale=s, outlier_frac=f)


# the unmasked twin-critic comparison's allowance: what the three seeds of test_captured_update_plan_replays_through_the_oracle and the three learners of the population
# test measure (tolerance ledger) + 25 %
CRITIC_UNMASKED_FRAC = 1.25e-3   # measured (round 5, gpurun_out/r05a/fractions.json): 0 on the single learner's seeds, 1.0007e-3 = ONE flipped ReLU row on learner 1 of the population


def compare_learner(o, nets, plan, k, tag=''):
  """Every persistent tensor of the learner after k updates, at the tests/gpu_util.py bounds."""
  actor, critic, target, log_alpha, disc = nets
  ao, co, to, do = plan._keep[4], plan._keep[5], plan._keep[6], plan._keep[8]
  s = 1e-5 * k
  # Twin critics after a CHAIN of updates: one ReLU pre-activation within rounding of 0 that takes the other sign moves a whole row of a W2 (256 of 145k elements = 1.8e-3 ... the
  # measured worst case over the learners of these tests is 1.0e-3 of the elements, round 3; everything else uses the 5e-4 default of tests/gpu_util.py). The run prints the
  # largest measured fractions at its end (conftest.pytest_terminal_summary).
  # (gate=False: the fraction is recorded, not warned about - compare_learner_masked is the gate of these tests; this comparison documents how far two correct evaluations drift)
  close_params(N(actor.flat), o.st.actor, f'{tag}actor after {k}', LR, k); close_params(crit_from_flat(critic, critic.flat), o.st.critic, f'{tag}critic after {k}', LR, k, outlier_frac=CRITIC_UNMASKED_FRAC, gate=False)
  close_params(crit_from_flat(critic, target.flat), o.st.target, f'{tag}target after {k}', LR, k, outlier_frac=CRITIC_UNMASKED_FRAC, gate=False)
  close(N(log_alpha), o.st.log_alpha, f'{tag}log_alpha after {k}', atol_scale=s)
  close_sparse(N(ao.exp_avg), o.st.actor_m, f'{tag}actor exp_avg', atol_scale=s); close_sparse(N(ao.exp_avg_sq), o.st.actor_v, f'{tag}actor exp_avg_sq', atol_scale=s)
  close_sparse(crit_from_flat(critic, co.exp_avg), o.st.critic_m, f'{tag}critic exp_avg', atol_scale=s); close_sparse(crit_from_flat(critic, co.exp_avg_sq), o.st.critic_v, f'{tag}critic exp_avg_sq', atol_scale=s)
  close(N(to.exp_avg), o.st.alpha_m, f'{tag}alpha exp_avg', atol_scale=s)
  close(N(disc.flat), o.ds.pack(), f'{tag}discriminator after {k}', atol_scale=4e-6 * k)
  close(N(do.exp_avg), o.ds.m, f'{tag}discriminator exp_avg', atol_scale=4e-6 * k); close(N(do.exp_avg_sq), o.ds.v, f'{tag}discriminator exp_avg_sq', atol_scale=4e-6 * k)
  for nm, val in disc.views().items():
    close(N(val), getattr(o.ds, nm), f'{tag}{nm} after {k}', atol_scale=4e-6 * k)


def per_update_outputs(plan):
  return N(plan.idx), N(plan.eidx), N(plan.rewards), N(plan.logp), N(plan.q)


def reward_bracket(o, disc, rew, k, tag=''):
  """The relabelled rewards against a float64 evaluation, isolated from the (separately compared) parameter state: the oracle's reward function in float32 and in float64 on
  the HIP path's OWN discriminator state after update k (weights, biases, u / v read back from the device) and the oracle's rows. AIRL's log D - log1p(-D) is
  ill-conditioned near D = 1/2, so the float32 oracle is itself off by up to ~1e-3 relative; the HIP rewards may be at most twice as far from float64 (+ 1e-6 of the scale)."""
  import copy
  ds = copy.deepcopy(o.ds)
  for nm, val in disc.views().items():
    getattr(ds, nm)[...] = N(val).reshape(getattr(ds, nm).shape)
  return bracket(rew, ogail.predict_reward(ds, o.last_x, 'AIRL'), ogail.predict_reward_f64(ds, o.last_x, 'AIRL'), f'{tag}relabelled rewards of update {k} vs float64')


def compare_outputs(got, want, k, tag=''):
  idx, eidx, rew, logp, q = got
  oidx, oeidx, orew, ologp, oq = want
  np.testing.assert_array_equal(idx, oidx, err_msg=f'{tag}agent index draw of update {k}'); np.testing.assert_array_equal(eidx, oeidx, err_msg=f'{tag}expert index draw of update {k}')
  s = 2e-6 * (k + 1)
  close(rew, orew, f'{tag}relabelled rewards of update {k}', rtol=1e-4, atol_scale=1e-5)   # coarse (two float32 evaluations on two float32 states); the tight statement is reward_bracket()
  close(logp, ologp, f'{tag}log pi of update {k}', atol_scale=s); close(q, oq, f'{tag}min Q of update {k}', atol_scale=s)


# ------------------------------------------------------------------------------------------------ a23: the single-learner plan bench.py times
@pytest.mark.parametrize('SEED', [3, 4, 5])
def test_captured_update_plan_replays_through_the_oracle(SEED):
  WARM, K = 2, 10
  il_training._NOISE.clear(); il_training._WS.clear()
  plan, nets, (tr, et) = bench.build(torch.device(DEV), 0, seed=SEED)
  o = OracleLearner(nets, plan, tr, et, index_seed=SEED)
  om = OracleLearner(nets, plan, tr, et, index_seed=SEED)   # its twin, replaying every update with the HIP path's ReLU decisions (the gate: compare_learner_masked)
  plan.record_relu_masks()
  for k in range(WARM):              # WARM eager updates (they count) ...
    plan.run(); torch.cuda.synchronize()
    o.update(k); om.update(k, masks=relu_masks(plan))
  plan.capture(warmup=0)             # ... then the two graphs
  assert plan.device_sync and plan.ring_mode and plan.inline_relabel and plan.graph_side is not None, 'this must be the schedule bench.py times: device hand-off, ring reads, inline relabel, two graphs'
  compare_learner(o, nets, plan, WARM, 'eager warm-up: ')
  for k in range(WARM, WARM + K):
    plan.replay()
    torch.cuda.synchronize()
    got = per_update_outputs(plan)
    compare_outputs(got, o.update(k), k)
    om.update(k, masks=relu_masks(plan))
    reward_bracket(o, nets[4], got[2], k)
  assert plan.sync_timeouts() == 0
  assert int(N(il_training._noise_counter(nets[0].flat.device))[0]) == WARM + K, 'one Philox counter tick per update'
  compare_learner_masked(om, nets, plan, WARM + K)
  compare_learner(o, nets, plan, WARM + K)
  final = [N(n.flat if hasattr(n, 'flat') else n) for n in nets]
  # test of the test: ONE row of one critic's W2 moved by one Adam step - the error a real defect in one sample's back-propagation would leave, and what the unmasked
  # comparison's allowance (1.5e-3 of the elements, each up to an Adam step) cannot tell from a ReLU flip - must fail the masked comparison
  keep = om.st.critic.copy()
  row = S + A + 1   # (any row of W2: offset H * IN + H + row * H in critic_1)
  om.st.critic[H * (S + A) + H + row * H: H * (S + A) + H + (row + 1) * H] += np.float32(LR)
  import warnings
  import gpu_util
  n_before = len(gpu_util.FRACTIONS)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    with pytest.raises(AssertionError, match='critic'):
      compare_learner_masked(om, nets, plan, WARM + K)
  del gpu_util.FRACTIONS[n_before:]   # (the deliberate failure is not a measurement)
  om.st.critic[:] = keep

  # the same replays back to back with no host synchronisation in between (the timed regime) end in the same bits
  il_training._NOISE.clear(); il_training._WS.clear()
  plan2, nets2, _ = bench.build(torch.device(DEV), 0, seed=SEED)
  plan2.capture(warmup=WARM)
  for _ in range(K):
    plan2.replay()
  torch.cuda.synchronize()
  assert plan2.sync_timeouts() == 0
  for a, n in zip(final, nets2):
    np.testing.assert_array_equal(a, N(n.flat if hasattr(n, 'flat') else n))


@pytest.mark.gpu
def test_expired_wait_poisons_the_learner_and_the_weights_stay():
  """[IL_SYNC_POISON] (round 6): a bounded device-side wait that gives up must not let its update reach the weights. The SAC branch is launched ALONE with a bound of four
  polls - its wait for the index draw and the relabel's wait for the discriminator step expire - and, separately, the discriminator branch alone: afterwards every
  parameter, Adam moment, the target network, log alpha and the spectral-norm vectors hold the bits they had before; the device flag, the time-out counter and the pinned
  host word are raised, and the next launch_direct() / replay() raises on the host instead of at the next logging interval."""
  il_training._NOISE.clear(); il_training._WS.clear()
  plan, nets, _ = bench.build(torch.device(DEV), 0, seed=13)
  for _ in range(3): plan.run()
  torch.cuda.synchronize()
  plan.watch_timeouts()
  plan.record_direct()
  for _ in range(3): plan.launch_direct()
  plan.join(); torch.cuda.synchronize()
  assert plan.sync_timeouts() == 0 and not plan.poisoned() and plan.timeouts_seen() == (0, 0)
  actor, critic, target, log_alpha, disc = nets
  ao, co, to, do = plan._keep[4], plan._keep[5], plan._keep[6], plan._keep[8]
  state = lambda: [N(t).copy() for t in (actor.flat, critic.flat, target.flat, log_alpha, disc.flat, disc.sn, ao.exp_avg, ao.exp_avg_sq, co.exp_avg, co.exp_avg_sq, to.exp_avg, to.exp_avg_sq,
                                         do.exp_avg, do.exp_avg_sq)]
  before = state()
  plan.sync[plan._sync_spin] = 4   # every [IL_SYNC_SPIN]-bounded wait gives up after four polls
  for fn, args in plan._direct_main:   # the SAC branch without its discriminator branch: no index draw, no discriminator step to wait for
    assert fn(*args) == 0
  torch.cuda.synchronize()
  assert plan.sync_timeouts() > 0 and plan.poisoned() and plan.timeouts_seen()[0] > 0
  for a, b in zip(before, state()): np.testing.assert_array_equal(a, b)
  with pytest.raises(RuntimeError, match='hand-off wait'): plan.launch_direct()
  for fn, args in plan._direct_side:   # ... and a discriminator step of the poisoned learner stores nothing either
    assert fn(*args) == 0
  torch.cuda.synchronize()
  for a, b in zip(before, state()): np.testing.assert_array_equal(a, b)
  plan.sync[plan._sync_spin] = 0
  plan.clear_poison(); torch.cuda.synchronize()
  assert not plan.poisoned() and plan.sync_timeouts() == 0 and plan.timeouts_seen()[0] == 0


@pytest.mark.gpu
def test_overlapped_launches_equal_the_graph_replays(monkeypatch):
  """IL_MAIN_OVERLAP=1 (round 6, off by default: measured slower): the SAC branch's four launches alternating over two streams with stage hand-offs on the device
  (il_sac_update_gather_overlap), issued back to back without a join, leave the bits of the in-order graph replays; no wait expired, the learner is not poisoned."""
  monkeypatch.setenv('IL_MAIN_OVERLAP', '1')
  test_direct_launches_equal_the_graph_replays(K=40, seed=11, expect_overlap=True)


def test_direct_launches_equal_the_graph_replays(K=6, seed=9, expect_overlap=False):
  """UpdatePlan.record_direct / launch_direct: the same two branches as direct launches (two library calls per update, no hipGraph) leave every persistent tensor and every
  per-update output with the bits of the graph replays."""
  finals = []
  for mode in ('graph', 'direct'):
    il_training._NOISE.clear(); il_training._WS.clear()
    plan, nets, _ = bench.build(torch.device(DEV), 0, seed=seed)
    for _ in range(2): plan.run()
    torch.cuda.synchronize()
    if mode == 'graph':
      plan.capture(warmup=0); step = plan.replay
    else:
      plan.record_direct(); step = plan.launch_direct
      assert len(plan._direct_side) == 1 and len(plan._direct_main) == 1, 'one library call per branch'
      assert bool(plan._direct_overlap) == bool(expect_overlap), 'the overlapped launches were expected and did not run (no third hardware queue?)' if expect_overlap else 'overlapped launches are opt-in'
      if expect_overlap:
        import functools
        step = functools.partial(plan.launch_direct, join=False)   # back to back: the next update's first launch is dispatched while this one's last still runs
    for _ in range(K): step()
    plan.join()
    torch.cuda.synchronize()
    assert plan.sync_timeouts() == 0 and not plan.poisoned()
    finals.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.logp), N(plan.q), N(plan.rewards), N(plan.idx), N(plan.eidx)])
  for a, b in zip(*finals):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_launcher_thread_issues_the_recorded_update(K=40, seed=13):
  """UpdatePlan.launch_async (round 6, csrc/launcher.hip): the recorded branches re-issued by a thread of the library - the caller returns at once - are the launches of
  launch_direct in the same order: every persistent tensor and per-update output keeps its bits; a host synchronisation needs launcher_wait() / join() first; run(),
  replay() and launch_direct() drain the launcher before they issue anything themselves."""
  finals = []
  for mode in ('direct', 'async', 'mixed'):
    il_training._NOISE.clear(); il_training._WS.clear()
    plan, nets, _ = bench.build(torch.device(DEV), 0, seed=seed)
    for _ in range(2): plan.run()
    torch.cuda.synchronize()
    plan.record_direct()
    if mode != 'direct':   # a second recording while a launcher exists: nothing of the launcher's own may end up in the recorded pass
      plan.launch_async(); plan.record_direct(); plan.launch_direct()
    else:
      plan.launch_direct(); plan.launch_direct()
    for k in range(K):
      if mode == 'direct' or (mode == 'mixed' and k % 7 == 3): plan.launch_direct()
      elif mode == 'mixed' and k % 7 == 5: plan.run()
      else: plan.launch_async()
    plan.join()
    assert int(_lib.lib().il_launcher_pending(plan._launcher)) == 0 if mode != 'direct' else True
    torch.cuda.synchronize()
    assert plan.sync_timeouts() == 0 and not plan.poisoned()
    finals.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.logp), N(plan.q), N(plan.rewards), N(plan.idx), N(plan.eidx)])
  for other in finals[1:]:
    for a, b in zip(finals[0], other):
      assert np.isfinite(a).all()
      np.testing.assert_array_equal(a, b)


def test_side_stream_workgroups_never_share_a_cu_with_a_pair_workgroup():
  """DESIGN.md 3.2 "whole-CU LDS": the pair-mode kernels ask for 160 KB of LDS and k_gail_reduce for 1 KB it never touches so that the dispatcher cannot co-locate a
  workgroup of the discriminator branch with a pair workgroup (whose weight stream its loads would queue behind: the first pair build LOST 4 % to exactly that). The
  launch stamps record where every workgroup ran (XCD, shader engine / array, CU) and when: in the timed schedule no k_gail_grad / k_gail_reduce workgroup overlaps in
  time with a k_sac_chain_pair workgroup on the same CU, and the pair kernels' workgroups have a CU each."""
  il_training._NOISE.clear(); il_training._WS.clear()
  plan, nets, _ = bench.build(torch.device(DEV), 0, seed=21)
  for _ in range(5): plan.run()
  torch.cuda.synchronize()
  plan.record_direct()
  _lib.check(_lib.lib().il_kernel_stamps_clear())
  for burst in range(5):   # five readings of the last update of a burst of back-to-back updates
    for _ in range(100): plan.launch_direct()
    torch.cuda.synchronize()
    rows = {k: _lib.kernel_stamp_rows(k) for k in ('k_sac_chain_pair', 'k_policy_critic_pair', 'k_gail_grad', 'k_gail_reduce')}
    chain, pc = rows['k_sac_chain_pair'], rows['k_policy_critic_pair']
    assert len(chain) >= 10 * (256 // 16) and len(pc) >= 4 * (256 // 16)
    assert len({p for _, _, p in chain}) == len(chain), 'two workgroups of k_sac_chain_pair on one CU: the 160 KB LDS request no longer reserves whole CUs'
    assert len({p for _, _, p in pc}) == len(pc)
    overlapped = 0
    for side in ('k_gail_grad', 'k_gail_reduce'):
      for b2, e2, p2 in rows[side]:
        for b, e, p in chain:
          if p == p2 and b < e2 and b2 < e:
            raise AssertionError(f'{side} workgroup on CU {p:#x} during [{b2:.2f}, {e2:.2f}] us overlaps a k_sac_chain_pair workgroup there during [{b:.2f}, {e:.2f}] us')
          overlapped += int(b < e2 and b2 < e)
    assert overlapped > 0, 'the discriminator branch must overlap the forward / critic-loss launch in time (otherwise this test asserts nothing)'
  assert plan.sync_timeouts() == 0


# ------------------------------------------------------------------------------------------------ f1: the population launches
def test_batched_population_replays_through_the_oracle():
  Lp, K = 3, 6
  il_training._NOISE.clear(); il_training._WS.clear()
  built = [bench.build(torch.device(DEV), 0, seed=100 + l, learner_id=100 + l) for l in range(Lp)]
  oracles = [OracleLearner(nets, plan, tr, et, index_seed=100 + l) for l, (plan, nets, (tr, et)) in enumerate(built)]
  masked = [OracleLearner(nets, plan, tr, et, index_seed=100 + l) for l, (plan, nets, (tr, et)) in enumerate(built)]   # twins replaying with the launches' ReLU decisions
  assert len({o.key for o in oracles}) == Lp
  for plan, _, _ in built: plan.record_relu_masks()   # (before the population copies the descriptors to the device)
  pop = il.BatchedPopulationPlan([b[0] for b in built])
  pop.run()                      # one eager update (builds the lane-ordered weight copies), then the captured launches
  torch.cuda.synchronize()
  for o, om, (plan, _, _) in zip(oracles, masked, built):
    o.update(0); om.update(0, masks=relu_masks(plan))
  pop.capture()
  for k in range(1, K):
    pop.replay()
    torch.cuda.synchronize()
    for l, (o, om, (plan, nets, _)) in enumerate(zip(oracles, masked, built)):
      got = per_update_outputs(plan)
      compare_outputs(got, o.update(k), k, f'learner {l}: ')
      om.update(k, masks=relu_masks(plan))
      reward_bracket(o, nets[4], got[2], k, f'learner {l}: ')
  for l, (o, om, (plan, nets, _)) in enumerate(zip(oracles, masked, built)):
    compare_learner_masked(om, nets, plan, K, f'learner {l}, masked oracle: ')
    compare_learner(o, nets, plan, K, f'learner {l}: ')


# ------------------------------------------------------------------------------------------------ f2: the acting launch
@pytest.mark.parametrize('schedule', ['exact', 'fused'])
def test_acting_launch_replays_through_the_oracle(schedule):
  """il_act_step (append + absorbing wrap + actor(state).sample() in one launch, train.py:151-168) against ReplayOracle + oracle.nets, with the Philox
  draws of every act recorded: ring contents bit-exact (the stored action = the returned action), actions at rtol 1e-5 of the oracle's."""
  cap, steps = 41, 70
  cfg = Cfg(hidden_size=H, depth=2, activation='relu')
  torch.manual_seed(17)
  actor = il.SoftActor(S, A, cfg, device=DEV)
  rs = np.random.RandomState(8)
  actor.flat.copy_(torch.from_numpy(gi.mlp_params(rs, S, H, 2, 2 * A, out_scale=0.3)).to(DEV))
  mem = il.ReplayMemory(cap, S, A, True, device=DEV)
  omem = oreplay.ReplayOracle(cap, S, A, True)
  layers = onets.unpack(N(actor.flat), onets.mlp_shapes(S, H, 2, 2 * A))
  w = il.ActingWorker(actor, mem)
  key = int(w._seed.value)

  def oracle_action(obs):
    """models.py:90-94 with the draw this act consumed (counter = the actor's act-call count after the launch)."""
    eps = record_noise(key, actor._act_calls & 0xFFFFFFFF, philox.STREAM_ACT, A)
    out, _ = onets.mlp_forward(layers, obs[None, :])
    mean, _, _, std = onets.actor_head(out, A)
    return np.tanh(mean + std * eps[None, :])[0]

  def obs_row():
    o = rs.standard_normal(S).astype(np.float32); o[-1] = 0
    return o
  obs = obs_row()
  act = N(w.act(obs))[0]
  close(act, oracle_action(obs), 'first action')
  for t in range(1, steps + 1):
    nxt, rew, term, tout = obs_row(), float(rs.standard_normal()), t in (9, 33, 58), t in (21, 47)
    omem.append(t, obs, act, rew, nxt, term, tout)
    if term and not tout:
      omem.wrap_for_absorbing_states()                       # train.py:161
    nobs = obs_row() if (term or tout) else nxt              # env.reset()
    if schedule == 'exact':
      w.append(t, nxt, rew, term, tout)
      a2 = N(w.act(nobs))[0]
    else:
      a2 = N(w.step(t, nxt, rew, term, tout, obs=nobs))[0]
    close(a2, oracle_action(nobs), f'action at step {t}')
    obs, act = nobs, a2
  torch.cuda.synchronize()
  assert (mem.idx, mem.full, mem.num_trajectories) == (omem.idx, omem.full, omem.num_trajectories) and omem.full, 'the script wraps the ring'
  assert N(mem._ring_state).tolist() == [omem.idx, int(omem.full), cap]
  for f in oreplay.FIELDS:
    np.testing.assert_array_equal(N(getattr(mem, f)), getattr(omem, f), err_msg=f)
