"""`-m gpu`: the captured update plan of EVERY algorithm (UpdatePlan: GMMIL, RED, DRIL, AdRIL / SQIL, PWIL, SAC with an expert memory, mixed batches, BC auxiliary
loss) against the per-function entry points called in the order of the reference loop (train.py:173-203) - the sequence the parity tests of the individual
functions pin to the reference.  Same seeds, same index stream, same Philox counters: bit-identical learners.  DRIL draws its dropout masks on chip; the plan's
masks are recorded with il_noise_fill and fed to the per-function call."""
import ctypes as C

import numpy as np
import pytest
import torch

import inputs as gi

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
  import imitation_learning_amd as il
  from imitation_learning_amd import _lib
  from imitation_learning_amd import training as il_training
  from gpu_util import DEV, N, Cfg, fill_memory

S, A, B = gi.DIMS['halfcheetah'][0], gi.DIMS['halfcheetah'][1], 64


GAIL_VARIANTS = {   # the discriminator variants whose per-function entry points run INSIDE the plan (UpdatePlan._variant): name -> (imitation overrides, discriminator overrides)
    'pu_margin': (dict(loss_function='PUGAIL', nonnegative_margin=0.05), {}),
    'sublogp': (dict(loss_function='BCE'), dict(subtract_log_policy=True)),
    'shaping': (dict(loss_function='PUGAIL'), dict(reward_shaping=True)),
    'deep': (dict(loss_function='BCE', entropy_bonus=0.02), dict(depth=2, activation='tanh')),
    'shaping_deep_sublogp': (dict(loss_function='BCE'), dict(reward_shaping=True, subtract_log_policy=True, depth=2, activation='tanh', hidden_size=64)),
}


def build(algorithm, seed, mixed=False, bc_aux=False, balanced=True, update_freq=1250, variant=None, net=None, critic_net=None):
  torch.manual_seed(seed); il.seed(seed)
  il_training._NOISE.clear(); il_training._WS.clear()
  cfg = Cfg(net or dict(hidden_size=128, depth=2, activation='relu'))
  actor, critic = il.SoftActor(S, A, cfg, device=DEV), il.TwinCritic(S, A, Cfg(critic_net) if critic_net else cfg, device=DEV)
  target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=DEV)
  ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
  rs = np.random.RandomState(seed)
  mem = il.ReplayMemory(4000, S, A, True, device=DEV); fill_memory(mem, gi.transitions(rs, 3000, S, A), 3000)
  mem.num_trajectories = 7
  emem = il.ReplayMemory(1000, S, A, True, device=DEV); fill_memory(emem, gi.transitions(rs, 1000, S, A, state_shift=0.5, weighted=True), 1000)
  emem.num_trajectories = 4
  disc, dopt = None, None
  dcfg = Cfg(hidden_size=32, depth=1, activation='tanh', input_dropout=0.1, dropout=0.2)
  if algorithm == 'GMMIL':
    disc = il.GMMILDiscriminator(S, A, Cfg(state_only=False))
  elif algorithm == 'RED':
    disc = il.REDDiscriminator(S, A, Cfg(state_only=False, reward_bandwidth_scale=None, discriminator=dcfg), device=DEV)
    e = emem.sample(B); disc.set_sigma(e['states'], e['actions']); disc.eval()
  elif algorithm == 'DRIL':
    disc = il.SoftActor(S, A, dcfg, device=DEV)
    disc.set_uncertainty_threshold(emem['states'][:200], emem['actions'][:200], 0.6)
  elif algorithm == 'AdRIL':
    disc = il.RewardRelabeller(update_freq, balanced)
  elif algorithm == 'GAIL':
    io, do_ = GAIL_VARIANTS[variant] if variant else ({}, {})
    icfg = Cfg(dict(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=0.7, entropy_bonus=0.0, mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=float('inf')), **io)
    icfg['discriminator'] = Cfg(dict(hidden_size=32, depth=1, activation='relu', reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'), **do_)
    disc = il.GAILDiscriminator(S, A, icfg, 0.97, device=DEV)
    disc.test_extra = dict(discriminator_optimiser=il.AdamW(disc, lr=3e-5, weight_decay=10), imitation_cfg=icfg)
  il.seed(seed)   # the set-up above consumed index draws: restart the stream so that both runs see the same one
  nets = (actor, critic, log_alpha, target)   # the argument order of sac_update / UpdatePlan
  return nets, (ao, co, to), mem, emem, disc


def per_function_update(algorithm, nets, opts, mem, emem, disc, step, mixed, bc_aux):
  """train.py:173-203 through the per-function entry points (what train.py ran for these algorithms before the plans existed)."""
  actor, critic, log_alpha, target = nets
  t, e = mem.sample(B), emem.sample(B)
  if mixed and algorithm in ('DRIL', 'GMMIL', 'RED'): il.mix_expert_agent_transitions(t, e)
  masks = None
  if algorithm == 'AdRIL':
    disc.resample_and_relabel(t, e, step, mem.num_trajectories, emem.num_trajectories)
  elif algorithm == 'GAIL':   # train.py:178-194
    x = disc.test_extra
    il.adversarial_imitation_update(actor, disc, t, e, x['discriminator_optimiser'], x['imitation_cfg'])
    dc = x['imitation_cfg'].discriminator
    t['rewards'] = disc.predict_reward(**il.make_gail_input(t['states'], t['actions'], t['next_states'], t['terminals'], actor, dc.reward_shaping, dc.subtract_log_policy))
  elif algorithm == 'GMMIL':
    t['rewards'] = disc.predict_reward(t['states'], t['actions'], e['states'], e['actions'], t['weights'].contiguous(), e['weights'].contiguous())
  elif algorithm == 'RED':
    t['rewards'] = disc.predict_reward(t['states'], t['actions'])
  elif algorithm == 'DRIL':
    # the masks the plan's k_dril_unc draws on chip for this update: noise stream (key, 0x40000000 + update counter), keep = u >= p
    key, ctr = torch.initial_seed() & (2**64 - 1), 0x40000000 + int(N(il_training._noise_counter(actor.flat.device))[0])
    def keep(stream, n, p):
      out = torch.empty(n, device=DEV)
      _lib.check(_lib.lib().il_noise_fill(C.c_uint64(key), ctr, stream, n, _lib.ptr(out), _lib.stream_ptr()))
      return (out >= p).float()
    masks = (keep(5, 5 * B * S, disc.p_in).view(5 * B, S), keep(6, 5 * B * disc.hidden, disc.p).view(5 * B, disc.hidden))
    t['rewards'] = disc.predict_reward(t['states'], t['actions'], masks=masks)
  if bc_aux: il.behavioural_cloning_update(actor, e, opts[0])
  logp, q = il.sac_update(actor, critic, log_alpha, target, t, *opts, 0.97, -0.5 * A, 0.99)
  return t['rewards'].clone(), logp, q


CASES = [('GMMIL', False, False, {}), ('GMMIL', True, False, {}), ('RED', False, False, {}), ('RED', True, True, {}), ('DRIL', False, False, {}), ('DRIL', True, False, {}),
         ('AdRIL', False, False, dict(balanced=True)), ('AdRIL', False, False, dict(balanced=False)), ('AdRIL', False, True, dict(balanced=True, update_freq=0)),
         ('PWIL', False, False, {}), ('SAC', False, True, {})] + [('GAIL', False, False, dict(variant=v)) for v in GAIL_VARIANTS]


@pytest.mark.parametrize('algorithm,mixed,bc_aux,kw', CASES)
def test_plan_of_every_algorithm_equals_the_per_function_sequence(algorithm, mixed, bc_aux, kw, launch='graph'):
  K, step0 = 5, 2400   # AdRIL: steps 2400.. straddle a round boundary of update_freq = 1250 (rows are stamped 1..3000)
  nets, opts, mem, emem, disc = build(algorithm, 11, **kw)
  want = [per_function_update(algorithm, nets, opts, mem, emem, disc, step0 + 37 * k, mixed, bc_aux) for k in range(K)]
  ref_state = [N(n.flat if hasattr(n, 'flat') else n) for n in nets]
  ref_disc = N(disc.flat) if algorithm == 'GAIL' else None

  nets, opts, mem, emem, disc = build(algorithm, 11, **kw)
  plan = il.UpdatePlan(algorithm, *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc, mix_expert=mixed, bc_aux=bc_aux, **getattr(disc, 'test_extra', {}))
  if algorithm == 'GAIL': assert plan._variant and not plan.device_sync, 'these discriminator variants run their per-function entry points inside the plan, on stream dependencies'
  got = []
  for k in range(K):
    if algorithm == 'AdRIL': plan.relabel_args(step0 + 37 * k, mem.num_trajectories)
    if k == 0:
      plan.run()    # first update eagerly (GMMIL: fixes the bandwidths), the rest as graph replays - or, launch='direct' (round 6), as recorded library calls re-issued on this stream
      if launch == 'direct':
        assert plan.direct_launch_ok()
        plan.record_direct()
        assert (plan._direct_side == [] and len(plan._direct_main) >= 2) if not plan.device_sync else len(plan._direct_side) == 1   # one stream: one list; SAC / PWIL: the resident draw on the second stream
      else:
        plan.capture(warmup=0)
    else:
      (plan.launch_direct if launch == 'direct' else plan.replay)()
    torch.cuda.synchronize()
    got.append((plan.transitions['rewards'].clone(), plan.logp.clone(), plan.q.clone()))
  for k, (w, g) in enumerate(zip(want, got)):
    for name, a, b in zip(('rewards', 'log pi', 'min Q'), w, g):
      np.testing.assert_array_equal(N(a), N(b), err_msg=f'{algorithm} update {k}: {name}')
  if algorithm == 'GAIL': ref_state.append(ref_disc); nets = tuple(nets) + (disc,)
  for i, (a, n) in enumerate(zip(ref_state, nets)):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, N(n.flat if hasattr(n, 'flat') else n), err_msg=f'{algorithm}: tensor {i} after {K} updates')
  if algorithm == 'GMMIL':
    assert disc.gamma_1 is not None and disc.gamma_2 is not None
  if algorithm == 'DRIL':
    r = N(got[-1][0]); assert set(np.unique(r)) <= {-1.0, 1.0} and 0 < (r > 0).mean() < 1, 'the threshold should split the batch'


@pytest.mark.parametrize('algorithm,mixed,bc_aux,kw', [c for c in CASES if c[0] != 'GAIL'])
def test_one_stream_plans_as_direct_launches_equal_the_per_function_sequence(algorithm, mixed, bc_aux, kw):
  """UpdatePlan.record_direct / launch_direct for the one-stream plans (round 6): GMMIL, RED, DRIL, AdRIL / SQIL, PWIL, SAC with the BC auxiliary step - library calls on the
  caller's stream and nothing else - re-issued without a hipGraph: bit-identical to the per-function sequence, like the graph replays."""
  test_plan_of_every_algorithm_equals_the_per_function_sequence(algorithm, mixed, bc_aux, kw, launch='direct')


@pytest.mark.parametrize('algorithm,mixed', [('SAC', False), ('RED', True), ('DRIL', False)])
def test_data_parallel_bc_aux_equals_the_plain_plan_on_one_rank(algorithm, mixed):
  """imitation.bc_aux_loss under DataParallelUpdate (round 6; train.py:201 between the reward step and sac_update): il_bc_step(IL_FLAG_GRADS_ONLY) -> [mean over ranks of the
  actor bucket] -> il_adam_step. With one rank the exchange is the identity, so the learner must evolve bit for bit like the plain plan (whose BC step applies AdamW in the
  gradient launch's epilogue) - eagerly and as a captured graph. DRIL ships with bc_aux_loss=true: before this it had no data-parallel form."""
  from imitation_learning_amd.parallel import DataParallelUpdate
  outs = []
  for mode in ('plan', 'dp', 'dp_graph'):
    nets, opts, mem, emem, disc = build(algorithm, 17)
    plan = il.UpdatePlan(algorithm, *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc, mix_expert=mixed, bc_aux=True)
    runner = plan if mode == 'plan' else DataParallelUpdate(plan)
    for k in range(4):
      if mode == 'dp_graph' and k == 1: runner.capture(warmup=0)
      (runner.replay if mode == 'dp_graph' and k >= 1 else runner.run)()
    torch.cuda.synchronize()
    outs.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(opts[0].exp_avg), N(opts[0].exp_avg_sq), N(opts[0].step_count[:1]), N(plan.logp), N(plan.q), N(plan.transitions['rewards'])])
  for mode, got in zip(('dp', 'dp_graph'), outs[1:]):
    for i, (a, b) in enumerate(zip(outs[0], got)):
      assert np.isfinite(a).all()
      np.testing.assert_array_equal(a, b, err_msg=f'{algorithm} {mode}: tensor {i}')
  assert int(outs[0][6][0]) == 8   # the actor's optimiser stepped twice per update: the BC auxiliary step and the policy step


@pytest.mark.parametrize('variant', list(GAIL_VARIANTS) + ['mixup_alpha'])
def test_data_parallel_gail_variants_equal_the_plain_plan_on_one_rank(variant):
  """The discriminator variants under DataParallelUpdate (round 6; training.py:100-114,130-132): a finite PUGAIL margin, subtract_log_policy, reward shaping, depth-2 / tanh
  discriminators and Mixup with mixup_alpha != 1 leave their gradient in the optimiser's arena (IL_FLAG_GRADS_ONLY), the arena is averaged over ranks, the AdamW step is applied
  from it. One rank: the exchange is the identity, the learner and its discriminator must evolve bit for bit like the plain plan - eagerly and as a captured graph."""
  from imitation_learning_amd.parallel import DataParallelUpdate
  kw = dict(variant=variant) if variant != 'mixup_alpha' else {}
  outs = []
  for mode in ('plan', 'dp', 'dp_graph'):
    nets, opts, mem, emem, disc = build('GAIL', 23, **kw)
    if variant == 'mixup_alpha':
      disc.test_extra['imitation_cfg']['loss_function'], disc.test_extra['imitation_cfg']['mixup_alpha'] = 'Mixup', 0.4
    plan = il.UpdatePlan('GAIL', *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc, **disc.test_extra)
    assert (plan._variant or plan._beta_alpha is not None) and not plan.device_sync
    runner = plan if mode == 'plan' else DataParallelUpdate(plan)
    if mode != 'plan': assert runner.variant and not runner.handoff
    for k in range(4):
      if mode == 'dp_graph' and k == 1: runner.capture(warmup=0)
      (runner.replay if mode == 'dp_graph' and k >= 1 else runner.run)()
    torch.cuda.synchronize()
    do = disc.test_extra['discriminator_optimiser']
    outs.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(disc.flat), N(disc.sn) if getattr(disc, 'sn', None) is not None else np.zeros(1), N(do.exp_avg), N(do.exp_avg_sq), N(do.step_count[:1]),
                                                                            N(plan.logp), N(plan.q), N(plan.transitions['rewards'])])
  for mode, got in zip(('dp', 'dp_graph'), outs[1:]):
    for i, (a, b) in enumerate(zip(outs[0], got)):
      assert np.isfinite(a).all()
      np.testing.assert_array_equal(a, b, err_msg=f'GAIL {variant} {mode}: tensor {i}')
  assert int(outs[0][8][0]) == 4


def test_plan_rejects_what_it_cannot_capture():
  nets, opts, mem, emem, disc = build('GMMIL', 3)
  with pytest.raises(AssertionError):
    il.UpdatePlan('GMMIL', *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, discriminator=disc)                      # no expert memory
  plan = il.UpdatePlan('GMMIL', *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc)
  with pytest.raises(AssertionError, match='run\\(\\) once before capture'):
    plan.capture(warmup=0)                                                                                       # the median heuristic needs an eager first update
  torch.cuda.synchronize()
  nets, opts, mem, emem, disc = build('AdRIL', 3)
  plan = il.UpdatePlan('AdRIL', *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc)
  with pytest.raises(AssertionError, match='relabel_args'):
    plan.run()


@pytest.mark.parametrize('variant', ['deep', 'shaping', 'shaping_deep_sublogp'])
def test_discriminator_variants_draw_fresh_noise_every_update(variant):
  """training.py:117-119: the gradient-penalty epsilon is a fresh torch.rand per update. The deep / shaped kernels key their on-chip draw by the learner's update counter (the
  one sac_update's actor step advances): the same discriminator step from the same parameters differs between update counters and repeats for equal counters."""
  outs = []
  for bump in (0, 0, 1):
    nets, opts, mem, emem, disc = build('GAIL', 5, variant=variant)
    actor = nets[0]
    ctr = il_training._noise_counter(actor.flat.device)
    ctr.fill_(bump)
    t, e = mem.sample(B), emem.sample(B)
    x = disc.test_extra
    il.adversarial_imitation_update(actor, disc, t, e, x['discriminator_optimiser'], x['imitation_cfg'])
    outs.append(N(disc.flat).copy())
  assert np.isfinite(outs[0]).all()
  np.testing.assert_array_equal(outs[0], outs[1])
  assert not np.array_equal(outs[0], outs[2]), 'the gradient-penalty draw must follow the update counter'
  # and the counter it reads is the one a SAC update advances
  nets, opts, mem, emem, disc = build('GAIL', 5, variant=variant)
  ctr = il_training._noise_counter(nets[0].flat.device)
  before = int(N(ctr)[0])
  il.sac_update(*nets, mem.sample(B), *opts, 0.97, -0.5 * A, 0.99)
  assert int(N(ctr)[0]) == before + 1


GENERAL_NETS = dict(d3_tanh=(dict(hidden_size=96, depth=3, activation='tanh'), None), mixed=(dict(hidden_size=32, depth=1, activation='relu'), dict(hidden_size=72, depth=3, activation='sigmoid')))


@pytest.mark.parametrize('algorithm,mixed,bc_aux,nets_name', [('SAC', False, True, 'd3_tanh'), ('GAIL', False, False, 'd3_tanh'), ('GMMIL', True, False, 'mixed'), ('AdRIL', False, False, 'mixed')])
def test_general_shape_plan_equals_the_per_function_sequence(algorithm, mixed, bc_aux, nets_name):
  """models.py:48-69 builds actor / critic networks of any depth and activation: UpdatePlan runs such shapes on one stream (device draws + gather, the reward step,
  il_sac_update_general) and captures them as ONE hipGraph - bit-identical to the per-function entry points called in the order of train.py:173-203."""
  K, step0 = 4, 2400
  net, cnet = GENERAL_NETS[nets_name]
  kw = dict(net=net, critic_net=cnet)
  nets, opts, mem, emem, disc = build(algorithm, 13, **kw)
  assert nets[0].general or nets[1].general
  want = [per_function_update(algorithm, nets, opts, mem, emem, disc, step0 + 37 * k, mixed, bc_aux) for k in range(K)]
  ref_state = [N(n.flat if hasattr(n, 'flat') else n) for n in nets] + ([N(disc.flat)] if algorithm == 'GAIL' else [])
  nets, opts, mem, emem, disc = build(algorithm, 13, **kw)
  plan = il.UpdatePlan(algorithm, *nets, mem, *opts, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc, mix_expert=mixed, bc_aux=bc_aux, **getattr(disc, 'test_extra', {}))
  assert plan.general and not plan.device_sync and plan.side is None
  got = []
  for k in range(K):
    if algorithm == 'AdRIL': plan.relabel_args(step0 + 37 * k, mem.num_trajectories)
    if k == 0:
      plan.run(); plan.capture(warmup=0)
    else:
      plan.replay()
    torch.cuda.synchronize()
    got.append((plan.transitions['rewards'].clone(), plan.logp.clone(), plan.q.clone()))
  for k, (w, g) in enumerate(zip(want, got)):
    for name, a, b in zip(('rewards', 'log pi', 'min Q'), w, g):
      np.testing.assert_array_equal(N(a), N(b), err_msg=f'{algorithm} update {k}: {name}')
  for i, (a, n) in enumerate(zip(ref_state, tuple(nets) + ((disc,) if algorithm == 'GAIL' else ()))):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, N(n.flat if hasattr(n, 'flat') else n), err_msg=f'{algorithm}: tensor {i} after {K} updates')
