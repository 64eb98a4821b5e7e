"""`-m gpu` parity tests: the HIP path (through the C ABI, via the Python host layer) against the numpy oracle on the same
seeded inputs, and against the committed reference-generated golden vectors.  Tolerances: rtol 1e-5 (north_star) plus an absolute
floor relative to each tensor's scale; bit-exact for index draws and replay rows.  /root/reference is never touched here."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import inputs as gi
from oracle import gail as ogail
from oracle import gmmil as ogmmil
from oracle import nets as onets
from oracle import pwil as opwil
from oracle import replay as oreplay
from oracle import sac as osac
from oracle.mt19937 import MT19937, sample_indices

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
  import imitation_learning_amd as il
  from imitation_learning_amd import _lib
  from imitation_learning_amd import memory as il_memory
  from imitation_learning_amd import training as il_training
  from gpu_util import DEV, N, T, Cfg, bracket, close, close_params, crit_from_flat, fill_memory, make_disc, make_sac, make_sac_oracle, tbatch


def load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


def test_native_library_loaded():
  name = C.create_string_buffer(64); cu = C.c_int(0)
  _lib.check(_lib.lib().il_device_info(name, 64, C.byref(cu)))
  assert name.value.decode().startswith('gfx950'), name.value
  assert cu.value == 256


# ------------------------------------------------------------------------------------------------ replay
@pytest.mark.parametrize('name,seed,size,fill', [('partial', 0, 1000, 300), ('wrapped', 1, 64, 150), ('expert', 2, 500, None)])
def test_replay_matches_reference_bit_exact(golden_dir, name, seed, size, fill):
  g = load(golden_dir, 'replay')
  S, A = 5, 2
  rs = np.random.RandomState(100 + seed)
  if fill is None:
    tr = gi.transitions(rs, size, S, A)
    mem = il.ReplayMemory(size, S, A, True, transitions={**{k: torch.from_numpy(v) for k, v in tr.items() if k != 'absorbing'}, 'num_trajectories': 3}, device=DEV)
  else:
    mem = il.ReplayMemory(size, S, A, True, device=DEV)
    tr = gi.transitions(rs, fill, S, A)
    for i in range(fill):
      if i % 2:  # exercise both the host-staged and the device-resident append paths
        mem.append(i + 1, torch.from_numpy(tr['states'][i:i + 1]), torch.from_numpy(tr['actions'][i:i + 1]), float(tr['rewards'][i]), torch.from_numpy(tr['next_states'][i:i + 1]), bool(tr['terminals'][i]), False)
      else:
        mem.append(i + 1, T(tr['states'][i:i + 1]), T(tr['actions'][i:i + 1]), float(tr['rewards'][i]), T(tr['next_states'][i:i + 1]), bool(tr['terminals'][i]), False)
      if i % 37 == 36:
        mem.wrap_for_absorbing_states()
  assert [mem.idx, int(mem.full), mem.num_trajectories, mem.size] == g[f'{name}_state'].tolist()
  for k in oreplay.FIELDS:
    np.testing.assert_array_equal(N(getattr(mem, k))[:min(size, 200)], g[f'{name}_mem_{k}'], err_msg=k)
  il.seed(seed)
  np.testing.assert_array_equal(mem._sample_idx_tensor(256).numpy(), g[f'{name}_idx'])
  il.seed(seed)
  batch = mem.sample(32)
  for k, v in batch.items():
    np.testing.assert_array_equal(N(v), g[f'{name}_batch_{k}'], err_msg=k)
  # device-side draw: same stream, same rejection rule, across several calls (twist boundaries included)
  il.seed(seed)
  gen = MT19937(seed)
  idx_out, rows_out = torch.empty(300, dtype=torch.int32, device=DEV), torch.empty(300, mem.row, device=DEV)
  for call in range(6):
    mem.sample_device(300, idx_out, rows_out)
    want = sample_indices(gen, 300, mem.size, mem.idx, mem.full)
    np.testing.assert_array_equal(N(idx_out), np.array(want), err_msg=f'device draw, call {call}')
    np.testing.assert_array_equal(N(rows_out), N(mem.ring)[np.array(want)])


@pytest.mark.parametrize('n_src,cap,pre', [(50, 64, 10), (50, 64, 40), (150, 64, 7), (128, 64, 0)])
def test_transfer_transitions_matches_sequential_appends(n_src, cap, pre):
  """memory.py:46-48 re-appends row by row; the bulk device copy must leave the ring, cursor, `full` and trajectory count exactly as that loop does - including a
  source LARGER than the destination (train.py:30 clamps memory.size to `steps`), where the last write to a slot wins."""
  S, A = 5, 2
  rs = np.random.RandomState(n_src + cap)
  tr = gi.transitions(rs, n_src, S, A, terminal_frac=0.1, weighted=True)
  src = il.ReplayMemory(n_src, S, A, True, transitions={**{k: torch.from_numpy(v) for k, v in tr.items() if k != 'absorbing'}, 'num_trajectories': 3}, device=DEV)
  osrc = oreplay.ReplayOracle(n_src, S, A, True, transitions={**tr, 'num_trajectories': 3})
  dst, odst = il.ReplayMemory(cap, S, A, True, device=DEV), oreplay.ReplayOracle(cap, S, A, True)
  filler = gi.transitions(rs, pre, S, A) if pre else None
  for i in range(pre):
    args = (i + 1, filler['states'][i:i + 1], filler['actions'][i:i + 1], float(filler['rewards'][i]), filler['next_states'][i:i + 1], False, False)
    dst.append(args[0], *(torch.from_numpy(a) if isinstance(a, np.ndarray) else a for a in args[1:])); odst.append(*args)
  dst.transfer_transitions(src); odst.transfer_transitions(osrc)
  assert (dst.idx, dst.full, dst.num_trajectories) == (odst.idx, odst.full, odst.num_trajectories)
  written = cap if odst.full else odst.idx
  for k in oreplay.FIELDS:
    np.testing.assert_array_equal(N(getattr(dst, k))[:written], getattr(odst, k)[:written], err_msg=k)
  assert N(dst._ring_state).tolist() == [odst.idx, int(odst.full), cap]


def test_replay_full_size_gather_property():
  """BASELINE size (capacity 1e6, HalfCheetah rows): gather == fancy indexing, checked through a checksum of every field."""
  S, A = gi.DIMS['halfcheetah']
  mem = il.ReplayMemory(1_000_000, S, A, True, device=DEV)
  mem.ring.copy_(torch.randn(mem.ring.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(0)))
  mem.idx, mem.full = 123_457, True
  mem._sync_ring_state()
  il.seed(7)
  idx = mem._sample_idx_tensor(4096)
  assert int(idx.max()) < 1_000_000 and int(idx.min()) >= 0 and 123_456 not in idx.tolist()
  rows = mem.gather(idx)
  assert torch.equal(rows, mem.ring[idx.to(DEV).long()])


# ------------------------------------------------------------------------------------------------ SAC
SAC_CASES = [('sac_halfcheetah', (3, 'halfcheetah', 256, 256, 3)), ('sac_hopper_h64', (4, 'hopper', 64, 96, 3)), ('sac_ant_b64', (5, 'ant', 256, 64, 2))]


@pytest.mark.parametrize('name,args', SAC_CASES)
def test_sac_update_matches_oracle_and_reference(golden_dir, name, args):
  g, c = load(golden_dir, name), gi.sac_case(*args)
  actor, critic, target, log_alpha, ao, co, to = make_sac(c)
  st = make_sac_oracle(c)
  for k in range(1, args[-1] + 1):
    b = c['batches'][k - 1]
    logp, q = il.sac_update(actor, critic, log_alpha, target, tbatch(b), ao, co, to, c['discount'], c['entropy_target'], c['polyak'], eps_next=T(c['eps_next'][k - 1]),
                            eps_cur=T(c['eps_cur'][k - 1]))
    ologp, oq = osac.sac_update(st, b, c['eps_next'][k - 1], c['eps_cur'][k - 1], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'],
                                lr=c['lr'], weight_decay=c['weight_decay'])
    torch.cuda.synchronize()
    s = 1e-5 * k
    close(N(logp), ologp, f'logp step {k}', atol_scale=2e-6 * k); close(N(q), oq, f'q step {k}', atol_scale=2e-6 * k)
    close_params(N(actor.flat), st.actor, f'actor step {k}', c['lr'], k); close_params(crit_from_flat(critic, critic.flat), st.critic, f'critic step {k}', c['lr'], k)
    close_params(crit_from_flat(critic, target.flat), st.target, f'target step {k}', c['lr'], k); close(N(log_alpha), st.log_alpha, f'log_alpha step {k}')
    close(N(ao.exp_avg), st.actor_m, f'actor m step {k}', atol_scale=s); close(N(ao.exp_avg_sq), st.actor_v, f'actor v step {k}', atol_scale=s)
    close(crit_from_flat(critic, co.exp_avg), st.critic_m, f'critic m step {k}', atol_scale=s)
    # and against the reference-generated vectors directly
    close(N(logp), g[f'logp_{k}'], f'golden logp {k}', atol_scale=2e-6 * k); close(N(q), g[f'q_{k}'], f'golden q {k}', atol_scale=2e-6 * k)
    close_params(gi.strided(N(actor.flat)), g[f'actor_{k}'], f'golden actor {k}', c['lr'], k)
    close_params(gi.strided(crit_from_flat(critic, critic.flat)), g[f'critic_{k}'], f'golden critic {k}', c['lr'], k)
    close_params(gi.strided(crit_from_flat(critic, target.flat)), g[f'target_{k}'], f'golden target {k}', c['lr'], k)
    close(N(log_alpha), g[f'log_alpha_{k}'], f'golden log_alpha {k}')
  assert int(ao.step_count[0]) == args[-1] and int(co.step_count[0]) == args[-1] and int(to.step_count[0]) == args[-1]


@pytest.mark.parametrize('name,args', SAC_CASES)
def test_sac_gradients_match_oracle(golden_dir, name, args):
  """IL_FLAG_GRADS_ONLY path (what the data-parallel all-reduce sees): gradients as tensors at the strict bound (rtol 1e-5), then the split Adam tail == fused - for every
  case (HalfCheetah 256 / 256, Hopper, Ant) and at EVERY step of the case, not only the first: before steps 2, 3 the oracle takes over the kernels' parameters and moments
  (so that the gradients of that step are compared at equal parameters instead of through the Adam-step allowance of `close_params`)."""
  g, c = load(golden_dir, name), gi.sac_case(*args)
  actor, critic, target, log_alpha, ao, co, to = make_sac(c)
  st = make_sac_oracle(c)
  d = il_training.sac_descriptor(actor, critic, log_alpha, target, c['B'], ao, co, to, c['discount'], c['entropy_target'], c['polyak'])
  L, s = _lib.lib(), _lib.stream_ptr()
  logp, q = torch.empty(c['B'], device=DEV), torch.empty(c['B'], device=DEV)
  for k in range(len(c['batches'])):
    if k > 0:   # equal parameters and moments on both sides
      st.actor[...] = N(actor.flat); st.critic[...] = crit_from_flat(critic, critic.flat); st.target[...] = crit_from_flat(critic, target.flat); st.log_alpha[...] = N(log_alpha)
      st.actor_m[...] = N(ao.exp_avg); st.actor_v[...] = N(ao.exp_avg_sq); st.critic_m[...] = crit_from_flat(critic, co.exp_avg); st.critic_v[...] = crit_from_flat(critic, co.exp_avg_sq)
      st.alpha_m[...] = N(to.exp_avg); st.alpha_v[...] = N(to.exp_avg_sq)
    b = c['batches'][k]
    _, _, gr = osac.sac_update(st, b, c['eps_next'][k], c['eps_cur'][k], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'], lr=c['lr'],
                               weight_decay=c['weight_decay'], return_grads=True)
    tb = tbatch(b); bd = il_memory.batch_desc(tb)
    e1, e2 = T(c['eps_next'][k]), T(c['eps_cur'][k])
    _lib.check(L.il_sac_critic_step(C.byref(d), C.byref(bd), _lib.ptr(e1), _lib.IL_FLAG_GRADS_ONLY, s))
    close(crit_from_flat(critic, co.grad), gr['critic'], f'critic grad step {k + 1}')
    if k == 0: close(gi.strided(crit_from_flat(critic, co.grad)), g['g_critic_1'], 'golden critic grad')
    _lib.check(L.il_sac_apply_critic_grads(C.byref(d), s))
    _lib.check(L.il_sac_actor_step(C.byref(d), C.byref(bd), _lib.ptr(e2), _lib.ptr(logp), _lib.ptr(q), _lib.IL_FLAG_GRADS_ONLY, s))
    close(N(ao.grad), gr['actor'], f'actor grad step {k + 1}'); close(N(to.grad), gr['alpha'], f'alpha grad step {k + 1}')
    if k == 0: close(gi.strided(N(ao.grad)), g['g_actor_1'], 'golden actor grad'); close(N(to.grad), g['g_alpha_1'], 'golden alpha grad')
    _lib.check(L.il_sac_apply_actor_grads(C.byref(d), s))
    torch.cuda.synchronize()
    close_params(N(actor.flat), st.actor, 'actor after split step', c['lr']); close_params(crit_from_flat(critic, critic.flat), st.critic, 'critic after split step', c['lr'])
    close_params(crit_from_flat(critic, target.flat), st.target, 'target after split step', c['lr']); close(N(log_alpha), st.log_alpha, 'log_alpha after split step')


@pytest.mark.parametrize('name', sorted(gi.GENERAL_SAC_CASES))
def test_general_shape_sac_matches_oracle_and_reference(golden_dir, name):
  """Actor / critic shapes outside the fused kernels (models.py:48-69 `_create_fcnn`: any depth, relu / tanh / sigmoid) run layer by layer through csrc/general.hip:
  depth 3 / tanh, depth 1 / sigmoid and a 320-wide depth-2 ReLU network - sac_update for the case's steps against the oracle and against the vectors the reference
  produced, then acting (greedy, a sample with fed noise and its log-probability), log pi of given actions and two behavioural_cloning_update steps."""
  g, c = load(golden_dir, name), gi.sac_case(**gi.GENERAL_SAC_CASES[name])
  actor, critic, target, log_alpha, ao, co, to = make_sac(c)
  assert (actor.general or critic.general) and (actor.depth, actor.activation) == (c['depth'], c['activation']) and (critic.hidden, critic.depth, critic.activation) == (c['critic_hidden'], c['critic_depth'], c['critic_activation'])
  st = make_sac_oracle(c)
  b0 = tbatch(c['batches'][0])
  # acting and log-probabilities on the initial actor
  close(N(actor.get_greedy_action(b0['states'])), g['act_greedy'], 'golden greedy action', atol_scale=2e-6)
  a, lp = actor(b0['states']).sample_with_log_prob(eps=T(c['eps_cur'][0]))
  close(N(a), g['act_sample'], 'golden sample', atol_scale=2e-6); close(N(lp), g['act_sample_logp'], 'golden log-prob of the sample', rtol=1e-4, atol_scale=1e-5)
  close(N(actor.log_prob(b0['states'], b0['actions'])), g['act_logp_given'], 'golden log-prob of given actions', rtol=1e-4, atol_scale=1e-5)
  steps = len(c['batches'])
  for k in range(1, steps + 1):
    b = c['batches'][k - 1]
    logp, q = il.sac_update(actor, critic, log_alpha, target, tbatch(b), ao, co, to, c['discount'], c['entropy_target'], c['polyak'], eps_next=T(c['eps_next'][k - 1]),
                            eps_cur=T(c['eps_cur'][k - 1]))
    ologp, oq = osac.sac_update(st, b, c['eps_next'][k - 1], c['eps_cur'][k - 1], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'],
                                lr=c['lr'], weight_decay=c['weight_decay'])
    torch.cuda.synchronize()
    s = 1e-5 * k
    close(N(logp), ologp, f'logp step {k}', atol_scale=2e-6 * k); close(N(q), oq, f'q step {k}', atol_scale=2e-6 * k)
    close_params(N(actor.flat), st.actor, f'general actor step {k}', c['lr'], k); close_params(crit_from_flat(critic, critic.flat), st.critic, f'general critic step {k}', c['lr'], k)
    close_params(crit_from_flat(critic, target.flat), st.target, f'general target step {k}', c['lr'], k); close(N(log_alpha), st.log_alpha, f'log_alpha step {k}')
    close(N(ao.exp_avg), st.actor_m, f'actor m step {k}', atol_scale=s); close(crit_from_flat(critic, co.exp_avg), st.critic_m, f'critic m step {k}', atol_scale=s)
    close(N(logp), g[f'logp_{k}'], f'golden logp {k}', atol_scale=2e-6 * k); close(N(q), g[f'q_{k}'], f'golden q {k}', atol_scale=2e-6 * k)
    close_params(gi.strided(N(actor.flat)), g[f'actor_{k}'], f'golden general actor {k}', c['lr'], k)
    close_params(gi.strided(crit_from_flat(critic, critic.flat)), g[f'critic_{k}'], f'golden general critic {k}', c['lr'], k)
    close_params(gi.strided(crit_from_flat(critic, target.flat)), g[f'target_{k}'], f'golden general target {k}', c['lr'], k)
    close(N(log_alpha), g[f'log_alpha_{k}'], f'golden log_alpha {k}')
  assert int(ao.step_count[0]) == steps and int(co.step_count[0]) == steps and int(to.step_count[0]) == steps
  # behavioural cloning from the initial parameters (training.py:57-64)
  actor2 = make_sac(gi.sac_case(**gi.GENERAL_SAC_CASES[name]))[0]
  opt = il.AdamW(actor2, lr=2.5e-4, weight_decay=0.01)
  for k in (1, 2):
    il.behavioural_cloning_update(actor2, tbatch(c['batches'][k % steps]), opt)
    torch.cuda.synchronize()
    if k == 1: close(gi.strided(N(opt.grad)), g['bc_g_actor_1'], 'golden behavioural-cloning gradient')
    close_params(gi.strided(N(actor2.flat)), g[f'bc_actor_{k}'], f'golden general actor after BC step {k}', 2.5e-4, k)
  # round 5: UpdatePlan takes these shapes on one stream (tests/test_update_plans_gpu.py::test_general_shape_plan_equals_the_per_function_sequence); the population launches
  # and the data-parallel runner still refuse them loudly
  gplan = il.UpdatePlan('SAC', actor, critic, log_alpha, target, il.ReplayMemory(64, c['S'], c['A'], True, device=DEV), ao, co, to, 16, 0.97, -1.0, 0.99, learner_id=77)
  assert gplan.general and gplan.side is None and not gplan.device_sync
  with pytest.raises(NotImplementedError):
    il.BatchedPopulationPlan([gplan, gplan])
  from imitation_learning_amd import parallel
  with pytest.raises(NotImplementedError):
    parallel.DataParallelUpdate(gplan)


@pytest.mark.parametrize('hidden,depth,activation,batch,critic', [(128, 3, 'tanh', 128, None), (64, 2, 'sigmoid', 256, (128, 1, 'tanh')), (256, 3, 'tanh', 256, None)])
def test_general_shape_tile_engine_matches_oracle_at_block_batches(hidden, depth, activation, batch, critic):
  """(round 6) general.hip's tile engine at batches that are multiples of 128, where its optimiser launches run dw_block.hpp's 32 x 32 block jobs (the single learner's
  form) and - at hidden widths that are multiples of 64 - every H x H layer runs from its lane-ordered copies, two panels per wave: two sac_update steps against the oracle
  (models.py:48-69 shapes: depth 3 / tanh, a sigmoid actor beside a depth-1 tanh critic, and depth 3 / tanh / 256 - the shape bench.py's `secondary` line times)."""
  kw = dict(seed=51, env='halfcheetah', hidden=hidden, batch=batch, steps=2, depth=depth, activation=activation)
  if critic is not None: kw['critic'] = critic
  c = gi.sac_case(**kw)
  actor, critic_net, target, log_alpha, ao, co, to = make_sac(c)
  assert actor.general or critic_net.general
  st = make_sac_oracle(c)
  for k in (1, 2):
    b = c['batches'][k - 1]
    logp, q = il.sac_update(actor, critic_net, log_alpha, target, tbatch(b), ao, co, to, c['discount'], c['entropy_target'], c['polyak'], eps_next=T(c['eps_next'][k - 1]), eps_cur=T(c['eps_cur'][k - 1]))
    ologp, oq = osac.sac_update(st, b, c['eps_next'][k - 1], c['eps_cur'][k - 1], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'], lr=c['lr'], weight_decay=c['weight_decay'])
    torch.cuda.synchronize()
    close(N(logp), ologp, f'logp step {k}', atol_scale=2e-6 * k); close(N(q), oq, f'q step {k}', atol_scale=2e-6 * k)
    close_params(N(actor.flat), st.actor, f'block-batch actor step {k}', c['lr'], k); close_params(crit_from_flat(critic_net, critic_net.flat), st.critic, f'block-batch critic step {k}', c['lr'], k)
    close_params(crit_from_flat(critic_net, target.flat), st.target, f'block-batch target step {k}', c['lr'], k); close(N(log_alpha), st.log_alpha, f'log_alpha step {k}')
    close(N(ao.exp_avg), st.actor_m, f'actor m step {k}', atol_scale=1e-5 * k); close(crit_from_flat(critic_net, co.exp_avg), st.critic_m, f'critic m step {k}', atol_scale=1e-5 * k)


def test_bc_update_matches_oracle_and_reference(golden_dir):
  g = load(golden_dir, 'bc_hopper')
  S, A = gi.DIMS['hopper']
  rs = np.random.RandomState(7)
  p = gi.mlp_params(rs, S, 256, 2, 2 * A, out_scale=0.3)
  actor = il.SoftActor(S, A, Cfg(hidden_size=256, depth=2, activation='relu'), device=DEV)
  actor.flat.copy_(T(p))
  opt = il.AdamW(actor, lr=2.5e-4, weight_decay=0.01)
  m, v = np.zeros_like(p), np.zeros_like(p)
  shapes = onets.mlp_shapes(S, 256, 2, 2 * A)
  f64 = load(golden_dir, 'f64_brackets')
  for k in range(1, 4):
    b = gi.transitions(rs, 256, S, A, weighted=True)
    b['actions'][:3] = np.array([1.0, -1.0, 0.9999999])[:, None]
    if k == 1:   # why log pi is compared at 1e-4 below: atanh at the clamp (|a| = 1 - 1e-6) amplifies ulps. Bracket on the initial parameters (identical on both sides):
      bracket(N(actor.log_prob(T(b['states']), T(b['actions']))), f64['bc_hopper.logp_init_f32'], f64['bc_hopper.logp_init'], 'bc log pi (initial parameters)')
    loss = il.behavioural_cloning_update(actor, tbatch(b), opt)
    oloss = osac.bc_update(p, m, v, k, shapes, A, b, lr=2.5e-4, weight_decay=0.01)
    close(N(loss), oloss, f'bc loss {k}', rtol=1e-5, atol_scale=1e-5)
    close_params(N(actor.flat), p, f'bc actor {k}', 2.5e-4, k); close(N(opt.exp_avg), m, f'bc m {k}', atol_scale=1e-5 * k)
    close_params(gi.strided(N(actor.flat)), g[f'actor_{k}'], f'golden bc actor {k}', 2.5e-4, k)
    close(N(actor.log_prob(T(b['states']), T(b['actions']))), g[f'logp_{k}'], f'golden bc logp {k}', rtol=1e-4, atol_scale=1e-5)


def test_actor_act_matches_oracle():
  c = gi.sac_case(3, 'halfcheetah', 256, 256, 1)
  actor = make_sac(c)[0]
  for n in (1, 5, 16, 37):
    s, eps = c['batches'][0]['states'][:n], c['eps_cur'][0][:n]
    out, _ = onets.mlp_forward(onets.unpack(c['actor'], onets.mlp_shapes(c['S'], c['H'], 2, 2 * c['A'])), s)
    mean, _, _, std = onets.actor_head(out, c['A'])
    x = mean + eps * std
    a, lp = actor(T(s)).sample_with_log_prob(T(eps))
    close(N(a), np.tanh(x), f'act sample n={n}', atol_scale=4e-6); close(N(lp), onets.tanh_gaussian_logp(x, mean, std), f'act logp n={n}', atol_scale=4e-6)
    close(N(actor.get_greedy_action(T(s))), np.tanh(mean), f'greedy n={n}', atol_scale=4e-6)
  # Philox path: finite, in (-1, 1), reproducible distribution moments
  a = actor(T(c['batches'][0]['states'])).sample()
  assert torch.isfinite(a).all() and float(a.abs().max()) <= 1.0 and 0.05 < float(a.std()) < 1.0


def test_adam_and_polyak_kernels():
  rs = np.random.RandomState(0)
  n = 100_003
  p, gr = rs.standard_normal(n).astype(np.float32), (rs.standard_normal(n) * rs.uniform(1e-6, 1, n)).astype(np.float32)
  m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
  pt = T(p.copy())
  opt = il.AdamW(pt, lr=3e-4, weight_decay=0.1)
  for t in range(1, 4):
    opt.step(T(gr * t))
    onets.adam_step(p, gr * t, m, v, t, 3e-4, 0.1)
    np.testing.assert_allclose(N(pt), p, rtol=2e-7, atol=1e-9); np.testing.assert_allclose(N(opt.exp_avg_sq), v, rtol=2e-7, atol=0)
  tgt = rs.standard_normal(n).astype(np.float32)
  tt = T(tgt.copy())
  _lib.check(_lib.lib().il_polyak(_lib.ptr(tt), _lib.ptr(pt), n, 0.995, _lib.stream_ptr()))
  onets.polyak(tgt, p, 0.995)
  np.testing.assert_allclose(N(tt), tgt, rtol=2e-7, atol=1e-9)


# ------------------------------------------------------------------------------------------------ GAIL
GAIL_CASES = [
    ('gail_default', dict(seed=31), dict(lr=3e-5, weight_decay=10, grad_penalty=1.0, entropy_bonus=0.0)),
    ('gail_h128_ent', dict(seed=32, hidden=128), dict(lr=7.3e-5, weight_decay=6.35, grad_penalty=0.32, entropy_bonus=0.0155)),
    ('gail_nosn_nogp', dict(seed=33, env='hopper', hidden=32, batch=128, spectral_norm=False), dict(lr=3e-4, weight_decay=0.0, grad_penalty=0.0, entropy_bonus=0.0)),
]


@pytest.mark.parametrize('name,case,hp', GAIL_CASES)
def test_gail_update_matches_oracle_and_reference(golden_dir, name, case, hp):
  g, c, f64 = load(golden_dir, name), gi.gail_case(**case), load(golden_dir, 'f64_brackets')
  d, ods, icfg = make_disc(c)
  icfg.update(loss_function='BCE', grad_penalty=hp['grad_penalty'], entropy_bonus=hp['entropy_bonus'], mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=float('inf'))
  opt = il.AdamW(d, lr=hp['lr'], weight_decay=hp['weight_decay'])
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  for k in range(1, len(c['policy']) + 1):
    pb, eb = c['policy'][k - 1], c['expert'][k - 1]
    d.train()
    il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(c['eps'][k - 1]))
    d.eval()
    ogr = ogail.gail_update(ods, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][k - 1], return_grads=True, **hp)
    close(N(opt.grad), ogr, f'disc grad {k}', atol_scale=4e-6 * k); close(N(opt.grad), g[f'g_{k}'], f'golden disc grad {k}', atol_scale=4e-6 * k)
    close(N(d.flat), ods.pack(), f'disc params {k}', atol_scale=4e-6 * k); close(N(d.flat), g[f'p_{k}'], f'golden disc params {k}', atol_scale=4e-6 * k)
    close(N(opt.exp_avg), ods.m, f'disc m {k}', atol_scale=4e-6 * k); close(N(opt.exp_avg_sq), ods.v, f'disc v {k}', atol_scale=4e-6 * k)
    if c['spectral_norm']:
      for nm, val in d.views().items():
        close(N(val), getattr(ods, nm), f'{nm} {k}'); close(N(val), g[f'{nm}_{k}'], f'golden {nm} {k}')
    for rf in ('AIRL', 'GAIL', 'FAIRL'):
      d.reward_function = rf
      r = d.predict_reward(T(pb['states']), T(pb['actions']))
      close(N(r), ogail.predict_reward(ods, cat(pb), rf), f'reward {rf} {k}', rtol=1e-4, atol_scale=1e-5)
      close(N(r), g[f'reward_{rf}_{k}'], f'golden reward {rf} {k}', rtol=1e-4, atol_scale=1e-5)
    d.reward_function = 'AIRL'
    close(N(d(T(pb['states']), T(pb['actions']))), g[f'logits_{k}'], f'golden logits {k}', atol_scale=4e-6)
    # why rewards are compared at 1e-4: log(D) - log1p(-D) crosses 0 near D = 1/2 (the reference's own float32 result is off by up to 1.5e-3 elementwise against its
    # float64 evaluation). Bracket on the REFERENCE's parameters / u / v of this step, so that both float32 results evaluate the same function:
    keep = d.flat.clone(), {nm: val.clone() for nm, val in d.views().items()} if c['spectral_norm'] else {}
    d.flat.copy_(T(g[f'p_{k}']))
    for nm, val in (d.views().items() if c['spectral_norm'] else ()):
      val.copy_(T(g[f'{nm}_{k}']))
    for rf in ('AIRL', 'GAIL', 'FAIRL'):
      d.reward_function = rf
      bracket(N(d.predict_reward(T(pb['states']), T(pb['actions']))), g[f'reward_{rf}_{k}'], f64[f'{name}.reward_{rf}_{k}'], f'{name} reward {rf} {k}')
    d.reward_function = 'AIRL'
    d.flat.copy_(keep[0])
    for nm, val in (d.views().items() if c['spectral_norm'] else ()):
      val.copy_(keep[1][nm])


# ------------------------------------------------------------------------------------------------ GMMIL / PWIL
@pytest.mark.parametrize('name,dims', [('small', (64, 48, 24)), ('ant', (256, 256, 120))])
def test_gmmil_matches_oracle_and_reference(golden_dir, name, dims):
  g = load(golden_dir, 'gmmil')
  X, E, w, we = gi.gmmil_case(11, *dims)
  D = dims[2]
  S = D - 8 if D > 8 else D - 2
  disc = il.GMMILDiscriminator(S, D - S, Cfg(state_only=False))
  args = (T(X[:, :S]), T(X[:, S:]), T(E[:, :S]), T(E[:, S:]), T(w), T(we))
  close(N(il_training.gmmil_sqdist(disc, *args[:4]))[:16], g[f'{name}_sqdist_xe'], 'sqdist')
  r, sim, self_sim = il_training.gmmil_predict_reward(disc, *args, return_parts=True)
  np.testing.assert_allclose([disc.gamma_1, disc.gamma_2], g[f'{name}_gammas'], rtol=1e-5)
  _, osim, oself = ogmmil.gmmil_reward(X, E, w, we, disc.gamma_1, disc.gamma_2, return_parts=True)
  close(N(sim), osim, 'similarity'); close(N(self_sim), oself, 'self similarity')
  assert np.abs(N(r) - g[f'{name}_reward_first']).max() <= 1e-5 * np.abs(osim).max()
  X2, _, w2, _ = gi.gmmil_case(12, *dims)
  r2 = disc.predict_reward(T(X2[:, :S]), T(X2[:, S:]), args[2], args[3], T(w2), args[5])
  assert np.abs(N(r2) - g[f'{name}_reward_second']).max() <= 1e-5 * np.abs(osim).max()


def test_gmmil_full_size_properties():
  """BASELINE config 4 (B=1024, Ant dims D=120): permutation equivariance and a row-subset check against the oracle."""
  X, E, w, we = gi.gmmil_case(5, 1024, 1024, 120, weighted=True)
  disc = il.GMMILDiscriminator(112, 8, Cfg(state_only=False))
  disc.gamma_1, disc.gamma_2 = 0.37, 0.91
  f = lambda X_, w_: il_training.gmmil_predict_reward(disc, T(X_[:, :112]), T(X_[:, 112:]), T(E[:, :112]), T(E[:, 112:]), T(w_), T(we), return_parts=True)
  r, sim, self_sim = f(X, w)
  perm = np.random.RandomState(0).permutation(1024)
  rp, simp, selfp = f(X[perm], w[perm])
  close(N(simp), N(sim)[perm], 'perm similarity', atol_scale=4e-6); close(N(selfp), N(self_sim)[perm], 'perm self similarity', atol_scale=4e-6)
  # the oracle on 64 query rows against the full sets (self term needs all rows: use the oracle's chunked distance)
  wn, wen = w / w.sum(dtype=np.float32), we / we.sum(dtype=np.float32)
  rows = perm[:64]
  dxe, dxx = ogmmil.squared_distance(X[rows], E), ogmmil.squared_distance(X[rows], X)
  osim = sum(wn[rows] * (np.exp(np.float32(-gm) * dxe) @ wen) for gm in (0.37, 0.91))
  oself = sum(wn[rows] * (np.exp(np.float32(-gm) * dxx) @ wn) for gm in (0.37, 0.91))
  close(N(sim)[rows], osim, 'full-size similarity'); close(N(self_sim)[rows], oself, 'full-size self similarity')


@pytest.mark.parametrize('offset', [0.0, 50.0, 1000.0])
@pytest.mark.parametrize('dims', [(1024, 1024, 120, 112), (300, 200, 35, 29), (64, 48, 24, 16), (130, 257, 128, 120), (70, 33, 15, 12)])
def test_gmmil_centred_gram_form_is_as_close_to_float64_as_the_direct_form(dims, offset):
  """k_gmmil_mfma (round 6, the default reward launch for D <= 128): pair distances as |x - c|^2 + |y - c|^2 - 2 (x - c).(y - c) on the matrix pipes. The uncentred Gram form
  loses the digits the data's offset takes (tests/test_gmmil_centred_form.py: rewards off by 2e-5 at offset 1000); centred on a mean of expert rows every term is of the size
  of the spread. Checked against float64 at the bound of the other GMMIL tests (1e-5 max|similarity|, measured ~2e-7 of it) with observations around 0, 50 and 1000, at the
  timed size, ragged shapes, rows that are not whole 16-byte lanes, D = 128 exactly and D < 16; three calls each (self-resetting arrival counters)."""
  n1, n2, D, S = dims
  if not torch.cuda.is_available() and n1 > 300:
    pytest.skip('the timed size takes minutes on the host emulator; the other shapes cover the code paths')
  X, E, w, we = gi.gmmil_case(31, n1, n2, D, weighted=True)
  X, E = (X + np.float32(offset)).astype(np.float32), (E + np.float32(offset)).astype(np.float32)
  d64 = lambda a, b: ((a.astype(np.float64)[:, None, :] - b.astype(np.float64)[None, :, :]) ** 2).mean(2) if a.shape[0] * b.shape[0] <= 1 << 17 else \
      np.concatenate([((a[i:i + 64].astype(np.float64)[:, None, :] - b.astype(np.float64)[None, :, :]) ** 2).mean(2) for i in range(0, a.shape[0], 64)])
  dxe, dxx, dee = d64(X, E), d64(X, X), d64(E, E)
  g1, g2 = 1.0 / (np.median(dxe) + 1e-8), 1.0 / (np.median(dee) + 1e-8)
  wn, wen = w.astype(np.float64) / w.astype(np.float64).sum(), we.astype(np.float64) / we.astype(np.float64).sum()
  sim64 = sum(wn * (np.exp(-gm * dxe) @ wen) for gm in (g1, g2))
  self64 = sum(wn * (np.exp(-gm * dxx) @ wn) for gm in (g1, g2))
  disc = il.GMMILDiscriminator(S, D - S, Cfg(state_only=False))
  disc.gamma_1, disc.gamma_2 = float(np.float32(g1)), float(np.float32(g2))
  g1f, g2f = np.float64(np.float32(g1)), np.float64(np.float32(g2))
  sim64 = sum(wn * (np.exp(-gm * dxe) @ wen) for gm in (g1f, g2f)); self64 = sum(wn * (np.exp(-gm * dxx) @ wn) for gm in (g1f, g2f))
  args = (T(X[:, :S]), T(X[:, S:]), T(E[:, :S]), T(E[:, S:]), T(w), T(we))
  for _ in range(3):
    r, sim, self_sim = il_training.gmmil_predict_reward(disc, *args, return_parts=True)
  bound = 1e-5 * np.abs(sim64).max()
  assert np.abs(N(sim) - sim64).max() <= bound and np.abs(N(self_sim) - self64).max() <= bound, (np.abs(N(sim) - sim64).max() / bound, np.abs(N(self_sim) - self64).max() / bound)
  assert np.abs(N(r) - (sim64 - self64)).max() <= bound, np.abs(N(r) - (sim64 - self64)).max() / bound


GMMIL_FORMS_WORKER = r'''
import hashlib, sys
sys.path[:0] = [sys.argv[1], sys.argv[1] + '/tests', sys.argv[1] + '/tests/golden']
import numpy as np, torch
import inputs as gi
import imitation_learning_amd as il
from imitation_learning_amd import training as T_
from gpu_util import T, N, Cfg
h = hashlib.sha256()
for seed, (n1, n2, D), S in ((5, (1024, 1024, 120), 112), (6, (300, 200, 35), 29), (7, (64, 48, 24), 16), (8, (130, 257, 132), 124), (9, (130, 257, 128), 120), (10, (40, 700, 152), 144)):   # Ant at the timed size; ragged, rows that are not whole 16-byte lanes; small; D > 128; ragged with D % 8 == 0 (the scalar-row kernel: 2 column tiles of 256, the second nearly empty); its largest D
  X, E, w, we = gi.gmmil_case(seed, n1, n2, D, weighted=True)
  disc = il.GMMILDiscriminator(S, D - S, Cfg(state_only=False))
  disc.gamma_1, disc.gamma_2 = 0.37, 0.91
  args = (T(X[:, :S]), T(X[:, S:]), T(E[:, :S]), T(E[:, S:]))
  for _ in range(3):   # the arrival counters are self-resetting: the third call must see what the first saw
    r, sim, self_sim = T_.gmmil_predict_reward(disc, *args, T(w), T(we), return_parts=True)
  d = T_.gmmil_sqdist(disc, *args)
  for t in (r, sim, self_sim, d): h.update(np.ascontiguousarray(N(t)).tobytes())
print('DIGEST', h.hexdigest())
'''


def test_gmmil_launch_forms_are_bit_identical(tmp_path):
  """k_gmmil_sx (round 5: the row operand in scalar registers, 32 x 256 pairs per workgroup), k_gmmil_resident (all features of both tiles resident in LDS, fence-free arrival),
  k_gmmil_direct (chunked ring) and k_gmmil_pack + k_gmmil_tile keep every
  pair's accumulation order over the features, the 64-column partial sums and the tile-ordered final sums: the same bits for rewards, both similarities and the distance
  matrix, at the timed size, for ragged shapes, for rows that are not whole 16-byte lanes and for D > 128. (The switches are read once per process: one process per form.)"""
  import subprocess, sys
  script = tmp_path / 'forms.py'
  script.write_text(GMMIL_FORMS_WORKER)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  digests = {}
  for name, env in (('scalar rows', {}), ('resident', dict(IL_GMMIL_SX='0')), ('direct', dict(IL_GMMIL_SX='0', IL_GMMIL_RESIDENT='0')), ('pack+tile', dict(IL_GMMIL_DIRECT='0'))):
    r = subprocess.run([sys.executable, str(script), root], env=dict(os.environ, IL_GMMIL_MFMA='0', **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    digests[name] = [l for l in r.stdout.splitlines() if l.startswith('DIGEST')][-1].split()[1]
  assert len(set(digests.values())) == 1, digests


def test_pwil_matches_oracle_and_reference(golden_dir):
  g = load(golden_dir, 'pwil')
  Nn, D, steps, Th = 400, 10, 260, 120
  atoms, agent = gi.pwil_case(21, Nn, D, steps)
  S, A = D - 3, 3
  mem = il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn),
                                                          next_states=torch.from_numpy(atoms[:, :S]), terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn),
                                                          num_trajectories=4), device=DEV)
  d = il.PWILDiscriminator(S, A, Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, Th)
  close(N(d.data_scale), g['scale'], 'scale'); close(N(d.data_offset), g['offset'], 'offset')
  o = opwil.PwilOracle(atoms, Th, 5, 5)
  rewards, orewards = [], []
  for t in range(steps):
    rewards.append(d.compute_reward(T(agent[t:t + 1, :S]), T(agent[t:t + 1, S:])))
    orewards.append(o.compute_reward(agent[t]))
    if t % Th == Th - 1:
      d.reset(); o.reset()
  np.testing.assert_allclose(rewards, orewards, rtol=2e-5)
  np.testing.assert_allclose(rewards, g['rewards'], rtol=2e-5)
  bracket(rewards, g['rewards'], load(golden_dir, 'f64_brackets')['pwil.rewards'], 'PWIL rewards')   # exp(-beta T / sqrt(D) * cost) turns 1e-8 of cost into 1e-5 of reward
  assert int((d.expert_weights >= 0).sum()) == int(g['remaining'][0])


# ------------------------------------------------------------------------------------------------ whole update block
@pytest.mark.parametrize('name,Nn,Th', [('one_workgroup', 3000, 10), ('two_launches_serial_merge', 6000, 30), ('step', 6000, 600)])
def test_pwil_every_launch_path_matches_oracle(name, Nn, Th):
  """il_pwil_reward picks its kernels from the atoms one step can consume (m = ceil(N / T) + 2) and the number of 256-atom chunks G: the one-launch k_pwil_step when
  G m <= 4096 candidates fit in LDS (every shipped configuration: T = 1000), k_pwil_select + the serial merge beyond that, the one-workgroup k_pwil_reward when m > 256
  (short horizons). All three against the oracle over a few episodes with resets: rewards at the bound of the other PWIL tests, the remaining atoms exactly."""
  D, A, steps = 10, 3, 2 * Th + 7 if Th <= 30 else 40
  S = D - A
  atoms, agent = gi.pwil_case(31, Nn, D, steps)
  mem = il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn),
                                                          next_states=torch.from_numpy(atoms[:, :S]), terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn),
                                                          num_trajectories=4), device=DEV)
  d = il.PWILDiscriminator(S, A, Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, Th)
  m, G = int(np.ceil(Nn / Th)) + 2, -(-Nn // 256)
  assert {'one_workgroup': m > 256, 'two_launches_serial_merge': m <= 256 and G * m > 4096, 'step': G * m <= 4096}[name], (m, G)
  o = opwil.PwilOracle(atoms, Th, 5, 5)
  got, want = [], []
  for k in range(steps):
    got.append(float(d.compute_reward(T(agent[k:k + 1, :S]), T(agent[k:k + 1, S:]))))
    want.append(o.compute_reward(agent[k]))
    if k % Th == Th - 1 or (Th > 30 and k == 17):
      d.reset(); o.reset()
  np.testing.assert_allclose(got, want, rtol=2e-5)
  assert int((d.expert_weights >= 0).sum()) == len(o.weights)


def _make_plan(algorithm, seed, device_draw=True, loss='BCE', entropy_bonus=0.0, B=256, margin=float('inf'), reward_function='AIRL', mixup_alpha=1):
  S, A = gi.DIMS['halfcheetah']
  torch.manual_seed(seed)
  cfg = Cfg(hidden_size=256, depth=2, activation='relu')
  actor, critic = il.SoftActor(S, A, cfg, device=DEV), il.TwinCritic(S, A, cfg, device=DEV)
  target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=DEV)
  ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
  rs = np.random.RandomState(seed)
  mem = il.ReplayMemory(20000, S, A, True, device=DEV); fill_memory(mem, gi.transitions(rs, 5000, S, A), 5000)
  emem = il.ReplayMemory(2000, S, A, True, device=DEV); fill_memory(emem, gi.transitions(rs, 2000, S, A, state_shift=0.5), 2000)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function=loss, grad_penalty=1.0, entropy_bonus=entropy_bonus, mixup_alpha=mixup_alpha, pos_class_prior=0.7, nonnegative_margin=margin,
             discriminator=Cfg(hidden_size=64, depth=1, activation='relu', reward_shaping=False, subtract_log_policy=False, reward_function=reward_function))
  disc = il.GAILDiscriminator(S, A, icfg, 0.97, device=DEV)
  do = il.AdamW(disc, lr=3e-5, weight_decay=10)
  plan = il.UpdatePlan(algorithm, actor, critic, log_alpha, target, mem, ao, co, to, B, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc, discriminator_optimiser=do,
                       imitation_cfg=icfg, device_index_draw=device_draw)
  return plan, (actor, critic, target, log_alpha, disc)


@pytest.mark.parametrize('reward_function', ['AIRL', 'GAIL', 'FAIRL'])
def test_inline_relabel_heads_equal_the_reward_kernel(reward_function):
  """models.py:177-180 inside the chained SAC launch (disc_reward.hpp: the rows a critic tile already holds are relabelled by the discriminator the other branch has just
  stepped) against `predict_reward` (k_gail_reward) on the same rows and the same, updated discriminator: the three reward heads, bit for bit."""
  il.seed(37); il_training._NOISE.clear()
  plan, nets = _make_plan('GAIL', 19, reward_function=reward_function, B=64)
  for _ in range(2):
    plan.run()
  torch.cuda.synchronize()
  assert plan.inline_relabel and plan.sync_timeouts() == 0
  t = plan.transitions
  want = nets[4].predict_reward(t['states'].contiguous(), t['actions'].contiguous())
  torch.cuda.synchronize()
  assert np.isfinite(N(plan.rewards)).all() and (N(plan.rewards) > 0).any() == (reward_function != 'FAIRL' or (N(want) > 0).any())
  np.testing.assert_array_equal(N(plan.rewards), N(want))


def _beta_draws(seed, counter, alpha, n):
  """il_noise_fill_beta for the update counter `counter` (include/il_hip.h): what a captured Mixup update with mixup_alpha != 1 consumes."""
  ctr, out = torch.tensor([counter], dtype=torch.int32, device=DEV), torch.empty(n, device=DEV)
  _lib.check(_lib.lib().il_noise_fill_beta(C.c_uint64(seed), _lib.ptr(ctr), alpha, n, _lib.ptr(out), _lib.stream_ptr()))
  return N(out)


@pytest.mark.parametrize('alpha', [0.05, 0.3, 0.5, 2.0, 7.5])   # 0.05: both gamma variates underflow in fp32 for ~1e-4 of the rows (log-space guard)
def test_device_beta_draws_are_beta_distributed(alpha):
  """training.py:105-107 draws the Mixup coefficients from Beta(alpha, alpha) with torch's CPU sampler; a captured update draws them on the device (Philox + Marsaglia-Tsang,
  il_noise_fill_beta): the same distribution - Kolmogorov-Smirnov against scipy's Beta, mean 1/2, variance 1 / (4 (2 alpha + 1)) - inside (0, 1), a pure function of
  (key, counter, index), different and uncorrelated for another counter or key."""
  from scipy import stats
  n = 200_000
  u = _beta_draws(12345, 7, alpha, n).astype(np.float64)
  assert np.isfinite(u).all() and 0.0 <= u.min() and u.max() <= 1.0
  if alpha >= 0.2:
    assert stats.kstest(u, stats.beta(alpha, alpha).cdf).pvalue > 1e-3
  else:
    # Beta(0.05, 0.05) puts ~20 % of its mass within 6e-8 of 1, where float32 has no numbers left (they round to 1.0; float32 resolves the same mass near 0 down to 1e-38):
    # the lower half against the truncated law, the halves' weights against 1/2
    # and ~1 % below 1e-38, which float32 flushes to 0): the law truncated to (1e-30, 0.5) on the draws in it, and the weights of the three pieces
    F, cut = stats.beta(alpha, alpha).cdf, 1e-30
    lo = u[(u > cut) & (u < 0.5)]
    assert abs((u < 0.5).mean() - 0.5) < 5 * np.sqrt(0.25 / n) and abs((u <= cut).mean() - F(cut)) < 5 * np.sqrt(F(cut) / n)
    assert stats.kstest(lo, lambda x: (F(x) - F(cut)) / (0.5 - F(cut))).pvalue > 1e-3
  var = 1.0 / (4.0 * (2.0 * alpha + 1.0))
  assert abs(u.mean() - 0.5) < 5 * np.sqrt(var / n) and abs(u.var() - var) < 0.02 * var
  np.testing.assert_array_equal(u, _beta_draws(12345, 7, alpha, n).astype(np.float64))
  for other in (_beta_draws(12345, 8, alpha, n), _beta_draws(12346, 7, alpha, n)):
    assert abs(np.corrcoef(u, other)[0, 1]) < 5 / np.sqrt(n)


@pytest.mark.parametrize('alpha', [0.4, 2.5])
def test_update_plan_mixup_with_beta_coefficients_drawn_on_the_device(alpha):
  """loss_function = Mixup with mixup_alpha != 1 in the captured plan: the Beta(alpha, alpha) coefficients of update k are drawn by a launch captured ahead of the discriminator
  step (they are exactly il_noise_fill_beta's draws for counter k, new every update), the graph replays evolve the learner bit for bit like eager launches, and the
  discriminator step each update took is the oracle's Mixup step (oracle/gail.py, pinned to the reference) on the rows the plan gathered with those coefficients and the
  gradient-penalty uniforms of the same counter."""
  B, K = 64, 4

  def gp_uniforms(key, k):   # the gradient-penalty U(0, 1) of update k (include/il_hip.h il_noise_fill, IL_NOISE_GP = 3)
    out = torch.empty(B, device=DEV)
    _lib.check(_lib.lib().il_noise_fill(C.c_uint64(key), k, 3, B, _lib.ptr(out), _lib.stream_ptr()))
    return N(out)
  results = []
  for mode in ('eager', 'graph'):
    il.seed(43); il_training._NOISE.clear()
    plan, nets = _make_plan('GAIL', 27, loss='Mixup', mixup_alpha=alpha, B=B, entropy_bonus=0.05)
    assert not plan.device_sync and plan._beta_alpha == alpha, 'the draw reads the update counter on the device: stream-ordered schedule'
    disc = nets[4]
    ods = ogail.DiscState(disc.in_dim, 64, True)
    ods.unpack_into(N(disc.flat)); v = disc.views()
    for kk in ('u1', 'v1', 'u2', 'v2'): getattr(ods, kk)[...] = N(v[kk])
    key = int(plan.disc.noise_seed)
    for k in range(K):
      if mode == 'graph' and k == 1: plan.capture(warmup=0)
      (plan.replay if (mode == 'graph' and k >= 1) else plan.run)()
      torch.cuda.synchronize()
      eps = N(plan.eps_mix).copy()
      np.testing.assert_array_equal(eps, _beta_draws(key, k, alpha, B), err_msg=f'update {k}: the coefficients are the Mixup stream\'s Beta draws of counter {k}')
      if mode == 'eager':
        cat = lambda t: np.concatenate([N(t['states']), N(t['actions'])], axis=1)
        t, e = plan.transitions, plan.expert_transitions
        ogail.gail_update(ods, cat(t), N(t['weights']), cat(e), N(e['weights']), gp_uniforms(key, k), lr=3e-5, weight_decay=10, grad_penalty=1.0, entropy_bonus=0.05,
                          loss_function='Mixup', eps_mix=eps)
        close_params(N(disc.flat)[:ods.pack().size], ods.pack(), f'mixup alpha {alpha}: discriminator after update {k + 1}', 3e-5, k + 1)
    results.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.logp), N(plan.rewards)])
  for a, b in zip(*results):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('algorithm', ['SAC', 'GAIL'])
def test_update_plan_graph_replay_equals_eager(algorithm):
  """A captured hipGraph of the whole update must evolve the learner exactly like eager launches (same Philox counters, same MT stream)."""
  results = []
  for mode in ('eager', 'graph'):
    il.seed(11)
    il_training._NOISE.clear()
    plan, nets = _make_plan(algorithm, 5)
    if mode == 'graph':
      plan.capture(warmup=0)   # capture itself does not execute kernels
      for _ in range(5):
        plan.replay()
    else:
      for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    assert plan.sync_timeouts() == 0 and (algorithm != 'GAIL' or plan.device_sync), 'the device-side hand-off between the two branches must be live and never time out'
    results.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.idx), N(plan.logp)])
  for a, b in zip(*results):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('extra_streams', [0, 1, 2, 3, 5])
def test_capture_warmup_runs_on_the_probed_stream_pair(extra_streams):
  """HIP multiplexes streams onto a few hardware queues (round-robin at creation). capture() used to warm up on a fresh stream; when that stream shared a
  queue with the plan's side stream the two branches serialised, the device-side waits expired (48 per update, seen in the full suite) and the warm-up updates
  consumed stale rewards. The warm-up now runs on the caller's stream - the pair the probe validated - whatever else the process created before."""
  keep = [torch.cuda.Stream() for _ in range(extra_streams)]
  for s in keep:
    with torch.cuda.stream(s):
      torch.zeros(1, device=DEV)
  il.seed(1); il_training._NOISE.clear()
  plan, nets = _make_plan('GAIL', 5)
  plan.capture(warmup=2)
  for _ in range(3):
    plan.replay()
  torch.cuda.synchronize()
  assert plan.sync_timeouts() == 0
  assert plan.device_sync, 'a side stream that shares the caller\'s hardware queue must be replaced, not silently traded for the slower stream-dependency schedule'
  del keep


def test_update_plan_host_and_device_index_draws_agree():
  outs = []
  for device_draw in (True, False):
    il.seed(3)
    il_training._NOISE.clear()
    plan, nets = _make_plan('GAIL', 9, device_draw)
    for _ in range(3):
      plan.run()
    torch.cuda.synchronize()
    outs.append((N(plan.idx), N(plan.eidx), N(nets[0].flat)))
  for a, b in zip(*outs):
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('exchange', ['none', 'peer_windows', 'peer_windows_in_apply'])
def test_data_parallel_path_equals_fused_path_on_one_rank(monkeypatch, exchange):
  """DataParallelUpdate (IL_FLAG_GRADS_ONLY kernels -> [all-reduce] -> apply kernels) must evolve a learner like the fused UpdatePlan:
  with one rank the all-reduce is the identity, so any difference would be a bug in the split path the multi-GPU run uses.
  exchange = 'peer_windows': the peer-window exchange with a world of one rank (the gradients travel through the window's slot and back), one il_peer_allreduce_mean launch per
  sync point; 'peer_windows_in_apply': the critic / actor exchanges inside the apply launches (il_sac_dp_phase_peer, IL_PEER_APPLY=1); 'none': no exchange is enqueued."""
  from imitation_learning_amd.parallel import DataParallelUpdate
  if exchange != 'none':
    monkeypatch.setenv('IL_PEER_EXCHANGE', 'require')
    monkeypatch.setenv('IL_PEER_APPLY', '1' if exchange == 'peer_windows_in_apply' else '0')
  outs = []
  for dp in (False, True):
    il.seed(21)
    il_training._NOISE.clear()
    plan, nets = _make_plan('GAIL', 13)
    runner = DataParallelUpdate(plan) if dp else plan
    for _ in range(4):
      runner.run()
    torch.cuda.synchronize()
    outs.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.logp), N(plan.q), N(plan.rewards)])
  for i, (a, b) in enumerate(zip(*outs)):
    assert np.isfinite(a).all() and np.isfinite(b).all()
    np.testing.assert_array_equal(a, b, err_msg=f'tensor {i}')
  # and the captured DP graph replays like its eager form
  il.seed(21); il_training._NOISE.clear()
  plan, nets = _make_plan('GAIL', 13)
  dp = DataParallelUpdate(plan).capture(warmup=0)
  for _ in range(4):
    dp.replay()
  torch.cuda.synchronize()
  for a, n in zip(outs[1], nets):
    np.testing.assert_array_equal(a, N(n.flat if hasattr(n, 'flat') else n))
  assert (dp.peer is not None) == (exchange != 'none') and dp.exchange_timeouts() == 0


# ------------------------------------------------------------------------------------------------ size-independent properties / edge cases
def test_sac_gradient_of_concatenated_batch_is_mean_of_shard_gradients():
  """BASELINE size (B=256 per shard, HalfCheetah dims): the data-parallel identity on the real kernels -- gradients from
  IL_FLAG_GRADS_ONLY on two 256-row shards average to the gradient on the 512-row concatenation (every loss is a mean)."""
  c = gi.sac_case(41, 'halfcheetah', 256, 512, 1)
  big = c['batches'][0]
  grads = {}
  for name, sl in (('all', slice(0, 512)), ('lo', slice(0, 256)), ('hi', slice(256, 512))):
    actor, critic, target, log_alpha, ao, co, to = make_sac(c)
    tb = tbatch({k: v[sl] for k, v in big.items()})
    n = tb['states'].size(0)
    d = il_training.sac_descriptor(actor, critic, log_alpha, target, n, ao, co, to, c['discount'], c['entropy_target'], c['polyak'])
    bd = il_memory.batch_desc(tb)
    e1 = T(c['eps_next'][0][sl])
    _lib.check(_lib.lib().il_sac_critic_step(C.byref(d), C.byref(bd), _lib.ptr(e1), _lib.IL_FLAG_GRADS_ONLY, _lib.stream_ptr()))
    torch.cuda.synchronize()
    grads[name] = N(co.grad)
  mean = 0.5 * (grads['lo'] + grads['hi'])
  close(mean, grads['all'], 'critic grad: mean of shards vs concatenated batch', atol_scale=4e-6)


@pytest.mark.parametrize('env,hidden,batch', [('halfcheetah', 128, 1024), ('walker2d', 192, 48), ('hopper', 256, 16)])
def test_sac_update_other_shapes(env, hidden, batch):
  """Largest tuned batch (1024), the smallest legal batch (one 16-row tile) and non-power-of-two hidden sizes, one update vs the oracle."""
  c = gi.sac_case(50 + batch, env, hidden, batch, 1)
  actor, critic, target, log_alpha, ao, co, to = make_sac(c)
  st = make_sac_oracle(c)
  b = c['batches'][0]
  logp, q = il.sac_update(actor, critic, log_alpha, target, tbatch(b), ao, co, to, c['discount'], c['entropy_target'], c['polyak'], eps_next=T(c['eps_next'][0]), eps_cur=T(c['eps_cur'][0]))
  ologp, oq = osac.sac_update(st, b, c['eps_next'][0], c['eps_cur'][0], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'], lr=c['lr'])
  close(N(logp), ologp, 'logp', atol_scale=4e-6); close(N(q), oq, 'q', atol_scale=4e-6)
  close_params(N(actor.flat), st.actor, 'actor', c['lr']); close_params(crit_from_flat(critic, critic.flat), st.critic, 'critic', c['lr'])
  close_params(crit_from_flat(critic, target.flat), st.target, 'target', c['lr']); close(N(log_alpha), st.log_alpha, 'log_alpha')


def test_unsupported_shapes_and_options_fail_loudly():
  # shapes outside the fused kernels run through csrc/general.hip (test_general_shape_sac_matches_oracle_and_reference); what that engine does not cover still raises
  assert il.SoftActor(18, 6, Cfg(hidden_size=256, depth=3, activation='relu')).general and il.SoftActor(18, 6, Cfg(hidden_size=100, depth=2, activation='relu')).general
  assert il.TwinCritic(18, 6, Cfg(hidden_size=256, depth=2, activation='tanh')).general and not il.TwinCritic(18, 6, Cfg(hidden_size=256, depth=2, activation='relu')).general
  assert il.SoftActor(18, 12, Cfg(hidden_size=256, depth=2, activation='relu')).general   # 2A > 16: the fused head is too narrow
  with pytest.raises(NotImplementedError):
    il.SoftActor(18, 6, Cfg(hidden_size=256, depth=9, activation='relu'))
  with pytest.raises(NotImplementedError):
    il.TwinCritic(18, 6, Cfg(hidden_size=4096, depth=2, activation='relu'))
  with pytest.raises(ValueError):
    il.SoftActor(18, 6, Cfg(hidden_size=256, depth=2, activation='gelu'))
  with pytest.raises(NotImplementedError):
    il.TwinCritic(18, 6, Cfg(hidden_size=256, depth=2, activation='relu', dropout=0.1))
  c = gi.sac_case(3, 'halfcheetah', 256, 24, 1)   # batch not a multiple of the 16-row tile
  actor, critic, target, log_alpha, ao, co, to = make_sac(c)
  with pytest.raises(RuntimeError, match='multiple of 16'):
    il.sac_update(actor, critic, log_alpha, target, tbatch(c['batches'][0]), ao, co, to, 0.99, -6.0, 0.995)
  g = gi.gail_case(31)
  d, _, icfg = make_disc(g)
  icfg.update(loss_function='Hinge', grad_penalty=1.0, entropy_bonus=0.0)
  with pytest.raises(ValueError, match='Hinge'):
    il.adversarial_imitation_update(None, d, tbatch(g['policy'][0]), tbatch(g['expert'][0]), il.AdamW(d, lr=3e-5, weight_decay=10), icfg)
  with pytest.raises(TypeError):
    il.sac_update(actor, critic, log_alpha, target, {k: v.cpu() for k, v in tbatch(gi.sac_case(3, 'halfcheetah', 256, 32, 1)['batches'][0]).items()}, ao, co, to, 0.99, -6.0, 0.995)


def test_gail_ragged_batch_and_state_only():
  """Discriminator kernels accept a batch that is not a multiple of the 16-row tile (last tile ragged) and the state_only variant."""
  g = gi.gail_case(61, env='hopper', hidden=64, batch=40, steps=1)
  for state_only in (False, True):
    icfg = Cfg(state_only=state_only, spectral_norm=True, loss_function='BCE', grad_penalty=0.5, entropy_bonus=0.01,
               discriminator=Cfg(hidden_size=64, depth=1, activation='relu', reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
    d = il.GAILDiscriminator(g['S'], g['A'], icfg, 0.97, device=DEV)
    D = g['S'] if state_only else g['D']
    ods = ogail.DiscState(D, 64, True)
    ods.unpack_into(N(d.flat)); v = d.views()
    for k in ('u1', 'v1', 'u2', 'v2'):
      getattr(ods, k)[...] = N(v[k])
    opt = il.AdamW(d, lr=1e-4, weight_decay=1.0)
    pb, eb = g['policy'][0], g['expert'][0]
    cat = (lambda b: b['states']) if state_only else (lambda b: np.concatenate([b['states'], b['actions']], axis=1))
    il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(g['eps'][0]))
    ogr = ogail.gail_update(ods, cat(pb), pb['weights'], cat(eb), eb['weights'], g['eps'][0], lr=1e-4, weight_decay=1.0, grad_penalty=0.5, entropy_bonus=0.01, return_grads=True)
    close(N(opt.grad), ogr, f'disc grad (state_only={state_only})', atol_scale=4e-6); close(N(d.flat), ods.pack(), f'disc params (state_only={state_only})', atol_scale=4e-6)
    close(N(d.predict_reward(T(pb['states']), T(pb['actions']))), ogail.predict_reward(ods, cat(pb)), 'reward', rtol=1e-4, atol_scale=1e-5)


def _population_learners(n):
  """n independent GAIL learners (own networks, rings, index streams, Philox counters) as UpdatePlans, reproducibly."""
  il_training._NOISE.clear(); il_training._WS.clear()
  plans, nets_all = [], []
  for l in range(n):
    S, A = gi.DIMS['halfcheetah']
    torch.manual_seed(30 + l)
    cfg = Cfg(hidden_size=256, depth=2, activation='relu')
    actor, critic = il.SoftActor(S, A, cfg, device=DEV), il.TwinCritic(S, A, cfg, device=DEV)
    target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=DEV)
    ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
    rs = np.random.RandomState(30 + l)
    mem = il.ReplayMemory(20000, S, A, True, device=DEV); fill_memory(mem, gi.transitions(rs, 5000, S, A), 5000)
    emem = il.ReplayMemory(2000, S, A, True, device=DEV); fill_memory(emem, gi.transitions(rs, 2000, S, A, state_shift=0.5), 2000)
    mem.index_rng = il.IndexStream(100 + l)
    icfg = Cfg(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=1.0, entropy_bonus=0.0,
               discriminator=Cfg(hidden_size=64, depth=1, activation='relu', reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
    disc = il.GAILDiscriminator(S, A, icfg, 0.97, device=DEV)
    do = il.AdamW(disc, lr=3e-5, weight_decay=10)
    plans.append(il.UpdatePlan('GAIL', actor, critic, log_alpha, target, mem, ao, co, to, 256, 0.97, -0.5 * A, 0.99, expert_memory=emem, discriminator=disc,
                               discriminator_optimiser=do, imitation_cfg=icfg, overlap=False, learner_id=l))
    nets_all.append((actor, critic, target, log_alpha, disc))
  return plans, nets_all


def _population_state(plans, nets_all):
  torch.cuda.synchronize()
  return [[N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(p.idx), N(p.logp), N(p.rewards)] for nets, p in zip(nets_all, plans)]


def test_batched_population_equals_independent_learners():
  """The population axis must not couple learners: L learners advanced by the il_*_population launches evolve exactly like the same
  L learners advanced one by one (same seeds, own index streams, own Philox counters)."""
  results = []
  for batched in (False, True):
    plans, nets_all = _population_learners(3)
    if batched:
      pop = il.BatchedPopulationPlan(plans)
      for _ in range(3):
        pop.run()
    else:
      for _ in range(3):
        for p in plans:
          p.run()
    results.append(_population_state(plans, nets_all))
  for l, (a_l, b_l) in enumerate(zip(*results)):
    for i, (a, b) in enumerate(zip(a_l, b_l)):
      assert np.isfinite(a).all()
      np.testing.assert_array_equal(a, b, err_msg=f'learner {l}, tensor {i}')
  assert not np.array_equal(results[0][0][0], results[0][1][0])  # the learners really are different


def test_population_xcd_decode_and_groups_are_bit_identical():
  """With 8 or more learners the population launches re-decode the linear workgroup id so that learner l sits on XCD l % 8 (pop_ids: full groups of 8 learners interleaved, the
  rest in natural order), and `groups` cuts the population into sub-populations replayed as parallel graph branches. Both are pure re-labelling / re-scheduling: 18 learners
  (two full groups + a tail of two) advanced by one population, by two sub-populations of nine (one full group + a tail each) through a captured graph, and one by one must
  end in the same bits."""
  results = {}
  for how in ('one by one', 'population', 'two groups, captured'):
    plans, nets_all = _population_learners(18)
    if how == 'one by one':
      for _ in range(3):
        for p in plans:
          p.run()
    elif how == 'population':
      pop = il.BatchedPopulationPlan(plans, groups=1)
      for _ in range(3):
        pop.run()
    else:
      pop = il.BatchedPopulationPlan(plans, groups=2)
      pop.run()
      torch.cuda.synchronize()
      pop.capture()
      for _ in range(2):
        pop.replay()
    results[how] = _population_state(plans, nets_all)
  ref = results['one by one']
  for how in ('population', 'two groups, captured'):
    for l, (a_l, b_l) in enumerate(zip(ref, results[how])):
      for i, (a, b) in enumerate(zip(a_l, b_l)):
        np.testing.assert_array_equal(a, b, err_msg=f'{how}: learner {l}, tensor {i}')


@pytest.mark.gpu
def test_population_launch_switches_are_bit_identical():
  """The population path's optional schedules - forward / critic loss chained per tile inside the population launch (IL_POP_CHAIN=1), full-width tile kernels
  (IL_POP_TILE_THREADS=1024), dW without the LDS-staged blocks (IL_POP_DW_LDS=0), no second stream (IL_POP_OVERLAP=0) - run the same arithmetic per element."""
  import subprocess, sys, json
  code = (
      "import sys, json, hashlib, numpy as np, torch; sys.path[:0] = ['.', 'tests', 'tests/golden']\n"
      "import imitation_learning_amd as il, bench\n"
      "from test_gpu_parity import N\n"
      "built = [bench.build(torch.device('cuda'), 0, seed=50 + l, learner_id=50 + l) for l in range(3)]\n"
      "pop = il.BatchedPopulationPlan([b[0] for b in built])\n"
      "for _ in range(4): pop.run()\n"
      "torch.cuda.synchronize()\n"
      "h = hashlib.sha256()\n"
      "for plan, nets, _ in built:\n"
      "  for n in list(nets) + [plan.logp, plan.q, plan.rewards]: h.update(np.ascontiguousarray(N(n.flat if hasattr(n, 'flat') else n)).tobytes())\n"
      "print(json.dumps(dict(digest=h.hexdigest())))\n")
  digests = {}
  for name, env in (('default', {}), ('chain', dict(IL_POP_CHAIN='1', IL_POP_OVERLAP='0')), ('full width', dict(IL_POP_TILE_THREADS='1024')), ('dw tiles', dict(IL_POP_DW_LDS='0')),
                    ('one stream', dict(IL_POP_OVERLAP='0'))):
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (name, r.stderr[-2000:])
    digests[name] = json.loads(r.stdout.strip().splitlines()[-1])['digest']
  assert len(set(digests.values())) == 1, digests


# ---------------------------------------------------------------------------------------------
# acting worker: il_act_step == actor(state).sample() + memory.append + wrap_for_absorbing_states
# ---------------------------------------------------------------------------------------------
def _acting_pair(absorbing, capacity=37, S=18, A=6, H=256):
  cfg = Cfg(hidden_size=H, depth=2, activation='relu')
  torch.manual_seed(3)
  actor_a, actor_b = il.SoftActor(S, A, cfg, device=DEV), il.SoftActor(S, A, cfg, device=DEV)
  actor_a.flat.copy_(torch.randn_like(actor_a.flat) * 0.08); actor_b.flat.copy_(actor_a.flat)
  return actor_a, actor_b, il.ReplayMemory(capacity, S, A, absorbing, device=DEV), il.ReplayMemory(capacity, S, A, absorbing, device=DEV)


def _episode_script(rs, n, S, absorbing):
  """(next_obs, reward, true_terminal, timeout) per step with a few episode ends of both kinds."""
  out = []
  for t in range(1, n + 1):
    obs = rs.standard_normal(S).astype(np.float32)
    if absorbing: obs[-1] = 0.0
    out.append((obs, float(rs.standard_normal()), t in (7, 31, 52), t in (19, 44)))
  return out


@pytest.mark.gpu
@pytest.mark.parametrize('absorbing', [True, False])
@pytest.mark.parametrize('schedule', ['exact', 'fused', 'overlap'])
def test_acting_worker_matches_separate_calls(absorbing, schedule):
  """One launch per env step (mailbox in pinned memory, cursor on the device) against the per-function path: same actions (same
  Philox offsets), bit-identical ring including absorbing wraps and ring wrap-around, same host-side cursor / trajectory count."""
  actor_a, actor_b, mem_a, mem_b = _acting_pair(absorbing)
  S = mem_a.state_size
  rs = np.random.RandomState(5)
  script = _episode_script(rs, 60, S, absorbing)
  first = rs.standard_normal(S).astype(np.float32); first[-1] = 0.0 if absorbing else first[-1]
  resets = [rs.standard_normal(S).astype(np.float32) * 0.1 for _ in range(8)]
  if absorbing:
    for r in resets: r[-1] = 0.0

  # reference order with the per-function entry points
  acts_a, obs, k = [], torch.from_numpy(first).unsqueeze(0), 0
  for t, (nxt, rew, term, tout) in enumerate(script, 1):
    a = actor_a(obs).sample()
    acts_a.append(N(a))
    nxt_t = torch.from_numpy(nxt).unsqueeze(0)
    mem_a.append(t, obs, a.cpu(), rew, nxt_t, term, tout)
    if term or tout:
      if absorbing and term and not tout: mem_a.wrap_for_absorbing_states()
      obs = torch.from_numpy(resets[k]).unsqueeze(0); k += 1
    else:
      obs = nxt_t

  w = il.ActingWorker(actor_b, mem_b, mirror=schedule == 'overlap')
  acts_b, k = [], 0
  if schedule == 'exact':
    obs = first
    for t, (nxt, rew, term, tout) in enumerate(script, 1):
      acts_b.append(N(w.act(obs)))
      w.append(t, nxt, rew, term, tout)
      if term or tout: obs = resets[k]; k += 1
      else: obs = nxt
  elif schedule == 'fused':
    a = w.act(first)
    for t, (nxt, rew, term, tout) in enumerate(script, 1):
      acts_b.append(N(a))
      ended = term or tout
      a = w.step(t, nxt, rew, term, tout, obs=resets[k] if ended else None)
      k += int(ended)
  else:  # act on its own stream from the published snapshot (== the live parameters here: nothing updates them), appends on the main stream
    obs, a = first, w.act(first)
    for t, (nxt, rew, term, tout) in enumerate(script, 1):
      acts_b.append(N(a))
      w.post(t, obs, a, nxt, rew, term, tout)
      w.enqueue_append()
      if t % 3 == 0: w.enqueue_append()   # a replayed launch without a new post must append nothing
      ended = term or tout
      obs = resets[k] if ended else nxt
      k += int(ended)
      a = w.act(obs)
  torch.cuda.synchronize()
  np.testing.assert_array_equal(np.concatenate(acts_a), np.concatenate(acts_b))
  np.testing.assert_array_equal(N(mem_a.ring), N(mem_b.ring))
  assert (mem_a.idx, mem_a.full, mem_a.num_trajectories) == (mem_b.idx, mem_b.full, mem_b.num_trajectories)
  assert N(mem_b._ring_state).tolist() == [mem_b.idx, int(mem_b.full), mem_b.size]
  assert mem_a.full, 'the script is meant to wrap the ring'


@pytest.mark.gpu
def test_acting_worker_greedy_and_loud_failure():
  actor_a, actor_b, mem_a, mem_b = _acting_pair(True)
  obs = np.random.RandomState(1).standard_normal(mem_a.state_size).astype(np.float32)
  w = il.ActingWorker(actor_b, mem_b)
  np.testing.assert_array_equal(N(w.act(obs, greedy=True)), N(actor_a.get_greedy_action(torch.from_numpy(obs))))
  L = _lib.lib()
  assert L.il_act_step(None, 18, 6, 256, None, None, None, None, 0, 0, None, 0, None) != 0 and b'il_act_step' in L.il_last_error()


# ---------------------------------------------------------------------------------------------
# AdRIL / SQIL relabeller and batch mixing (models.py:287-318): bit-exact against the reference-generated fixture and the oracle
# ---------------------------------------------------------------------------------------------
def _packed(batch, S, A):
  """A ReplayMemory.sample-style dict (views into packed device rows) holding `batch`."""
  n = batch['rewards'].shape[0]
  rows = torch.zeros(n, int(_lib.lib().il_ring_row_floats(S, A)), device=DEV)
  views = il_memory.batch_views(rows, S, A, True)
  for k in il_memory.FIELDS:
    views[k].copy_(T(batch[k]))
  return views


@pytest.mark.gpu
@pytest.mark.parametrize('name,update_freq,balanced', [('adril_balanced', 1250, True), ('adril_halves', 1250, False), ('sqil_balanced', 0, True), ('sqil_halves', 0, False)])
def test_reward_relabeller_bit_exact(golden_dir, name, update_freq, balanced):
  from oracle import adril as oadril
  g = load(golden_dir, 'adril')
  S, A = gi.DIMS['hopper']
  rel, orel = il.RewardRelabeller(update_freq, balanced), oadril.RelabellerOracle(update_freq, balanced)
  for call in range(3):
    pol, exp = gi.adril_batches(40 + call, 64, S, A)
    tp, te = _packed(pol, S, A), _packed(exp, S, A)
    rel.resample_and_relabel(tp, te, gi.ADRIL_STEP + call * 700, gi.ADRIL_TRAJ + call, 7)
    orel.resample_and_relabel(pol, exp, gi.ADRIL_STEP + call * 700, gi.ADRIL_TRAJ + call, 7)
    for k in il_memory.FIELDS + ('absorbing',):
      got = N(tp[k])
      assert got.tobytes() == g[f'{name}.{call}.{k}'].tobytes(), (name, call, k)   # includes the sign of -0.0 rewards
      assert got.tobytes() == np.asarray(pol[k], np.float32).tobytes()


@pytest.mark.gpu
def test_mix_expert_agent_transitions_bit_exact(golden_dir):
  g = load(golden_dir, 'adril')
  S, A = gi.DIMS['hopper']
  pol, exp = gi.adril_batches(50, 64, S, A)
  tp, te = _packed(pol, S, A), _packed(exp, S, A)
  il.mix_expert_agent_transitions(tp, te)
  for k in il_memory.FIELDS + ('absorbing',):
    assert N(tp[k]).tobytes() == g[f'mix.{k}'].tobytes(), k
  with pytest.raises(TypeError):
    il.mix_expert_agent_transitions(tbatch(pol), tbatch(exp))   # loose tensors, not views into packed rows


# ---------------------------------------------------------------------------------------------
# RED (models.py:252-284, training.py:68-75) against the reference-generated fixture and the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name,kw', [(n, kw) for n, kw, _, _ in gi.RED_CASES], ids=[n for n, *_ in gi.RED_CASES])
def test_red_matches_reference(golden_dir, name, kw):
  """REDDiscriminator on the HIP path against the reference fixture and the oracle: conf/algorithm/RED.yaml (depth 1, relu) and the shapes of
  conf/optimised_hyperparameters/RED_*.yaml (depth 2 / tanh / predictor dropout, the reference's keep-masks fed back in)."""
  from oracle import red as ored
  g = load(golden_dir, 'red')
  c = gi.red_case(**kw)
  lr, wd = (float(x) for x in g[f'{name}.hyper'])
  icfg = Cfg(state_only=False, reward_bandwidth_scale=None,
             discriminator=Cfg(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], input_dropout=c['p_in'], dropout=c['p']))
  d = il.REDDiscriminator(c['S'], c['A'], icfg, device=DEV)
  # Sequential slots of the Linear layers: Dropout modules and activations occupy slots too (models.py:49-70)
  step = 2 + (c['p'] > 0)
  lin = [int(c['p_in'] > 0) + step * l for l in range(c['depth'] + 1)]
  assert set(d.state_dict()) == ({f'predictor.embedding.{i}.{p}' for i in lin for p in ('weight', 'bias')}
                                 | {f'target.embedding.{2 * l}.{p}' for l in range(c['depth'] + 1) for p in ('weight', 'bias')})
  d.flat.copy_(T(c['predictor'])); d.target_flat.copy_(T(c['target']))
  opt = il.AdamW(d, lr=lr, weight_decay=wd)
  st = ored.RedState(c['D'], c['H'], c['depth'], c['activation'], c['p_in'], c['p']); st.predictor[:] = c['predictor']; st.target[:] = c['target']
  tm = lambda masks: tuple(T(m) for m in masks) if masks else None
  for k, (b, masks) in enumerate(zip(c['batches'], c['masks']), 1):
    loss = il.target_estimation_update(d, tbatch(b), opt, want_loss=True, masks=tm(masks))
    oloss = ored.target_estimation_update(st, np.concatenate([b['states'], b['actions']], 1), b['weights'], lr=lr, weight_decay=wd, masks=masks)
    close(N(loss)[0], oloss, f'{name} loss {k}')
    close_params(N(d.flat), g[f'{name}.predictor.{k}'], f'{name} predictor after update {k} (reference)', lr, steps=k)
    close_params(N(d.flat), st.predictor, f'{name} predictor after update {k} (oracle)', lr, steps=k)
  close(N(opt.exp_avg), g[f'{name}.exp_avg'], f'{name} exp_avg', rtol=1e-4)
  assert int(opt.step_count[0]) == len(c['batches'])
  assert np.array_equal(N(d.target_flat), c['target']), 'the target network is frozen'
  # bandwidth + reward on the reference's post-training predictor (isolates the two calls from Adam-amplified differences)
  d.flat.copy_(T(g[f'{name}.predictor.{len(c["batches"])}']))
  e, q = tbatch(c['sigma_batch']), tbatch(c['query'])
  d.set_sigma(e['states'], e['actions'], masks=tm(c['sigma_masks']))   # still in train mode (train.py:128)
  assert abs(d.sigma_1 - float(g[f'{name}.sigma_1'][0])) <= 1e-5 * d.sigma_1
  d.eval()                                                              # train.py:147
  close(N(d.predict_reward(q['states'], q['actions'])), g[f'{name}.reward'], f'{name} reward', rtol=2e-5)   # ragged: B + 16 rows
  pred, targ = d(q['states'][:5], q['actions'][:5])
  x = np.concatenate([c['query']['states'][:5], c['query']['actions'][:5]], 1)
  st.predictor[:] = g[f'{name}.predictor.{len(c["batches"])}']
  op, ot, _ = ored.forward(st, x)
  close(N(pred), op, f'{name} predictor embedding'); close(N(targ), ot, f'{name} target embedding')
  if c['p'] > 0:   # on-chip masks: train-mode forwards differ from call to call and from the eval-mode forward, at the keep rate the config asks for
    d.train()
    p1, _ = d(q['states'], q['actions']); p2, _ = d(q['states'], q['actions'])
    assert not np.array_equal(N(p1), N(p2)) and np.isfinite(N(p1)).all()


@pytest.mark.gpu
def test_red_loud_failures():
  for bad in (dict(depth=3, activation='relu'), dict(depth=1, activation='sigmoid')):
    icfg = Cfg(state_only=False, reward_bandwidth_scale=None, discriminator=Cfg(hidden_size=32, input_dropout=0, dropout=0, **bad))
    with pytest.raises(NotImplementedError):
      il.REDDiscriminator(11, 3, icfg, device=DEV)
  L = _lib.lib()
  assert L.il_red_step(None, None, None, None, None, 0, None, 0, None) != 0 and b'il_red' in L.il_last_error()


# ---------------------------------------------------------------------------------------------
# DRIL dropout policy ensemble (models.py:84-120 + training.py:57-64) against the reference-generated fixture and the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name,kw', [(n, kw) for n, kw, _, _ in gi.DRIL_CASES], ids=[n for n, *_ in gi.DRIL_CASES])
def test_dril_matches_reference(golden_dir, name, kw):
  """The DRIL policy ensemble on the HIP path against the reference fixture and the oracle: conf/algorithm/DRIL.yaml (depth 1, tanh) and the shapes of
  conf/optimised_hyperparameters/DRIL_*.yaml (depth 2 / relu), the reference's keep-masks fed back in."""
  from oracle import dril as odril
  g = load(golden_dir, 'dril')
  c = gi.dril_case(**kw)
  lr, wd = (float(x) for x in g[f'{name}.hyper'])
  d = il.SoftActor(c['S'], c['A'], Cfg(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], input_dropout=c['p_in'], dropout=c['p']), device=DEV)
  lin = [1 + 3 * l for l in range(c['depth'] + 1)]
  assert type(d).__name__ == 'DropoutSoftActor' and list(d.state_dict()) == [f'actor.{i}.{p}' for i in lin for p in ('weight', 'bias')]
  d.flat.copy_(T(c['params']))
  opt = il.AdamW(d, lr=lr, weight_decay=wd)
  ds = odril.DrilState(c['S'], c['A'], c['H'], c['p_in'], c['p'], c['depth'], c['activation']); ds.params[:] = c['params']
  tm = lambda masks: tuple(T(m) for m in masks)
  for k, b in enumerate(c['batches'], 1):
    masks = gi.dril_masks(c, 'm', k - 1)
    loss = il.behavioural_cloning_update(d, tbatch(b), opt, masks=tm(masks))
    oloss = odril.bc_update(ds, b, *masks, lr=lr, weight_decay=wd)
    close(N(loss)[0], oloss, f'{name} BC loss {k}')
    close_params(N(d.flat), g[f'{name}.params.{k}'], f'{name} params after update {k} (reference)', lr, steps=k)
    close_params(N(d.flat), ds.params, f'{name} params after update {k} (oracle)', lr, steps=k)
  close(N(opt.exp_avg), g[f'{name}.exp_avg'], f'{name} exp_avg', rtol=1e-4)
  # uncertainty / threshold / reward on the reference's trained parameters. The variance of 5 nearly equal probabilities cancels
  # leading digits, so it is compared at 1e-4 of the largest value in the batch rather than 1e-5 per element.
  d.flat.copy_(T(g[f'{name}.params.{len(c["batches"])}']))
  e, q = tbatch(c['expert']), tbatch(c['query'])
  em, qm = tm(gi.dril_masks(c, 'e_m')), tm(gi.dril_masks(c, 'q_m'))
  ue = N(d._get_action_uncertainty(e['states'], e['actions'], masks=em))
  ref_ue = g[f'{name}.expert_uncertainty']
  assert np.abs(ue - ref_ue).max() <= 1e-4 * np.abs(ref_ue).max()
  f64 = load(golden_dir, 'f64_brackets')
  # (factor 4: the variance cancels the leading digits of five probabilities p = exp(log pi) ~ O(1), so both float32 results sit a few ulp OF p^2 from float64 -
  # 1e-7 absolute against a variance of 3e-2 - and the ratio between two libm's atanh / exp at that level is noise; the bound still scales with the reference's error)
  bracket(ue, ref_ue, f64[f'{name}.expert_uncertainty'], f'{name} expert uncertainty', factor=4.0)
  bracket(N(d._get_action_uncertainty(q['states'], q['actions'], masks=qm)), g[f'{name}.query_uncertainty'], f64[f'{name}.query_uncertainty'], f'{name} query uncertainty', factor=4.0)
  d.set_uncertainty_threshold(e['states'], e['actions'], 0.9, masks=em)
  assert abs(d.q - float(g[f'{name}.q'][0])) <= 1e-4 * max(abs(d.q), np.abs(ref_ue).max())
  d.q = float(g[f'{name}.q'][0])
  r = N(d.predict_reward(q['states'], q['actions'], masks=qm))   # 37 rows: ragged tiles
  ref_r, ref_u = g[f'{name}.reward'], g[f'{name}.query_uncertainty']
  decided = np.abs(ref_u - d.q) > 1e-4 * np.abs(ref_u).max()   # rows whose uncertainty is not within rounding of the threshold
  assert decided.sum() >= len(ref_r) - 2 and np.array_equal(r[decided], ref_r[decided]) and set(np.unique(r)) <= {-1.0, 1.0}


# imitation.discriminator of every conf/optimised_hyperparameters/{RED,DRIL}_*_trajectories.yaml of the reference (hidden, depth, activation, input_dropout, dropout)
SHIPPED_SHAPES = {'RED_5': (128, 2, 'relu', 0.0776, 0.3771), 'RED_10': (32, 1, 'tanh', 0.3861, 0.6857), 'RED_25': (64, 2, 'tanh', 0.0534, 0.4138),
                  'DRIL_5': (64, 1, 'tanh', 0.2122, 0.2069), 'DRIL_10': (32, 2, 'relu', 0.0906, 0.4719), 'DRIL_25': (32, 2, 'relu', 0.4033, 0.5565)}


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(SHIPPED_SHAPES))
def test_every_shipped_red_dril_shape_runs_at_ant_dims(name):
  """The tuned configurations the reference ships must all run on the HIP path, at the largest environment (Ant: the LDS-heaviest tiles), with on-chip masks
  and a ragged batch: two updates and the reward, everything finite and the parameters moving."""
  H, depth, act, p_in, p = SHIPPED_SHAPES[name]
  S, A = gi.DIMS['ant']
  rs = np.random.RandomState(5)
  b = gi.transitions(rs, 200, S, A, state_shift=0.5, weighted=True)
  b['actions'] = np.clip(b['actions'], -0.97, 0.97).astype(np.float32)
  mcfg = Cfg(hidden_size=H, depth=depth, activation=act, input_dropout=p_in, dropout=p)
  if name.startswith('RED'):
    d = il.REDDiscriminator(S, A, Cfg(state_only=False, reward_bandwidth_scale=None, discriminator=mcfg), device=DEV)
    opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
    before = N(d.flat).copy()
    for _ in range(2): il.target_estimation_update(d, tbatch(b), opt)
    d.set_sigma(T(b['states'][:64]), T(b['actions'][:64])); d.eval()
    r = N(d.predict_reward(T(b['states']), T(b['actions'])))
    assert ((r > 0) & (r <= 1)).all()
  else:
    d = il.SoftActor(S, A, mcfg, device=DEV)
    opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
    before = N(d.flat).copy()
    for _ in range(2): il.behavioural_cloning_update(d, tbatch(b), opt)
    d.set_uncertainty_threshold(T(b['states']), T(b['actions']), 0.9)
    r = N(d.predict_reward(T(b['states']), T(b['actions'])))
    assert set(np.unique(r)) <= {-1.0, 1.0}
  after = N(d.flat)
  assert np.isfinite(after).all() and np.isfinite(r).all() and not np.array_equal(before, after)


@pytest.mark.gpu
def test_dril_onchip_masks_are_bernoulli_and_change_per_call():
  c = gi.dril_case(71, 'hopper', 64, 64, 1)
  d = il.SoftActor(c['S'], c['A'], Cfg(hidden_size=64, depth=1, activation='tanh', input_dropout=0.1, dropout=0.1), device=DEV)
  d.flat.copy_(T(c['params']))
  e = tbatch(c['expert'])
  u1, u2 = N(d._get_action_uncertainty(e['states'], e['actions'])), N(d._get_action_uncertainty(e['states'], e['actions']))
  assert np.isfinite(u1).all() and (u1 >= 0).all() and (u1 > 0).mean() > 0.9 and not np.array_equal(u1, u2)
  nodrop = il.DropoutSoftActor(c['S'], c['A'], Cfg(hidden_size=64, depth=1, activation='tanh', input_dropout=0, dropout=0), device=DEV)
  nodrop.flat.copy_(T(c['params']))
  assert float(nodrop._get_action_uncertainty(e['states'], e['actions']).abs().max()) < 1e-12   # no dropout: the 5 members agree (up to the rounding of their mean)


# ---------------------------------------------------------------------------------------------
# GAIL loss variants (training.py:100-113) and subtract_log_policy (models.py:144,175) against the reference fixture and the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name,loss,sub', [('pugail', 'PUGAIL', False), ('mixup', 'Mixup', False), ('sublogp', 'BCE', True), ('mixup_sublogp', 'Mixup', True)])
def test_gail_loss_variants_match_reference(golden_dir, name, loss, sub):
  g = load(golden_dir, 'gail_variants')
  c = gi.gail_case(35, env='hopper', hidden=32, batch=96, steps=2)
  x = gi.gail_extras(35, c)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function=loss, grad_penalty=0.5, mixup_alpha=0.7, entropy_bonus=0.02, pos_class_prior=0.7, nonnegative_margin=float('inf'),
             discriminator=Cfg(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=False, subtract_log_policy=sub, reward_function='AIRL'))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  ods = ogail.DiscState(c['D'], c['H'], True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ods, k)[...] = c[k]
  d.flat.copy_(T(ods.pack()))
  for k, v in d.views().items():
    v.copy_(T(c[k]))
  actor = il.SoftActor(c['S'], c['A'], Cfg(hidden_size=64, depth=2, activation='relu'), device=DEV)
  actor.flat.copy_(T(x['actor']))
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
  for i in range(2):
    p, e = tbatch(c['policy'][i]), tbatch(c['expert'][i])
    if sub:
      close(N(actor.log_prob(p['states'], p['actions'])), g[f'{name}.logp_policy_{i + 1}'], 'log pi (policy batch)')
      close(N(actor.log_prob(e['states'], e['actions'])), g[f'{name}.logp_expert_{i + 1}'], 'log pi (expert batch)')
    il.adversarial_imitation_update(actor, d, p, e, opt, icfg, eps_gp=T(c['eps'][i]), eps_mix=T(x['eps_mix'][i]) if loss == 'Mixup' else None)
    close(N(opt.grad), g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1}', rtol=1e-5, atol_scale=1e-5)
    close_params(N(d.flat), g[f'{name}.p_{i + 1}'], f'{name} parameters {i + 1}', 1e-3, steps=i + 1)
    d.flat.copy_(T(g[f'{name}.p_{i + 1}']))   # continue from the reference's parameters: later steps then test one step each
    r = d.predict_reward(**il.make_gail_input(p['states'], p['actions'], p['next_states'], p['terminals'], actor, False, sub))
    close(N(r), g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=2e-5, atol_scale=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['clamped', 'open'])
def test_gail_pugail_finite_margin_matches_reference(golden_dir, name):
  """PUGAIL with a finite nonnegative_margin (training.py:100-102): the value pass + clamp decision of il_gail_disc_step against the reference fixture and the oracle,
  once with the unlabelled term clamped away and once with the gradient passing."""
  g = load(golden_dir, 'gail_pu_margin')
  c = gi.gail_case(37, env='halfcheetah', hidden=64, batch=128, steps=2)
  margin = float(g[f'{name}.margin'][0])
  d, ods, icfg = make_disc(c)
  icfg.update(loss_function='PUGAIL', grad_penalty=0.5, entropy_bonus=0.01, mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=margin)
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  for i in range(2):
    pb, eb = c['policy'][i], c['expert'][i]
    d.train()
    il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(c['eps'][i]))
    d.eval()
    ogr = ogail.gail_update(ods, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.5, entropy_bonus=0.01, return_grads=True,
                            loss_function='PUGAIL', pos_class_prior=0.7, nonnegative_margin=margin)
    close(N(opt.grad), g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1} (reference)', atol_scale=4e-6 * (i + 1)); close(N(opt.grad), ogr, f'{name} gradient {i + 1} (oracle)', atol_scale=4e-6 * (i + 1))
    close(N(d.flat), g[f'{name}.p_{i + 1}'], f'{name} parameters {i + 1}', atol_scale=4e-6 * (i + 1))
  assert int(opt.step_count[0]) == 2, 'the value pass must not tick the optimiser'
  plan, _ = _make_plan('GAIL', 3, loss='PUGAIL', margin=0.1)   # the captured plan runs the same per-function entry point inside (tests/test_update_plans_gpu.py)
  assert plan._variant and not plan.device_sync


@pytest.mark.gpu
def test_gail_variants_loud_failures():
  c = gi.gail_case(35, env='hopper', hidden=32, batch=96, steps=1)
  mk = lambda **kw: Cfg(state_only=False, spectral_norm=True, loss_function='PUGAIL', grad_penalty=0.5, mixup_alpha=1, entropy_bonus=0.0, pos_class_prior=0.7, nonnegative_margin=kw.get('margin', float('inf')),
                        discriminator=Cfg(hidden_size=32, depth=1, activation='relu', reward_shaping=kw.get('shaping', False), subtract_log_policy=False, reward_function='AIRL'))
  assert type(il.GAILDiscriminator(c['S'], c['A'], mk(shaping=True), 0.97, device=DEV)).__name__ == 'ShapedGAILDiscriminator'
  deep_shaping = mk(shaping=True); deep_shaping['discriminator'] = Cfg(deep_shaping['discriminator'], depth=2)   # a depth-2 / tanh potential: the general kernels
  assert type(il.GAILDiscriminator(c['S'], c['A'], deep_shaping, 0.97, device=DEV)).__name__ == 'ShapedDeepGAILDiscriminator'
  for bad in (dict(depth=3), dict(depth=2, hidden_size=192), dict(depth=1, activation='sigmoid')):   # what has no kernel raises at construction; there is no torch fallback to fall into
    cfg = mk(shaping=True); cfg['discriminator'] = Cfg(cfg['discriminator'], **bad)
    with pytest.raises(NotImplementedError):
      il.GAILDiscriminator(c['S'], c['A'], cfg, 0.97, device=DEV)


@pytest.mark.gpu
@pytest.mark.parametrize('loss,replays', [('BCE', 400), ('Mixup', 60), ('PUGAIL', 60)])
def test_device_handoff_equals_stream_dependencies(monkeypatch, loss, replays):
  """The two-graph update whose branches hand over through device counters (rows read from the rings, relabel inline) must evolve the learner bit for bit like
  the one-graph update with a fork / join on gathered rows and the stand-alone relabel kernel; no bounded wait may have timed out. Mixup (alpha = 1: on-chip
  U(0,1) coefficients, the loss of conf/optimised_hyperparameters/GAIL_{5,10}_trajectories.yaml) and PUGAIL run the same fused plan."""
  results = []
  for device_sync in ('1', '0'):
    monkeypatch.setenv('IL_DEVICE_SYNC', device_sync)
    il.seed(23)
    il_training._NOISE.clear()
    plan, nets = _make_plan('GAIL', 8, loss=loss, entropy_bonus=0.0 if loss == 'BCE' else 0.05)
    assert plan.device_sync == (device_sync == '1')
    plan.capture(warmup=0)
    for _ in range(replays):
      plan.replay()
    torch.cuda.synchronize()
    assert plan.sync_timeouts() == 0
    results.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.idx), N(plan.logp), N(plan.rewards)])
  for a, b in zip(*results):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('B,ring', [(128, True), (512, True), (1024, False)])
def test_update_plan_batch_sizes_of_the_tuned_configs(monkeypatch, B, ring):
  """training.batch_size of conf/optimised_hyperparameters/*.yaml is 128 ... 1024: up to 512 the chained launch is co-resident and the plan reads its rows through
  il_batch.gather; at 1024 (64 tiles x 6 workgroups > 256 CUs) it gathers first and keeps the separate kernels. Either way the device hand-off must equal the
  stream-dependency schedule bit for bit."""
  results = []
  for device_sync in ('1', '0'):
    monkeypatch.setenv('IL_DEVICE_SYNC', device_sync)
    il.seed(29); il_training._NOISE.clear()
    plan, nets = _make_plan('GAIL', 10, B=B)
    if device_sync == '1': assert plan.device_sync and plan.ring_mode == ring and plan.inline_relabel == ring
    for _ in range(6): plan.run()
    torch.cuda.synchronize()
    assert plan.sync_timeouts() == 0
    results.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.logp), N(plan.rewards)])
  for a, b in zip(*results):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('switch,B', [('IL_RING_GATHER', 256), ('IL_INLINE_RELABEL', 256), ('IL_SAC_CHAIN', 256), ('IL_PC_SPLIT', 256), ('IL_RESIDENT_SAMPLER', 256), ('IL_RESIDENT_SAMPLER', 80),
                                      ('IL_SAC_CHAIN', 80), ('IL_RING_GATHER', 80), ('IL_PC_SPLIT', 48), ('IL_CHAIN_XCD_NETS', 256), ('IL_CHAIN_XCD_NETS', 80),
                                      ('IL_PAIR', 256), ('IL_PAIR', 128), ('IL_PAIR', 80), ('IL_STAGE_ROWS', 256), ('IL_STAGE_ROWS', 80), ('IL_EARLY_DRAW', 256), ('IL_EARLY_DRAW', 48),
                                      ('IL_MAIN_OVERLAP', 256), ('IL_MAIN_OVERLAP', 128)])   # IL_MAIN_OVERLAP (round 6): the SAC branch's four launches alternating over two streams (il_sac_update_gather_overlap) against in-order launches   # IL_PAIR: the column-split pairs of k_sac_chain_pair / k_policy_critic_pair against the 16-wave workgroups   # 80 / 48 rows: 5 / 3 tiles, the non-XCD-aware role decode; IL_CHAIN_XCD_NETS: one network per XCD (off by default)
def test_schedule_switches_are_bit_identical(monkeypatch, switch, B):
  """Every schedule of the update (rows through il_batch.gather vs a gather kernel, inline relabel vs k_gail_reward, chained vs separate forward / critic-loss
  launches, helper-split vs second-arriver policy tail) runs the same arithmetic per element: switching one off must not change a bit.
  The C-side switches are read once per process, so they are compared through a subprocess."""
  import subprocess, sys, json
  code = (
      "import sys, json, hashlib, numpy as np, torch; sys.path[:0] = ['.', 'tests', 'tests/golden']\n"
      "import imitation_learning_amd as il\n"
      "from imitation_learning_amd import training as T\n"
      "from test_gpu_parity import _make_plan, N\n"
      "il.seed(31); T._NOISE.clear()\n"
      f"plan, nets = _make_plan('GAIL', 17, B={B})\n"
      "for _ in range(6): plan.run()\n"
      "torch.cuda.synchronize()\n"
      "assert plan.sync_timeouts() == 0\n"
      "h = hashlib.sha256()\n"
      "for n in list(nets) + [plan.logp, plan.q, plan.rewards, plan.idx]: h.update(np.ascontiguousarray(N(n.flat if hasattr(n, 'flat') else n)).tobytes())\n"
      "print(json.dumps(dict(digest=h.hexdigest(), ring=plan.ring_mode, inline=plan.inline_relabel, staged=plan.staged_rows, overlap=bool(plan._ov_active), poisoned=plan.poisoned())))\n")
  outs = []
  for value in ('1', '0'):
    env = dict(os.environ, **{switch: value})
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
  if switch == 'IL_STAGE_ROWS': assert outs[0]['staged'] and not outs[1]['staged']
  if switch == 'IL_MAIN_OVERLAP': assert outs[0]['overlap'] and not outs[1]['overlap'], 'the overlapped launches did not run (no third hardware queue?)'
  assert not outs[0]['poisoned'] and not outs[1]['poisoned']
  assert outs[0]['ring'] and outs[0]['inline'], 'the default schedule reads rows through il_batch.gather and relabels inline'
  if switch == 'IL_RING_GATHER': assert not outs[1]['ring']
  if switch == 'IL_INLINE_RELABEL': assert outs[1]['ring'] and not outs[1]['inline']
  assert outs[0]['digest'] == outs[1]['digest']


@pytest.mark.gpu
@pytest.mark.parametrize('switch,values', [('IL_POP_SPLIT_TAIL', ('1', '0')), ('IL_POP_DISC_TPW', ('4', '1')), ('IL_POP_DISC_TPW', ('3', '16')), ('IL_POP_FUSE_POLYAK', ('1', '0'))])
def test_population_switches_are_bit_identical(switch, values):
  """The round-4 forms of the population launches - the policy backward as a launch of its own (k_actor_bwd_pop) instead of the tail of each tile's second critic
  workgroup, several tiles per workgroup of k_gail_grad_pop (3: a ragged last workgroup; 16: a whole call in one workgroup) - only move work between workgroups:
  three learners after two population updates must not differ in a bit from the other setting (and, by test_batched_population_equals_independent_learners, from
  independent learners). IL_POP_FUSE_POLYAK (round 5): the target step of the critics' H x H layers inside their optimiser blocks instead of the actor launch's tail. The
  switches are read once per process: compared through subprocesses."""
  import subprocess, sys, json
  code = (
      "import sys, json, hashlib, numpy as np, torch; sys.path[:0] = ['.', 'tests', 'tests/golden']\n"
      "import imitation_learning_amd as il\n"
      "from test_gpu_parity import _population_learners, _population_state\n"
      "plans, nets_all = _population_learners(3)\n"
      "pop = il.BatchedPopulationPlan(plans)\n"
      "for _ in range(2): pop.run()\n"
      "torch.cuda.synchronize()\n"
      "h = hashlib.sha256()\n"
      "for learner in _population_state(plans, nets_all):\n"
      "  for a in learner: h.update(np.ascontiguousarray(a).tobytes())\n"
      "print(json.dumps(dict(digest=h.hexdigest())))\n")
  outs = []
  for value in values:
    env = dict(os.environ, **{switch: value})
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
  assert outs[0]['digest'] == outs[1]['digest']


@pytest.mark.gpu
def test_expert_data_ingest_on_device(tmp_path):
  """SURVEY.md 8 f3 on the GPU (reference environments.py:63-125 ends in a ReplayMemory): `dataset_to_memory(..., device='cuda')` for the 8 (absorbing, subsample,
  trajectories) cases of tests/golden/dataset.npz - every field of the DEVICE-resident ring bit-equal to what D4RLEnv.get_dataset built from the same raw arrays, the
  bookkeeping (num_trajectories, idx, full, len) equal, and 4,096 rows drawn from it (`sample`: host draw + il_replay_gather; `sample_device`: device draw + gather)
  equal to the fixture's rows at the drawn indices."""
  from imitation_learning_amd import environments
  g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dataset.npz'))
  raw = gi.raw_d4rl_dataset(81)
  path = str(tmp_path / 'expert.npz')
  np.savez(path, **raw)
  fields = ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights', 'step')
  for absorbing in (True, False):
    for subsample in (1, 3):
      for trajectories in (0, 2):
        for source in (raw, environments.load_dataset_file(path)):
          il.seed(17)
          mem = environments.dataset_to_memory(source, absorbing, trajectories, subsample, device=DEV)
          assert mem.ring.device.type == torch.device(DEV).type
          tag = f'abs{int(absorbing)}_sub{subsample}_traj{trajectories}'
          for k in fields:
            assert N(getattr(mem, k)).tobytes() == g[f'{tag}.{k}'].tobytes(), (tag, k)
          assert [mem.num_trajectories, mem.idx, int(mem.full), len(mem)] == g[f'{tag}.meta'].tolist(), tag
        n = 4096
        idx = mem._sample_idx_tensor(n)
        hi = mem.size if mem.full else mem.idx - 1
        assert int(idx.min()) >= 0 and int(idx.max()) < hi
        got, ix = mem.gather(idx), N(idx).astype(np.int64)
        views = il_memory.batch_views(got, mem.state_size, mem.action_size, mem.absorbing)
        for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'weights'):
          want = g[f'{tag}.{k}'][ix]
          assert N(views[k]).reshape(want.shape).tobytes() == want.tobytes(), (tag, k, 'sample')
        didx, drows = torch.empty(n, dtype=torch.int32, device=DEV), torch.empty(n, mem.row, device=DEV)
        dv = mem.sample_device(n, didx, drows)
        torch.cuda.synchronize()
        dix = N(didx).astype(np.int64)
        assert dix.min() >= 0 and dix.max() < hi
        for k in ('states', 'actions', 'next_states', 'weights'):
          want = g[f'{tag}.{k}'][dix]
          assert N(dv[k]).reshape(want.shape).tobytes() == want.tobytes(), (tag, k, 'sample_device')


@pytest.mark.gpu
def test_pair_mode_hops_stay_bit_identical_over_many_replays():
  """k_sac_chain_pair / k_policy_critic_pair hand 16 x 128 halves between workgroups through L2 (same XCD: plain stores behind a drained flag) or write-through stores
  (any placement), read with L1-bypassing loads - no fence. A stale or torn hop would change a hidden activation and, within an update, every parameter: 1,500 captured
  replays back to back (the timed regime: launches overlap, consumers L1-warm from the previous replay) must end in the bits of the 16-wave workgroups (IL_PAIR=0)."""
  import subprocess, sys, json
  code = (
      "import sys, json, hashlib, numpy as np, torch; sys.path[:0] = ['.', 'tests', 'tests/golden']\n"
      "import imitation_learning_amd as il\n"
      "from imitation_learning_amd import training as T\n"
      "from test_gpu_parity import _make_plan, N\n"
      "il.seed(41); T._NOISE.clear()\n"
      "plan, nets = _make_plan('GAIL', 23)\n"
      "plan.capture(warmup=2)\n"
      "for _ in range(1500): plan.replay()\n"
      "torch.cuda.synchronize()\n"
      "assert plan.sync_timeouts() == 0\n"
      "h = hashlib.sha256()\n"
      "for n in list(nets) + [plan.logp, plan.q, plan.rewards, plan.idx]: h.update(np.ascontiguousarray(N(n.flat if hasattr(n, 'flat') else n)).tobytes())\n"
      "print(json.dumps(dict(digest=h.hexdigest(), finite=bool(np.isfinite(N(nets[0].flat)).all()))))\n")
  outs = []
  for value in ('1', '0'):
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, IL_PAIR=value), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
  assert outs[0]['finite'] and outs[0]['digest'] == outs[1]['digest']


@pytest.mark.gpu
def test_batch_gather_is_rejected_where_it_is_not_honoured():
  """il_batch.gather silently ignored would read the wrong rows: every entry point that indexes batch rows directly must refuse it."""
  from imitation_learning_amd import _lib
  plan, nets = _make_plan('GAIL', 19)
  plan.run(); torch.cuda.synchronize()
  ring_batch = plan._ring_batches()[0]
  rc = _lib.lib().il_sac_update(C.byref(plan.sac), C.byref(ring_batch), None, None, _lib.ptr(plan.logp), _lib.ptr(plan.q), 0, _lib.stream_ptr())
  assert rc != 0 and b'gather' in _lib.lib().il_last_error()
  with pytest.raises(RuntimeError, match='gather'):
    _lib.check(_lib.lib().il_sac_actor_step(C.byref(plan.sac), C.byref(ring_batch), None, _lib.ptr(plan.logp), _lib.ptr(plan.q), 0, _lib.stream_ptr()))
  with pytest.raises(RuntimeError, match='gather'):   # and il_sac_update_gather insists on it for `ring`
    _lib.check(_lib.lib().il_sac_update_gather(C.byref(plan.sac), C.byref(plan.pb), C.byref(plan.pb), None, None, None, None, None, _lib.ptr(plan.logp), _lib.ptr(plan.q), 0, _lib.stream_ptr()))


# ---------------------------------------------------------------------------------------------
# GAIL discriminators of depth 1-2 with relu / tanh (gail_deep.hip) against the reference fixture and the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', [n for n, *_ in gi.GAIL_DEEP_CASES])
def test_gail_deep_discriminator_matches_reference(golden_dir, name):
  from oracle import gail_deep as ogd
  g, f64 = load(golden_dir, 'gail_deep'), load(golden_dir, 'f64_brackets')
  _, kw, loss, (lr, wd, gp, ent), rf = next(c for c in gi.GAIL_DEEP_CASES if c[0] == name)
  c = gi.gail_deep_case(**kw)
  icfg = Cfg(state_only=False, spectral_norm=c['spectral_norm'], loss_function=loss, grad_penalty=gp, mixup_alpha=0.7, entropy_bonus=ent, pos_class_prior=0.7, nonnegative_margin=float('inf'),
             discriminator=Cfg(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], reward_shaping=False, subtract_log_policy=False, reward_function=rf))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  assert type(d).__name__ == 'DeepGAILDiscriminator' and list(d.state_dict())[:2] == [str(x) for x in g[f'{name}.param_names'][:2]]
  ds = ogd.DeepDiscState(c['D'], c['H'], c['depth'], c['activation'], c['spectral_norm'])
  for l in range(c['depth'] + 1):
    ds.W[l][...] = c['W'][l]; ds.b[l][...] = c['b'][l]; ds.u[l][...] = c['u'][l]; ds.v[l][...] = c['v'][l]
  d.flat.copy_(T(ds.pack()))
  if c['spectral_norm']: d.sn.copy_(T(ds.pack_sn()))
  opt = il.AdamW(d, lr=lr, weight_decay=wd)
  cat = lambda b: np.concatenate([b['states'], b['actions']], 1)
  for i in range(len(c['policy'])):
    pb, eb = c['policy'][i], c['expert'][i]
    if i:   # every step starts from the reference's state (isolates the step from Adam-amplified differences)
      d.flat.copy_(T(g[f'{name}.p_{i}'])); ds.unpack_into(g[f'{name}.p_{i}'])
      if c['spectral_norm']: d.sn.copy_(T(g[f'{name}.sn_{i}'])); ds.unpack_sn(g[f'{name}.sn_{i}'])
    il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(c['eps'][i]), eps_mix=T(c['eps_mix'][i]))
    ogr = ogd.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][i], lr=lr, weight_decay=wd, grad_penalty=gp, entropy_bonus=ent, return_grads=True,
                          loss_function=loss, eps_mix=c['eps_mix'][i])
    close(N(opt.grad), g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1} (reference)', rtol=2e-5, atol_scale=1e-5)
    close(N(opt.grad), ogr, f'{name} gradient {i + 1} (oracle)', rtol=2e-5, atol_scale=1e-5)
    bracket(N(opt.grad), g[f'{name}.g_{i + 1}'], f64[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1}')   # the same update from the same float32 state, evaluated by the reference in float64
    if c['spectral_norm']:
      close(N(d.sn), g[f'{name}.sn_{i + 1}'], f'{name} u / v after update {i + 1}', rtol=2e-5, atol_scale=1e-5)
    d.flat.copy_(T(g[f'{name}.p_{i + 1}']))
    r = d.predict_reward(T(pb['states']), T(pb['actions']))
    close(N(r), g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=5e-5, atol_scale=1e-5)
    bracket(N(r), g[f'{name}.reward_{i + 1}'], f64[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}')
  assert int(opt.step_count[0]) == len(c['policy'])


# ---------------------------------------------------------------------------------------------
# GAIL with reward shaping (models.py:152-180, reward_shaping=True) against the reference fixture and the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name,sn,loss', [('sn_bce', True, 'BCE'), ('plain_pugail', False, 'PUGAIL')])
def test_gail_reward_shaping_matches_reference(golden_dir, name, sn, loss):
  from oracle import gail_shaped as ogs
  g = load(golden_dir, 'gail_shaped')
  c = gi.gail_shaped_case(91, 'hopper', 32, 96, 2, sn)
  icfg = Cfg(state_only=False, spectral_norm=sn, loss_function=loss, grad_penalty=0.7, mixup_alpha=1, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=float('inf'),
             discriminator=Cfg(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  assert type(d).__name__ == 'ShapedGAILDiscriminator'
  assert [n for n, _ in d.named_parameters()] == list(g[f'{name}.param_names'])
  ods = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, sn)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ods, k)[...] = c[k]
  d.flat.copy_(T(ods.pack()))
  if sn:
    for k, v in d.views().items():
      v.copy_(T(c[k]))
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
  for i in range(2):
    p, e = tbatch(c['policy'][i]), tbatch(c['expert'][i])
    il.adversarial_imitation_update(None, d, p, e, opt, icfg, eps_gp=T(c['eps'][i]))
    ogr = ogs.gail_update(ods, c['policy'][i], c['expert'][i], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function=loss, return_grads=True)
    close(N(opt.grad), g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1} (reference)', rtol=1e-5, atol_scale=1e-5)
    close(N(opt.grad), ogr, f'{name} gradient {i + 1} (oracle)', rtol=1e-5, atol_scale=1e-5)
    close_params(N(d.flat), g[f'{name}.p_{i + 1}'], f'{name} parameters {i + 1}', 1e-3, steps=i + 1)
    if sn:
      for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
        close(N(d.views()[k]), g[f'{name}.{k}_{i + 1}'], f'{name} {k} after update {i + 1}', rtol=1e-5, atol_scale=1e-5)
    d.flat.copy_(T(g[f'{name}.p_{i + 1}'])); ods.unpack_into(g[f'{name}.p_{i + 1}'].copy())
    r = d.predict_reward(**il.make_gail_input(p['states'], p['actions'], p['next_states'], p['terminals'], None, True, False))
    close(N(r), g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=2e-5, atol_scale=1e-5)


@pytest.mark.gpu
def test_gail_reward_shaping_mixup_matches_reference(golden_dir):
  """Reward shaping under loss_function = Mixup (one call on the convex combination of every field, fractional terminals), spectral norm, penalty, entropy bonus."""
  from oracle import gail_shaped as ogs
  g = load(golden_dir, 'gail_shaped_mixup')
  c = gi.gail_shaped_case(95, 'hopper', 32, 96, 2, True)
  em = gi.mixup_draws(1095, 96, 2)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function='Mixup', grad_penalty=0.7, mixup_alpha=0.7, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=float('inf'),
             discriminator=Cfg(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  assert type(d).__name__ == 'ShapedGAILDiscriminator'
  ods = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, True)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ods, k)[...] = c[k]
  d.flat.copy_(T(ods.pack()))
  for k, v in d.views().items():
    v.copy_(T(c[k]))
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
  for i in range(2):
    p, e = tbatch(c['policy'][i]), tbatch(c['expert'][i])
    il.adversarial_imitation_update(None, d, p, e, opt, icfg, eps_gp=T(c['eps'][i]), eps_mix=T(em[i]))
    ogr = ogs.gail_update(ods, c['policy'][i], c['expert'][i], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function='Mixup', return_grads=True,
                          eps_mix=em[i])
    close(N(opt.grad), g[f'g_{i + 1}'], f'shaped mixup gradient {i + 1} (reference)', rtol=1e-5, atol_scale=1e-5)
    close(N(opt.grad), ogr, f'shaped mixup gradient {i + 1} (oracle)', rtol=1e-5, atol_scale=1e-5)
    close_params(N(d.flat), g[f'p_{i + 1}'], f'shaped mixup parameters {i + 1}', 1e-3, steps=i + 1)
    for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
      close(N(d.views()[k]), g[f'{k}_{i + 1}'], f'shaped mixup {k} after update {i + 1}', rtol=1e-5, atol_scale=1e-5)
    d.flat.copy_(T(g[f'p_{i + 1}'])); ods.unpack_into(g[f'p_{i + 1}'].copy())
    r = d.predict_reward(**il.make_gail_input(p['states'], p['actions'], p['next_states'], p['terminals'], None, True, False))
    close(N(r), g[f'reward_{i + 1}'], f'shaped mixup reward {i + 1}', rtol=2e-5, atol_scale=1e-5)
  # alpha = 1 without given draws: the coefficients come from the on-chip Philox stream (IL_STREAM_MIX), the update must run and change the parameters
  icfg1 = Cfg(icfg); icfg1['mixup_alpha'] = 1
  before = N(d.flat)
  il.adversarial_imitation_update(None, d, tbatch(c['policy'][0]), tbatch(c['expert'][0]), opt, icfg1)
  assert np.isfinite(N(d.flat)).all() and not np.array_equal(before, N(d.flat))


# ---------------------------------------------------------------------------------------------
# GAIL with reward shaping and a depth 1-2 / relu / tanh potential (gail_shaped_deep.hip) against the reference fixture and the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', [n for n, *_ in gi.GAIL_SHAPED_DEEP_CASES])
def test_gail_reward_shaping_general_potential_matches_reference(golden_dir, name):
  """BCE / PUGAIL (infinite margin, and a finite one that binds) / Mixup with fractional terminals, spectral norm on and off, state_only, the three reward heads, a
  log-policy offset: gradients, parameters, u / v buffers and rewards of `adversarial_imitation_update` + `predict_reward` on the reference's own outputs."""
  from oracle import gail_shaped_deep as osd
  from test_oracle_golden import _shaped_deep_state
  g = load(golden_dir, 'gail_shaped_deep')
  _, kw, loss, (lr, wd, gp, ent), rf, margin = next(c for c in gi.GAIL_SHAPED_DEEP_CASES if c[0] == name)
  c = gi.gail_shaped_deep_case(**kw)
  icfg = Cfg(state_only=c['state_only'], spectral_norm=c['spectral_norm'], loss_function=loss, grad_penalty=gp, mixup_alpha=0.7, entropy_bonus=ent, pos_class_prior=0.7, nonnegative_margin=margin,
             discriminator=Cfg(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], reward_shaping=True, subtract_log_policy=False, reward_function=rf))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  assert type(d).__name__ == 'ShapedDeepGAILDiscriminator' and [n for n, _ in d.named_parameters()] == list(g[f'{name}.param_names'])
  ods = _shaped_deep_state(c)
  d.flat.copy_(T(ods.pack()))
  if c['spectral_norm']: d.sn.copy_(T(ods.pack_sn()))
  opt = il.AdamW(d, lr=lr, weight_decay=wd)
  for i in range(len(c['policy'])):
    pb, eb = c['policy'][i], c['expert'][i]
    if i:   # every step starts from the reference's state (isolates the step from Adam-amplified differences)
      d.flat.copy_(T(g[f'{name}.p_{i}'])); ods.unpack_into(g[f'{name}.p_{i}'].copy())
      if c['spectral_norm']: d.sn.copy_(T(g[f'{name}.sn_{i}'])); ods.unpack_sn(g[f'{name}.sn_{i}'].copy())
    il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(c['eps'][i]), eps_mix=T(c['eps_mix'][i]))
    ogr = osd.gail_update(ods, pb, eb, c['eps'][i], lr=lr, weight_decay=wd, grad_penalty=gp, entropy_bonus=ent, return_grads=True, loss_function=loss, pos_class_prior=0.7,
                          nonnegative_margin=margin, eps_mix=c['eps_mix'][i])
    close(N(opt.grad), g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1} (reference)', rtol=2e-5, atol_scale=1e-5)
    close(N(opt.grad), ogr, f'{name} gradient {i + 1} (oracle)', rtol=2e-5, atol_scale=1e-5)
    close_params(N(d.flat), g[f'{name}.p_{i + 1}'], f'{name} parameters {i + 1}', lr, steps=1, outlier_frac=2e-3)   # <= 3 of ~1,500 elements (one would already be 6.6e-4)
    if c['spectral_norm']:
      close(N(d.sn), g[f'{name}.sn_{i + 1}'], f'{name} u / v after update {i + 1}', rtol=2e-5, atol_scale=1e-5)
    d.flat.copy_(T(g[f'{name}.p_{i + 1}']))
    p = tbatch(pb)
    r = d.predict_reward(**il.make_gail_input(p['states'], p['actions'], p['next_states'], p['terminals'], None, True, False))
    close(N(r), g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=5e-5, atol_scale=1e-5)
    d.subtract_log_policy = True
    r = d.predict_reward(p['states'], p['actions'], p['next_states'], p['terminals'], log_policy=T(c['logp_policy'][i]))
    d.subtract_log_policy = False
    close(N(r), g[f'{name}.reward_logp_{i + 1}'], f'{name} reward with a log-policy offset {i + 1}', rtol=5e-5, atol_scale=1e-5)
  assert int(opt.step_count[0]) == len(c['policy'])   # the PUGAIL value pass does not tick the optimiser


@pytest.mark.gpu
def test_gail_reward_shaping_general_kernels_equal_the_depth1_relu_kernels(golden_dir, monkeypatch):
  """IL_SHAPED_GENERAL=1 sends the default potential through gail_shaped_deep.hip: the two implementations of the same update agree with the reference fixture and with
  each other (gradients to 2e-5 of the scale: different tile heights and summation orders), and on the on-chip draws (the same Philox streams)."""
  g = load(golden_dir, 'gail_shaped')
  c = gi.gail_shaped_case(91, 'hopper', 32, 96, 2, True)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=0.7, mixup_alpha=1, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=float('inf'),
             discriminator=Cfg(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
  from oracle import gail_shaped as ogs
  ods = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, True)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2'):
    getattr(ods, k)[...] = c[k]
  sn0 = np.concatenate([c[k] for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2')])
  grads = {}
  for general in ('0', '1'):
    monkeypatch.setenv('IL_SHAPED_GENERAL', general)
    il.seed(41); il_training._NOISE.clear()
    d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
    assert type(d).__name__ == ('ShapedDeepGAILDiscriminator' if general == '1' else 'ShapedGAILDiscriminator')
    d.flat.copy_(T(ods.pack())); d.sn.copy_(T(sn0))
    opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
    il.adversarial_imitation_update(None, d, tbatch(c['policy'][0]), tbatch(c['expert'][0]), opt, icfg, eps_gp=T(c['eps'][0]))
    close(N(opt.grad), g['sn_bce.g_1'], f'general={general} gradient (reference)', rtol=2e-5, atol_scale=1e-5)
    close(N(d.sn), np.concatenate([g[f'sn_bce.{k}_1'] for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2')]), f'general={general} u / v', rtol=2e-5, atol_scale=1e-5)
    il.adversarial_imitation_update(None, d, tbatch(c['policy'][1]), tbatch(c['expert'][1]), opt, icfg)   # on-chip gradient-penalty draws
    grads[general] = N(opt.grad)
  close(grads['1'], grads['0'], 'general vs depth-1 kernels, on-chip draws', rtol=5e-5, atol_scale=2e-5)


# ---------------------------------------------------------------------------------------------
# PUGAIL with a finite nonnegative_margin (training.py:100-102) on the depth-2 / tanh and the reward-shaping discriminator
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', ['clamped', 'open'])
def test_gail_deep_pugail_finite_margin_matches_reference(golden_dir, name):
  from oracle import gail_deep as ogd
  g = load(golden_dir, 'gail_pu_margin_general')
  margin = float(g[f'deep.{name}.margin'][0])
  c = gi.gail_deep_case(seed=111, env='hopper', hidden=32, batch=96, steps=2, depth=2, activation='tanh', spectral_norm=True)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function='PUGAIL', grad_penalty=0.6, mixup_alpha=1, entropy_bonus=0.02, pos_class_prior=0.7, nonnegative_margin=margin,
             discriminator=Cfg(hidden_size=c['H'], depth=2, activation='tanh', reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  assert type(d).__name__ == 'DeepGAILDiscriminator'
  ds = ogd.DeepDiscState(c['D'], c['H'], 2, 'tanh', True)
  for l in range(3):
    ds.W[l][...] = c['W'][l]; ds.b[l][...] = c['b'][l]; ds.u[l][...] = c['u'][l]; ds.v[l][...] = c['v'][l]
  d.flat.copy_(T(ds.pack())); d.sn.copy_(T(ds.pack_sn()))
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
  cat = lambda b: np.concatenate([b['states'], b['actions']], 1)
  for i in range(2):
    pb, eb = c['policy'][i], c['expert'][i]
    if i:
      d.flat.copy_(T(g[f'deep.{name}.p_{i}'])); ds.unpack_into(g[f'deep.{name}.p_{i}'])
      d.sn.copy_(T(g[f'deep.{name}.sn_{i}'])); ds.unpack_sn(g[f'deep.{name}.sn_{i}'])
    il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(c['eps'][i]))
    ogr = ogd.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.6, entropy_bonus=0.02, return_grads=True,
                          loss_function='PUGAIL', pos_class_prior=0.7, nonnegative_margin=margin)
    close(N(opt.grad), g[f'deep.{name}.g_{i + 1}'], f'deep {name} gradient {i + 1} (reference)', rtol=2e-5, atol_scale=1e-5)
    close(N(opt.grad), ogr, f'deep {name} gradient {i + 1} (oracle)', rtol=2e-5, atol_scale=1e-5)
    close(N(d.sn), g[f'deep.{name}.sn_{i + 1}'], f'deep {name} u / v after update {i + 1}', rtol=2e-5, atol_scale=1e-5)
  assert int(opt.step_count[0]) == 2   # the value pass does not tick the optimiser
  # the two margins give different gradients at the first update (one side of the clamp each), so a kernel that ignored the margin fails one of the two cases
  assert np.abs(g['deep.clamped.g_1'] - g['deep.open.g_1']).max() > 1e-3 * np.abs(g['deep.open.g_1']).max()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['clamped', 'open'])
def test_gail_shaped_pugail_finite_margin_matches_reference(golden_dir, name):
  from oracle import gail_shaped as ogs
  g = load(golden_dir, 'gail_pu_margin_general')
  margin = float(g[f'shaped.{name}.margin'][0])
  c = gi.gail_shaped_case(93, 'hopper', 32, 96, 2, True)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function='PUGAIL', grad_penalty=0.7, mixup_alpha=1, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=margin,
             discriminator=Cfg(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device=DEV)
  assert type(d).__name__ == 'ShapedGAILDiscriminator'
  ods = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, True)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ods, k)[...] = c[k]
  d.flat.copy_(T(ods.pack()))
  for k, v in d.views().items():
    v.copy_(T(c[k]))
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.1)
  for i in range(2):   # the second update of 'clamped' is on the open side of the clamp (value_2 > -margin): the decision is taken per update, on the device
    il.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg, eps_gp=T(c['eps'][i]))
    ogr = ogs.gail_update(ods, c['policy'][i], c['expert'][i], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function='PUGAIL', return_grads=True,
                          pos_class_prior=0.7, nonnegative_margin=margin)
    close(N(opt.grad), g[f'shaped.{name}.g_{i + 1}'], f'shaped {name} gradient {i + 1} (reference)', rtol=1e-5, atol_scale=1e-5)
    close(N(opt.grad), ogr, f'shaped {name} gradient {i + 1} (oracle)', rtol=1e-5, atol_scale=1e-5)
    close_params(N(d.flat), g[f'shaped.{name}.p_{i + 1}'], f'shaped {name} parameters {i + 1}', 1e-3, steps=i + 1)
    for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
      close(N(d.views()[k]), g[f'shaped.{name}.{k}_{i + 1}'], f'shaped {name} {k} after update {i + 1}', rtol=1e-5, atol_scale=1e-5)
    d.flat.copy_(T(g[f'shaped.{name}.p_{i + 1}'])); ods.unpack_into(g[f'shaped.{name}.p_{i + 1}'].copy())
  assert int(opt.step_count[0]) == 2
  assert np.abs(g['shaped.clamped.g_1'] - g['shaped.open.g_1']).max() > 1e-3 * np.abs(g['shaped.open.g_1']).max()


# ---------------------------------------------------------------------------------------------
# the small-network kernels at the largest environment (Ant: S = 112, A = 8) and ragged batches, against the oracle (no reference fixture at these sizes)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_red_dril_shaped_at_ant_dims_match_oracle():
  from oracle import dril as odril, gail_shaped as ogs, red as ored
  # RED, hidden 64, ragged batch of 100
  c = gi.red_case(63, 'ant', 64, 100, 2)
  icfg = Cfg(state_only=False, reward_bandwidth_scale=None, discriminator=Cfg(hidden_size=64, depth=1, activation='relu', input_dropout=0, dropout=0))
  d = il.REDDiscriminator(c['S'], c['A'], icfg, device=DEV)
  d.flat.copy_(T(c['predictor'])); d.target_flat.copy_(T(c['target']))
  opt = il.AdamW(d, lr=1e-3, weight_decay=0.0)
  st = ored.RedState(c['D'], c['H']); st.predictor[:] = c['predictor']; st.target[:] = c['target']
  for k, b in enumerate(c['batches'], 1):
    il.target_estimation_update(d, tbatch(b), opt)
    ored.target_estimation_update(st, np.concatenate([b['states'], b['actions']], 1), b['weights'], lr=1e-3, weight_decay=0.0)
    close_params(N(d.flat), st.predictor, f'RED ant predictor {k}', 1e-3, steps=k)
  # DRIL, hidden 64, ragged batch of 80
  c = gi.dril_case(73, 'ant', 64, 80, 2)
  a = il.SoftActor(c['S'], c['A'], Cfg(hidden_size=64, depth=1, activation='tanh', input_dropout=0.1, dropout=0.1), device=DEV)
  a.flat.copy_(T(c['params']))
  opt = il.AdamW(a, lr=1e-3, weight_decay=0.0)
  ds = odril.DrilState(c['S'], c['A'], 64, 0.1, 0.1); ds.params[:] = c['params']
  for k, (b, m0, m1) in enumerate(zip(c['batches'], c['m0'], c['m1']), 1):
    il.behavioural_cloning_update(a, tbatch(b), opt, masks=(T(m0), T(m1)))
    odril.bc_update(ds, b, m0, m1, lr=1e-3, weight_decay=0.0)
    close_params(N(a.flat), ds.params, f'DRIL ant params {k}', 1e-3, steps=k)
  q = tbatch(c['query'])
  u = N(a._get_action_uncertainty(q['states'], q['actions'], masks=(T(c['q_m0']), T(c['q_m1']))))
  ou = odril.uncertainty(ds, c['query']['states'], c['query']['actions'], c['q_m0'], c['q_m1'])
  a.flat.copy_(T(ds.params)); u = N(a._get_action_uncertainty(q['states'], q['actions'], masks=(T(c['q_m0']), T(c['q_m1']))))
  assert np.abs(u - ou).max() <= 1e-4 * max(np.abs(ou).max(), 1e-30)
  # reward-shaping GAIL, hidden 64, ragged batch of 72, state_only
  c = gi.gail_shaped_case(92, 'ant', 64, 72, 1, True)
  icfg = Cfg(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=1.0, mixup_alpha=1, entropy_bonus=0.0, pos_class_prior=0.7, nonnegative_margin=float('inf'),
             discriminator=Cfg(hidden_size=64, depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='GAIL'))
  dd = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.99, device=DEV)
  ods = ogs.ShapedState(c['S'], c['A'], 64, 0.99, True)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ods, k)[...] = c[k]
  dd.flat.copy_(T(ods.pack()))
  for k, v in dd.views().items():
    v.copy_(T(c[k]))
  opt = il.AdamW(dd, lr=1e-3, weight_decay=0.0)
  il.adversarial_imitation_update(None, dd, tbatch(c['policy'][0]), tbatch(c['expert'][0]), opt, icfg, eps_gp=T(c['eps'][0]))
  og = ogs.gail_update(ods, c['policy'][0], c['expert'][0], c['eps'][0], lr=1e-3, weight_decay=0.0, grad_penalty=1.0, return_grads=True)
  close(N(opt.grad), og, 'shaped GAIL ant gradient', rtol=1e-5, atol_scale=1e-5)
  p = tbatch(c['policy'][0])
  dd.flat.copy_(T(ods.pack()))
  close(N(dd.predict_reward(p['states'], p['actions'], p['next_states'], p['terminals'])), ogs.predict_reward(ods, c['policy'][0], 'GAIL'), 'shaped GAIL ant reward', rtol=2e-5, atol_scale=1e-5)
