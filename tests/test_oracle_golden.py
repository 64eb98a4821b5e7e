"""Pins the numpy oracle to the reference: every fixture in tests/golden/*.npz was produced by
the unmodified reference on CPU torch (tests/golden/make_golden.py).  Tolerance: rtol 1e-5 as
BASELINE.json's north_star states, plus an absolute floor scaled to each tensor's magnitude
(sums of O(256) float32 products differ in the last ulps between MKL, OpenBLAS and MFMA order)."""
import os

import numpy as np
import pytest

import inputs as gi
from oracle import gail, gmmil, nets, pwil, replay, sac
from oracle.mt19937 import MT19937

RTOL = 1e-5


def close(a, b, name, rtol=RTOL, atol_scale=2e-6):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert a.shape == b.shape, (name, a.shape, b.shape)
  atol = atol_scale * max(float(np.abs(b).max()), 1e-30)
  err = np.abs(a - b) - rtol * np.abs(b)
  assert float(err.max()) <= atol, f'{name}: max excess err {err.max():.3e} (atol {atol:.3e}) at {int(err.argmax())}'


def load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name + '.npz'), allow_pickle=False)


# ------------------------------------------------------------------ replay (bit-exact)
@pytest.mark.parametrize('name,seed,size,fill', [('partial', 0, 1000, 300), ('wrapped', 1, 64, 150), ('expert', 2, 500, None)])
def test_replay_bit_exact(golden_dir, name, seed, size, fill):
  g = load(golden_dir, 'replay')
  S, A = 5, 2
  rs = np.random.RandomState(100 + seed)
  if fill is None:
    tr = gi.transitions(rs, size, S, A)
    mem = replay.ReplayOracle(size, S, A, True, transitions={**tr, 'num_trajectories': 3})
  else:
    mem = replay.ReplayOracle(size, S, A, True)
    tr = gi.transitions(rs, fill, S, A)
    for i in range(fill):
      mem.append(i + 1, tr['states'][i], tr['actions'][i], float(tr['rewards'][i]), tr['next_states'][i], bool(tr['terminals'][i]), False)
      if i % 37 == 36:
        mem.wrap_for_absorbing_states()
  assert [mem.idx, int(mem.full), mem.num_trajectories, mem.size] == g[f'{name}_state'].tolist()
  for k in replay.FIELDS:
    np.testing.assert_array_equal(getattr(mem, k)[:min(size, 200)], g[f'{name}_mem_{k}'], err_msg=k)
  np.testing.assert_array_equal(np.array(mem.sample_idx(MT19937(seed), 256)), g[f'{name}_idx'])
  batch = mem.sample(MT19937(seed), 32)
  for k, v in batch.items():
    np.testing.assert_array_equal(v, g[f'{name}_batch_{k}'], err_msg=k)


# ------------------------------------------------------------------ SAC
def run_sac_oracle(c, steps):
  st = sac.SacState(c['S'], c['A'], c['H'], c.get('depth', 2), c.get('activation', 'relu'), c.get('critic_hidden'), c.get('critic_depth'), c.get('critic_activation'))
  st.actor[:], st.critic[:], st.target[:], st.log_alpha[:] = c['actor'], c['critic'], c['target'], c['log_alpha']
  outs = []
  for i in range(steps):
    logp, q, gr = sac.sac_update(st, c['batches'][i], c['eps_next'][i], c['eps_cur'][i], discount=c['discount'], entropy_target=c['entropy_target'],
                                 polyak_factor=c['polyak'], lr=c['lr'], weight_decay=c['weight_decay'], return_grads=True)
    outs.append((logp.copy(), q.copy(), gr, {k: getattr(st, k).copy() for k in ('actor', 'critic', 'target', 'log_alpha', 'actor_m', 'actor_v', 'critic_m', 'critic_v')}))
  return outs


@pytest.mark.parametrize('name,args', [('sac_halfcheetah', (3, 'halfcheetah', 256, 256, 3)), ('sac_hopper_h64', (4, 'hopper', 64, 96, 3)), ('sac_ant_b64', (5, 'ant', 256, 64, 2))])
def test_sac_update_matches_reference(golden_dir, name, args):
  g, c = load(golden_dir, name), gi.sac_case(*args)
  outs = run_sac_oracle(c, args[-1])
  for k, (logp, q, gr, snap) in enumerate(outs, start=1):
    # gradients are compared at step 1 only as tensors (later steps start from already-rounded state) -- and at every step loosely via the parameters
    close(logp, g[f'logp_{k}'], f'logp_{k}', atol_scale=2e-6 * k)
    close(q, g[f'q_{k}'], f'q_{k}', atol_scale=2e-6 * k)
    if k == 1:
      close(gi.strided(gr['critic']), g['g_critic_1'], 'g_critic_1')
      close(gi.strided(gr['actor']), g['g_actor_1'], 'g_actor_1')
      close(gr['alpha'], g['g_alpha_1'], 'g_alpha_1')
    # Adam turns ulp-level gradient noise into O(lr * noise/|g|) parameter noise: the per-step bound is lr (3e-4) * 1e-3 relative-gradient noise
    for nm in ('actor', 'critic'):
      close(gi.strided(snap[nm]), g[f'{nm}_{k}'], f'{nm}_{k}', atol_scale=1e-5 * k)
      close(gi.strided(snap[nm + '_m']), g[f'{nm}_m_{k}'], f'{nm}_m_{k}', atol_scale=1e-5 * k)
      close(gi.strided(snap[nm + '_v']), g[f'{nm}_v_{k}'], f'{nm}_v_{k}', atol_scale=1e-5 * k)
    close(gi.strided(snap['target']), g[f'target_{k}'], f'target_{k}', atol_scale=1e-5 * k)
    close(snap['log_alpha'], g[f'log_alpha_{k}'], f'log_alpha_{k}')


def check_sac_against_golden(g, outs):
  for k, (logp, q, gr, snap) in enumerate(outs, start=1):
    close(logp, g[f'logp_{k}'], f'logp_{k}', atol_scale=2e-6 * k)
    close(q, g[f'q_{k}'], f'q_{k}', atol_scale=2e-6 * k)
    if k == 1:
      close(gi.strided(gr['critic']), g['g_critic_1'], 'g_critic_1')
      close(gi.strided(gr['actor']), g['g_actor_1'], 'g_actor_1')
      close(gr['alpha'], g['g_alpha_1'], 'g_alpha_1')
    for nm in ('actor', 'critic'):
      close(gi.strided(snap[nm]), g[f'{nm}_{k}'], f'{nm}_{k}', atol_scale=1e-5 * k)
      close(gi.strided(snap[nm + '_m']), g[f'{nm}_m_{k}'], f'{nm}_m_{k}', atol_scale=1e-5 * k)
      close(gi.strided(snap[nm + '_v']), g[f'{nm}_v_{k}'], f'{nm}_v_{k}', atol_scale=1e-5 * k)
    close(gi.strided(snap['target']), g[f'target_{k}'], f'target_{k}', atol_scale=1e-5 * k)
    close(snap['log_alpha'], g[f'log_alpha_{k}'], f'log_alpha_{k}')


def general_actor_oracle(c):
  """Acting and behavioural cloning of a general-shape actor through the oracle: what make_golden.gen_sac_general recorded from the reference (models.py:90-105, training.py:57-64)."""
  A, act = c['A'], c['activation']
  shapes = nets.mlp_shapes(c['S'], c['H'], c['depth'], 2 * A)
  p = c['actor'].copy()
  b = c['batches'][0]
  out, _ = nets.mlp_forward(nets.unpack(p, shapes), b['states'], activation=act)
  mean, _, _, std = nets.actor_head(out, A)
  x = c['eps_cur'][0] * std + mean                        # torch.normal: z * std + mean (Normal.sample)
  res = dict(act_greedy=np.tanh(mean), act_sample=np.tanh(x), act_sample_logp=nets.tanh_gaussian_logp(x, mean, std))   # (TanhTransform(cache_size=1): log_prob of the sample uses the cached pre-image x)
  xg = np.arctanh(np.clip(b['actions'], np.float32(-1 + 1e-6), np.float32(1 - 1e-6))).astype(np.float32)
  res['act_logp_given'] = nets.tanh_gaussian_logp(xg, mean, std)
  m, v = np.zeros_like(p), np.zeros_like(p)
  for k in (1, 2):
    bb = c['batches'][k % len(c['batches'])]
    _, gr, _ = sac.bc_update(p, m, v, k, shapes, A, bb, lr=2.5e-4, weight_decay=0.01, return_grads=True, activation=act)
    res[f'bc_actor_{k}'], res[f'bc_g_actor_{k}'] = p.copy(), gr.copy()
  return res


@pytest.mark.parametrize('name', sorted(gi.GENERAL_SAC_CASES))
def test_general_shape_sac_matches_reference(golden_dir, name):
  """models.py:48-69 `_create_fcnn` builds any depth with relu / tanh / sigmoid: the oracle's general form (nets.mlp_forward(activation=)) against the reference run on
  depth 3 / tanh, depth 1 / sigmoid and a 320-wide depth-2 ReLU network - sac_update for three steps, acting, log-probabilities and behavioural cloning."""
  c = gi.sac_case(**gi.GENERAL_SAC_CASES[name])
  g = load(golden_dir, name)
  check_sac_against_golden(g, run_sac_oracle(c, len(c['batches'])))
  res = general_actor_oracle(c)
  close(res['act_greedy'], g['act_greedy'], 'act_greedy', atol_scale=2e-6)
  close(res['act_sample'], g['act_sample'], 'act_sample', atol_scale=2e-6)
  close(res['act_sample_logp'], g['act_sample_logp'], 'act_sample_logp', rtol=1e-4, atol_scale=1e-5)
  close(res['act_logp_given'], g['act_logp_given'], 'act_logp_given', rtol=1e-4, atol_scale=1e-5)
  close(gi.strided(res['bc_g_actor_1']), g['bc_g_actor_1'], 'bc_g_actor_1')
  for k in (1, 2): close(gi.strided(res[f'bc_actor_{k}']), g[f'bc_actor_{k}'], f'bc_actor_{k}', atol_scale=1e-5 * k)


def test_bc_update_matches_reference(golden_dir):
  g = load(golden_dir, 'bc_hopper')
  S, A = gi.DIMS['hopper']
  rs = np.random.RandomState(7)
  p = gi.mlp_params(rs, S, 256, 2, 2 * A, out_scale=0.3)
  m, v = np.zeros_like(p), np.zeros_like(p)
  shapes = nets.mlp_shapes(S, 256, 2, 2 * A)
  for k in range(1, 4):
    b = gi.transitions(rs, 256, S, A, weighted=True)
    b['actions'][:3] = np.array([1.0, -1.0, 0.9999999])[:, None]
    _, gr, _ = sac.bc_update(p, m, v, k, shapes, A, b, lr=2.5e-4, weight_decay=0.01, return_grads=True)
    if k == 1:
      close(gi.strided(gr), g['g_actor_1'], 'g_actor_1')
    close(gi.strided(p), g[f'actor_{k}'], f'actor_{k}', atol_scale=1e-5 * k)
    close(gi.strided(m), g[f'actor_m_{k}'], f'actor_m_{k}', atol_scale=1e-5 * k)
    # log-prob of the updated actor on the same batch (SoftActor.log_prob, models.py:97-99)
    out, _ = nets.mlp_forward(nets.unpack(p, shapes), b['states'])
    mean, _, _, std = nets.actor_head(out, A)
    x = np.arctanh(np.clip(b['actions'], np.float32(-1 + 1e-6), np.float32(1 - 1e-6)))
    close(nets.tanh_gaussian_logp(x, mean, std), g[f'logp_{k}'], f'logp_{k}', rtol=1e-4, atol_scale=1e-5)


# ------------------------------------------------------------------ GAIL
def make_disc(c):
  ds = gail.DiscState(c['D'], c['H'], c['spectral_norm'])
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = c[k]
  return ds


@pytest.mark.parametrize('name,case,hp', [
    ('gail_default', dict(seed=31), dict(lr=3e-5, weight_decay=10, grad_penalty=1.0, entropy_bonus=0.0)),
    ('gail_h128_ent', dict(seed=32, hidden=128), dict(lr=7.3e-5, weight_decay=6.35, grad_penalty=0.32, entropy_bonus=0.0155)),
    ('gail_nosn_nogp', dict(seed=33, env='hopper', hidden=32, batch=128, spectral_norm=False), dict(lr=3e-4, weight_decay=0.0, grad_penalty=0.0, entropy_bonus=0.0)),
])
def test_gail_update_matches_reference(golden_dir, name, case, hp):
  g, c = load(golden_dir, name), gi.gail_case(**case)
  ds = make_disc(c)
  want = ['g.0.bias', 'g.0.parametrizations.weight.original', 'g.2.bias', 'g.2.parametrizations.weight.original'] if c['spectral_norm'] else ['g.0.weight', 'g.0.bias', 'g.2.weight', 'g.2.bias']
  assert g['param_names'].tolist() == want
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  for k in range(1, len(c['policy']) + 1):
    pb, eb = c['policy'][k - 1], c['expert'][k - 1]
    gr = gail.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][k - 1], return_grads=True, **hp)
    close(gr, g[f'g_{k}'], f'g_{k}', atol_scale=2e-6 * k)
    close(ds.pack(), g[f'p_{k}'], f'p_{k}', atol_scale=2e-6 * k)
    close(ds.m, g[f'm_{k}'], f'm_{k}', atol_scale=2e-6 * k); close(ds.v, g[f'v_{k}'], f'v_{k}', atol_scale=2e-6 * k)
    if c['spectral_norm']:
      for nm in ('u1', 'v1', 'u2', 'v2'):
        close(getattr(ds, nm), g[f'{nm}_{k}'], f'{nm}_{k}')
    close(gail.disc_logits(ds, cat(pb)), g[f'logits_{k}'], f'logits_{k}', atol_scale=4e-6)
    for rf in ('AIRL', 'GAIL', 'FAIRL'):
      close(gail.predict_reward(ds, cat(pb), rf), g[f'reward_{rf}_{k}'], f'reward_{rf}_{k}', rtol=1e-4, atol_scale=1e-5)


# ------------------------------------------------------------------ GMMIL / PWIL
@pytest.mark.parametrize('name,dims', [('small', (64, 48, 24)), ('ant', (256, 256, 120))])
def test_gmmil_matches_reference(golden_dir, name, dims):
  g = load(golden_dir, 'gmmil')
  X, E, w, we = gi.gmmil_case(11, *dims)
  g1, g2 = gmmil.median_gammas(X, E, w, we)
  np.testing.assert_allclose([g1, g2], g[f'{name}_gammas'], rtol=1e-6)
  close(gmmil.squared_distance(X, E)[:16], g[f'{name}_sqdist_xe'], 'sqdist')
  r, sim, self_sim = gmmil.gmmil_reward(X, E, w, we, g1, g2, return_parts=True)
  # reward = difference of two near-equal sums: absolute floor relative to the summands, not to the difference
  assert np.abs(r - g[f'{name}_reward_first']).max() <= 1e-5 * np.abs(sim).max()
  X2, _, w2, _ = gi.gmmil_case(12, *dims)
  r2, sim2, _ = gmmil.gmmil_reward(X2, E, w2, we, g1, g2, return_parts=True)
  assert np.abs(r2 - g[f'{name}_reward_second']).max() <= 1e-5 * np.abs(sim2).max()


def test_pwil_matches_reference(golden_dir):
  g = load(golden_dir, 'pwil')
  N, D, steps, Th = 400, 10, 260, 120
  atoms, agent = gi.pwil_case(21, N, D, steps)
  o = pwil.PwilOracle(atoms, Th, 5, 5)
  close(o.scale, g['scale'], 'scale'); close(o.offset, g['offset'], 'offset')
  rewards = []
  for t in range(steps):
    rewards.append(o.compute_reward(agent[t]))
    if t % Th == Th - 1:
      o.reset()
  np.testing.assert_allclose(rewards, g['rewards'], rtol=1e-5)
  assert o.weights.size == int(g['remaining'][0])


def test_adril_relabeller_oracle_matches_reference_fixture(golden_dir):
  """oracle/adril.py against RewardRelabeller / mix_expert_agent_transitions outputs of the reference (bitwise, incl. -0.0)."""
  from oracle import adril as oadril
  g = np.load(os.path.join(golden_dir, 'adril.npz'))
  S, A = gi.DIMS['hopper']
  for name, update_freq, balanced in (('adril_balanced', 1250, True), ('adril_halves', 1250, False), ('sqil_balanced', 0, True), ('sqil_halves', 0, False)):
    rel = oadril.RelabellerOracle(update_freq, balanced)
    for call in range(3):
      pol, exp = gi.adril_batches(40 + call, 64, S, A)
      rel.resample_and_relabel(pol, exp, gi.ADRIL_STEP + call * 700, gi.ADRIL_TRAJ + call, 7)
      for k, v in pol.items():
        assert np.asarray(v, np.float32).tobytes() == g[f'{name}.{call}.{k}'].tobytes(), (name, call, k)
  pol, exp = gi.adril_batches(50, 64, S, A)
  oadril.mix(pol, exp)
  for k, v in pol.items():
    assert np.asarray(v, np.float32).tobytes() == g[f'mix.{k}'].tobytes(), k


@pytest.mark.parametrize('name,kw', [(n, kw) for n, kw, _, _ in gi.RED_CASES], ids=[n for n, *_ in gi.RED_CASES])
def test_red_oracle_matches_reference_fixture(golden_dir, name, kw):
  """oracle/red.py against REDDiscriminator + target_estimation_update outputs of the reference (predictor after each AdamW step,
  kernel-median bandwidth, rewards) for depth 1-2, relu / tanh, with and without the predictor's dropout (the reference's masks are inputs)."""
  from oracle import red
  g = np.load(os.path.join(golden_dir, 'red.npz'))
  c = gi.red_case(**kw)
  lr, wd = (float(x) for x in g[f'{name}.hyper'])
  st = red.RedState(c['D'], c['H'], c['depth'], c['activation'], c['p_in'], c['p']); st.predictor[:] = c['predictor']; st.target[:] = c['target']
  for k, (b, masks) in enumerate(zip(c['batches'], c['masks']), 1):
    red.target_estimation_update(st, np.concatenate([b['states'], b['actions']], 1), b['weights'], lr=lr, weight_decay=wd, masks=masks)
    ref = g[f'{name}.predictor.{k}']
    assert np.max(np.abs(st.predictor - ref) - 1e-5 * np.abs(ref)) <= (1e-5 + 2 * lr * (c['p'] > 0)) * np.abs(ref).max()   # Adam turns ulp-level noise on ~0 gradients (dropped units) into O(lr)
    assert np.mean(np.abs(st.predictor - ref) > 1e-5 * np.abs(ref) + 1e-6) < 5e-3
  e, q = c['sigma_batch'], c['query']
  st.predictor[:] = g[f'{name}.predictor.{len(c["batches"])}']
  sigma = red.set_sigma(st, np.concatenate([e['states'], e['actions']], 1), masks=c['sigma_masks'])
  assert abs(sigma - float(g[f'{name}.sigma_1'][0])) <= 1e-5 * sigma
  np.testing.assert_allclose(red.predict_reward(st, np.concatenate([q['states'], q['actions']], 1)), g[f'{name}.reward'], rtol=2e-5)


@pytest.mark.parametrize('name,kw', [(n, kw) for n, kw, _, _ in gi.DRIL_CASES], ids=[n for n, *_ in gi.DRIL_CASES])
def test_dril_oracle_matches_reference_fixture(golden_dir, name, kw):
  """oracle/dril.py against the reference SoftActor (DRIL configs: depth 1-2, tanh / relu; train mode) fed the same dropout masks: BC updates, MC-dropout
  uncertainty, +-1 reward."""
  from oracle import dril
  g = np.load(os.path.join(golden_dir, 'dril.npz'))
  c = gi.dril_case(**kw)
  lr, wd = (float(x) for x in g[f'{name}.hyper'])
  ds = dril.DrilState(c['S'], c['A'], c['H'], c['p_in'], c['p'], c['depth'], c['activation']); ds.params[:] = c['params']
  for k, b in enumerate(c['batches'], 1):
    dril.bc_update(ds, b, *gi.dril_masks(c, 'm', k - 1), lr=lr, weight_decay=wd)
    ref = g[f'{name}.params.{k}']
    assert np.max(np.abs(ds.params - ref) - 1e-5 * np.abs(ref)) <= (1e-5 + 2 * lr * (c['depth'] == 2 or c['activation'] == 'relu')) * np.abs(ref).max()   # Adam on ~0 gradients of dropped / dead units
    assert np.mean(np.abs(ds.params - ref) > 1e-5 * np.abs(ref) + 1e-6) < 5e-3
  e, q = c['expert'], c['query']
  ds.params[:] = g[f'{name}.params.{len(c["batches"])}']
  ue, ref_ue = dril.uncertainty(ds, e['states'], e['actions'], *gi.dril_masks(c, 'e_m')), g[f'{name}.expert_uncertainty']
  assert np.abs(ue - ref_ue).max() <= 1e-4 * np.abs(ref_ue).max()
  ds.q = float(g[f'{name}.q'][0])
  r, ref_r, ref_u = dril.predict_reward(ds, q['states'], q['actions'], *gi.dril_masks(c, 'q_m')), g[f'{name}.reward'], g[f'{name}.query_uncertainty']
  decided = np.abs(ref_u - ds.q) > 1e-4 * np.abs(ref_u).max()
  assert decided.sum() >= len(ref_r) - 2 and np.array_equal(r[decided], ref_r[decided])


@pytest.mark.parametrize('name,loss,sub', [('pugail', 'PUGAIL', False), ('mixup', 'Mixup', False), ('sublogp', 'BCE', True), ('mixup_sublogp', 'Mixup', True)])
def test_gail_variant_oracle_matches_reference_fixture(golden_dir, name, loss, sub):
  """oracle/gail.py with loss_function PUGAIL / Mixup and with the subtract_log_policy offsets against adversarial_imitation_update of the reference."""
  g = np.load(os.path.join(golden_dir, 'gail_variants.npz'))
  c = gi.gail_case(35, env='hopper', hidden=32, batch=96, steps=2)
  x = gi.gail_extras(35, c)
  ds = gail.DiscState(c['D'], c['H'], True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = c[k]
  for i in range(2):
    p, e = c['policy'][i], c['expert'][i]
    xp, xe = np.concatenate([p['states'], p['actions']], 1), np.concatenate([e['states'], e['actions']], 1)
    lp = g[f'{name}.logp_policy_{i + 1}'] if sub else None
    le = g[f'{name}.logp_expert_{i + 1}'] if sub else None
    lm = g[f'{name}.logp_mix_{i + 1}'] if sub and loss == 'Mixup' else None
    gr = gail.gail_update(ds, xp, p['weights'], xe, e['weights'], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.5, entropy_bonus=0.02, return_grads=True,
                          loss_function=loss, pos_class_prior=0.7, eps_mix=x['eps_mix'][i], logp_policy=None if loss == 'Mixup' else lp, logp_expert=None if loss == 'Mixup' else le, logp_mix=lm)
    ref = g[f'{name}.g_{i + 1}']
    assert np.abs(gr - ref).max() <= 1e-5 * np.abs(ref).max()
    np.testing.assert_allclose(gail.predict_reward(ds, xp, 'AIRL', log_policy=lp), g[f'{name}.reward_{i + 1}'], rtol=3e-5, atol=1e-6)


@pytest.mark.parametrize('name,sn,loss', [('sn_bce', True, 'BCE'), ('plain_pugail', False, 'PUGAIL')])
def test_gail_shaped_oracle_matches_reference_fixture(golden_dir, name, sn, loss):
  """oracle/gail_shaped.py (reward shaping: g + (1 - t)(discount h(s') - h(s)), two power iterations of h per call, closed-form gradient penalty)
  against the reference's autograd: gradients, parameters after AdamW, spectral-norm buffers, rewards."""
  from oracle import gail_shaped as ogs
  g = np.load(os.path.join(golden_dir, 'gail_shaped.npz'))
  c = gi.gail_shaped_case(91, 'hopper', 32, 96, 2, sn)
  ds = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, sn)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = c[k]
  for i in range(2):
    gr = ogs.gail_update(ds, c['policy'][i], c['expert'][i], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function=loss, return_grads=True)
    ref = g[f'{name}.g_{i + 1}']
    assert np.abs(gr - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(ds.pack() - g[f'{name}.p_{i + 1}']).max() <= 2e-6
    if sn:
      for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
        np.testing.assert_allclose(getattr(ds, k), g[f'{name}.{k}_{i + 1}'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ogs.predict_reward(ds, c['policy'][i]), g[f'{name}.reward_{i + 1}'], rtol=5e-5, atol=2e-6)

def test_gail_shaped_oracle_mixup_matches_reference_fixture(golden_dir):
  """Reward shaping under Mixup (training.py:104-113 on every field of the transitions, fractional terminals): oracle/gail_shaped.py against the reference's autograd."""
  from oracle import gail_shaped as ogs
  g = np.load(os.path.join(golden_dir, 'gail_shaped_mixup.npz'))
  c = gi.gail_shaped_case(95, 'hopper', 32, 96, 2, True)
  em = gi.mixup_draws(1095, 96, 2)
  ds = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, True)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = c[k]
  for i in range(2):
    gr = ogs.gail_update(ds, c['policy'][i], c['expert'][i], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function='Mixup', return_grads=True,
                         eps_mix=em[i])
    ref = g[f'g_{i + 1}']
    assert np.abs(gr - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(ds.pack() - g[f'p_{i + 1}']).max() <= 2e-6
    for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
      np.testing.assert_allclose(getattr(ds, k), g[f'{k}_{i + 1}'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ogs.predict_reward(ds, c['policy'][i]), g[f'reward_{i + 1}'], rtol=5e-5, atol=2e-6)


def _deep_state(c):
  from oracle import gail_deep as ogd
  ds = ogd.DeepDiscState(c['D'], c['H'], c['depth'], c['activation'], c['spectral_norm'])
  for l in range(c['depth'] + 1):
    ds.W[l][...] = c['W'][l]; ds.b[l][...] = c['b'][l]; ds.u[l][...] = c['u'][l]; ds.v[l][...] = c['v'][l]
  return ds


@pytest.mark.parametrize('name', [n for n, *_ in gi.GAIL_DEEP_CASES])
def test_gail_deep_oracle_matches_reference_fixture(golden_dir, name):
  """oracle/gail_deep.py (depth 1-2, relu / tanh discriminators: closed-form backward incl. the double backward of the gradient penalty) against
  adversarial_imitation_update + predict_reward of the reference."""
  from oracle import gail_deep as ogd
  g = np.load(os.path.join(golden_dir, 'gail_deep.npz'))
  _, kw, loss, (lr, wd, gp, ent), rf = next(c for c in gi.GAIL_DEEP_CASES if c[0] == name)
  c = gi.gail_deep_case(**kw)
  ds = _deep_state(c)
  cat = lambda b: np.concatenate([b['states'], b['actions']], 1)
  for i in range(len(c['policy'])):
    pb, eb = c['policy'][i], c['expert'][i]
    ds.unpack_into(g[f'{name}.p_{i}']) if i else None            # start every step from the reference's parameters (isolates the step)
    if i and c['spectral_norm']: ds.unpack_sn(g[f'{name}.sn_{i}'])
    grad = ogd.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][i], lr=lr, weight_decay=wd, grad_penalty=gp, entropy_bonus=ent, return_grads=True,
                           loss_function=loss, eps_mix=c['eps_mix'][i])
    ref = g[f'{name}.g_{i + 1}']
    np.testing.assert_allclose(grad, ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())
    if c['spectral_norm']:
      np.testing.assert_allclose(ds.pack_sn(), g[f'{name}.sn_{i + 1}'], rtol=1e-4, atol=1e-6)
    ds.unpack_into(g[f'{name}.p_{i + 1}'])
    np.testing.assert_allclose(ogd.predict_reward(ds, cat(pb), rf), g[f'{name}.reward_{i + 1}'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', ['clamped', 'open'])
def test_gail_deep_and_shaped_oracles_pugail_finite_margin(golden_dir, name):
  """training.py:100-102 with a finite nonnegative_margin on the depth-2 / tanh and on the reward-shaping discriminator (fixture: gail_pu_margin_general.npz, one margin
  that clamps the unlabelled term away at the first update and one that never does): gradients, parameters, u / v buffers."""
  from oracle import gail_deep as ogd, gail_shaped as ogs
  g = np.load(os.path.join(golden_dir, 'gail_pu_margin_general.npz'))
  margin = float(g[f'deep.{name}.margin'][0])
  assert (float(g[f'deep.{name}.value_1'][0]) < -margin) == (name == 'clamped') and (float(g[f'shaped.{name}.value_1'][0]) < -margin) == (name == 'clamped')
  c = gi.gail_deep_case(seed=111, env='hopper', hidden=32, batch=96, steps=2, depth=2, activation='tanh', spectral_norm=True)
  ds = _deep_state(c)
  cat = lambda b: np.concatenate([b['states'], b['actions']], 1)
  for i in range(2):
    pb, eb = c['policy'][i], c['expert'][i]
    if i: ds.unpack_into(g[f'deep.{name}.p_{i}']); ds.unpack_sn(g[f'deep.{name}.sn_{i}'])
    grad = ogd.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.6, entropy_bonus=0.02, return_grads=True,
                           loss_function='PUGAIL', pos_class_prior=0.7, nonnegative_margin=margin)
    ref = g[f'deep.{name}.g_{i + 1}']
    np.testing.assert_allclose(grad, ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())
    np.testing.assert_allclose(ds.pack_sn(), g[f'deep.{name}.sn_{i + 1}'], rtol=1e-4, atol=1e-6)
  c = gi.gail_shaped_case(93, 'hopper', 32, 96, 2, True)
  ss = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, True)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ss, k)[...] = c[k]
  for i in range(2):
    gr = ogs.gail_update(ss, c['policy'][i], c['expert'][i], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function='PUGAIL', return_grads=True,
                         pos_class_prior=0.7, nonnegative_margin=margin)
    ref = g[f'shaped.{name}.g_{i + 1}']
    assert np.abs(gr - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(ss.pack() - g[f'shaped.{name}.p_{i + 1}']).max() <= 2e-6
    for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
      np.testing.assert_allclose(getattr(ss, k), g[f'shaped.{name}.{k}_{i + 1}'], rtol=1e-5, atol=1e-6)


def _shaped_deep_state(c):
  from oracle import gail_shaped_deep as osd
  ds = osd.ShapedDeepState(c['S'], c['A'], c['H'], 0.97, c['depth'], c['activation'], c['spectral_norm'], c['state_only'])
  ds.Wg[...] = c['Wg']; ds.bg[...] = c['bg']; ds.ug[...] = c['ug']; ds.vg[...] = c['vg']
  for l in range(c['depth'] + 1):
    ds.h.W[l][...] = c['W'][l]; ds.h.b[l][...] = c['b'][l]; ds.h.u[l][...] = c['u'][l]; ds.h.v[l][...] = c['v'][l]
  return ds


@pytest.mark.parametrize('name', [n for n, *_ in gi.GAIL_SHAPED_DEEP_CASES])
def test_gail_shaped_deep_oracle_matches_reference_fixture(golden_dir, name):
  """oracle/gail_shaped_deep.py (reward shaping with a depth 1-2 / relu / tanh potential: two power iterations of every layer of h per call, chain rule per use,
  the gradient penalty's double backward through the second use) against adversarial_imitation_update + predict_reward of the reference."""
  from oracle import gail_shaped_deep as osd
  g = np.load(os.path.join(golden_dir, 'gail_shaped_deep.npz'))
  _, kw, loss, (lr, wd, gp, ent), rf, margin = next(c for c in gi.GAIL_SHAPED_DEEP_CASES if c[0] == name)
  c = gi.gail_shaped_deep_case(**kw)
  ds = _shaped_deep_state(c)
  if margin != float('inf'):
    assert float(g[f'{name}.value_1'][0]) < -margin, 'the fixture is meant to have the clamp binding at the first update'
  for i in range(len(c['policy'])):
    if i:   # start every step from the reference's parameters (isolates the step)
      ds.unpack_into(g[f'{name}.p_{i}'])
      if c['spectral_norm']: ds.unpack_sn(g[f'{name}.sn_{i}'])
    grad = osd.gail_update(ds, c['policy'][i], c['expert'][i], c['eps'][i], lr=lr, weight_decay=wd, grad_penalty=gp, entropy_bonus=ent, return_grads=True, loss_function=loss,
                           pos_class_prior=0.7, nonnegative_margin=margin, eps_mix=c['eps_mix'][i])
    ref = g[f'{name}.g_{i + 1}']
    np.testing.assert_allclose(grad, ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())
    assert np.abs(ds.pack() - g[f'{name}.p_{i + 1}']).max() <= 6e-6   # lr <= 1e-3; Adam's early steps are lr * m / sqrt(v): the ratio amplifies the gradient's last bits where |g| is small
    if c['spectral_norm']:
      np.testing.assert_allclose(ds.pack_sn(), g[f'{name}.sn_{i + 1}'], rtol=1e-4, atol=1e-6)
    ds.unpack_into(g[f'{name}.p_{i + 1}'])
    np.testing.assert_allclose(osd.predict_reward(ds, c['policy'][i], rf), g[f'{name}.reward_{i + 1}'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(osd.predict_reward(ds, c['policy'][i], rf, log_policy=c['logp_policy'][i]), g[f'{name}.reward_logp_{i + 1}'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('sn,loss', [(True, 'BCE'), (False, 'PUGAIL'), (True, 'Mixup')])
def test_gail_shaped_deep_oracle_equals_the_depth1_relu_oracle(sn, loss):
  """The general restatement reduces to oracle/gail_shaped.py (itself pinned to the reference) for a depth-1 ReLU potential: same gradients, parameters, buffers, rewards."""
  from oracle import gail_shaped as ogs, gail_shaped_deep as osd
  c = gi.gail_shaped_case(97, 'hopper', 32, 96, 2, sn)
  em = gi.mixup_draws(1097, 96, 2)
  a = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, sn)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(a, k)[...] = c[k]
  b = osd.ShapedDeepState(c['S'], c['A'], c['H'], 0.97, 1, 'relu', sn)
  b.Wg[...] = c['Wg']; b.bg[...] = c['bg']; b.ug[...] = c['ug']; b.vg[...] = c['vg']
  for l, n in enumerate(('1', '2')):
    b.h.W[l][...] = c['W' + n]; b.h.b[l][...] = c['b' + n]; b.h.u[l][...] = c['u' + n]; b.h.v[l][...] = c['v' + n]
  kw = dict(lr=1e-3, weight_decay=0.1, grad_penalty=0.7, entropy_bonus=0.01, loss_function=loss, return_grads=True)
  for i in range(2):
    ga = ogs.gail_update(a, c['policy'][i], c['expert'][i], c['eps'][i], eps_mix=em[i], **kw)
    gb = osd.gail_update(b, c['policy'][i], c['expert'][i], c['eps'][i], eps_mix=em[i], **kw)
    np.testing.assert_allclose(gb, ga, rtol=1e-5, atol=1e-6 * np.abs(ga).max())
    np.testing.assert_allclose(b.pack(), a.pack(), rtol=1e-5, atol=2e-6)
    if sn:
      np.testing.assert_allclose(b.pack_sn(), np.concatenate([a.ug, a.vg, a.u1, a.v1, a.u2, a.v2]), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(osd.predict_reward(b, c['policy'][i]), ogs.predict_reward(a, c['policy'][i]), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ Philox (the on-chip noise source's restatement)
def test_philox_restatement_matches_random123_known_answers():
  """Known-answer vectors of Random123's philox4x32_10 (kat_vectors: all-zero, all-ones and the pi-digits counter / key)."""
  from oracle import philox
  kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
         ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
         ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
  for ctr, key, want in kat:
    assert tuple(int(x) for x in philox.philox4x32_10(*ctr, *key)) == want
  z = philox.normal(7, 3, philox.STREAM_EPS_CUR, 1 << 18).astype(np.float64)
  assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
  u = philox.uniform(7, 3, philox.STREAM_GP, 1 << 18)
  assert 0 <= u.min() and u.max() < 1 and abs(float(u.mean()) - 0.5) < 0.01


@pytest.mark.parametrize('name', ['clamped', 'open'])
def test_oracle_pugail_finite_margin_matches_reference(name):
  """training.py:100-102 with a finite nonnegative_margin: the clamp decides batch-wide whether the unlabelled term has a gradient (fixture: one margin each way)."""
  g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'gail_pu_margin.npz'))
  c = gi.gail_case(37, env='halfcheetah', hidden=64, batch=128, steps=2)
  ds = gail.DiscState(c['D'], c['H'], True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = c[k]
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  margin = float(g[f'{name}.margin'][0])
  assert (float(g[f'{name}.value_1'][0]) < -margin) == (name == 'clamped')
  for i in range(2):
    pb, eb = c['policy'][i], c['expert'][i]
    gr = gail.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][i], lr=1e-3, weight_decay=0.1, grad_penalty=0.5, entropy_bonus=0.01, return_grads=True,
                          loss_function='PUGAIL', pos_class_prior=0.7, nonnegative_margin=margin)
    np.testing.assert_allclose(gr, g[f'{name}.g_{i + 1}'], rtol=1e-5, atol=4e-6 * np.abs(g[f'{name}.g_{i + 1}']).max())
    np.testing.assert_allclose(ds.pack(), g[f'{name}.p_{i + 1}'], rtol=1e-5, atol=4e-6 * (i + 1) * np.abs(g[f'{name}.p_{i + 1}']).max())
