"""Parity at the sizes the benchmarks run, against reference outputs generated at those sizes (tests/golden/timed_sizes.npz, make_golden.py gen_timed_sizes):
PWIL against N = 25,000 atoms / D = 24 / T = 1000 over 1,100 steps including a reset() (reference models.py:216-249; at this size the merge wave of
k_pwil_merge owns more than one candidate list per lane and stages candidates in LDS), GMMIL.predict_reward at B = 1024 / D = 120 (models.py:189-201,
BASELINE.json configs[3]) as a full reward vector, and one adversarial_imitation_update at B = 1024 with the tuned GAIL_5 hyper-parameters (Mixup, spectral
norm, gradient penalty, entropy bonus; training.py:85-134).  CPU half: the oracle against the same fixtures."""
import os

import numpy as np
import pytest
import torch

import inputs as gi
from oracle import gail as ogail
from oracle import gmmil as ogmmil
from oracle import pwil as opwil

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TIMED_GAIL = dict(lr=0.0002778119723405689, weight_decay=8.46588535234332, grad_penalty=0.2799364347010851, entropy_bonus=0.24145587952807546)


def fixture():
  return np.load(os.path.join(GOLDEN, 'timed_sizes.npz'))


def eps_mix_1024():
  return np.random.RandomState(4036).uniform(size=1024).astype(np.float32)


# ------------------------------------------------------------------------------------------------ CPU: the oracle at these sizes
def test_oracle_pwil_25k_atoms_matches_reference():
  t = fixture()
  atoms, agent = gi.pwil_case(22, 25000, 24, 1100)
  o = opwil.PwilOracle(atoms, 1000, 5, 5)
  r = []
  for k in range(1100):
    r.append(o.compute_reward(agent[k]))
    if k % 1000 == 999:
      o.reset()
  np.testing.assert_allclose(r, t['pwil25k.rewards'], rtol=2e-6)   # the reference's own float32 vs float64: 7.6e-7
  assert len(o.weights) == int(t['pwil25k.remaining'][0])


def test_oracle_gmmil_b1024_matches_reference():
  t = fixture()
  X, E, w, we = gi.gmmil_case(13, 1024, 1024, 120)
  g1, g2 = ogmmil.median_gammas(X, E, w, we)
  np.testing.assert_allclose([g1, g2], t['gmmil1024.gammas'], rtol=1e-5)
  r, sim, _ = ogmmil.gmmil_reward(X, E, w, we, g1, g2, return_parts=True)
  assert np.abs(r - t['gmmil1024.reward_first']).max() <= 1e-5 * np.abs(sim).max()   # the reward is a difference of two near-equal similarities
  X2, _, w2, _ = gi.gmmil_case(14, 1024, 1024, 120)
  r2 = ogmmil.gmmil_reward(X2, E, w2, we, g1, g2)
  assert np.abs(r2 - t['gmmil1024.reward_second']).max() <= 1e-5 * np.abs(sim).max()


def test_oracle_gail_b1024_mixup_matches_reference():
  t = fixture()
  c = gi.gail_case(36, env='halfcheetah', hidden=64, batch=1024, steps=1)
  ds = ogail.DiscState(c['D'], c['H'], True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = c[k]
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  pb, eb = c['policy'][0], c['expert'][0]
  g = ogail.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], c['eps'][0], return_grads=True, loss_function='Mixup', eps_mix=eps_mix_1024(), **TIMED_GAIL)
  np.testing.assert_allclose(g, t['gail1024.g_1'], rtol=1e-5, atol=4e-6 * np.abs(t['gail1024.g_1']).max())
  np.testing.assert_allclose(ds.pack(), t['gail1024.p_1'], rtol=1e-5, atol=4e-6 * np.abs(t['gail1024.p_1']).max())
  for nm in ('u1', 'v1', 'u2', 'v2'):
    np.testing.assert_allclose(getattr(ds, nm), t[f'gail1024.{nm}_1'], rtol=1e-5, atol=2e-6)
  ds.unpack_into(t['gail1024.p_1'].copy())
  r = ogail.predict_reward(ds, cat(pb), 'AIRL')
  f64 = t['gail1024.reward_1_f64']
  assert np.abs(r - f64).max() <= 2 * np.abs(t['gail1024.reward_1'] - f64).max() + 2e-7 * np.abs(f64).max()


# ------------------------------------------------------------------------------------------------ GPU: the HIP path at these sizes
if torch.cuda.is_available():
  import imitation_learning_amd as il
  from imitation_learning_amd import training as il_training
  from gpu_util import DEV, N, T, Cfg, bracket, close, make_disc, tbatch


@pytest.mark.gpu
def test_pwil_25k_atoms_matches_reference():
  t = fixture()
  Nn, D, steps, Th, A = 25000, 24, 1100, 1000, 6
  S = D - A
  atoms, agent = gi.pwil_case(22, Nn, D, steps)
  mem = il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn),
                                                          next_states=torch.from_numpy(atoms[:, :S]), terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn),
                                                          num_trajectories=25), device=DEV)
  d = il.PWILDiscriminator(S, A, Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, Th)
  rewards = []
  for k in range(steps):
    rewards.append(d.compute_reward(T(agent[k:k + 1, :S]), T(agent[k:k + 1, S:])))
    if k % Th == Th - 1:
      d.reset()
  np.testing.assert_allclose(rewards, t['pwil25k.rewards'], rtol=2e-5)   # the bound of the N = 400 test; rewards span 1e-4 .. 0.1
  bracket(rewards, t['pwil25k.rewards'], t['pwil25k.rewards_f64'], 'PWIL rewards at 25k atoms', factor=4.0)
  assert int((d.expert_weights >= 0).sum()) == int(t['pwil25k.remaining'][0])


@pytest.mark.gpu
def test_gmmil_b1024_full_reward_vector_matches_reference():
  t = fixture()
  X, E, w, we = gi.gmmil_case(13, 1024, 1024, 120)
  S = 112
  disc = il.GMMILDiscriminator(S, 8, Cfg(state_only=False))
  args = (T(X[:, :S]), T(X[:, S:]), T(E[:, :S]), T(E[:, S:]), T(w), T(we))
  r, sim, self_sim = il_training.gmmil_predict_reward(disc, *args, return_parts=True)
  np.testing.assert_allclose([disc.gamma_1, disc.gamma_2], t['gmmil1024.gammas'], rtol=1e-5)
  scale = float(np.abs(N(sim)).max())
  assert np.abs(N(r) - t['gmmil1024.reward_first']).max() <= 1e-5 * scale
  X2, _, w2, _ = gi.gmmil_case(14, 1024, 1024, 120)
  disc.gamma_1, disc.gamma_2 = (float(x) for x in t['gmmil1024.gammas'])   # the reference's frozen bandwidths: both float32 results then evaluate the same function
  r2 = N(disc.predict_reward(T(X2[:, :S]), T(X2[:, S:]), args[2], args[3], T(w2), args[5]))
  assert np.abs(r2 - t['gmmil1024.reward_second']).max() <= 1e-5 * scale
  bracket(r2, t['gmmil1024.reward_second'], t['gmmil1024.reward_second_f64'], 'GMMIL reward at B = 1024')


@pytest.mark.gpu
def test_gail_b1024_mixup_update_matches_reference():
  t = fixture()
  c = gi.gail_case(36, env='halfcheetah', hidden=64, batch=1024, steps=1)
  d, ods, icfg = make_disc(c)
  icfg.update(loss_function='Mixup', grad_penalty=TIMED_GAIL['grad_penalty'], entropy_bonus=TIMED_GAIL['entropy_bonus'], mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=float('inf'))
  opt = il.AdamW(d, lr=TIMED_GAIL['lr'], weight_decay=TIMED_GAIL['weight_decay'])
  pb, eb = c['policy'][0], c['expert'][0]
  d.train()
  il.adversarial_imitation_update(None, d, tbatch(pb), tbatch(eb), opt, icfg, eps_gp=T(c['eps'][0]), eps_mix=T(eps_mix_1024()))
  d.eval()
  close(N(opt.grad), t['gail1024.g_1'], 'B = 1024 Mixup gradient', atol_scale=4e-6)
  close(N(d.flat), t['gail1024.p_1'], 'B = 1024 Mixup parameters', atol_scale=4e-6)
  for nm, val in d.views().items():
    close(N(val), t[f'gail1024.{nm}_1'], f'B = 1024 {nm}')
  d.flat.copy_(T(t['gail1024.p_1']))
  for nm, val in d.views().items():
    val.copy_(T(t[f'gail1024.{nm}_1']))
  r = N(d.predict_reward(T(pb['states']), T(pb['actions'])))
  close(r, t['gail1024.reward_1'], 'B = 1024 reward', rtol=1e-4, atol_scale=1e-5)
  bracket(r, t['gail1024.reward_1'], t['gail1024.reward_1_f64'], 'B = 1024 reward')
