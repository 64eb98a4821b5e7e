#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the UNMODIFIED reference on CPU torch.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The reference's three hot-path modules import with a 12-line `omegaconf` stub
(the real package is absent; it is only used for a type annotation and
attribute/.get access).  Noise is recorded, not re-implemented: the three draw
sites (`torch.normal` in Normal.sample, `_standard_normal` in rsample,
`torch.rand_like` for the gradient-penalty epsilon) are wrapped so the values
the reference consumed are exactly the ones stored in / rebuilt from inputs.py.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import inputs as gi  # noqa: E402

REF = os.environ.get('IL_REFERENCE', '/root/reference')
stub = tempfile.mkdtemp()
os.makedirs(os.path.join(stub, 'omegaconf'))
with open(os.path.join(stub, 'omegaconf', '__init__.py'), 'w') as f:
  f.write('''
class DictConfig(dict):
  def __getattr__(self, k):
    try: v = self[k]
    except KeyError: raise AttributeError(k)
    return DictConfig(v) if isinstance(v, dict) and not isinstance(v, DictConfig) else v
  def __setattr__(self, k, v): self[k] = v
class OmegaConf: pass
''')
sys.path.insert(0, stub)
sys.path.insert(0, REF)
from omegaconf import DictConfig  # noqa: E402
import memory as ref_memory  # noqa: E402
import models as ref_models  # noqa: E402
import training as ref_training  # noqa: E402

torch.set_num_threads(1)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
N_ = lambda t: t.detach().cpu().numpy().copy()


class NoiseFeed:
  """Feeds pre-drawn noise to the reference's three draw sites, in call order."""

  def __init__(self):
    self.normal, self.rsample, self.rand = [], [], []
    self._orig = (torch.normal, torch.distributions.normal._standard_normal, torch.rand_like)

  def __enter__(self):
    def normal(mean, std, *a, **k):
      return self.normal.pop(0) * std + mean  # same op order as at::normal (z * std + mean)

    def std_normal(shape, dtype, device):
      return self.rsample.pop(0)

    def rand_like(x, *a, **k):
      return self.rand.pop(0)
    torch.normal, torch.distributions.normal._standard_normal, torch.rand_like = normal, std_normal, rand_like
    return self

  def __exit__(self, *a):
    torch.normal, torch.distributions.normal._standard_normal, torch.rand_like = self._orig


def flat(module):
  return N_(torch.nn.utils.parameters_to_vector(module.parameters()))


def opt_state(opt, key):
  return np.concatenate([N_(opt.state[p][key]).ravel() for g in opt.param_groups for p in g['params']])


def tbatch(b):
  return {k: T(v) for k, v in b.items()}


def tbatch64(b):
  return {k: T(v).double() for k, v in b.items()}


def twin64(module):
  """A float64 copy of a reference module (same parameters / buffers, widened): the reference's own code evaluated in double precision."""
  import copy
  return copy.deepcopy(module).double()


# ---------------------------------------------------------------- replay
def gen_replay():
  out = {}
  # index streams: (seed, size, prefill rows) for a not-full ring, a wrapped ring, an expert buffer (full, idx 0)
  for name, seed, size, fill in (('partial', 0, 1000, 300), ('wrapped', 1, 64, 150), ('expert', 2, 500, None)):
    S, A = 5, 2
    rs = np.random.RandomState(100 + seed)
    if fill is None:
      tr = gi.transitions(rs, size, S, A)
      mem = ref_memory.ReplayMemory(size, S, A, True, transitions={**{k: T(v) for k, v in tr.items() if k != 'absorbing'}, 'num_trajectories': 3})
    else:
      mem = ref_memory.ReplayMemory(size, S, A, True)
      tr = gi.transitions(rs, fill, S, A)
      for i in range(fill):
        mem.append(i + 1, T(tr['states'][i:i + 1]), T(tr['actions'][i:i + 1]), float(tr['rewards'][i]), T(tr['next_states'][i:i + 1]), bool(tr['terminals'][i]), False)
        if i % 37 == 36:
          mem.wrap_for_absorbing_states()
    np.random.seed(seed)
    idxs = [mem._sample_idx() for _ in range(256)]
    out[f'{name}_idx'] = np.array(idxs, np.int64)
    np.random.seed(seed)
    batch = mem.sample(32)
    for k, v in batch.items():
      out[f'{name}_batch_{k}'] = N_(v)
    out[f'{name}_state'] = np.array([mem.idx, int(mem.full), mem.num_trajectories, mem.size])
    for k in ('step', 'states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
      out[f'{name}_mem_{k}'] = N_(getattr(mem, k))[:min(size, 200)]
  np.savez_compressed(os.path.join(HERE, 'replay.npz'), **out)


# ---------------------------------------------------------------- SAC
def build_sac(c):
  cfg = DictConfig(hidden_size=c['H'], depth=c.get('depth', 2), activation=c.get('activation', 'relu'))
  ccfg = DictConfig(hidden_size=c.get('critic_hidden', c['H']), depth=c.get('critic_depth', c.get('depth', 2)), activation=c.get('critic_activation', c.get('activation', 'relu')))
  actor, critic = ref_models.SoftActor(c['S'], c['A'], cfg), ref_models.TwinCritic(c['S'], c['A'], ccfg)
  torch.nn.utils.vector_to_parameters(T(c['actor']), actor.parameters())
  torch.nn.utils.vector_to_parameters(T(c['critic']), critic.parameters())
  target = ref_models.create_target_network(critic)
  torch.nn.utils.vector_to_parameters(T(c['target']), target.parameters())
  log_alpha = T(c['log_alpha'].copy()).requires_grad_()
  return actor, critic, target, log_alpha


def gen_sac(name, c):
  actor, critic, target, log_alpha = build_sac(c)
  ao = torch.optim.AdamW(actor.parameters(), lr=c['lr'], weight_decay=c['weight_decay'])
  co = torch.optim.AdamW(critic.parameters(), lr=c['lr'], weight_decay=c['weight_decay'])
  to = torch.optim.Adam([log_alpha], lr=c['lr'])
  grads = {}

  def snap(opt, key, params):
    orig = opt.step

    def step(*a, **k):
      grads.setdefault(key, []).append(np.concatenate([N_(p.grad).ravel() for p in params]))
      return orig(*a, **k)
    opt.step = step
  snap(ao, 'actor', list(actor.parameters())); snap(co, 'critic', list(critic.parameters())); snap(to, 'alpha', [log_alpha])
  out = {}
  for i, b in enumerate(c['batches']):
    with NoiseFeed() as nf:
      nf.normal.append(T(c['eps_next'][i])); nf.rsample.append(T(c['eps_cur'][i]))
      logp, q = ref_training.sac_update(actor, critic, log_alpha, target, tbatch(b), ao, co, to, c['discount'], c['entropy_target'], c['polyak'])
      assert not nf.normal and not nf.rsample
    k = i + 1
    out[f'logp_{k}'], out[f'q_{k}'] = N_(logp), N_(q)
    out[f'log_alpha_{k}'] = N_(log_alpha)
    for nm, mod, opt in (('actor', actor, ao), ('critic', critic, co)):
      out[f'{nm}_{k}'] = gi.strided(flat(mod))
      out[f'{nm}_m_{k}'] = gi.strided(opt_state(opt, 'exp_avg'))
      out[f'{nm}_v_{k}'] = gi.strided(opt_state(opt, 'exp_avg_sq'))
      out[f'{nm}_norm_{k}'] = np.array([np.linalg.norm(flat(mod).astype(np.float64))])
    out[f'target_{k}'] = gi.strided(flat(target))
    out[f'g_actor_{k}'] = gi.strided(grads['actor'][i]); out[f'g_critic_{k}'] = gi.strided(grads['critic'][i]); out[f'g_alpha_{k}'] = grads['alpha'][i]
    out[f'g_actor_norm_{k}'] = np.array([np.linalg.norm(grads['actor'][i].astype(np.float64))])
    out[f'g_critic_norm_{k}'] = np.array([np.linalg.norm(grads['critic'][i].astype(np.float64))])
  np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)


def gen_sac_general(name, kw):
  """General actor / critic shapes (models.py:48-69: any depth, relu / tanh / sigmoid): the same quantities as gen_sac plus, on the INITIAL actor, what acting and
  behavioural cloning need - greedy action, a sample and its log-probability with fed noise, log pi of given actions - and two behavioural_cloning_update steps on a copy."""
  gen_sac(name, gi.sac_case(**kw))
  out = dict(np.load(os.path.join(HERE, f'{name}.npz')))
  c = gi.sac_case(**kw)   # (a fresh case: the reference's optimisers stepped the first one's parameter arrays in place)
  actor, _, _, _ = build_sac(c)
  b = c['batches'][0]
  st, ac = T(b['states']), T(b['actions'])
  with torch.no_grad():
    out['act_greedy'] = N_(actor.get_greedy_action(st))
    with NoiseFeed() as nf:
      nf.normal.append(T(c['eps_cur'][0]))
      pol = actor(st)
      a = pol.sample()
      out['act_sample'], out['act_sample_logp'] = N_(a), N_(pol.log_prob(a))
    out['act_logp_given'] = N_(actor.log_prob(st, ac))
  opt = torch.optim.AdamW(actor.parameters(), lr=2.5e-4, weight_decay=0.01)
  for k in (1, 2):
    bb = c['batches'][k % len(c['batches'])]
    ref_training.behavioural_cloning_update(actor, tbatch(bb), opt)
    out[f'bc_actor_{k}'] = gi.strided(flat(actor)); out[f'bc_g_actor_{k}'] = gi.strided(np.concatenate([N_(p.grad).ravel() for p in actor.parameters()]))
    out[f'bc_actor_norm_{k}'] = np.array([np.linalg.norm(flat(actor).astype(np.float64))])
  np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)


def gen_bc(name, env, hidden, batch, steps, f64=None):
  S, A = gi.DIMS[env]
  rs = np.random.RandomState(7)
  p0 = gi.mlp_params(rs, S, hidden, 2, 2 * A, out_scale=0.3)
  actor = ref_models.SoftActor(S, A, DictConfig(hidden_size=hidden, depth=2, activation='relu'))
  torch.nn.utils.vector_to_parameters(T(p0), actor.parameters())
  opt = torch.optim.AdamW(actor.parameters(), lr=2.5e-4, weight_decay=0.01)
  out = {}
  for k in range(1, steps + 1):
    b = gi.transitions(rs, batch, S, A, weighted=True)
    b['actions'][:3] = np.array([1.0, -1.0, 0.9999999])[:, None]  # exercise the clamp
    if f64 is not None and k == 1:   # log pi on the INITIAL parameters (known to the tests from the seed): float32 and float64
      with torch.no_grad():
        f64[f'{name}.logp_init_f32'] = N_(actor.log_prob(T(b['states']), T(b['actions'])))
        f64[f'{name}.logp_init'] = N_(twin64(actor).log_prob(T(b['states']).double(), T(b['actions']).double()))
    ref_training.behavioural_cloning_update(actor, tbatch(b), opt)
    out[f'actor_{k}'] = gi.strided(flat(actor)); out[f'actor_m_{k}'] = gi.strided(opt_state(opt, 'exp_avg')); out[f'actor_v_{k}'] = gi.strided(opt_state(opt, 'exp_avg_sq'))
    out[f'g_actor_{k}'] = gi.strided(np.concatenate([N_(p.grad).ravel() for p in actor.parameters()]))
    with torch.no_grad():
      out[f'logp_{k}'] = N_(actor.log_prob(T(b['states']), T(b['actions'])))
  if f64 is not None: return
  np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)


# ---------------------------------------------------------------- GAIL
def build_disc(c, reward_function='AIRL'):
  icfg = DictConfig(state_only=False, spectral_norm=c['spectral_norm'],
                    discriminator=DictConfig(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=False, subtract_log_policy=False, reward_function=reward_function))
  d = ref_models.GAILDiscriminator(c['S'], c['A'], icfg, 0.97)
  with torch.no_grad():
    if c['spectral_norm']:
      for li, (W, b, u, v) in ((0, (c['W1'], c['b1'], c['u1'], c['v1'])), (2, (c['W2'], c['b2'], c['u2'], c['v2']))):
        d.g[li].parametrizations.weight.original.copy_(T(W)); d.g[li].bias.copy_(T(b))
        d.g[li].parametrizations.weight[0]._u.copy_(T(u)); d.g[li].parametrizations.weight[0]._v.copy_(T(v))
    else:
      d.g[0].weight.copy_(T(c['W1'])); d.g[0].bias.copy_(T(c['b1'])); d.g[2].weight.copy_(T(c['W2'])); d.g[2].bias.copy_(T(c['b2']))
  return d, icfg


def gen_gail(name, c, *, lr, weight_decay, grad_penalty, entropy_bonus, f64=None):
  d, icfg = build_disc(c)
  icfg.update(loss_function='BCE', grad_penalty=grad_penalty, mixup_alpha=1, entropy_bonus=entropy_bonus, pos_class_prior=0.7, nonnegative_margin=float('inf'))
  opt = torch.optim.AdamW(d.parameters(), lr=lr, weight_decay=weight_decay)
  out = {'param_names': np.array([n for n, _ in d.named_parameters()])}
  for i in range(len(c['policy'])):
    d.train()
    with NoiseFeed() as nf:
      nf.rand.append(T(c['eps'][i]))
      ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
    d.eval()
    k = i + 1
    out[f'g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()])
    out[f'p_{k}'] = flat(d); out[f'm_{k}'] = opt_state(opt, 'exp_avg'); out[f'v_{k}'] = opt_state(opt, 'exp_avg_sq')
    if c['spectral_norm']:
      for li, nm in ((0, '1'), (2, '2')):
        out[f'u{nm}_{k}'] = N_(d.g[li].parametrizations.weight[0]._u); out[f'v{nm}_{k}'] = N_(d.g[li].parametrizations.weight[0]._v)
    b = c['policy'][i]
    with torch.inference_mode():
      for rf in ('AIRL', 'GAIL', 'FAIRL'):
        d.reward_function = rf
        out[f'reward_{rf}_{k}'] = N_(d.predict_reward(T(b['states']), T(b['actions'])))
        if f64 is not None:
          d64 = twin64(d); d64.reward_function = rf
          f64[f'{name}.reward_{rf}_{k}'] = N_(d64.predict_reward(T(b['states']).double(), T(b['actions']).double()))
      out[f'logits_{k}'] = N_(d(T(b['states']), T(b['actions'])))
  if f64 is not None: return
  np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)


def gen_gail_variants():
  """adversarial_imitation_update with loss_function=PUGAIL / Mixup and with subtract_log_policy (training.py:100-113, models.py:139-144,173-175):
  gradients, parameters after AdamW, rewards. Mixup's Beta(alpha, alpha) draws are fed (Beta.sample patched), like the other noise."""
  out = {}
  for name, loss, sub in (('pugail', 'PUGAIL', False), ('mixup', 'Mixup', False), ('sublogp', 'BCE', True), ('mixup_sublogp', 'Mixup', True)):
    c = gi.gail_case(35, env='hopper', hidden=32, batch=96, steps=2)
    x = gi.gail_extras(35, c)
    d, icfg = build_disc(c)
    icfg.discriminator.update(subtract_log_policy=sub); icfg['discriminator'] = icfg.discriminator
    d.subtract_log_policy = sub
    icfg.update(loss_function=loss, grad_penalty=0.5, mixup_alpha=0.7, entropy_bonus=0.02, pos_class_prior=0.7, nonnegative_margin=float('inf'))
    actor = ref_models.SoftActor(c['S'], c['A'], DictConfig(hidden_size=64, depth=2, activation='relu'))
    torch.nn.utils.vector_to_parameters(T(x['actor']), actor.parameters())
    opt = torch.optim.AdamW(d.parameters(), lr=1e-3, weight_decay=0.1)
    for i in range(len(c['policy'])):
      d.train()
      feed = [T(x['eps_mix'][i])]
      orig = torch.distributions.Beta.sample
      torch.distributions.Beta.sample = lambda self, *a, **k: feed.pop(0)
      try:
        with NoiseFeed() as nf:
          nf.rand.append(T(c['eps'][i]))
          ref_training.adversarial_imitation_update(actor, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      finally:
        torch.distributions.Beta.sample = orig
      d.eval()
      k = i + 1
      out[f'{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()])
      out[f'{name}.p_{k}'] = flat(d)
      b = c['policy'][i]
      with torch.inference_mode():
        inp = ref_models.make_gail_input(T(b['states']), T(b['actions']), T(b['next_states']), T(b['terminals']), actor, False, sub)
        out[f'{name}.reward_{k}'] = N_(d.predict_reward(**inp))
        if sub:
          out[f'{name}.logp_policy_{k}'] = N_(actor.log_prob(T(b['states']), T(b['actions'])))
          e = c['expert'][i]
          out[f'{name}.logp_expert_{k}'] = N_(actor.log_prob(T(e['states']), T(e['actions'])))
          if loss == 'Mixup':
            em = T(x['eps_mix'][i]).unsqueeze(1)
            out[f'{name}.logp_mix_{k}'] = N_(actor.log_prob(em * T(e['states']) + (1 - em) * T(b['states']), em * T(e['actions']) + (1 - em) * T(b['actions'])))
  np.savez_compressed(os.path.join(HERE, 'gail_variants.npz'), **out)


def gen_gail_pu_margin():
  """adversarial_imitation_update with loss_function=PUGAIL and a FINITE nonnegative_margin (training.py:100-102): one margin that clamps the unlabelled term away
  (its gradient vanishes) and one that does not; gradients, parameters, and the clamped value itself."""
  out = {}
  for name, margin in (('clamped', 0.02), ('open', 1.5)):
    c = gi.gail_case(37, env='halfcheetah', hidden=64, batch=128, steps=2)
    d, icfg = build_disc(c)
    icfg.update(loss_function='PUGAIL', grad_penalty=0.5, mixup_alpha=1, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=margin)
    opt = torch.optim.AdamW(d.parameters(), lr=1e-3, weight_decay=0.1)
    for i in range(2):
      d.train()
      seen = []
      orig = torch.clamp
      def spy(x, *a, **k):
        if x.dim() == 0: seen.append(float(x))
        return orig(x, *a, **k)
      torch.clamp = spy
      try:
        with NoiseFeed() as nf:
          nf.rand.append(T(c['eps'][i]))
          ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      finally:
        torch.clamp = orig
      d.eval()
      k = i + 1
      out[f'{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()]); out[f'{name}.p_{k}'] = flat(d)
      out[f'{name}.value_{k}'] = np.array(seen[:1], np.float64)
    out[f'{name}.margin'] = np.array([margin])
  np.savez_compressed(os.path.join(HERE, 'gail_pu_margin.npz'), **out)


def deep_disc(c, icfg):
  """The reference's GAILDiscriminator for a gail_deep_case, weights / biases / u / v set from the case; and the indices of its Linear layers in d.g."""
  d = ref_models.GAILDiscriminator(c['S'], c['A'], icfg, 0.97)
  lin = [2 * l for l in range(c['depth'] + 1)]
  with torch.no_grad():
    for l, li in enumerate(lin):
      if c['spectral_norm']:
        d.g[li].parametrizations.weight.original.copy_(T(c['W'][l])); d.g[li].bias.copy_(T(c['b'][l]))
        d.g[li].parametrizations.weight[0]._u.copy_(T(c['u'][l])); d.g[li].parametrizations.weight[0]._v.copy_(T(c['v'][l]))
      else:
        d.g[li].weight.copy_(T(c['W'][l])); d.g[li].bias.copy_(T(c['b'][l]))
  return d, lin


def shaped_disc(c, icfg, sn):
  """The reference's reward-shaping GAILDiscriminator for a gail_shaped_case."""
  d = ref_models.GAILDiscriminator(c['S'], c['A'], icfg, 0.97)
  with torch.no_grad():
    if sn:
      d.g.parametrizations.weight.original.copy_(T(c['Wg'])); d.g.bias.copy_(T(c['bg']))
      d.g.parametrizations.weight[0]._u.copy_(T(c['ug'])); d.g.parametrizations.weight[0]._v.copy_(T(c['vg']))
      for li, (W, b, u, v) in ((0, (c['W1'], c['b1'], c['u1'], c['v1'])), (2, (c['W2'], c['b2'], c['u2'], c['v2']))):
        d.h[li].parametrizations.weight.original.copy_(T(W)); d.h[li].bias.copy_(T(b))
        d.h[li].parametrizations.weight[0]._u.copy_(T(u)); d.h[li].parametrizations.weight[0]._v.copy_(T(v))
    else:
      d.g.weight.copy_(T(c['Wg'])); d.g.bias.copy_(T(c['bg']))
      d.h[0].weight.copy_(T(c['W1'])); d.h[0].bias.copy_(T(c['b1'])); d.h[2].weight.copy_(T(c['W2'])); d.h[2].bias.copy_(T(c['b2']))
  return d


class ClampSpy:
  """Records the scalar arguments of torch.clamp (training.py:102: the value the PUGAIL margin is compared with)."""
  def __enter__(self):
    self.seen, self.orig = [], torch.clamp
    def spy(x, *a, **k):
      if x.dim() == 0: self.seen.append(float(x))
      return self.orig(x, *a, **k)
    torch.clamp = spy
    return self
  def __exit__(self, *exc):
    torch.clamp = self.orig


PU_MARGINS = (('clamped', 0.02), ('open', 1.5))


def gen_gail_pu_margin_general():
  """PUGAIL with a FINITE nonnegative_margin (training.py:100-102) on the two other discriminator shapes: depth 2 / tanh / spectral norm (models.py:152-162) and
  reward shaping with spectral norm (models.py:163-176). One margin that clamps the unlabelled term away and one that does not; gradients, parameters, u / v
  buffers and the clamped value."""
  out = {}
  for name, margin in PU_MARGINS:
    c = gi.gail_deep_case(seed=111, env='hopper', hidden=32, batch=96, steps=2, depth=2, activation='tanh', spectral_norm=True)
    icfg = DictConfig(state_only=False, spectral_norm=True, loss_function='PUGAIL', grad_penalty=0.6, mixup_alpha=1, entropy_bonus=0.02, pos_class_prior=0.7, nonnegative_margin=margin,
                      discriminator=DictConfig(hidden_size=c['H'], depth=2, activation='tanh', reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
    d, lin = deep_disc(c, icfg)
    opt = torch.optim.AdamW(d.parameters(), lr=1e-3, weight_decay=0.1)
    for i in range(2):
      d.train()
      with ClampSpy() as spy, NoiseFeed() as nf:
        nf.rand.append(T(c['eps'][i]))
        ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      d.eval()
      k = i + 1
      out[f'deep.{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()]); out[f'deep.{name}.p_{k}'] = flat(d)
      out[f'deep.{name}.sn_{k}'] = np.concatenate([np.concatenate([N_(d.g[li].parametrizations.weight[0]._u), N_(d.g[li].parametrizations.weight[0]._v)]) for li in lin])
      out[f'deep.{name}.value_{k}'] = np.array(spy.seen[:1], np.float64)
    out[f'deep.{name}.margin'] = np.array([margin])

    c = gi.gail_shaped_case(93, 'hopper', 32, 96, 2, True)
    icfg = DictConfig(state_only=False, spectral_norm=True, loss_function='PUGAIL', grad_penalty=0.7, mixup_alpha=1, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=margin,
                      discriminator=DictConfig(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
    d = shaped_disc(c, icfg, True)
    opt = torch.optim.AdamW(d.parameters(), lr=1e-3, weight_decay=0.1)
    for i in range(2):
      d.train()
      with ClampSpy() as spy, NoiseFeed() as nf:
        nf.rand.append(T(c['eps'][i]))
        ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      d.eval()
      k = i + 1
      out[f'shaped.{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()]); out[f'shaped.{name}.p_{k}'] = flat(d)
      out[f'shaped.{name}.ug_{k}'] = N_(d.g.parametrizations.weight[0]._u); out[f'shaped.{name}.vg_{k}'] = N_(d.g.parametrizations.weight[0]._v)
      for li, nm in ((0, '1'), (2, '2')):
        out[f'shaped.{name}.u{nm}_{k}'] = N_(d.h[li].parametrizations.weight[0]._u); out[f'shaped.{name}.v{nm}_{k}'] = N_(d.h[li].parametrizations.weight[0]._v)
      out[f'shaped.{name}.value_{k}'] = np.array(spy.seen[:1], np.float64)
    out[f'shaped.{name}.margin'] = np.array([margin])
  np.savez_compressed(os.path.join(HERE, 'gail_pu_margin_general.npz'), **out)


def gen_gail_deep(f64=None):
  """GAILDiscriminator with depth 1-2 / relu / tanh (models.py:152-162, no reward shaping) under adversarial_imitation_update: gradients, parameters after
  AdamW, u / v buffers, rewards. The gradient-penalty and Mixup draws are fed like the other noise."""
  out = {}
  for name, kw, loss, (lr, wd, gp, ent), rf in gi.GAIL_DEEP_CASES:
    c = gi.gail_deep_case(**kw)
    icfg = DictConfig(state_only=False, spectral_norm=c['spectral_norm'], loss_function=loss, grad_penalty=gp, mixup_alpha=0.7, entropy_bonus=ent, pos_class_prior=0.7,
                      nonnegative_margin=float('inf'),
                      discriminator=DictConfig(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], reward_shaping=False, subtract_log_policy=False, reward_function=rf))
    d, lin = deep_disc(c, icfg)
    out[f'{name}.param_names'] = np.array([n for n, _ in d.named_parameters()])
    opt = torch.optim.AdamW(d.parameters(), lr=lr, weight_decay=wd)
    for i in range(len(c['policy'])):
      d.train()
      orig = torch.distributions.Beta.sample
      if f64 is not None:   # the same update from the same float32 state, evaluated in float64 (gradients only: the twin is discarded)
        d64 = twin64(d)
        feed = [T(c['eps_mix'][i]).double()]
        torch.distributions.Beta.sample = lambda self, *a, **k: feed.pop(0)
        try:
          with NoiseFeed() as nf:
            nf.rand.append(T(c['eps'][i]).double())
            ref_training.adversarial_imitation_update(None, d64, tbatch64(c['policy'][i]), tbatch64(c['expert'][i]), torch.optim.AdamW(d64.parameters(), lr=lr, weight_decay=wd), icfg)
        finally:
          torch.distributions.Beta.sample = orig
        f64[f'{name}.g_{i + 1}'] = np.concatenate([N_(p.grad).ravel() for p in d64.parameters()])
      feed = [T(c['eps_mix'][i])]
      torch.distributions.Beta.sample = lambda self, *a, **k: feed.pop(0)
      try:
        with NoiseFeed() as nf:
          nf.rand.append(T(c['eps'][i]))
          ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      finally:
        torch.distributions.Beta.sample = orig
      d.eval()
      k = i + 1
      out[f'{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()])
      out[f'{name}.p_{k}'] = flat(d)
      if c['spectral_norm']:
        out[f'{name}.sn_{k}'] = np.concatenate([np.concatenate([N_(d.g[li].parametrizations.weight[0]._u), N_(d.g[li].parametrizations.weight[0]._v)]) for li in lin])
      b = c['policy'][i]
      with torch.inference_mode():
        out[f'{name}.reward_{k}'] = N_(d.predict_reward(T(b['states']), T(b['actions'])))
        if f64 is not None:
          f64[f'{name}.reward_{k}'] = N_(twin64(d).predict_reward(T(b['states']).double(), T(b['actions']).double()))
    out[f'{name}.exp_avg'] = opt_state(opt, 'exp_avg')
  if f64 is not None: return
  np.savez_compressed(os.path.join(HERE, 'gail_deep.npz'), **out)


def gen_gail_shaped():
  """GAILDiscriminator with reward_shaping (models.py:152-180) under adversarial_imitation_update: gradients, parameters, u / v buffers, rewards."""
  out = {}
  for name, sn, loss in (('sn_bce', True, 'BCE'), ('plain_pugail', False, 'PUGAIL')):
    c = gi.gail_shaped_case(91, 'hopper', 32, 96, 2, sn)
    icfg = DictConfig(state_only=False, spectral_norm=sn, loss_function=loss, grad_penalty=0.7, mixup_alpha=1, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=float('inf'),
                      discriminator=DictConfig(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
    d = shaped_disc(c, icfg, sn)
    names = [n for n, _ in d.named_parameters()]
    out[f'{name}.param_names'] = np.array(names)
    opt = torch.optim.AdamW(d.parameters(), lr=1e-3, weight_decay=0.1)
    for i in range(len(c['policy'])):
      d.train()
      with NoiseFeed() as nf:
        nf.rand.append(T(c['eps'][i]))
        ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      d.eval()
      k = i + 1
      out[f'{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()])
      out[f'{name}.p_{k}'] = flat(d)
      if sn:
        out[f'{name}.ug_{k}'] = N_(d.g.parametrizations.weight[0]._u); out[f'{name}.vg_{k}'] = N_(d.g.parametrizations.weight[0]._v)
        for li, nm in ((0, '1'), (2, '2')):
          out[f'{name}.u{nm}_{k}'] = N_(d.h[li].parametrizations.weight[0]._u); out[f'{name}.v{nm}_{k}'] = N_(d.h[li].parametrizations.weight[0]._v)
      b = c['policy'][i]
      with torch.inference_mode():
        out[f'{name}.reward_{k}'] = N_(d.predict_reward(T(b['states']), T(b['actions']), T(b['next_states']), T(b['terminals'])))
  np.savez_compressed(os.path.join(HERE, 'gail_shaped.npz'), **out)


def gen_gail_shaped_mixup():
  """Reward shaping under loss_function=Mixup (training.py:104-113: ONE discriminator call on the convex combination of every field of the transitions - the mixed
  terminal is fractional), spectral norm, gradient penalty, entropy bonus: gradients, parameters, u / v buffers, rewards. The Beta draws are fed like the other noise."""
  out = {}
  c = gi.gail_shaped_case(95, 'hopper', 32, 96, 2, True)
  em = gi.mixup_draws(1095, 96, 2)
  icfg = DictConfig(state_only=False, spectral_norm=True, loss_function='Mixup', grad_penalty=0.7, mixup_alpha=0.7, entropy_bonus=0.01, pos_class_prior=0.7, nonnegative_margin=float('inf'),
                    discriminator=DictConfig(hidden_size=c['H'], depth=1, activation='relu', reward_shaping=True, subtract_log_policy=False, reward_function='AIRL'))
  d = shaped_disc(c, icfg, True)
  opt = torch.optim.AdamW(d.parameters(), lr=1e-3, weight_decay=0.1)
  for i in range(2):
    d.train()
    orig = torch.distributions.Beta.sample
    feed = [T(em[i])]
    torch.distributions.Beta.sample = lambda self, *a, **k: feed.pop(0)
    try:
      with NoiseFeed() as nf:
        nf.rand.append(T(c['eps'][i]))
        ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
    finally:
      torch.distributions.Beta.sample = orig
    d.eval()
    k = i + 1
    out[f'g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()]); out[f'p_{k}'] = flat(d)
    out[f'ug_{k}'] = N_(d.g.parametrizations.weight[0]._u); out[f'vg_{k}'] = N_(d.g.parametrizations.weight[0]._v)
    for li, nm in ((0, '1'), (2, '2')):
      out[f'u{nm}_{k}'] = N_(d.h[li].parametrizations.weight[0]._u); out[f'v{nm}_{k}'] = N_(d.h[li].parametrizations.weight[0]._v)
    b = c['policy'][i]
    with torch.inference_mode():
      out[f'reward_{k}'] = N_(d.predict_reward(T(b['states']), T(b['actions']), T(b['next_states']), T(b['terminals'])))
  np.savez_compressed(os.path.join(HERE, 'gail_shaped_mixup.npz'), **out)


def shaped_deep_disc(c, icfg):
  """The reference's reward-shaping GAILDiscriminator for a gail_shaped_deep_case; and the indices of the Linear layers of its potential d.h."""
  d = ref_models.GAILDiscriminator(c['S'], c['A'], icfg, 0.97)
  lin = [2 * l for l in range(c['depth'] + 1)]
  with torch.no_grad():
    if c['spectral_norm']:
      d.g.parametrizations.weight.original.copy_(T(c['Wg'])); d.g.bias.copy_(T(c['bg']))
      d.g.parametrizations.weight[0]._u.copy_(T(c['ug'])); d.g.parametrizations.weight[0]._v.copy_(T(c['vg']))
      for l, li in enumerate(lin):
        d.h[li].parametrizations.weight.original.copy_(T(c['W'][l])); d.h[li].bias.copy_(T(c['b'][l]))
        d.h[li].parametrizations.weight[0]._u.copy_(T(c['u'][l])); d.h[li].parametrizations.weight[0]._v.copy_(T(c['v'][l]))
    else:
      d.g.weight.copy_(T(c['Wg'])); d.g.bias.copy_(T(c['bg']))
      for l, li in enumerate(lin):
        d.h[li].weight.copy_(T(c['W'][l])); d.h[li].bias.copy_(T(c['b'][l]))
  return d, lin


def gen_gail_shaped_deep():
  """Reward shaping with a depth 1-2 / relu / tanh potential (models.py:157-160 with discriminator.depth / activation from conf/hyperparameter_search_space/GAIL.yaml)
  under adversarial_imitation_update: BCE, PUGAIL (infinite and finite margin), Mixup; gradients, parameters after AdamW, u / v buffers, rewards. The first case also
  runs a third update with subtract_log_policy-style logit offsets fed through `log_policy` (models.py:175)."""
  out = {}
  for name, kw, loss, (lr, wd, gp, ent), rf, margin in gi.GAIL_SHAPED_DEEP_CASES:
    c = gi.gail_shaped_deep_case(**kw)
    sn = c['spectral_norm']
    icfg = DictConfig(state_only=c['state_only'], spectral_norm=sn, loss_function=loss, grad_penalty=gp, mixup_alpha=0.7, entropy_bonus=ent, pos_class_prior=0.7, nonnegative_margin=margin,
                      discriminator=DictConfig(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], reward_shaping=True, subtract_log_policy=False, reward_function=rf))
    d, lin = shaped_deep_disc(c, icfg)
    out[f'{name}.param_names'] = np.array([n for n, _ in d.named_parameters()])
    opt = torch.optim.AdamW(d.parameters(), lr=lr, weight_decay=wd)
    for i in range(len(c['policy'])):
      d.train()
      orig = torch.distributions.Beta.sample
      feed = [T(c['eps_mix'][i])]
      torch.distributions.Beta.sample = lambda self, *a, **k: feed.pop(0)
      try:
        with ClampSpy() as spy, NoiseFeed() as nf:
          nf.rand.append(T(c['eps'][i]))
          ref_training.adversarial_imitation_update(None, d, tbatch(c['policy'][i]), tbatch(c['expert'][i]), opt, icfg)
      finally:
        torch.distributions.Beta.sample = orig
      d.eval()
      k = i + 1
      out[f'{name}.g_{k}'] = np.concatenate([N_(p.grad).ravel() for p in d.parameters()]); out[f'{name}.p_{k}'] = flat(d)
      if sn:
        out[f'{name}.sn_{k}'] = np.concatenate([N_(d.g.parametrizations.weight[0]._u), N_(d.g.parametrizations.weight[0]._v)] +
                                               [np.concatenate([N_(d.h[li].parametrizations.weight[0]._u), N_(d.h[li].parametrizations.weight[0]._v)]) for li in lin])
      if loss == 'PUGAIL' and margin != float('inf'):
        out[f'{name}.value_{k}'] = np.array(spy.seen[:1], np.float64)
      b = c['policy'][i]
      with torch.inference_mode():
        out[f'{name}.reward_{k}'] = N_(d.predict_reward(T(b['states']), T(b['actions']), T(b['next_states']), T(b['terminals'])))
        d.subtract_log_policy = True    # models.py:175 (the flag only selects `f - log_policy`): the reward head with an offset, on the same weights
        out[f'{name}.reward_logp_{k}'] = N_(d.predict_reward(T(b['states']), T(b['actions']), T(b['next_states']), T(b['terminals']), log_policy=T(c['logp_policy'][i])))
        d.subtract_log_policy = False
    out[f'{name}.exp_avg'] = opt_state(opt, 'exp_avg')
  np.savez_compressed(os.path.join(HERE, 'gail_shaped_deep.npz'), **out)


# ---------------------------------------------------------------- GMMIL / PWIL
def gen_gmmil():
  out = {}
  for name, (B1, B2, D) in (('small', (64, 48, 24)), ('ant', (256, 256, 120))):
    X, E, w, we = gi.gmmil_case(11, B1, B2, D)
    S = D - 8 if D > 8 else D - 2
    d = ref_models.GMMILDiscriminator(S, D - S, DictConfig(state_only=False))
    args = (T(X[:, :S]), T(X[:, S:]), T(E[:, :S]), T(E[:, S:]), T(w), T(we))
    out[f'{name}_reward_first'] = N_(d.predict_reward(*args))
    out[f'{name}_gammas'] = np.array([d.gamma_1, d.gamma_2], np.float64)
    X2, _, w2, _ = gi.gmmil_case(12, B1, B2, D)
    out[f'{name}_reward_second'] = N_(d.predict_reward(T(X2[:, :S]), T(X2[:, S:]), T(E[:, :S]), T(E[:, S:]), T(w2), T(we)))
    out[f'{name}_sqdist_xe'] = N_(ref_models._squared_distance(T(X), T(E)))[:16]
  np.savez_compressed(os.path.join(HERE, 'gmmil.npz'), **out)


def run_pwil(seed, N, D, steps, Th, A, double=False):
  """PWILDiscriminator.compute_reward over `steps` agent atoms with a reset() every Th (models.py:216-249). double=True: the same code on float64 atoms."""
  atoms, agent = gi.pwil_case(seed, N, D, steps)
  S = D - A
  mem = ref_memory.ReplayMemory(N, S, A, False, transitions=dict(states=T(atoms[:, :S]), actions=T(atoms[:, S:]), rewards=torch.zeros(N), next_states=T(atoms[:, :S]), terminals=torch.zeros(N), timeouts=torch.zeros(N), weights=torch.ones(N), num_trajectories=4))
  cast = (lambda t: t.double()) if double else (lambda t: t)
  if double:
    mem.states, mem.actions = mem.states.double(), mem.actions.double()
  d = ref_models.PWILDiscriminator(S, A, DictConfig(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, Th)
  rewards = []
  for t in range(steps):
    rewards.append(d.compute_reward(cast(T(agent[t:t + 1, :S])), cast(T(agent[t:t + 1, S:]))))
    if t % Th == Th - 1:
      d.reset()
  return d, np.array(rewards, np.float64)


def gen_pwil():
  d, rewards = run_pwil(21, 400, 10, 260, 120, 3)
  np.savez_compressed(os.path.join(HERE, 'pwil.npz'), rewards=np.array(rewards, np.float64), scale=N_(d.data_scale), offset=N_(d.data_offset), remaining=np.array([d.expert_weights.numel()]))


def gen_adril():
  """RewardRelabeller (models.py:293-318) and mix_expert_agent_transitions (models.py:287-290) on seeded batches."""
  S, A, B = gi.DIMS['hopper'][0], gi.DIMS['hopper'][1], 64
  out = {}
  for name, update_freq, balanced in (('adril_balanced', 1250, True), ('adril_halves', 1250, False), ('sqil_balanced', 0, True), ('sqil_halves', 0, False)):
    rel = ref_models.RewardRelabeller(update_freq, balanced)
    for call in range(3):
      pol, exp = gi.adril_batches(40 + call, B, S, A)
      tp, te = tbatch(pol), tbatch(exp)
      rel.resample_and_relabel(tp, te, step=gi.ADRIL_STEP + call * 700, num_trajectories=gi.ADRIL_TRAJ + call, num_expert_trajectories=7)
      for k, v in tp.items():
        out[f'{name}.{call}.{k}'] = N_(v)
  pol, exp = gi.adril_batches(50, B, S, A)
  tp, te = tbatch(pol), tbatch(exp)
  ref_models.mix_expert_agent_transitions(tp, te)
  for k, v in tp.items():
    out[f'mix.{k}'] = N_(v)
  np.savez_compressed(os.path.join(HERE, 'adril.npz'), **out)


def gen_red():
  """REDDiscriminator (models.py:252-284) + target_estimation_update (training.py:68-75): updates (train mode: the predictor's dropout masks are fed in
  module order), set_sigma (still train mode, train.py:128), then `eval()` and predict_reward (train.py:147)."""
  out = {}
  for name, kw, lr, wd in gi.RED_CASES:
    c = gi.red_case(**kw)
    cfg = DictConfig(state_only=False, reward_bandwidth_scale=None,
                     discriminator=DictConfig(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], input_dropout=c['p_in'], dropout=c['p']))
    d = ref_models.REDDiscriminator(c['S'], c['A'], cfg)
    torch.nn.utils.vector_to_parameters(T(c['predictor']), d.predictor.parameters())
    torch.nn.utils.vector_to_parameters(T(c['target']), d.target.parameters())
    opt = torch.optim.AdamW(d.predictor.parameters(), lr=lr, weight_decay=wd)
    for k, (b, masks) in enumerate(zip(c['batches'], c['masks']), 1):
      with DropoutFeed([T(m) for m in masks]):
        ref_training.target_estimation_update(d, tbatch(b), opt)
      out[f'{name}.predictor.{k}'] = flat(d.predictor)
    out[f'{name}.exp_avg'], out[f'{name}.exp_avg_sq'] = opt_state(opt, 'exp_avg'), opt_state(opt, 'exp_avg_sq')
    with torch.inference_mode():
      e = tbatch(c['sigma_batch'])
      with DropoutFeed([T(m) for m in c['sigma_masks']]):
        d.set_sigma(e['states'], e['actions'])
      out[f'{name}.sigma_1'] = np.array([d.sigma_1], np.float64)
      d.eval()
      q = tbatch(c['query'])
      out[f'{name}.reward'] = N_(d.predict_reward(q['states'], q['actions']))
    out[f'{name}.hyper'] = np.array([lr, wd], np.float64)
  np.savez_compressed(os.path.join(HERE, 'red.npz'), **out)


class DropoutFeed:
  """Feeds pre-drawn keep-masks to F.dropout (nn.Dropout.forward), in call order, with ATen's arithmetic (mask / (1 - p), then multiply)."""

  def __init__(self, masks):
    self.masks, self._orig = list(masks), torch.nn.functional.dropout

  def __enter__(self):
    def dropout(input, p=0.5, training=True, inplace=False):
      if not training or p == 0: return input
      return input * (self.masks.pop(0) / (1 - p))
    torch.nn.functional.dropout = dropout
    return self

  def __exit__(self, *a):
    torch.nn.functional.dropout = self._orig
    assert not self.masks, 'unused dropout masks'


def gen_dril(f64=None):
  """SoftActor with the DRIL discriminator config (models.py:84-120) in train mode: BC updates, uncertainty, threshold, reward; depth 1-2, tanh / relu."""
  out = {}
  for name, kw, lr, wd in gi.DRIL_CASES:
    c = gi.dril_case(**kw)
    cfg = DictConfig(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], input_dropout=c['p_in'], dropout=c['p'])
    d = ref_models.SoftActor(c['S'], c['A'], cfg)
    lin = [1 + 3 * l for l in range(c['depth'] + 1)]   # Dropout, [Linear, Dropout, act] x depth, Linear
    assert [k for k in d.state_dict()] == [f'actor.{i}.{p}' for i in lin for p in ('weight', 'bias')]
    torch.nn.utils.vector_to_parameters(T(c['params']), d.parameters())
    opt = torch.optim.AdamW(d.parameters(), lr=lr, weight_decay=wd)
    for k, b in enumerate(c['batches'], 1):
      with DropoutFeed([T(m) for m in gi.dril_masks(c, 'm', k - 1)]):
        ref_training.behavioural_cloning_update(d, tbatch(b), opt)
      out[f'{name}.params.{k}'] = flat(d)
    out[f'{name}.exp_avg'] = opt_state(opt, 'exp_avg')
    with torch.inference_mode():
      e, q = tbatch(c['expert']), tbatch(c['query'])
      em, qm = [T(m) for m in gi.dril_masks(c, 'e_m')], [T(m) for m in gi.dril_masks(c, 'q_m')]
      with DropoutFeed(list(em)):
        out[f'{name}.expert_uncertainty'] = N_(d._get_action_uncertainty(e['states'], e['actions']))
      with DropoutFeed(list(em)):
        d.set_uncertainty_threshold(e['states'], e['actions'], 0.9)
      out[f'{name}.q'] = np.array([d.q], np.float64)
      with DropoutFeed(list(qm)):
        out[f'{name}.reward'] = N_(d.predict_reward(q['states'], q['actions']))
      with DropoutFeed(list(qm)):
        out[f'{name}.query_uncertainty'] = N_(d._get_action_uncertainty(q['states'], q['actions']))
      if f64 is not None:
        d64 = twin64(d)
        for tag, bb, mm in (('expert', e, em), ('query', q, qm)):
          with DropoutFeed([m.double() for m in mm]):
            f64[f'{name}.{tag}_uncertainty'] = N_(d64._get_action_uncertainty(bb['states'].double(), bb['actions'].double()))
    out[f'{name}.hyper'] = np.array([lr, wd], np.float64)
  if f64 is not None: return
  np.savez_compressed(os.path.join(HERE, 'dril.npz'), **out)


def gen_dataset():
  """D4RLEnv.get_dataset (environments.py:63-125) on a raw D4RL-format dataset. environments.py itself cannot be imported (gym, d4rl), so the
  method's source is extracted with `ast` and executed unmodified on a stand-in object that has the two attributes it reads."""
  import ast, types
  src = open(os.path.join(REF, 'environments.py')).read()
  fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == 'get_dataset')
  ns = dict(torch=torch, np=np, ReplayMemory=ref_memory.ReplayMemory)
  exec(compile(ast.Module(body=[fn], type_ignores=[]), 'environments.py', 'exec'), ns)
  raw = gi.raw_d4rl_dataset(81)
  out = {}
  for absorbing in (True, False):
    for subsample in (1, 3):
      for trajectories in (0, 2):
        env = types.SimpleNamespace(dataset={k: v.copy() for k, v in raw.items()}, absorbing=absorbing)
        np.random.seed(17)
        mem = ns['get_dataset'](env, trajectories=trajectories, subsample=subsample)
        tag = f'abs{int(absorbing)}_sub{subsample}_traj{trajectories}'
        for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights', 'step'):
          out[f'{tag}.{k}'] = N_(getattr(mem, k))
        out[f'{tag}.meta'] = np.array([mem.num_trajectories, mem.idx, int(mem.full), len(mem)], np.int64)
  np.savez_compressed(os.path.join(HERE, 'dataset.npz'), **out)


# ---------------------------------------------------------------- the sizes bench.py times
TIMED_GAIL = dict(lr=0.0002778119723405689, weight_decay=8.46588535234332, grad_penalty=0.2799364347010851, entropy_bonus=0.24145587952807546)  # conf/optimised_hyperparameters/GAIL_5_trajectories.yaml


def timed_gail_eps_mix():
  return np.random.RandomState(4036).uniform(size=1024).astype(np.float32)   # mixup_alpha = 1 (conf/algorithm/GAIL.yaml): Beta(1, 1) = U(0, 1)


def gen_timed_sizes():
  """Reference outputs at the sizes the benchmarks run: PWIL against N = 25,000 atoms, D = 24, T = 1000 over 1,100 steps including a
  reset(); GMMIL.predict_reward at B = 1024, D = 120 (BASELINE.json configs[3]), the full reward vector; one adversarial_imitation_update at B = 1024 with
  the tuned GAIL_5 hyper-parameters (Mixup, spectral norm, gradient penalty, entropy bonus)."""
  out = {}
  d, rewards = run_pwil(22, 25000, 24, 1100, 1000, 6)
  out['pwil25k.rewards'], out['pwil25k.remaining'] = rewards, np.array([d.expert_weights.numel()])
  _, out['pwil25k.rewards_f64'] = run_pwil(22, 25000, 24, 1100, 1000, 6, double=True)

  X, E, w, we = gi.gmmil_case(13, 1024, 1024, 120)
  S = 112
  g = ref_models.GMMILDiscriminator(S, 8, DictConfig(state_only=False))
  args = (T(X[:, :S]), T(X[:, S:]), T(E[:, :S]), T(E[:, S:]), T(w), T(we))
  out['gmmil1024.reward_first'] = N_(g.predict_reward(*args))
  out['gmmil1024.gammas'] = np.array([g.gamma_1, g.gamma_2], np.float64)
  X2, _, w2, _ = gi.gmmil_case(14, 1024, 1024, 120)
  out['gmmil1024.reward_second'] = N_(g.predict_reward(T(X2[:, :S]), T(X2[:, S:]), args[2], args[3], T(w2), args[5]))
  g64 = ref_models.GMMILDiscriminator(S, 8, DictConfig(state_only=False))
  g64.gamma_1, g64.gamma_2 = g.gamma_1, g.gamma_2   # same frozen bandwidths; the kernel sums in float64
  r64 = []
  for i in range(0, 1024, 128):   # row blocks: [128, 1024, 120] float64 temporaries instead of [1024, 1024, 120]; the self term needs all rows -> weights of the full set
    wn, wen = (T(w2) / T(w2).sum()).double(), (T(we) / T(we).sum()).double()
    xb, xall, eall = T(X2[i:i + 128]).double(), T(X2).double(), T(E).double()
    sim = sum((wn[i:i + 128, None] * torch.exp(-gam * ref_models._squared_distance(xb, eall)) * wen[None, :]).sum(1) for gam in (g.gamma_1, g.gamma_2))
    slf = sum((wn[i:i + 128, None] * torch.exp(-gam * ref_models._squared_distance(xb, xall)) * wn[None, :]).sum(1) for gam in (g.gamma_1, g.gamma_2))
    r64.append(N_(sim - slf))
  out['gmmil1024.reward_second_f64'] = np.concatenate(r64)

  c = gi.gail_case(36, env='halfcheetah', hidden=64, batch=1024, steps=1)
  dd, icfg = build_disc(c)
  icfg.update(loss_function='Mixup', grad_penalty=TIMED_GAIL['grad_penalty'], mixup_alpha=1, entropy_bonus=TIMED_GAIL['entropy_bonus'], pos_class_prior=0.7, nonnegative_margin=float('inf'))
  opt = torch.optim.AdamW(dd.parameters(), lr=TIMED_GAIL['lr'], weight_decay=TIMED_GAIL['weight_decay'])
  dd.train()
  feed, orig = [T(timed_gail_eps_mix())], torch.distributions.Beta.sample
  torch.distributions.Beta.sample = lambda self, *a, **k: feed.pop(0)
  try:
    with NoiseFeed() as nf:
      nf.rand.append(T(c['eps'][0]))
      ref_training.adversarial_imitation_update(None, dd, tbatch(c['policy'][0]), tbatch(c['expert'][0]), opt, icfg)
  finally:
    torch.distributions.Beta.sample = orig
  dd.eval()
  out['gail1024.g_1'] = np.concatenate([N_(p.grad).ravel() for p in dd.parameters()]); out['gail1024.p_1'] = flat(dd)
  for li, nm in ((0, '1'), (2, '2')):
    out[f'gail1024.u{nm}_1'] = N_(dd.g[li].parametrizations.weight[0]._u); out[f'gail1024.v{nm}_1'] = N_(dd.g[li].parametrizations.weight[0]._v)
  b = c['policy'][0]
  with torch.inference_mode():
    out['gail1024.reward_1'] = N_(dd.predict_reward(T(b['states']), T(b['actions'])))
    out['gail1024.reward_1_f64'] = N_(twin64(dd).predict_reward(T(b['states']).double(), T(b['actions']).double()))
  np.savez_compressed(os.path.join(HERE, 'timed_sizes.npz'), **out)


def gen_f64():
  """Float64 evaluations of the reference at the comparison sites whose float32 tolerance the GPU tests widen beyond rtol 1e-5 (rewards = a difference of
  logs near D = 1/2, log pi of actions at the clamp, second-order gradients of deep discriminators, exp(-1000 cost), a variance of 5 exponentials): the tests
  then bound |hip - f64| by the reference's own |f32 - f64| instead of a hand-picked number."""
  f64 = {}
  gen_gail('gail_default', gi.gail_case(31), lr=3e-5, weight_decay=10, grad_penalty=1.0, entropy_bonus=0.0, f64=f64)
  gen_gail('gail_h128_ent', gi.gail_case(32, hidden=128), lr=7.3e-5, weight_decay=6.35, grad_penalty=0.32, entropy_bonus=0.0155, f64=f64)
  gen_gail('gail_nosn_nogp', gi.gail_case(33, env='hopper', hidden=32, batch=128, spectral_norm=False), lr=3e-4, weight_decay=0.0, grad_penalty=0.0, entropy_bonus=0.0, f64=f64)
  gen_bc('bc_hopper', 'hopper', 256, 256, 3, f64=f64)
  gen_gail_deep(f64=f64)
  gen_dril(f64=f64)
  _, f64['pwil.rewards'] = run_pwil(21, 400, 10, 260, 120, 3, double=True)
  np.savez_compressed(os.path.join(HERE, 'f64_brackets.npz'), **f64)


if __name__ == '__main__':
  only = set(sys.argv[1:])  # e.g. `make_golden.py adril` regenerates just that fixture
  want = lambda tag: not only or tag in only
  if want('replay'): gen_replay()
  if want('sac'):
    gen_sac('sac_halfcheetah', gi.sac_case(3, 'halfcheetah', 256, 256, 3))
    gen_sac('sac_hopper_h64', gi.sac_case(4, 'hopper', 64, 96, 3))
    gen_sac('sac_ant_b64', gi.sac_case(5, 'ant', 256, 64, 2))
  if want('sac_general'):
    for name, kw in gi.GENERAL_SAC_CASES.items(): gen_sac_general(name, kw)
  if want('bc'): gen_bc('bc_hopper', 'hopper', 256, 256, 3)
  if want('gail'):
    gen_gail('gail_default', gi.gail_case(31), lr=3e-5, weight_decay=10, grad_penalty=1.0, entropy_bonus=0.0)
    gen_gail('gail_h128_ent', gi.gail_case(32, hidden=128), lr=7.3e-5, weight_decay=6.35, grad_penalty=0.32, entropy_bonus=0.0155)
    gen_gail('gail_nosn_nogp', gi.gail_case(33, env='hopper', hidden=32, batch=128, spectral_norm=False), lr=3e-4, weight_decay=0.0, grad_penalty=0.0, entropy_bonus=0.0)
  if want('gail_variants'): gen_gail_variants()
  if want('gail_pu_margin'): gen_gail_pu_margin()
  if want('gail_pu_margin_general'): gen_gail_pu_margin_general()
  if want('gail_shaped'): gen_gail_shaped()
  if want('gail_shaped_mixup'): gen_gail_shaped_mixup()
  if want('gail_shaped_deep'): gen_gail_shaped_deep()
  if want('gail_deep'): gen_gail_deep()
  if want('gmmil'): gen_gmmil()
  if want('pwil'): gen_pwil()
  if want('adril'): gen_adril()
  if want('red'): gen_red()
  if want('dril'): gen_dril()
  if want('dataset'): gen_dataset()
  if want('timed_sizes'): gen_timed_sizes()
  if want('f64'): gen_f64()
