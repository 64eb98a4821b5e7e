"""Deterministic synthetic inputs shared by `make_golden.py` (which feeds them to the
reference) and by the tests (which feed them to the oracle / the HIP path).

Everything comes from numpy's legacy `RandomState(seed)` stream, which is frozen
across numpy versions, so the inputs never have to be stored in the fixtures.
Shapes follow SURVEY.md §8(d): D4RL-shaped transitions with an absorbing bit.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32

DIMS = {  # state size incl. absorbing bit, action size (environments.py:27, gym MuJoCo dims)
    'halfcheetah': (18, 6), 'walker2d': (18, 6), 'hopper': (12, 3), 'ant': (112, 8),
    'wide': (21, 12),   # not a gym task: an action space wider than the fused head (2A > 16) for the general-shape kernels
}


def mlp_params(rs, in_dim, hidden, depth, out_dim, out_scale=1.0):
  """Flat parameter vector in torch order, fan-in scaled weights, small non-zero biases."""
  dims = [in_dim] + [hidden] * depth + [out_dim]
  parts = []
  for i in range(len(dims) - 1):
    scale = (2.0 if i < len(dims) - 2 else out_scale) / np.sqrt(dims[i])
    parts.append((rs.standard_normal((dims[i + 1], dims[i])) * scale).astype(f32).ravel())
    parts.append((rs.standard_normal(dims[i + 1]) * 0.05).astype(f32))
  return np.concatenate(parts)


def transitions(rs, n, S, A, *, state_shift=0.0, absorbing_frac=0.02, terminal_frac=0.01, weighted=False):
  """n D4RL-shaped rows (fields of memory.py:17) + the `absorbing` flag of memory.py:62."""
  st = (rs.standard_normal((n, S)) + state_shift).astype(f32)
  nx = (rs.standard_normal((n, S)) + state_shift).astype(f32)
  st[:, -1] = 0; nx[:, -1] = 0
  ac = rs.uniform(-1, 1, (n, A)).astype(f32)
  n_abs = int(round(n * absorbing_frac))
  if n_abs:
    rows = rs.choice(n, n_abs, replace=False)
    st[rows] = 0; st[rows, -1] = 1; nx[rows] = 0; nx[rows, -1] = 1; ac[rows] = 0
  out = dict(
      step=np.arange(1, n + 1, dtype=f32), states=st, actions=ac, rewards=rs.standard_normal(n).astype(f32), next_states=nx,
      terminals=(rs.uniform(size=n) < terminal_frac).astype(f32), timeouts=np.zeros(n, f32),
      weights=(rs.uniform(0.5, 1.5, n).astype(f32) if weighted else np.ones(n, f32)))
  out['absorbing'] = st[:, -1].copy()
  return out


def sac_case(seed, env='halfcheetah', hidden=256, batch=256, steps=3, depth=2, activation='relu', critic=None):
  """critic: (hidden, depth, activation) of the twin critics when they differ from the actor's (reinforcement.critic of the reference's configuration)."""
  S, A = DIMS[env]
  ch, cd, ca = critic or (hidden, depth, activation)
  rs = np.random.RandomState(seed)
  actor = mlp_params(rs, S, hidden, depth, 2 * A, out_scale=0.3)
  crit = np.concatenate([mlp_params(rs, S + A, ch, cd, 1) for _ in range(2)])
  target = (crit + rs.standard_normal(crit.size).astype(f32) * f32(0.01)).astype(f32)
  log_alpha = np.array([-0.3], f32)
  batches = [transitions(rs, batch, S, A, weighted=True) for _ in range(steps)]
  eps_next = [rs.standard_normal((batch, A)).astype(f32) for _ in range(steps)]
  eps_cur = [rs.standard_normal((batch, A)).astype(f32) for _ in range(steps)]
  return dict(S=S, A=A, H=hidden, B=batch, depth=depth, activation=activation, critic_hidden=ch, critic_depth=cd, critic_activation=ca, actor=actor, critic=crit, target=target,
              log_alpha=log_alpha, batches=batches, eps_next=eps_next, eps_cur=eps_cur, discount=0.97, entropy_target=-0.5 * A, polyak=0.99, lr=3e-4, weight_decay=0.0)


# General actor / critic shapes (models.py:48-69 `_create_fcnn`: any depth, relu / tanh / sigmoid) - the shapes outside the fused kernels. name -> sac_case arguments
GENERAL_SAC_CASES = {
    'sac_general_d3_tanh': dict(seed=41, env='hopper', hidden=48, batch=96, steps=3, depth=3, activation='tanh'),
    'sac_general_d1_sigmoid': dict(seed=42, env='halfcheetah', hidden=80, batch=64, steps=3, depth=1, activation='sigmoid'),
    'sac_general_d2_relu_h320': dict(seed=43, env='walker2d', hidden=320, batch=40, steps=2, depth=2, activation='relu'),
    'sac_general_wide_action': dict(seed=45, env='wide', hidden=128, batch=48, steps=2, depth=2, activation='relu'),   # the fused SHAPE, but 2A = 24 outputs: general kernels
    'sac_general_mixed': dict(seed=44, env='hopper', hidden=32, batch=50, steps=3, depth=1, activation='relu', critic=(72, 3, 'sigmoid')),   # reinforcement.actor != reinforcement.critic
}


def gail_case(seed, env='halfcheetah', hidden=64, batch=256, steps=3, spectral_norm=True):
  S, A = DIMS[env]
  D = S + A
  rs = np.random.RandomState(seed)
  W1 = (rs.standard_normal((hidden, D)) * np.sqrt(2.0 / D)).astype(f32)
  b1 = (rs.standard_normal(hidden) * 0.05).astype(f32)
  W2 = (rs.standard_normal((1, hidden)) / np.sqrt(hidden)).astype(f32)
  b2 = (rs.standard_normal(1) * 0.05).astype(f32)
  unit = lambda x: (x / np.linalg.norm(x)).astype(f32)
  u1, v1, u2, v2 = unit(rs.standard_normal(hidden)), unit(rs.standard_normal(D)), unit(rs.standard_normal(1)), unit(rs.standard_normal(hidden))
  pol = [transitions(rs, batch, S, A, weighted=True) for _ in range(steps)]
  exp = [transitions(rs, batch, S, A, state_shift=0.5, weighted=True) for _ in range(steps)]
  eps = [rs.uniform(size=batch).astype(f32) for _ in range(steps)]
  return dict(S=S, A=A, D=D, H=hidden, B=batch, W1=W1, b1=b1, W2=W2, b2=b2, u1=u1, v1=v1, u2=u2, v2=v2, policy=pol, expert=exp, eps=eps,
              spectral_norm=spectral_norm)


def gail_deep_case(seed, env, hidden, batch, steps, depth=2, activation='tanh', spectral_norm=True):
  """GAIL discriminator of any `_create_fcnn` shape: weights / biases / u / v per layer (hidden layers then the H -> 1 output), policy + expert batches,
  the U(0,1) draws of the gradient penalty and of Mixup, and log pi offsets for subtract_log_policy."""
  S, A = DIMS[env]
  D = S + A
  rs = np.random.RandomState(seed)
  dims = [D] + [hidden] * depth + [1]
  unit = lambda x: (x / np.linalg.norm(x)).astype(f32)
  W = [(rs.standard_normal((dims[i + 1], dims[i])) * np.sqrt((2.0 if i < depth else 1.0) / dims[i])).astype(f32) for i in range(depth + 1)]
  b = [(rs.standard_normal(dims[i + 1]) * 0.05).astype(f32) for i in range(depth + 1)]
  u = [unit(rs.standard_normal(dims[i + 1])) for i in range(depth + 1)]
  v = [unit(rs.standard_normal(dims[i])) for i in range(depth + 1)]
  pol = [transitions(rs, batch, S, A, weighted=True) for _ in range(steps)]
  exp = [transitions(rs, batch, S, A, state_shift=0.5, weighted=True) for _ in range(steps)]
  eps = [rs.uniform(size=batch).astype(f32) for _ in range(steps)]
  eps_mix = [rs.uniform(size=batch).astype(f32) for _ in range(steps)]
  return dict(S=S, A=A, D=D, H=hidden, B=batch, depth=depth, activation=activation, spectral_norm=spectral_norm, W=W, b=b, u=u, v=v, policy=pol, expert=exp, eps=eps, eps_mix=eps_mix)


GAIL_DEEP_CASES = (   # name, gail_deep_case arguments, loss_function, (lr, weight decay, grad_penalty, entropy_bonus), reward_function
    ('hopper_d2_tanh_sn', dict(seed=101, env='hopper', hidden=32, batch=96, steps=2, depth=2, activation='tanh', spectral_norm=True), 'BCE', (1e-3, 0.1, 0.6, 0.02), 'AIRL'),
    ('halfcheetah_d2_relu', dict(seed=102, env='halfcheetah', hidden=64, batch=64, steps=2, depth=2, activation='relu', spectral_norm=False), 'PUGAIL', (5e-4, 1.0, 1.0, 0.0), 'GAIL'),
    ('walker2d_d1_tanh_sn', dict(seed=103, env='walker2d', hidden=64, batch=80, steps=2, depth=1, activation='tanh', spectral_norm=True), 'Mixup', (1e-3, 0.0, 0.3, 0.05), 'FAIRL'),
    ('hopper_d2_relu_sn', dict(seed=104, env='hopper', hidden=32, batch=64, steps=2, depth=2, activation='relu', spectral_norm=True), 'BCE', (1e-3, 0.1, 1.0, 0.0), 'AIRL'),
)


def gail_shaped_deep_case(seed, env, hidden, batch, steps, depth=2, activation='tanh', spectral_norm=True, state_only=False):
  """Reward-shaping discriminator with a shaping potential of any `_create_fcnn` shape: g = Linear(Dg, 1), h = [Linear - act] x depth - Linear(H, 1) on the state,
  spectral-norm buffers per Linear, batches with ~30 % terminals, the U(0,1) draws of the gradient penalty, Beta draws for Mixup and log pi offsets."""
  S, A = DIMS[env]
  Dg = S if state_only else S + A
  rs = np.random.RandomState(seed)
  dims = [S] + [hidden] * depth + [1]
  unit = lambda x: (x / np.linalg.norm(x)).astype(f32)
  c = dict(S=S, A=A, Dg=Dg, H=hidden, B=batch, depth=depth, activation=activation, spectral_norm=spectral_norm, state_only=state_only,
           Wg=(rs.standard_normal((1, Dg)) / np.sqrt(Dg)).astype(f32), bg=(rs.standard_normal(1) * 0.05).astype(f32),
           ug=unit(rs.standard_normal(1)), vg=unit(rs.standard_normal(Dg)),
           W=[(rs.standard_normal((dims[i + 1], dims[i])) * np.sqrt((2.0 if i < depth else 1.0) / dims[i])).astype(f32) for i in range(depth + 1)],
           b=[(rs.standard_normal(dims[i + 1]) * 0.05).astype(f32) for i in range(depth + 1)],
           u=[unit(rs.standard_normal(dims[i + 1])) for i in range(depth + 1)], v=[unit(rs.standard_normal(dims[i])) for i in range(depth + 1)])
  c['policy'] = [transitions(rs, batch, S, A, weighted=True, terminal_frac=0.3) for _ in range(steps)]
  c['expert'] = [transitions(rs, batch, S, A, state_shift=0.5, weighted=True, terminal_frac=0.3) for _ in range(steps)]
  c['eps'] = [rs.uniform(size=batch).astype(f32) for _ in range(steps)]
  c['eps_mix'] = [rs.beta(0.7, 0.7, size=batch).astype(f32) for _ in range(steps)]
  c['logp_policy'] = [(rs.standard_normal(batch) * 0.5 - 1.0).astype(f32) for _ in range(steps)]
  c['logp_expert'] = [(rs.standard_normal(batch) * 0.5 - 1.0).astype(f32) for _ in range(steps)]
  return c


GAIL_SHAPED_DEEP_CASES = (   # name, gail_shaped_deep_case arguments, loss_function, (lr, weight decay, grad_penalty, entropy_bonus), reward_function, nonnegative_margin
    ('hopper_d2_tanh_sn', dict(seed=121, env='hopper', hidden=32, batch=96, steps=2, depth=2, activation='tanh', spectral_norm=True), 'BCE', (1e-3, 0.1, 0.7, 0.01), 'AIRL', float('inf')),
    ('halfcheetah_d2_relu', dict(seed=122, env='halfcheetah', hidden=64, batch=64, steps=2, depth=2, activation='relu', spectral_norm=False), 'PUGAIL', (5e-4, 1.0, 1.0, 0.0), 'GAIL', float('inf')),
    ('walker2d_d1_tanh_sn', dict(seed=123, env='walker2d', hidden=64, batch=80, steps=2, depth=1, activation='tanh', spectral_norm=True), 'Mixup', (1e-3, 0.0, 0.3, 0.05), 'FAIRL', float('inf')),
    ('hopper_d2_relu_sn_margin', dict(seed=124, env='hopper', hidden=32, batch=72, steps=2, depth=2, activation='relu', spectral_norm=True), 'PUGAIL', (1e-3, 0.1, 0.5, 0.02), 'AIRL', 0.02),
    ('hopper_d2_tanh_sn_state_only', dict(seed=125, env='hopper', hidden=32, batch=64, steps=2, depth=2, activation='tanh', spectral_norm=True, state_only=True), 'BCE', (1e-3, 0.1, 0.0, 0.01), 'AIRL',
     float('inf')),   # state_only: the reference's gradient penalty differentiates w.r.t. an action the discriminator never saw and raises, so grad_penalty = 0 here
)


def mixup_draws(seed, batch, steps):
  """The Beta(alpha, alpha) coefficients of `steps` Mixup updates (training.py:106), fed to the reference and to the HIP path alike."""
  rs = np.random.RandomState(seed)
  return [rs.beta(0.7, 0.7, size=batch).astype(f32) for _ in range(steps)]


def gmmil_case(seed, B1, B2, D, weighted=True):
  rs = np.random.RandomState(seed)
  X = rs.standard_normal((B1, D)).astype(f32)
  E = (rs.standard_normal((B2, D)) * 0.8 + 0.5).astype(f32)
  w = rs.uniform(0.5, 1.5, B1).astype(f32) if weighted else np.ones(B1, f32)
  we = rs.uniform(0.5, 1.5, B2).astype(f32) if weighted else np.ones(B2, f32)
  return X, E, w, we


def pwil_case(seed, N, D, steps):
  rs = np.random.RandomState(seed)
  atoms = (rs.standard_normal((N, D)) * rs.uniform(0.5, 2.0, D) + rs.standard_normal(D)).astype(f32)
  atoms[:, -1] = 0  # a constant feature: std == 0 -> scale 1 (models.py:207)
  agent = (rs.standard_normal((steps, D)) * 1.2).astype(f32)
  agent[:, -1] = 0
  return atoms, agent


def strided(x, stride=29):
  return np.ascontiguousarray(np.asarray(x).ravel()[::stride])


ADRIL_STEP, ADRIL_TRAJ = 2600, 4


def adril_batches(seed, B, S, A):
  """(policy batch, expert batch) for the AdRIL / SQIL relabeller: policy rows carry env-step stamps spread over three rounds of 1250."""
  rs = np.random.RandomState(seed)
  pol, exp = transitions(rs, B, S, A), transitions(rs, B, S, A, state_shift=0.5, weighted=True)
  pol['step'] = rs.randint(1, 3800, B).astype(f32)
  return pol, exp


def red_case(seed, env, hidden, batch, steps, depth=1, activation='relu', p_in=0.0, p=0.0):
  """RED predictor / frozen target (`_create_fcnn`: D -> H (-> H) -> D, ReLU / Tanh), weighted expert batches, a sigma batch and a query batch; with
  dropout also the predictor's keep-masks per update (input, hidden 1[, hidden 2]) and for the train-mode set_sigma forward."""
  S, A = DIMS[env]
  D = S + A
  rs = np.random.RandomState(seed)
  predictor, target = mlp_params(rs, D, hidden, depth, D), mlp_params(rs, D, hidden, depth, D)
  batches = [transitions(rs, batch, S, A, state_shift=0.5, weighted=True) for _ in range(steps)]
  c = dict(S=S, A=A, D=D, H=hidden, B=batch, depth=depth, activation=activation, p_in=p_in, p=p, predictor=predictor, target=target, batches=batches,
           sigma_batch=transitions(rs, batch, S, A, state_shift=0.5), query=transitions(rs, batch + 16, S, A))
  keep = lambda shape, pr: (rs.uniform(size=shape) >= pr).astype(f32)   # drawn after everything else: the dropout-free cases stay what they were
  masks = lambda n: ([keep((n, D), p_in)] if p_in > 0 else []) + ([keep((n, hidden), p) for _ in range(depth)] if p > 0 else [])
  c['masks'] = [masks(batch) for _ in range(steps)]
  c['sigma_masks'] = masks(batch)
  return c


RED_CASES = (   # name, red_case arguments, lr, weight decay; the last three mirror conf/optimised_hyperparameters/RED_{5,10,25}_trajectories.yaml
    ('hopper_h32', dict(seed=61, env='hopper', hidden=32, batch=64, steps=4), 3e-5, 0.0),
    ('halfcheetah_h64', dict(seed=62, env='halfcheetah', hidden=64, batch=256, steps=3), 1e-3, 0.01),
    ('hopper_d2_relu_drop', dict(seed=63, env='hopper', hidden=32, batch=96, steps=3, depth=2, activation='relu', p_in=0.1, p=0.3), 1e-3, 0.5),
    ('halfcheetah_d1_tanh_drop', dict(seed=64, env='halfcheetah', hidden=32, batch=64, steps=3, depth=1, activation='tanh', p_in=0.2, p=0.1), 5e-4, 0.0),
    ('hopper_d2_tanh_drop', dict(seed=65, env='hopper', hidden=64, batch=128, steps=3, depth=2, activation='tanh', p_in=0.05, p=0.4), 2.4e-4, 2.5),
)


def dril_case(seed, env, hidden, batch, steps, p_in=0.1, p=0.1, depth=1, activation='tanh'):
  """DRIL policy ensemble (`_create_fcnn`: Dropout-Linear(S,H)-Dropout-act(-Linear(H,H)-Dropout-act)-Linear(H,2A)), expert batches with their dropout
  keep-masks, and the masks of the 5-member Monte-Carlo ensemble for an expert set and a query set (rows in repeat_interleave order)."""
  S, A = DIMS[env]
  rs = np.random.RandomState(seed)
  params = mlp_params(rs, S, hidden, depth, 2 * A, out_scale=0.3)
  keep = lambda shape, pr: (rs.uniform(size=shape) >= pr).astype(f32)
  batches = [transitions(rs, batch, S, A, state_shift=0.5, weighted=True) for _ in range(steps)]
  for b in batches:
    b['actions'] = np.clip(b['actions'], -0.97, 0.97).astype(f32); b['actions'][:2] = np.array([1.0, -1.0], f32)[:, None]  # exercise the clamp
  expert, query = transitions(rs, 80, S, A, state_shift=0.5), transitions(rs, 37, S, A)
  c = dict(S=S, A=A, H=hidden, B=batch, p_in=p_in, p=p, depth=depth, activation=activation, params=params, batches=batches, m0=[keep((batch, S), p_in) for _ in range(steps)],
           m1=[keep((batch, hidden), p) for _ in range(steps)], expert=expert, query=query, e_m0=keep((80 * 5, S), p_in), e_m1=keep((80 * 5, hidden), p),
           q_m0=keep((37 * 5, S), p_in), q_m1=keep((37 * 5, hidden), p))
  if depth == 2:   # drawn last: the depth-1 cases stay what they were
    c.update(m2=[keep((batch, hidden), p) for _ in range(steps)], e_m2=keep((80 * 5, hidden), p), q_m2=keep((37 * 5, hidden), p))
  return c


def dril_masks(c, which, k=None):
  """The keep-masks of one call in module order: which = 'm' (update k), 'e_m' (expert set) or 'q_m' (query set)."""
  names = [f'{which}{i}' for i in range(1 + c['depth'])]
  return [c[n][k] if k is not None else c[n] for n in names]


DRIL_CASES = (   # name, dril_case arguments, lr, weight decay; the last two mirror conf/optimised_hyperparameters/DRIL_{10,25}_trajectories.yaml (depth 2, relu)
    ('hopper_h64', dict(seed=71, env='hopper', hidden=64, batch=64, steps=3), 3e-5, 0.0),
    ('halfcheetah_h32', dict(seed=72, env='halfcheetah', hidden=32, batch=128, steps=2, p_in=0.2, p=0.3), 1e-3, 0.01),
    ('hopper_d2_relu', dict(seed=74, env='hopper', hidden=32, batch=96, steps=3, p_in=0.4, p=0.55, depth=2, activation='relu'), 2.7e-4, 5.4),
    ('walker2d_d2_tanh', dict(seed=75, env='walker2d', hidden=64, batch=64, steps=2, p_in=0.1, p=0.2, depth=2, activation='tanh'), 1e-3, 0.0),
    ('halfcheetah_d1_relu', dict(seed=76, env='halfcheetah', hidden=128, batch=64, steps=2, p_in=0.05, p=0.3, depth=1, activation='relu'), 5e-4, 0.1),
)


def gail_extras(seed, c):
  """Extra inputs of the GAIL loss variants: Beta(alpha, alpha)-like mixup draws per step and an actor (for subtract_log_policy)."""
  rs = np.random.RandomState(seed + 1000)
  steps, B = len(c['policy']), c['B']
  return dict(eps_mix=[rs.beta(0.7, 0.7, B).astype(f32) for _ in range(steps)], actor=mlp_params(rs, c['S'], 64, 2, 2 * c['A'], out_scale=0.3))


def raw_d4rl_dataset(seed, obs_dim=5, act_dim=2):
  """A raw D4RL-format dataset (flat arrays + terminals / timeouts flags): 5 trajectories of different lengths, ended by true termination,
  timeout, termination, timeout, termination, plus a dangling tail without an end flag (dropped by the trajectory split)."""
  rs = np.random.RandomState(seed)
  lengths, ends = [7, 9, 4, 12, 6], ['terminal', 'timeout', 'terminal', 'timeout', 'terminal']
  n = sum(lengths) + 3
  d = dict(observations=rs.standard_normal((n, obs_dim)).astype(f32), actions=rs.uniform(-1, 1, (n, act_dim)).astype(f32),
           next_observations=rs.standard_normal((n, obs_dim)).astype(f32), terminals=np.zeros(n, f32), timeouts=np.zeros(n, f32))
  pos = 0
  for L, e in zip(lengths, ends):
    pos += L
    d['terminals' if e == 'terminal' else 'timeouts'][pos - 1] = 1
  return d


def gail_shaped_case(seed, env, hidden, batch, steps, spectral_norm):
  """Reward-shaping discriminator: g = Linear(S+A, 1), h = Linear(S, H)-ReLU-Linear(H, 1), spectral-norm buffers, batches with ~30 % terminals so that
  the (1 - terminal) factor is exercised, gradient-penalty draws."""
  S, A = DIMS[env]
  rs = np.random.RandomState(seed)
  D = S + A
  unit = lambda x: (x / np.linalg.norm(x)).astype(f32)
  c = dict(S=S, A=A, H=hidden, B=batch, spectral_norm=spectral_norm,
           Wg=(rs.standard_normal((1, D)) / np.sqrt(D)).astype(f32), bg=(rs.standard_normal(1) * 0.05).astype(f32),
           W1=(rs.standard_normal((hidden, S)) * np.sqrt(2.0 / S)).astype(f32), b1=(rs.standard_normal(hidden) * 0.05).astype(f32),
           W2=(rs.standard_normal((1, hidden)) / np.sqrt(hidden)).astype(f32), b2=(rs.standard_normal(1) * 0.05).astype(f32),
           ug=unit(rs.standard_normal(1)), vg=unit(rs.standard_normal(D)), u1=unit(rs.standard_normal(hidden)), v1=unit(rs.standard_normal(S)),
           u2=unit(rs.standard_normal(1)), v2=unit(rs.standard_normal(hidden)))
  c['policy'] = [transitions(rs, batch, S, A, weighted=True, terminal_frac=0.3) for _ in range(steps)]
  c['expert'] = [transitions(rs, batch, S, A, state_shift=0.5, weighted=True, terminal_frac=0.3) for _ in range(steps)]
  c['eps'] = [rs.uniform(size=batch).astype(f32) for _ in range(steps)]
  return c
