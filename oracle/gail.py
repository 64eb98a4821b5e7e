"""GAIL discriminator oracle (TEST ORACLE, numpy float32, closed-form backward).

Restates, for the default discriminator family (depth-1 MLP, ReLU, optional
spectral norm, BCE loss, gradient penalty, optional entropy bonus;
`conf/algorithm/GAIL.yaml`):
  * `adversarial_imitation_update` (training.py:85-134);
  * torch `_SpectralNorm` (nn/utils/parametrizations.py): train mode = one power
    iteration per weight access (3 discriminator calls per update), sigma = u^T W v,
    W/sigma with u, v constants in autograd; eval mode = no iteration;
  * `GAILDiscriminator.predict_reward` (models.py:177-180): AIRL / GAIL / FAIRL.

Parameter vector order = `discriminator.parameters()`:
  g.0.bias[H], g.0.parametrizations.weight.original[H,D], g.2.bias[1],
  g.2.parametrizations.weight.original[1,H]      (spectral_norm=True)
  g.0.weight[H,D], g.0.bias[H], g.2.weight[1,H], g.2.bias[1]   (spectral_norm=False)
The oracle keeps them as named arrays; `pack`/`unpack` give the torch order.
"""
from __future__ import annotations

import numpy as np

from . import nets
from .nets import f32


def _normalize(x, eps=1e-12):
  return (x / max(float(np.sqrt(np.sum(x * x, dtype=f32))), eps)).astype(f32)


class DiscState:
  def __init__(self, in_dim, hidden, spectral_norm=True):
    self.D, self.H, self.sn = in_dim, hidden, spectral_norm
    self.W1, self.b1 = np.zeros((hidden, in_dim), f32), np.zeros(hidden, f32)
    self.W2, self.b2 = np.zeros((1, hidden), f32), np.zeros(1, f32)
    self.u1, self.v1 = np.zeros(hidden, f32), np.zeros(in_dim, f32)
    self.u2, self.v2 = np.zeros(1, f32), np.zeros(hidden, f32)
    self.P = hidden * in_dim + hidden + hidden + 1
    self.m, self.v, self.t = np.zeros(self.P, f32), np.zeros(self.P, f32), 0

  # torch parameter order
  def names(self):
    return ('b1', 'W1', 'b2', 'W2') if self.sn else ('W1', 'b1', 'W2', 'b2')

  def pack(self, d=None):
    d = d or {k: getattr(self, k) for k in ('W1', 'b1', 'W2', 'b2')}
    return np.concatenate([np.asarray(d[k], f32).ravel() for k in self.names()])

  def unpack_into(self, flat):
    o = 0
    for k in self.names():
      arr = getattr(self, k)
      arr[...] = flat[o:o + arr.size].reshape(arr.shape); o += arr.size


def _power_iter(W, u, v):
  u = _normalize(W @ v)
  v = _normalize(W.T @ u)
  return u, v


def _sn_weights(ds: DiscState, train: bool):
  """One discriminator call's effective weights. Returns (W1h, W2h, ctx) and updates u, v in train mode."""
  if not ds.sn:
    return ds.W1, ds.W2, None
  if train:
    ds.u1, ds.v1 = _power_iter(ds.W1, ds.u1, ds.v1)
    ds.u2, ds.v2 = _power_iter(ds.W2, ds.u2, ds.v2)
  s1 = f32(np.dot(ds.u1, ds.W1 @ ds.v1))
  s2 = f32(np.dot(ds.u2, ds.W2 @ ds.v2))
  ctx = (ds.u1.copy(), ds.v1.copy(), s1, ds.u2.copy(), ds.v2.copy(), s2)
  return (ds.W1 / s1).astype(f32), (ds.W2 / s2).astype(f32), ctx


def _sn_backward(ds, ctx, G1h, G2h):
  """dL/dW from dL/dW_hat for one call: G/sigma - (<G, W>/sigma^2) u v^T."""
  if ctx is None:
    return G1h, G2h
  u1, v1, s1, u2, v2, s2 = ctx
  G1 = G1h / s1 - (np.sum(G1h * ds.W1, dtype=f32) / (s1 * s1)) * np.outer(u1, v1)
  G2 = G2h / s2 - (np.sum(G2h * ds.W2, dtype=f32) / (s2 * s2)) * np.outer(u2, v2)
  return G1.astype(f32), G2.astype(f32)


def _forward(W1h, b1, W2h, b2, x):
  h = x @ W1h.T + b1
  a = np.maximum(h, f32(0))
  z = a @ W2h[0] + b2[0]
  return h, a, z.astype(f32)


def _sigmoid(z):
  return (f32(1) / (f32(1) + np.exp(-z))).astype(f32)


def disc_logits(ds: DiscState, x, train=False):
  W1h, W2h, _ = _sn_weights(ds, train)
  return _forward(W1h, ds.b1, W2h, ds.b2, x.astype(f32))[2]


def gail_update(ds: DiscState, xp, wp, xe, we, eps_gp, *, lr, weight_decay, grad_penalty=1.0, entropy_bonus=0.0, return_grads=False, loss_function='BCE',
                pos_class_prior=0.7, eps_mix=None, logp_policy=None, logp_expert=None, logp_mix=None, nonnegative_margin=float('inf')):
  """One `adversarial_imitation_update` (training.py:85-134). xp/xe = cat(state, action) of policy / expert batch.
  loss_function: 'BCE' (:97-99), 'PUGAIL' with nonnegative_margin = inf (:100-102: the clamp never binds) or 'Mixup' (:104-113, eps_mix = the
  Beta(alpha, alpha) draws). logp_policy / logp_expert: log pi(a|s) of the two batches when subtract_log_policy (models.py:173-175: D = f - log pi;
  computed under no_grad, so it only shifts the logits)."""
  xp, xe, wp, we = xp.astype(f32), xe.astype(f32), wp.astype(f32), we.astype(f32)
  B = xp.shape[0]
  g = dict(W1=np.zeros_like(ds.W1), b1=np.zeros_like(ds.b1), W2=np.zeros_like(ds.W2), b2=np.zeros_like(ds.b2))
  zero = np.zeros(B, f32)
  if loss_function == 'BCE':      # d loss / d logit = w (c_sig * sigmoid - c_lab) / B
    calls = [(xp, wp, f32(1), zero, logp_policy), (xe, we, f32(1), zero + f32(1), logp_expert)]
  elif loss_function == 'PUGAIL':  # prior*BCE(D_e,1) + clamp(prior*BCE(D_e,0) - BCE(D_p,0), min=-margin)   (training.py:100-102)
    pr = f32(pos_class_prior)
    on = f32(1)
    if nonnegative_margin != float('inf'):   # the clamp passes the gradient only where its argument is not below the bound: a batch-wide decision on the logits of both calls
      import copy
      probe = copy.deepcopy(ds)              # the same two power iterations the calls below will run
      zs = []
      for x, off in ((xp, logp_policy), (xe, logp_expert)):
        W1h, W2h, _ = _sn_weights(probe, True)
        z = _forward(W1h, probe.b1, W2h, probe.b2, x)[2]
        zs.append(z if off is None else z - off.astype(f32))
      V = pr * np.mean(we * nets.softplus(zs[1]), dtype=f32) - np.mean(wp * nets.softplus(zs[0]), dtype=f32)
      on = f32(1) if V >= -nonnegative_margin else f32(0)
    calls = [(xp, wp, -on, zero, logp_policy), (xe, we, (f32(1) + on) * pr, zero + pr, logp_expert)]
  elif loss_function == 'Mixup':   # eps*BCE(D_mix,1) + (1-eps)*BCE(D_mix,0) on convex combinations
    em = eps_mix.astype(f32)
    calls = [(em[:, None] * xe + (f32(1) - em[:, None]) * xp, em * we + (f32(1) - em) * wp, f32(1), em, logp_mix)]   # logp_mix: log pi of the mixed rows (training.py:108)
    assert logp_policy is None and logp_expert is None
  else:
    raise ValueError(loss_function)

  # D_policy then D_expert (training.py:95) / D_mix: each call runs its own power iteration
  for x, w, c_sig, c_lab, off in calls:
    W1h, W2h, ctx = _sn_weights(ds, True)
    h, a, z = _forward(W1h, ds.b1, W2h, ds.b2, x)
    if off is not None:
      z = z - off.astype(f32)
    p = _sigmoid(z)
    dz = w * (c_sig * p - c_lab) / f32(B)              # BCE-with-logits, mean reduction, per-sample weight
    if entropy_bonus > 0:                             # training.py:130-132: -beta * mean(w * H(Bernoulli(logits=z)))
      dz = dz + f32(entropy_bonus) * w * z * p * (f32(1) - p) / f32(B)   # dH/dz = -z p (1-p)
    G2h = (dz @ a)[None, :]
    dh = dz[:, None] * W2h[0][None, :] * (h > 0)
    G1h = dh.T @ x
    G1, G2 = _sn_backward(ds, ctx, G1h.astype(f32), G2h.astype(f32))
    g['W1'] += G1; g['W2'] += G2; g['b1'] += dh.sum(axis=0); g['b2'] += dz.sum()

  if grad_penalty > 0:                                # training.py:117-127
    e = eps_gp.astype(f32)
    xm = e[:, None] * xe + (f32(1) - e[:, None]) * xp
    wm = e * we + (f32(1) - e) * wp
    W1h, W2h, ctx = _sn_weights(ds, True)
    h = xm @ W1h.T + ds.b1
    q = (h > 0) * W2h[0][None, :]                     # dD/dh
    gx = q @ W1h                                      # dD/dx  [B, D]
    c = (f32(2) * f32(grad_penalty) * wm / f32(B))
    cg = c[:, None] * gx
    G1h = q.T @ cg
    G2h = (((cg @ W1h.T) * (h > 0)).sum(axis=0))[None, :]
    G1, G2 = _sn_backward(ds, ctx, G1h.astype(f32), G2h.astype(f32))
    g['W1'] += G1; g['W2'] += G2

  flat_g = ds.pack(g)
  flat_p = ds.pack()
  ds.t += 1
  nets.adam_step(flat_p, flat_g, ds.m, ds.v, ds.t, lr, weight_decay)
  ds.unpack_into(flat_p)
  return flat_g if return_grads else None


def predict_reward_f64(ds: DiscState, x, reward_function='AIRL'):
  """`predict_reward` of the SAME float32 state evaluated in float64 (eval mode): the conditioning reference of a reward comparison. AIRL's log D - log1p(-D) loses
  digits near D = 1/2, so a float32 evaluation is itself off by up to ~1e-3 relative; a test that allows another float32 implementation `2 x |f32 - f64|` instead of a
  hand-picked rtol cannot be loosened by accident (tests/gpu_util.py bracket)."""
  d = np.float64
  W1, W2, b1, b2 = ds.W1.astype(d), ds.W2.astype(d), ds.b1.astype(d), ds.b2.astype(d)
  if ds.sn:
    W1 = W1 / np.dot(ds.u1.astype(d), W1 @ ds.v1.astype(d)); W2 = W2 / np.dot(ds.u2.astype(d), W2 @ ds.v2.astype(d))
  z = np.maximum(x.astype(d) @ W1.T + b1, 0.0) @ W2[0] + b2[0]
  D = 1.0 / (1.0 + np.exp(-z))
  h = -np.log1p(-D + 1e-6) if reward_function == 'GAIL' else np.log(D + 1e-6) - np.log1p(-D + 1e-6)
  return np.exp(h) * -h if reward_function == 'FAIRL' else h


def predict_reward(ds: DiscState, x, reward_function='AIRL', log_policy=None):
  """models.py:177-180, eval mode (no power iteration); log_policy: the subtract_log_policy offset (models.py:175)."""
  z = disc_logits(ds, x, train=False)
  D = _sigmoid(z if log_policy is None else z - log_policy.astype(f32))
  if reward_function == 'GAIL':
    h = -np.log1p(-D + f32(1e-6))
  else:
    h = np.log(D + f32(1e-6)) - np.log1p(-D + f32(1e-6))
  return (np.exp(h) * -h if reward_function == 'FAIRL' else h).astype(f32)
