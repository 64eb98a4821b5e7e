"""GAIL discriminator of any `_create_fcnn` shape (TEST ORACLE, numpy float32, closed-form backward incl. the double-backward of the gradient penalty).

Generalises oracle/gail.py (depth 1, ReLU) to `imitation.discriminator.depth` in {1, 2} and `activation` in {relu, tanh}
(conf/hyperparameter_search_space/GAIL.yaml), without reward shaping:  g = [SN(Linear) - act] x depth - SN(Linear(H, 1))  on x = cat(s, a)
(models.py:49-70, 152-162). Restates `adversarial_imitation_update` (training.py:85-134: BCE / PUGAIL / Mixup, entropy
bonus, gradient penalty), torch `_SpectralNorm` (one power iteration per weight access in train mode, sigma = u^T W v, u and v constants in autograd)
and `predict_reward` (models.py:177-180). Pinned by tests/golden/gail_deep.npz (reference outputs).

Gradient penalty (training.py:117-127): L = lambda * mean_r w_r ||g_r||^2 with g = dD/dx. With z_l = a_{l-1} W^_l^T + b_l, a_l = phi(z_l), D = a_L . w^_o + b_o:
  u_L = phi'(z_L) * w^_o,   s_l = u_l W^_l,   u_{l-1} = phi'(z_{l-1}) * s_l,   g = s_1.
Its parameter gradient needs the derivative of that backward pass: the terms through W^_l in s_l = u_l W^_l, and - for tanh only - the terms through
phi'(z_l), which re-enter the forward graph with phi''(z_l) (ReLU: phi'' = 0, so no bias gradient and no second-order terms).

Parameter vector order = `discriminator.parameters()`: per Linear (bias, parametrizations.weight.original) with spectral norm, (weight, bias) without.
"""
from __future__ import annotations

import numpy as np

from . import nets
from .gail import _normalize, _power_iter, _sigmoid
from .nets import f32


class DeepDiscState:
  def __init__(self, in_dim, hidden, depth=2, activation='tanh', spectral_norm=True):
    self.D, self.H, self.depth, self.activation, self.sn = in_dim, hidden, depth, activation, spectral_norm
    dims = [in_dim] + [hidden] * depth + [1]
    self.W = [np.zeros((dims[i + 1], dims[i]), f32) for i in range(depth + 1)]
    self.b = [np.zeros(dims[i + 1], f32) for i in range(depth + 1)]
    self.u = [np.zeros(dims[i + 1], f32) for i in range(depth + 1)]
    self.v = [np.zeros(dims[i], f32) for i in range(depth + 1)]
    self.P = sum(w.size + b.size for w, b in zip(self.W, self.b))
    self.m, self.vv, self.t = np.zeros(self.P, f32), np.zeros(self.P, f32), 0

  def pack(self, W=None, b=None):
    W, b = W or self.W, b or self.b
    parts = []
    for w, bb in zip(W, b):
      parts += [bb.ravel(), w.ravel()] if self.sn else [w.ravel(), bb.ravel()]
    return np.concatenate(parts).astype(f32)

  def unpack_into(self, flat):
    o = 0
    for w, bb in zip(self.W, self.b):
      for arr in ((bb, w) if self.sn else (w, bb)):
        arr[...] = flat[o:o + arr.size].reshape(arr.shape); o += arr.size

  def pack_sn(self):
    return np.concatenate([np.concatenate([u, v]) for u, v in zip(self.u, self.v)]).astype(f32)

  def unpack_sn(self, flat):
    o = 0
    for u, v in zip(self.u, self.v):
      u[...] = flat[o:o + u.size]; o += u.size
      v[...] = flat[o:o + v.size]; o += v.size


def _phi(z, act):
  return np.tanh(z).astype(f32) if act == 'tanh' else np.maximum(z, f32(0)).astype(f32)


def _dphi(a, act):     # phi'(z) in terms of a = phi(z)
  return (f32(1) - a * a).astype(f32) if act == 'tanh' else (a > 0).astype(f32)


def _d2phi(a, act):    # phi''(z) in terms of a
  return (f32(-2) * a * (f32(1) - a * a)).astype(f32) if act == 'tanh' else np.zeros_like(a)


def _sn_weights(ds: DeepDiscState, train: bool):
  """Effective weights of ONE discriminator call (every layer's weight is accessed once per call). Updates u, v in train mode."""
  if not ds.sn:
    return [w for w in ds.W], None
  Wh, ctx = [], []
  for l, W in enumerate(ds.W):
    if train:
      ds.u[l], ds.v[l] = _power_iter(W, ds.u[l], ds.v[l])
    s = f32(np.dot(ds.u[l], W @ ds.v[l]))
    Wh.append((W / s).astype(f32)); ctx.append((ds.u[l].copy(), ds.v[l].copy(), s))
  return Wh, ctx


def _sn_backward(ds, ctx, Gh):
  if ctx is None:
    return Gh
  out = []
  for W, G, (u, v, s) in zip(ds.W, Gh, ctx):
    out.append((G / s - (np.sum(G * W, dtype=f32) / (s * s)) * np.outer(u, v)).astype(f32))
  return out


def _forward(Wh, b, x, act):
  a, acts = x.astype(f32), [x.astype(f32)]
  for W, bb in zip(Wh[:-1], b[:-1]):
    a = _phi(a @ W.T + bb, act)
    acts.append(a)
  return acts, (a @ Wh[-1][0] + b[-1][0]).astype(f32)


def disc_logits(ds: DeepDiscState, x, train=False):
  Wh, _ = _sn_weights(ds, train)
  return _forward(Wh, ds.b, x, ds.activation)[1]


def _backward_first_order(Wh, acts, dz_out, act):
  """dL/dW^_l, dL/db_l for dL/dD = dz_out [B]."""
  L = len(Wh) - 1
  GW, Gb = [None] * (L + 1), [None] * (L + 1)
  GW[L] = (dz_out @ acts[L])[None, :].astype(f32); Gb[L] = np.array([dz_out.sum()], f32)
  zbar = (dz_out[:, None] * Wh[L][0][None, :]) * _dphi(acts[L], act)
  for l in range(L - 1, -1, -1):
    GW[l] = (zbar.T @ acts[l]).astype(f32); Gb[l] = zbar.sum(axis=0).astype(f32)
    if l > 0:
      zbar = (zbar @ Wh[l]) * _dphi(acts[l], act)
  return GW, Gb


def _input_gradient(Wh, acts, act):
  """The input-gradient pass g = dD/dx of one call: u[l] = dD/dz_{l+1} (hidden layer l, 0-based), s[l] = u[l] W^_l = dD/da_l; g = s[0]."""
  L = len(Wh) - 1
  u, s = [None] * L, [None] * L
  u[L - 1] = _dphi(acts[L], act) * Wh[L][0][None, :]
  for l in range(L - 1, -1, -1):
    s[l] = (u[l] @ Wh[l]).astype(f32)
    if l > 0:
      u[l - 1] = _dphi(acts[l], act) * s[l]
  return u, s


def _input_gradient_backward(Wh, acts, u, s, sbar, act):
  """Gradients w.r.t. the effective weights / biases of a loss that reaches the parameters only through g = dD/dx, given sbar = dL/dg [B, D]."""
  L = len(Wh) - 1
  GW = [np.zeros_like(w) for w in Wh]; Gb = [np.zeros(w.shape[0], f32) for w in Wh]
  sbar = sbar.astype(f32)
  zbar2 = [None] * L                                         # second-order terms entering the forward graph at z_{l+1}
  for l in range(L):
    GW[l] += (u[l].T @ sbar).astype(f32)                     # s_l = u_l W^_l
    ubar = (sbar @ Wh[l].T).astype(f32)
    if l < L - 1:
      zbar2[l] = ubar * s[l + 1] * _d2phi(acts[l + 1], act)  # u_l = phi'(z_{l+1}) * s_{l+1}
      sbar = ubar * _dphi(acts[l + 1], act)
    else:
      GW[L] += (ubar * _dphi(acts[L], act)).sum(axis=0)[None, :].astype(f32)   # u_{L-1} = phi'(z_L) * w^_o
      zbar2[l] = ubar * Wh[L][0][None, :] * _d2phi(acts[L], act)
  if act == 'tanh':                                          # phi'' != 0: back through the forward pass
    zbar = zbar2[L - 1]
    for l in range(L - 1, -1, -1):
      GW[l] += (zbar.T @ acts[l]).astype(f32); Gb[l] += zbar.sum(axis=0).astype(f32)
      if l > 0:
        zbar = (zbar @ Wh[l]) * _dphi(acts[l], act) + zbar2[l - 1]
  return GW, Gb


def _grad_penalty_grads(Wh, acts, c, act):
  """Gradients of sum_r c_r ||dD/dx_r||^2 w.r.t. the effective weights / biases. acts = [x, a_1 .. a_L]; layer index l = 0 .. L-1 hidden, L = output."""
  u, s = _input_gradient(Wh, acts, act)
  return _input_gradient_backward(Wh, acts, u, s, (f32(2) * c[:, None] * s[0]).astype(f32), act)   # dL/dg = 2 c g


def gail_update(ds: DeepDiscState, xp, wp, xe, we, eps_gp, *, lr, weight_decay, grad_penalty=1.0, entropy_bonus=0.0, return_grads=False, loss_function='BCE',
                pos_class_prior=0.7, eps_mix=None, logp_policy=None, logp_expert=None, nonnegative_margin=float('inf')):
  """One `adversarial_imitation_update` (training.py:85-134); arguments as in oracle/gail.py:gail_update."""
  xp, xe, wp, we = xp.astype(f32), xe.astype(f32), wp.astype(f32), we.astype(f32)
  B, act = xp.shape[0], ds.activation
  GW = [np.zeros_like(w) for w in ds.W]; Gb = [np.zeros_like(b) for b in ds.b]
  zero = np.zeros(B, f32)
  if loss_function == 'BCE':
    calls = [(xp, wp, f32(1), zero, logp_policy), (xe, we, f32(1), zero + f32(1), logp_expert)]
  elif loss_function == 'PUGAIL':
    pr = f32(pos_class_prior)
    on = f32(1)
    if nonnegative_margin != float('inf'):   # training.py:102: the clamp passes the gradient only where its argument is not below the bound (a batch-wide decision)
      import copy
      probe = copy.deepcopy(ds)              # the same two power iterations the calls below will run
      zs = []
      for x, off in ((xp, logp_policy), (xe, logp_expert)):
        Wh, _ = _sn_weights(probe, True)
        z = _forward(Wh, probe.b, x, act)[1]
        zs.append(z if off is None else z - off.astype(f32))
      V = pr * np.mean(we * nets.softplus(zs[1]), dtype=f32) - np.mean(wp * nets.softplus(zs[0]), dtype=f32)
      on = f32(1) if V >= -nonnegative_margin else f32(0)
    calls = [(xp, wp, -on, zero, logp_policy), (xe, we, (f32(1) + on) * pr, zero + pr, logp_expert)]
  elif loss_function == 'Mixup':
    em = eps_mix.astype(f32)
    calls = [(em[:, None] * xe + (f32(1) - em[:, None]) * xp, em * we + (f32(1) - em) * wp, f32(1), em, None)]
    assert logp_policy is None and logp_expert is None
  else:
    raise ValueError(loss_function)
  for x, w, c_sig, c_lab, off in calls:
    Wh, ctx = _sn_weights(ds, True)
    acts, z = _forward(Wh, ds.b, x, act)
    if off is not None:
      z = z - off.astype(f32)
    p = _sigmoid(z)
    dz = w * (c_sig * p - c_lab) / f32(B)
    if entropy_bonus > 0:
      dz = dz + f32(entropy_bonus) * w * z * p * (f32(1) - p) / f32(B)
    gW, gb = _backward_first_order(Wh, acts, dz.astype(f32), act)
    for l, G in enumerate(_sn_backward(ds, ctx, gW)):
      GW[l] += G; Gb[l] += gb[l]
  if grad_penalty > 0:
    e = eps_gp.astype(f32)
    xm = e[:, None] * xe + (f32(1) - e[:, None]) * xp
    wm = e * we + (f32(1) - e) * wp
    Wh, ctx = _sn_weights(ds, True)
    acts, _ = _forward(Wh, ds.b, xm, act)
    gW, gb = _grad_penalty_grads(Wh, acts, (f32(grad_penalty) * wm / f32(B)).astype(f32), act)
    for l, G in enumerate(_sn_backward(ds, ctx, gW)):
      GW[l] += G; Gb[l] += gb[l]
  flat_g, flat_p = ds.pack(GW, Gb), ds.pack()
  ds.t += 1
  nets.adam_step(flat_p, flat_g, ds.m, ds.vv, ds.t, lr, weight_decay)
  ds.unpack_into(flat_p)
  return flat_g if return_grads else None


def predict_reward(ds: DeepDiscState, x, reward_function='AIRL', log_policy=None):
  """models.py:177-180, eval mode (no power iteration)."""
  z = disc_logits(ds, x, train=False)
  D = _sigmoid(z if log_policy is None else z - log_policy.astype(f32))
  h = -np.log1p(-D + f32(1e-6)) if reward_function == 'GAIL' else np.log(D + f32(1e-6)) - np.log1p(-D + f32(1e-6))
  return (np.exp(h) * -h if reward_function == 'FAIRL' else h).astype(f32)
