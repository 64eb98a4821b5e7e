"""GAIL discriminator with reward shaping and a shaping potential of any `_create_fcnn` shape (TEST ORACLE, numpy float32, closed-form backward) -- restates
reference `models.py:152-180` for `reward_shaping=True` with `discriminator.depth` in {1, 2} and `activation` in {relu, tanh}, under `training.py:85-134`.

  f(s, a, s', t) = g(x) + (1 - t) (discount * h(s') - h(s)),   g = SN(Linear(Dg, 1)),   h = [SN(Linear) - act] x depth - SN(Linear(H, 1)) on the state.

oracle/gail_shaped.py is the depth-1 ReLU case written out by hand; this file composes the general potential from oracle/gail_deep.py's pieces
(`_forward`, `_backward_first_order`, `_input_gradient`, `_input_gradient_backward`, `_sn_weights`), and the two are tested against each other on the
depth-1 ReLU fixtures. What is specific to shaping:
  * torch's `_SpectralNorm` iterates once per weight ACCESS and `forward` evaluates g(x), h(s'), h(s) in that order: one call advances g once and every
    layer of h twice; the two uses of h are normalised by different sigmas and the chain rule dW = G^/sigma - <G^, W>/sigma^2 u v^T applies per use.
  * gradient penalty (training.py:117-127): only the mixed state and action carry a gradient, so  dD/ds = Wg_s^ + k dh/ds (second use, k = -(1 - t)),
    dD/da = Wg_a^;  the penalty reaches h's parameters through dh/ds only: dL/d(dh/ds) = 2 c k dD/ds with c = lambda w / B.
Parameter order = `discriminator.parameters()`: g (bias, original | weight, bias), then h's Linears in order, each (bias, original) with spectral norm and
(weight, bias) without. Spectral-norm buffers: ug | vg | per layer of h (u | v).   Pinned by tests/golden/gail_shaped_deep.npz (reference outputs).
"""
from __future__ import annotations

import copy

import numpy as np

from . import nets
from .gail import _power_iter, _sigmoid
from .gail_deep import (DeepDiscState, _backward_first_order, _forward, _input_gradient, _input_gradient_backward, _sn_backward, _sn_weights)
from .nets import f32


class ShapedDeepState:
  def __init__(self, S, A, H, discount, depth=2, activation='tanh', spectral_norm=True, state_only=False):
    self.S, self.A, self.H, self.sn, self.discount, self.state_only = S, A, H, spectral_norm, f32(discount), state_only
    self.Dg = S if state_only else S + A
    self.Wg, self.bg = np.zeros((1, self.Dg), f32), np.zeros(1, f32)
    self.ug, self.vg = np.zeros(1, f32), np.zeros(self.Dg, f32)
    self.h = DeepDiscState(S, H, depth, activation, spectral_norm)
    self.P = self.Dg + 1 + self.h.P
    self.m, self.v, self.t = np.zeros(self.P, f32), np.zeros(self.P, f32), 0

  def pack(self, Wg=None, bg=None, W=None, b=None):
    Wg, bg = (self.Wg if Wg is None else Wg), (self.bg if bg is None else bg)
    head = [bg.ravel(), Wg.ravel()] if self.sn else [Wg.ravel(), bg.ravel()]
    return np.concatenate(head + [self.h.pack(W, b)]).astype(f32)

  def unpack_into(self, flat):
    n = self.Dg + 1
    if self.sn: self.bg[...] = flat[:1]; self.Wg[...] = flat[1:n].reshape(1, -1)
    else: self.Wg[...] = flat[:n - 1].reshape(1, -1); self.bg[...] = flat[n - 1:n]
    self.h.unpack_into(flat[n:])

  def pack_sn(self):
    return np.concatenate([self.ug, self.vg, self.h.pack_sn()]).astype(f32)

  def unpack_sn(self, flat):
    self.ug[...] = flat[:1]; self.vg[...] = flat[1:1 + self.Dg]; self.h.unpack_sn(flat[1 + self.Dg:])


def _use_g(ds, train):
  if not ds.sn:
    return ds.Wg, None
  if train:
    ds.ug, ds.vg = _power_iter(ds.Wg, ds.ug, ds.vg)
  s = f32(np.dot(ds.ug, ds.Wg @ ds.vg))
  return (ds.Wg / s).astype(f32), (ds.ug.copy(), ds.vg.copy(), s)


def _chain_g(ds, ctx, Gh):
  if ctx is None:
    return Gh.astype(f32)
  u, v, s = ctx
  return (Gh / s - (np.sum(Gh * ds.Wg, dtype=f32) / (s * s)) * np.outer(u, v)).astype(f32)


def _split(ds, b):
  s, ns, t = b['states'].astype(f32), b['next_states'].astype(f32), b['terminals'].astype(f32)
  x = s if ds.state_only else np.concatenate([s, b['actions'].astype(f32)], axis=1)
  return x, s, ns, t, b['weights'].astype(f32)


def forward(ds: ShapedDeepState, x, s, ns, t, train=False):
  """f of one discriminator call in the reference's evaluation order: g(x), then h(s'), then h(s) (one more power iteration of every layer of h)."""
  act = ds.h.activation
  Wgh, cg = _use_g(ds, train)
  gx = (x @ Wgh[0] + ds.bg[0]).astype(f32)
  Whn, cn = _sn_weights(ds.h, train)
  acts_n, hn = _forward(Whn, ds.h.b, ns, act)
  Whs, cs = _sn_weights(ds.h, train)
  acts_s, hs = _forward(Whs, ds.h.b, s, act)
  f = gx + (f32(1) - t) * (ds.discount * hn - hs)
  return f.astype(f32), (Wgh, cg, (Whn, cn, acts_n), (Whs, cs, acts_s))


def gail_update(ds: ShapedDeepState, pol, exp, eps_gp, *, lr, weight_decay, grad_penalty=1.0, entropy_bonus=0.0, loss_function='BCE', pos_class_prior=0.7,
                logp_policy=None, logp_expert=None, return_grads=False, nonnegative_margin=float('inf'), eps_mix=None, logp_mix=None):
  """One `adversarial_imitation_update` with reward shaping; arguments as oracle/gail_shaped.py:gail_update."""
  B, act = pol['states'].shape[0], ds.h.activation
  gWg, gbg = np.zeros_like(ds.Wg), np.zeros_like(ds.bg)
  GW = [np.zeros_like(w) for w in ds.h.W]; Gb = [np.zeros_like(b) for b in ds.h.b]
  pu, pr, zero, on = loss_function == 'PUGAIL', f32(pos_class_prior), np.zeros(B, f32), f32(1)

  def h_backward(use, coef):
    Wh, ctx, acts = use
    gW, gb = _backward_first_order(Wh, acts, coef.astype(f32), act)
    for l, G in enumerate(_sn_backward(ds.h, ctx, gW)):
      GW[l] += G; Gb[l] += gb[l]

  if pu and nonnegative_margin != float('inf'):   # training.py:102: a batch-wide decision on the logits of both calls
    probe = copy.deepcopy(ds)                      # the same power iterations the calls below will run
    zs = []
    for b, off in ((pol, logp_policy), (exp, logp_expert)):
      x, s, ns, t, _ = _split(probe, b)
      f = forward(probe, x, s, ns, t, train=True)[0]
      zs.append(f if off is None else f - off.astype(f32))
    V = pr * np.mean(exp['weights'].astype(f32) * nets.softplus(zs[1]), dtype=f32) - np.mean(pol['weights'].astype(f32) * nets.softplus(zs[0]), dtype=f32)
    on = f32(1) if V >= -nonnegative_margin else f32(0)
  calls = [(pol, -on if pu else f32(1), zero, logp_policy), (exp, (f32(1) + on) * pr if pu else f32(1), zero + (pr if pu else f32(1)), logp_expert)]
  if loss_function == 'Mixup':   # training.py:104-113: ONE call on the convex combination of every field (the mixed terminal is fractional), label = the coefficient
    em = eps_mix.astype(f32)
    mixf = lambda a, b_: (em[:, None] * a.astype(f32) + (f32(1) - em[:, None]) * b_.astype(f32)) if a.ndim == 2 else (em * a.astype(f32) + (f32(1) - em) * b_.astype(f32))
    calls = [({k: mixf(exp[k], pol[k]) for k in ('states', 'actions', 'next_states', 'terminals', 'weights')}, f32(1), em, logp_mix)]
    assert logp_policy is None and logp_expert is None
  for b, c_sig, c_lab, off in calls:
    x, s, ns, t, w = _split(ds, b)
    f, (Wgh, cg, use_n, use_s) = forward(ds, x, s, ns, t, train=True)
    z = f if off is None else f - off.astype(f32)
    p = _sigmoid(z)
    dz = w * (c_sig * p - c_lab) / f32(B)
    if entropy_bonus > 0:
      dz = dz + f32(entropy_bonus) * w * z * p * (f32(1) - p) / f32(B)
    gWg += _chain_g(ds, cg, (dz @ x)[None, :].astype(f32)); gbg += dz.sum()
    h_backward(use_n, dz * (f32(1) - t) * ds.discount)
    h_backward(use_s, -dz * (f32(1) - t))

  if grad_penalty > 0:
    xp, sp, nsp, tp, wp = _split(ds, pol)
    xe, se, nse, te, we = _split(ds, exp)
    e = eps_gp.astype(f32)
    mix = lambda a, b_: (e[:, None] * a + (f32(1) - e[:, None]) * b_) if a.ndim == 2 else (e * a + (f32(1) - e) * b_)
    xm, sm, nsm, tm, wm = mix(xe, xp), mix(se, sp), mix(nse, nsp), mix(te, tp), mix(we, wp)
    _, (Wgh, cg, use_n, (Whs, cs, acts_s)) = forward(ds, xm, sm, nsm, tm, train=True)
    u, sl = _input_gradient(Whs, acts_s, act)               # dh/ds of the second use: the one the input gradient goes through
    k = -(f32(1) - tm)                                      # coefficient of h(s) in f
    gin = np.repeat(Wgh, B, axis=0).astype(f32)             # dD/dx  [B, Dg]
    gin[:, :ds.S] += k[:, None] * sl[0]
    c = f32(2) * f32(grad_penalty) * wm / f32(B)
    cg_in = (c[:, None] * gin).astype(f32)                  # d penalty / d(dD/dx)
    gWg += _chain_g(ds, cg, cg_in.sum(axis=0)[None, :].astype(f32))
    gW, gb = _input_gradient_backward(Whs, acts_s, u, sl, cg_in[:, :ds.S] * k[:, None], act)
    for l, G in enumerate(_sn_backward(ds.h, cs, gW)):
      GW[l] += G; Gb[l] += gb[l]

  flat_g, flat_p = ds.pack(gWg, gbg, GW, Gb), ds.pack()
  ds.t += 1
  nets.adam_step(flat_p, flat_g, ds.m, ds.v, ds.t, lr, weight_decay)
  ds.unpack_into(flat_p)
  return flat_g if return_grads else None


def predict_reward(ds: ShapedDeepState, b, reward_function='AIRL', log_policy=None):
  """models.py:177-180, eval mode (no power iteration: both uses of h see the same weights)."""
  x, s, ns, t, _ = _split(ds, b)
  f, _ = forward(ds, x, s, ns, t, train=False)
  D = _sigmoid(f if log_policy is None else f - log_policy.astype(f32))
  h = -np.log1p(-D + f32(1e-6)) if reward_function == 'GAIL' else np.log(D + f32(1e-6)) - np.log1p(-D + f32(1e-6))
  return (np.exp(h) * -h if reward_function == 'FAIRL' else h).astype(f32)
