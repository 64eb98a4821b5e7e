"""GAIL discriminator with reward shaping (TEST ORACLE, numpy float32, closed-form backward) -- restates reference `models.py:152-180` for
`reward_shaping=True` and `training.py:85-134` (BCE / PUGAIL losses, gradient penalty, entropy bonus, optional subtract_log_policy).

  f(s, a, s', t) = g(x) + (1 - t) (discount * h(s') - h(s)),   x = cat(s, a) (or s with state_only)
  g = Linear(D, 1);  h = Linear(S, H) -> ReLU -> Linear(H, 1);  every weight optionally under torch's `_SpectralNorm`.
In train mode each ACCESS of a parametrised weight runs one power iteration, and `forward` evaluates g(x), then h(s'), then h(s) (Python's left to
right order), so one discriminator call advances g's (u, v) once and h's twice, and h(s') / h(s) use DIFFERENT sigmas (after the first / second
iteration of that call). sigma = u^T W v with u, v constants in autograd: dL/dW = G/sigma - <G, W>/sigma^2 u v^T per use.
Gradient penalty (training.py:117-127): inputs with grad are the mixed state and action only, so
  dD/ds = Wg_s^ - (1 - t) (w2^ [a > 0]) W1^      (second use of h in the mix call),      dD/da = Wg_a^,
and the penalty's parameter gradient follows with the ReLU mask constant.
Parameter order = `discriminator.parameters()`: with spectral norm  g.bias, g.original, h.0.bias, h.0.original, h.2.bias, h.2.original;
without  g.weight, g.bias, h.0.weight, h.0.bias, h.2.weight, h.2.bias.   Pinned by tests/golden/gail_shaped.npz.
"""
from __future__ import annotations

import numpy as np

from . import nets
from .gail import _normalize, _power_iter, _sigmoid
from .nets import f32


class ShapedState:
  def __init__(self, S, A, H, discount, spectral_norm=True, state_only=False):
    self.S, self.A, self.H, self.sn, self.discount, self.state_only = S, A, H, spectral_norm, f32(discount), state_only
    self.Dg = S if state_only else S + A
    self.Wg, self.bg = np.zeros((1, self.Dg), f32), np.zeros(1, f32)
    self.W1, self.b1, self.W2, self.b2 = np.zeros((H, S), f32), np.zeros(H, f32), np.zeros((1, H), f32), np.zeros(1, f32)
    self.ug, self.vg = np.zeros(1, f32), np.zeros(self.Dg, f32)
    self.u1, self.v1, self.u2, self.v2 = np.zeros(H, f32), np.zeros(S, f32), np.zeros(1, f32), np.zeros(H, f32)
    self.P = self.Dg + 1 + H * S + H + H + 1
    self.m, self.v, self.t = np.zeros(self.P, f32), np.zeros(self.P, f32), 0

  def names(self):
    return ('bg', 'Wg', 'b1', 'W1', 'b2', 'W2') if self.sn else ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2')

  def pack(self, d=None):
    d = d or {k: getattr(self, k) for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2')}
    return np.concatenate([np.asarray(d[k], f32).ravel() for k in self.names()])

  def unpack_into(self, flat):
    o = 0
    for k in self.names():
      arr = getattr(self, k)
      arr[...] = flat[o:o + arr.size].reshape(arr.shape); o += arr.size


def _use(ds, name, train):
  """One access of weight `name` ('g', '1', '2'): (W_hat, ctx) with ctx = (u, v, sigma) for the chain rule; advances u, v in train mode."""
  W = getattr(ds, 'W' + name)
  if not ds.sn:
    return W, None
  u, v = getattr(ds, 'u' + name), getattr(ds, 'v' + name)
  if train:
    u, v = _power_iter(W, u, v)
    setattr(ds, 'u' + name, u); setattr(ds, 'v' + name, v)
  s = f32(np.dot(u, W @ v))
  return (W / s).astype(f32), (u.copy(), v.copy(), s)


def _chain(ds, name, ctx, Gh):
  if ctx is None:
    return Gh.astype(f32)
  u, v, s = ctx
  W = getattr(ds, 'W' + name)
  return (Gh / s - (np.sum(Gh * W, dtype=f32) / (s * s)) * np.outer(u, v)).astype(f32)


def _h(ds, s, train):
  W1h, c1 = _use(ds, '1', train)
  W2h, c2 = _use(ds, '2', train)
  pre = s @ W1h.T + ds.b1
  act = np.maximum(pre, f32(0))
  return (act @ W2h[0] + ds.b2[0]).astype(f32), (W1h, c1, W2h, c2, pre, act)


def forward(ds: ShapedState, x, s, ns, t, train=False):
  """f for one discriminator call, in the reference's evaluation order; returns f and what the backward needs."""
  Wgh, cg = _use(ds, 'g', train)
  gx = (x @ Wgh[0] + ds.bg[0]).astype(f32)
  hn, cn = _h(ds, ns, train)      # h(s') first ...
  hs, cs = _h(ds, s, train)       # ... then h(s): one more power iteration
  f = gx + (f32(1) - t) * (ds.discount * hn - hs)
  return f.astype(f32), (Wgh, cg, cn, cs)


def _h_backward(ds, g, cache, s, coef):
  """Accumulate d(sum_r coef_r h(s_r))/d params into g for one use of h."""
  W1h, c1, W2h, c2, pre, act = cache
  G2h = (coef @ act)[None, :]
  dpre = coef[:, None] * W2h[0][None, :] * (pre > 0)
  G1h = dpre.T @ s
  g['W1'] += _chain(ds, '1', c1, G1h.astype(f32)); g['W2'] += _chain(ds, '2', c2, G2h.astype(f32))
  g['b1'] += dpre.sum(axis=0); g['b2'] += coef.sum()


def _split(ds, b):
  s, ns, t = b['states'].astype(f32), b['next_states'].astype(f32), b['terminals'].astype(f32)
  x = s if ds.state_only else np.concatenate([s, b['actions'].astype(f32)], axis=1)
  return x, s, ns, t, b['weights'].astype(f32)


def gail_update(ds: ShapedState, pol, exp, eps_gp, *, lr, weight_decay, grad_penalty=1.0, entropy_bonus=0.0, loss_function='BCE', pos_class_prior=0.7,
                logp_policy=None, logp_expert=None, return_grads=False, nonnegative_margin=float('inf'), eps_mix=None, logp_mix=None):
  """One `adversarial_imitation_update` with reward shaping; pol / exp are transition dicts. nonnegative_margin: training.py:100-102 (PUGAIL), as in oracle/gail.py."""
  B = pol['states'].shape[0]
  g = {k: np.zeros_like(getattr(ds, k)) for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2')}
  pu = loss_function == 'PUGAIL'
  pr = f32(pos_class_prior)
  zero = np.zeros(B, f32)
  on = f32(1)
  if pu and nonnegative_margin != float('inf'):   # the clamp passes the gradient only where its argument is not below the bound: a batch-wide decision on the logits of both calls
    import copy
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
    f, (Wgh, cg, cn, cs) = forward(ds, x, s, ns, t, train=True)
    z = f if off is None else f - off.astype(f32)
    p = _sigmoid(z)
    dz = w * (c_sig * p - c_lab) / f32(B)
    if entropy_bonus > 0:
      dz = dz + f32(entropy_bonus) * w * z * p * (f32(1) - p) / f32(B)
    g['Wg'] += _chain(ds, 'g', cg, (dz @ x)[None, :].astype(f32)); g['bg'] += dz.sum()
    _h_backward(ds, g, cn, ns, dz * (f32(1) - t) * ds.discount)
    _h_backward(ds, g, cs, s, -dz * (f32(1) - t))

  if grad_penalty > 0:
    xp, sp, nsp, tp, wp = _split(ds, pol)
    xe, se, nse, te, we = _split(ds, exp)
    e = eps_gp.astype(f32)
    mix = lambda a, b_: (e[:, None] * a + (f32(1) - e[:, None]) * b_) if a.ndim == 2 else (e * a + (f32(1) - e) * b_)
    xm, sm, nsm, tm, wm = mix(xe, xp), mix(se, sp), mix(nse, nsp), mix(te, tp), mix(we, wp)
    _, (Wgh, cg, cn, cs) = forward(ds, xm, sm, nsm, tm, train=True)
    W1h, c1, W2h, c2, pre, act = cs                      # h(s): the use the input gradient goes through
    mask = (pre > 0).astype(f32)
    q = mask * W2h[0][None, :]                            # dh/dpre  [B, H]
    k = -(f32(1) - tm)                                    # coefficient of h(s) in f
    gin = np.repeat(Wgh, B, axis=0).astype(f32)           # dD/dx  [B, Dg]
    gin[:, :ds.S] += k[:, None] * (q @ W1h)
    c = f32(2) * f32(grad_penalty) * wm / f32(B)
    cg_in = c[:, None] * gin                              # d penalty / d(dD/dx)
    g['Wg'] += _chain(ds, 'g', cg, cg_in.sum(axis=0)[None, :].astype(f32))
    cs_in = cg_in[:, :ds.S] * k[:, None]                  # d penalty / d(q W1^)  [B, S]
    G1h = q.T @ cs_in
    G2h = (((cs_in @ W1h.T) * mask).sum(axis=0))[None, :]
    g['W1'] += _chain(ds, '1', c1, G1h.astype(f32)); g['W2'] += _chain(ds, '2', c2, G2h.astype(f32))

  flat_g, flat_p = ds.pack(g), ds.pack()
  ds.t += 1
  nets.adam_step(flat_p, flat_g, ds.m, ds.v, ds.t, lr, weight_decay)
  ds.unpack_into(flat_p)
  return flat_g if return_grads else None


def predict_reward(ds: ShapedState, b, reward_function='AIRL', log_policy=None):
  x, s, ns, t, _ = _split(ds, b)
  f, _ = forward(ds, x, s, ns, t, train=False)
  D = _sigmoid(f if log_policy is None else f - log_policy.astype(f32))
  h = -np.log1p(-D + f32(1e-6)) if reward_function == 'GAIL' else np.log(D + f32(1e-6)) - np.log1p(-D + f32(1e-6))
  return (np.exp(h) * -h if reward_function == 'FAIRL' else h).astype(f32)
