"""DRIL dropout policy ensemble (TEST ORACLE, numpy float32) -- restates reference `models.py:84-120` for the network built by
`_create_fcnn(S, H, depth=1, 2A, 'tanh', input_dropout=p_in, dropout=p)` (models.py:47-69) and `behavioural_cloning_update`
(training.py:57-64) applied to it (train.py:119, the "discriminator" of algorithm=DRIL stays in train mode, so dropout is active in
every call).

  forward(s; m0, m1):  x~ = s * m0 / (1 - p_in);  z = W1 x~ + b1;  h = tanh(z * m1 / (1 - p));  (mean | log_std) = W2 h + b2
  log_prob(s, a)    :  a clamped to +-(1 - 1e-6), tanh-Gaussian log density (same op order as oracle/nets.py)
  BC update         :  loss = mean_i w_i * (-log_prob_i), AdamW
  uncertainty(s, a) :  variance (unbiased) over 5 independent dropout masks of exp(log_prob)          (models.py:104-107)
  reward            :  +1 where uncertainty <= q, else -1; q = torch.quantile(uncertainty(expert data), cutoff)   (models.py:110-120)
Masks are inputs (kept = 1): the reference draws them inside F.dropout; tests/golden/make_golden.py records the ones it drew.
"""
from __future__ import annotations

import numpy as np

from . import nets

f32 = np.float32
ENSEMBLE = 5


class DrilState:
  def __init__(self, S, A, H, p_in, p):
    self.S, self.A, self.H, self.p_in, self.p = S, A, H, p_in, p
    self.shapes = nets.mlp_shapes(S, H, 1, 2 * A)
    P = nets.mlp_numel(S, H, 1, 2 * A)
    self.params, self.m, self.v, self.t = np.zeros(P, f32), np.zeros(P, f32), np.zeros(P, f32), 0
    self.q = None


def _scale(p):
  return f32(1) / f32(1 - p)  # ATen: noise.bernoulli_(1 - p).div_(1 - p)


def forward(ds: DrilState, s, m0, m1):
  (W1, b1), (W2, b2) = nets.unpack(ds.params, ds.shapes)
  xt = (s * (m0 * _scale(ds.p_in))).astype(f32) if ds.p_in > 0 else s.astype(f32)
  z = xt @ W1.T + b1
  zt = (z * (m1 * _scale(ds.p))).astype(f32) if ds.p > 0 else z
  h = np.tanh(zt).astype(f32)
  out = h @ W2.T + b2
  return out, (xt, h)


def log_prob(ds: DrilState, s, a, m0, m1):
  out, cache = forward(ds, s, m0, m1)
  mean, ls_raw, _, std = nets.actor_head(out, ds.A)
  a = np.clip(a, f32(-1 + 1e-6), f32(1 - 1e-6))
  x = np.arctanh(a).astype(f32)
  return nets.tanh_gaussian_logp(x, mean, std), (x, mean, ls_raw, std) + cache


def bc_update(ds: DrilState, batch, m0, m1, *, lr, weight_decay=0.0, return_grads=False):
  s, a, w = batch['states'], batch['actions'], batch['weights']
  B = s.shape[0]
  logp, (x, mean, ls_raw, std, xt, h) = log_prob(ds, s, a, m0, m1)
  (W1, b1), (W2, b2) = nets.unpack(ds.params, ds.shapes)
  up = (-w / f32(B))[:, None]
  d = x - mean
  dmean = up * d / (std * std)
  dstd = up * (d * d / (std * std * std) - f32(1) / std)
  dls = dstd * std * ((ls_raw >= nets.LOG_STD_MIN) & (ls_raw <= nets.LOG_STD_MAX))
  dout = np.concatenate([dmean, dls], axis=1).astype(f32)
  gW2, gb2 = dout.T @ h, dout.sum(axis=0)
  dzt = (dout @ W2) * (f32(1) - h * h)
  dz = dzt * (m1 * _scale(ds.p)) if ds.p > 0 else dzt
  gW1, gb1 = dz.T @ xt, dz.sum(axis=0)
  g = np.concatenate([gW1.ravel(), gb1, gW2.ravel(), gb2]).astype(f32)
  ds.t += 1
  nets.adam_step(ds.params, g, ds.m, ds.v, ds.t, lr, weight_decay)
  loss = np.mean(w * -logp, dtype=f32)
  return (loss, g, logp) if return_grads else loss


def uncertainty(ds: DrilState, s, a, m0, m1):
  """m0 [n*5, S], m1 [n*5, H] in repeat_interleave order (models.py:105)."""
  logp, _ = log_prob(ds, np.repeat(s, ENSEMBLE, axis=0), np.repeat(a, ENSEMBLE, axis=0), m0, m1)
  prob = np.exp(logp).astype(f32).reshape(-1, ENSEMBLE)
  return prob.var(axis=1, ddof=1).astype(f32)


def predict_reward(ds: DrilState, s, a, m0, m1):
  return np.where(uncertainty(ds, s, a, m0, m1) <= f32(ds.q), f32(1), f32(-1))
