"""DRIL dropout policy ensemble (TEST ORACLE, numpy float32) -- restates reference `models.py:84-120` for the network built by
`_create_fcnn(S, H, depth in {1, 2}, 2A, 'tanh' | 'relu', input_dropout=p_in, dropout=p)` (models.py:47-69) and `behavioural_cloning_update`
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
  def __init__(self, S, A, H, p_in, p, depth=1, activation='tanh'):
    self.S, self.A, self.H, self.p_in, self.p, self.depth, self.activation = S, A, H, p_in, p, depth, activation
    self.shapes = nets.mlp_shapes(S, H, depth, 2 * A)
    P = nets.mlp_numel(S, H, depth, 2 * A)
    self.params, self.m, self.v, self.t = np.zeros(P, f32), np.zeros(P, f32), np.zeros(P, f32), 0
    self.q = None


def _scale(p):
  return f32(1) / f32(1 - p)  # ATen: noise.bernoulli_(1 - p).div_(1 - p)


def _act(z, activation):
  return np.tanh(z).astype(f32) if activation == 'tanh' else np.maximum(z, f32(0)).astype(f32)


def forward(ds: DrilState, s, m0, *mh):
  """m0: input keep-mask, mh: one keep-mask per hidden layer (module order of `_create_fcnn`)."""
  layers = nets.unpack(ds.params, ds.shapes)
  assert len(mh) == ds.depth
  h = (s * (m0 * _scale(ds.p_in))).astype(f32) if ds.p_in > 0 else s.astype(f32)
  acts = [h]
  for (W, b), m in zip(layers[:-1], mh):
    z = h @ W.T + b
    zt = (z * (m * _scale(ds.p))).astype(f32) if ds.p > 0 else z
    h = _act(zt, ds.activation)
    acts.append(h)
  Wo, bo = layers[-1]
  return h @ Wo.T + bo, tuple(acts)


def log_prob(ds: DrilState, s, a, m0, *mh):
  out, acts = forward(ds, s, m0, *mh)
  mean, ls_raw, _, std = nets.actor_head(out, ds.A)
  a = np.clip(a, f32(-1 + 1e-6), f32(1 - 1e-6))
  x = np.arctanh(a).astype(f32)
  return nets.tanh_gaussian_logp(x, mean, std), (x, mean, ls_raw, std, acts)


def bc_update(ds: DrilState, batch, m0, *mh, lr, weight_decay=0.0, return_grads=False):
  s, a, w = batch['states'], batch['actions'], batch['weights']
  B = s.shape[0]
  logp, (x, mean, ls_raw, std, acts) = log_prob(ds, s, a, m0, *mh)
  layers = nets.unpack(ds.params, ds.shapes)
  up = (-w / f32(B))[:, None]
  d = x - mean
  dmean = up * d / (std * std)
  dstd = up * (d * d / (std * std * std) - f32(1) / std)
  dls = dstd * std * ((ls_raw >= nets.LOG_STD_MIN) & (ls_raw <= nets.LOG_STD_MAX))
  dout = np.concatenate([dmean, dls], axis=1).astype(f32)
  grads = [None] * len(layers)
  grads[-1] = (dout.T @ acts[-1], dout.sum(axis=0))
  dh = dout @ layers[-1][0]
  for l in range(len(layers) - 2, -1, -1):
    h = acts[l + 1]
    dzt = dh * (f32(1) - h * h) if ds.activation == 'tanh' else np.where(h > 0, dh, f32(0))
    dz = dzt * (mh[l] * _scale(ds.p)) if ds.p > 0 else dzt
    grads[l] = (dz.T @ acts[l], dz.sum(axis=0))
    dh = dz @ layers[l][0]
  g = np.concatenate([np.concatenate([gw.ravel(), gb.ravel()]) for gw, gb in grads]).astype(f32)
  ds.t += 1
  nets.adam_step(ds.params, g, ds.m, ds.v, ds.t, lr, weight_decay)
  loss = np.mean(w * -logp, dtype=f32)
  return (loss, g, logp) if return_grads else loss


def uncertainty(ds: DrilState, s, a, m0, *mh):
  """m0 [n*5, S], mh [n*5, H] per hidden layer, in repeat_interleave order (models.py:105)."""
  logp, _ = log_prob(ds, np.repeat(s, ENSEMBLE, axis=0), np.repeat(a, ENSEMBLE, axis=0), m0, *mh)
  prob = np.exp(logp).astype(f32).reshape(-1, ENSEMBLE)
  return prob.var(axis=1, ddof=1).astype(f32)


def predict_reward(ds: DrilState, s, a, m0, *mh):
  return np.where(uncertainty(ds, s, a, m0, *mh) <= f32(ds.q), f32(1), f32(-1))
