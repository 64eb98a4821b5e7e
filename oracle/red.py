"""Random Expert Distillation (TEST ORACLE, numpy float32) -- restates reference `models.py:252-284` and `training.py:68-75`.

REDDiscriminator = predictor and frozen random target, both `_create_fcnn` (models.py:49-70) MLPs on x = cat(state, action) (or the state alone with
state_only): [Dropout(p_in)] Linear(D,H) [Dropout(p)] act (Linear(H,H) [Dropout(p)] act) Linear(H,D), act in {ReLU, Tanh}; only the predictor has the
dropout layers, active in train mode (target_estimation_update and set_sigma, train.py:115-128; predict_reward follows `discriminator.eval()`).
`target_estimation_update` minimises mean_i w_i * mean_c (pred_ic - target_ic)^2 with AdamW on the predictor; `set_sigma` = 1 / median over the [n, n]
matrix of mean_c (pred_ic - target_jc)^2 (torch lower median); `predict_reward` = exp(-sigma_1 * mean_c (pred - target)^2).
Dropout masks are inputs (keep = 1 / drop = 0; ATen multiplies by mask / (1 - p)). Pinned by tests/golden/red.npz (reference outputs).
"""
from __future__ import annotations

import numpy as np

from . import nets
from .gmmil import squared_distance

f32 = np.float32


class RedState:
  def __init__(self, D, H, depth=1, activation='relu', p_in=0.0, p=0.0):
    self.D, self.H, self.depth, self.activation, self.p_in, self.p = D, H, depth, activation, float(p_in), float(p)
    self.shapes = nets.mlp_shapes(D, H, depth, D)
    P = nets.mlp_numel(D, H, depth, D)
    self.predictor, self.target = np.zeros(P, f32), np.zeros(P, f32)
    self.m, self.v, self.t = np.zeros(P, f32), np.zeros(P, f32), 0
    self.sigma_1 = None


def _act(z, activation):
  return np.tanh(z).astype(f32) if activation == 'tanh' else np.maximum(z, f32(0))


def _net_forward(rs: RedState, flat, x, masks=None):
  """masks: None (eval mode / the target) or [m_in?, m_h1?, m_h2?] in module order. Returns (out, activations [x~, h1, (h2)], keep-scales per hidden layer)."""
  layers = nets.unpack(flat, rs.shapes)
  masks = list(masks) if masks is not None else []
  h = x.astype(f32)
  if masks and rs.p_in > 0: h = (h * (masks.pop(0) / f32(1 - rs.p_in)).astype(f32)).astype(f32)
  acts, scales = [h], []
  for W, b in layers[:-1]:
    z = (h @ W.T + b).astype(f32)
    scale = None
    if masks and rs.p > 0:
      scale = (masks.pop(0) / f32(1 - rs.p)).astype(f32)
      z = (z * scale).astype(f32)
    h = _act(z, rs.activation)
    acts.append(h); scales.append(scale)
  Wo, bo = layers[-1]
  assert not masks, 'unused dropout masks'
  return (h @ Wo.T + bo).astype(f32), acts, scales


def forward(rs: RedState, x, masks=None):
  pred, acts, scales = _net_forward(rs, rs.predictor, x, masks)
  targ, _, _ = _net_forward(rs, rs.target, x, None)
  return pred, targ, (acts, scales)


def target_estimation_update(rs: RedState, x, w, *, lr, weight_decay, masks=None, return_grads=False):
  """training.py:68-75 (train mode). Returns the loss (and the flat gradient when asked)."""
  B, D = x.shape
  pred, targ, (acts, scales) = forward(rs, x, masks)
  err = pred - targ
  loss = (w * (err * err).mean(axis=1)).mean()
  g = (f32(2) * err * (w / f32(B * D))[:, None]).astype(f32)
  layers = nets.unpack(rs.predictor, rs.shapes)
  grads = [None] * len(layers)
  grads[-1] = ((g.T @ acts[-1]).astype(f32), g.sum(axis=0).astype(f32))
  dh = (g @ layers[-1][0]).astype(f32)
  for l in range(len(layers) - 2, -1, -1):
    h = acts[l + 1]
    dz = (dh * (f32(1) - h * h)).astype(f32) if rs.activation == 'tanh' else np.where(h > 0, dh, f32(0)).astype(f32)
    if scales[l] is not None: dz = (dz * scales[l]).astype(f32)
    grads[l] = ((dz.T @ acts[l]).astype(f32), dz.sum(axis=0).astype(f32))
    dh = (dz @ layers[l][0]).astype(f32)
  grad = np.concatenate([np.concatenate([gw.ravel(), gb.ravel()]) for gw, gb in grads]).astype(f32)
  rs.t += 1
  nets.adam_step(rs.predictor, grad, rs.m, rs.v, rs.t, lr, wd=weight_decay)
  return (f32(loss), grad) if return_grads else f32(loss)


def set_sigma(rs: RedState, x, masks=None):
  """models.py:274-277: only when no reward_bandwidth_scale was configured; the module is still in train mode here (train.py:128 precedes :147)."""
  if not rs.sigma_1:
    pred, targ, _ = forward(rs, x, masks)
    flat = np.sort(squared_distance(pred, targ).ravel())
    rs.sigma_1 = 1 / float(flat[(flat.size - 1) // 2])  # torch.median: lower of the two middle values
  return rs.sigma_1


def predict_reward(rs: RedState, x):
  pred, targ, _ = forward(rs, x)   # eval mode
  err = pred - targ
  return np.exp(-f32(rs.sigma_1) * (err * err).mean(axis=1)).astype(f32)
