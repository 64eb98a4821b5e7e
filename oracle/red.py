"""Random Expert Distillation (TEST ORACLE, numpy float32) -- restates reference `models.py:252-284` and `training.py:68-75`.

REDDiscriminator = predictor and frozen random target, both `Linear(D,H) -> ReLU -> Linear(H,D)` on x = cat(state, action)
(or the state alone with state_only).  `target_estimation_update` minimises mean_i w_i * mean_c (pred_ic - target_ic)^2 with AdamW
on the predictor; `set_sigma` = 1 / median over the [n, n] matrix of mean_c (pred_ic - target_jc)^2 (torch lower median);
`predict_reward` = exp(-sigma_1 * mean_c (pred - target)^2).  Pinned by tests/golden/red.npz (reference outputs).
"""
from __future__ import annotations

import numpy as np

from . import nets
from .gmmil import squared_distance

f32 = np.float32


class RedState:
  def __init__(self, D, H):
    self.D, self.H = D, H
    self.shapes = nets.mlp_shapes(D, H, 1, D)
    P = nets.mlp_numel(D, H, 1, D)
    self.predictor, self.target = np.zeros(P, f32), np.zeros(P, f32)
    self.m, self.v, self.t = np.zeros(P, f32), np.zeros(P, f32), 0
    self.sigma_1 = None


def forward(rs: RedState, x):
  pred, acts = nets.mlp_forward(nets.unpack(rs.predictor, rs.shapes), x)
  targ, _ = nets.mlp_forward(nets.unpack(rs.target, rs.shapes), x)
  return pred, targ, acts


def target_estimation_update(rs: RedState, x, w, *, lr, weight_decay, return_grads=False):
  """training.py:68-75. Returns the loss (and the flat gradient when asked)."""
  B, D = x.shape
  pred, targ, acts = forward(rs, x)
  err = pred - targ
  loss = (w * (err * err).mean(axis=1)).mean()
  dout = (f32(2) * err * (w / f32(B * D))[:, None]).astype(f32)
  grad, _ = nets.mlp_backward(nets.unpack(rs.predictor, rs.shapes), acts, dout, need_dx=False)
  rs.t += 1
  nets.adam_step(rs.predictor, grad, rs.m, rs.v, rs.t, lr, wd=weight_decay)
  return (f32(loss), grad) if return_grads else f32(loss)


def set_sigma(rs: RedState, x):
  """models.py:274-277: only when no reward_bandwidth_scale was configured."""
  if not rs.sigma_1:
    pred, targ, _ = forward(rs, x)
    flat = np.sort(squared_distance(pred, targ).ravel())
    rs.sigma_1 = 1 / float(flat[(flat.size - 1) // 2])  # torch.median: lower of the two middle values
  return rs.sigma_1


def predict_reward(rs: RedState, x):
  pred, targ, _ = forward(rs, x)
  err = pred - targ
  return np.exp(-f32(rs.sigma_1) * (err * err).mean(axis=1)).astype(f32)
