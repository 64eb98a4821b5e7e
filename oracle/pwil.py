"""PWIL greedy-coupling reward oracle (TEST ORACLE, numpy) -- reference `models.py:205-249`.

Literal restatement incl. row deletion (so "first index on ties" and the shrinking
index space behave as in the reference); `reset()` restores atoms and weights.
"""
from __future__ import annotations

from math import exp, sqrt

import numpy as np

from .nets import f32


class PwilOracle:
  def __init__(self, expert_atoms_raw, time_horizon, reward_scale, reward_bandwidth_scale):
    raw = np.asarray(expert_atoms_raw, f32)
    r64 = raw.astype(np.float64)   # torch's CPU reductions accumulate float32 in double (at::acc_type) and round the result once (models.py:204-205)
    inv_scale = r64.std(axis=0, ddof=1, keepdims=True).astype(f32)      # torch .std() is unbiased
    self.offset = (-r64.mean(axis=0, keepdims=True)).astype(f32)
    inv_scale[inv_scale == 0] = 1
    self.scale = (f32(1) / inv_scale).astype(f32)
    self.raw, self.T = raw, time_horizon
    self.reward_scale = reward_scale
    self.reward_bandwidth = reward_bandwidth_scale * time_horizon / sqrt(raw.shape[1])
    self.reset()

  def reset(self):
    self.atoms = (self.scale * (self.raw + self.offset)).astype(f32)
    n = self.raw.shape[0]
    self.weights = np.full(n, f32(1 / n), f32)

  def compute_reward(self, atom_raw):
    atom = (self.scale * (np.asarray(atom_raw, f32).reshape(1, -1) + self.offset)).astype(f32)
    weight, cost = 1 / self.T - 1e-6, 0.0
    d = self.atoms - atom
    dists = np.sqrt((d * d).sum(axis=1, dtype=f32)).astype(f32)
    while weight > 0:
      i = int(np.argmin(dists))
      ew = float(self.weights[i])
      if weight >= ew:
        cost += ew * float(dists[i])
        weight -= ew
        self.atoms, self.weights, dists = np.delete(self.atoms, i, 0), np.delete(self.weights, i, 0), np.delete(dists, i, 0)
      else:
        cost += weight * float(dists[i])
        self.weights[i] -= f32(weight)
        weight = 0
    return self.reward_scale * exp(-self.reward_bandwidth * cost)
