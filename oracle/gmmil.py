"""GMMIL reward oracle (TEST ORACLE, numpy float32) -- reference `models.py:25-44, 183-201`.

d(x, y) = mean_k (x_k - y_k)^2 (mean over features, NOT sum; models.py:25-28);
weighted median = first sorted element whose normalised cumulative weight >= 0.5
(models.py:40-44); gammas frozen from the first batch (models.py:193-195).
"""
from __future__ import annotations

import numpy as np

from .nets import f32


def squared_distance(x, y, chunk=128):
  x, y = x.astype(f32), y.astype(f32)
  out = np.empty((x.shape[0], y.shape[0]), f32)
  for i in range(0, x.shape[0], chunk):
    d = x[i:i + chunk, None, :] - y[None, :, :]
    out[i:i + chunk] = (d * d).mean(axis=2, dtype=f32)
  return out


def weighted_median(x, weights):
  flat, w = x.ravel(), weights.ravel()
  order = np.argsort(flat, kind='stable')
  wn = (w / w.sum(dtype=f32))[order]
  k = int(np.argmax(np.cumsum(wn, dtype=f32) >= 0.5))
  return flat[order][k]


def median_gammas(X, E, w, we):
  g1 = 1 / (float(weighted_median(squared_distance(X, E), np.outer(w, we))) + 1e-8)
  g2 = 1 / (float(weighted_median(squared_distance(E, E), np.outer(we, we))) + 1e-8)
  return g1, g2


def gmmil_reward(X, E, w, we, gamma_1, gamma_2, return_parts=False):
  wn, wen = (w / w.sum(dtype=f32)).astype(f32), (we / we.sum(dtype=f32)).astype(f32)
  dxe, dxx = squared_distance(X, E), squared_distance(X, X)
  sim = np.zeros(X.shape[0], f32)
  self_sim = np.zeros(X.shape[0], f32)
  for gam in (gamma_1, gamma_2):
    sim += wn * (np.exp(f32(-gam) * dxe) @ wen)
    self_sim += wn * (np.exp(f32(-gam) * dxx) @ wn)
  r = (sim - self_sim).astype(f32)
  return (r, sim, self_sim) if return_parts else r
