"""The formulation behind csrc/gmmil.hip's k_gmmil_mfma (round 6), pinned on the CPU in numpy float32: the GMMIL pair distances of reference models.py:25-28 written as a
Gram product, ssq = |x|^2 + |y|^2 - 2 x.y, lose the digits the data's OFFSET takes - which is why the kernels of rounds 1-5 used the direct difference form on the VALU -
while the same product on operands CENTRED on a mean of expert rows is as close to float64 as the direct form, whatever the offset. (The GPU-side check of the kernel
itself: tests/test_gpu_parity.py::test_gmmil_centred_gram_form_is_as_close_to_float64_as_the_direct_form, also run on the host emulator.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
import inputs as gi  # noqa: E402

f32 = np.float32


def _rewards(dxe, dxx, w, we, g1, g2, dt):
  wn, wen = (w / w.sum(dtype=dt)).astype(dt), (we / we.sum(dtype=dt)).astype(dt)
  sim = sum(wn * (np.exp(dt(-g) * dxe) @ wen) for g in (g1, g2))
  return sim - sum(wn * (np.exp(dt(-g) * dxx) @ wn) for g in (g1, g2)), sim


def _direct(X, Y, dt):
  out = np.empty((X.shape[0], Y.shape[0]), dt)
  for i in range(0, X.shape[0], 64):
    d = X[i:i + 64, None, :].astype(dt) - Y[None].astype(dt)
    out[i:i + 64] = (d * d).sum(2, dtype=dt) / dt(X.shape[1])
  return out


def _gram(X, Y, c):
  Xc, Yc = (X - c).astype(f32), (Y - c).astype(f32)
  nx, ny = (Xc * Xc).sum(1, dtype=f32), (Yc * Yc).sum(1, dtype=f32)
  return np.maximum((nx[:, None] + ny[None, :]) - f32(2) * (Xc @ Yc.T), f32(0)) / f32(X.shape[1])


@pytest.mark.parametrize('offset', [0.0, 50.0, 1000.0])
def test_centred_gram_form_keeps_the_direct_forms_accuracy_and_the_plain_gram_form_does_not(offset):
  X, E, w, we = gi.gmmil_case(5, 512, 512, 120)
  X, E = (X + f32(offset)).astype(f32), (E + f32(offset)).astype(f32)
  d64xe, d64xx = _direct(X, E, np.float64), _direct(X, X, np.float64)
  g1, g2 = 1 / np.median(d64xe), 1 / np.median(_direct(E, E, np.float64))
  r64, s64 = _rewards(d64xe, d64xx, w.astype(np.float64), we.astype(np.float64), g1, g2, np.float64)
  bound = 1e-5 * np.abs(s64).max()   # the bound of the GMMIL parity tests
  rows = (np.arange(8) * E.shape[0]) // 8
  c = E[rows].mean(0, dtype=f32)    # the kernel's centre: 8 rows spread over the expert batch
  err = lambda r: np.abs(r - r64).max()
  e_direct = err(_rewards(_direct(X, E, f32), _direct(X, X, f32), w, we, g1, g2, f32)[0])
  e_centred = err(_rewards(_gram(X, E, c), _gram(X, X, c), w, we, g1, g2, f32)[0])
  e_plain = err(_rewards(_gram(X, E, 0 * c), _gram(X, X, 0 * c), w, we, g1, g2, f32)[0])
  assert e_direct <= 0.1 * bound and e_centred <= 0.1 * bound, (e_direct / bound, e_centred / bound)
  assert e_centred <= 3 * e_direct + 1e-3 * bound
  if offset >= 50:
    assert e_plain > bound, e_plain / bound   # what the offset costs the uncentred product
