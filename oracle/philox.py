"""Philox4x32-10 counter RNG (TEST ORACLE, numpy) -- an independent restatement of the noise source of the HIP update kernels.

The reference draws its noise from torch's global CPU generator (`torch.manual_seed`, train.py:52: `policy.sample()` training.py:21,
`rsample()` :35, `rand_like` :118, `actor(state).sample()` train.py:152).  A generator stream cannot be reproduced on the GPU, and the
parity contract (BASELINE.json north_star) is "fixed seed + injected noise", so the kernels either take the noise as an input (the
per-function parity tests) or generate it on chip.  The on-chip source is Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11; the Random123 library's reference implementation publishes the known-answer vectors checked in
tests/test_oracle_golden.py) keyed by the 64-bit seed, with counter words {index, update counter, stream id, 0}:
  normal  = sqrt(-2 ln u1) cos(2 pi u2),  u = ((word >> 8) + 0.5) / 2^24   (open interval)     -- words 0, 1
  uniform = (word 0 >> 8) / 2^24                                            ([0, 1) like torch.rand)
This module restates that (integer part bit-exact; the float32 transform to within a few ulp of the device's libm) so that
`il_noise_fill` -- the recorder the timed-path parity tests use -- is itself pinned to something that is not the kernel source.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
STREAM_EPS_NEXT, STREAM_EPS_CUR, STREAM_GP, STREAM_ACT, STREAM_MIX = 1, 2, 3, 4, 7   # include/il_hip.h IL_NOISE_*


def philox4x32_10(c0, c1, c2, c3, k0, k1):
  """Vectorised over the counter words (uint32 arrays or scalars); returns the four output words."""
  c = [np.asarray(x, np.uint64) & np.uint64(0xFFFFFFFF) for x in np.broadcast_arrays(c0, c1, c2, c3)]
  k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
  mask, sh = np.uint64(0xFFFFFFFF), np.uint64(32)
  for _ in range(10):
    p0, p1 = M0 * c[0], M1 * c[2]
    c = [(p1 >> sh) ^ c[1] ^ np.uint64(k0), p1 & mask, (p0 >> sh) ^ c[3] ^ np.uint64(k1), p0 & mask]
    k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
  return [x.astype(np.uint32) for x in c]


def _words(seed, ctr, stream_id, n):
  idx = np.arange(n, dtype=np.uint64)
  return philox4x32_10(idx, np.uint64(ctr & 0xFFFFFFFF), np.uint64(stream_id), np.uint64(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def normal(seed: int, ctr: int, stream_id: int, n: int) -> np.ndarray:
  f32 = np.float32
  w = _words(seed, ctr, stream_id, n)
  u1 = ((w[0] >> np.uint32(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
  u2 = ((w[1] >> np.uint32(8)).astype(f32) + f32(0.5)) * f32(1.0 / 16777216.0)
  return (np.sqrt(f32(-2.0) * np.log(u1)) * np.cos(f32(6.28318530717958647692) * u2)).astype(f32)


def uniform(seed: int, ctr: int, stream_id: int, n: int) -> np.ndarray:
  w = _words(seed, ctr, stream_id, n)
  return ((w[0] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)
