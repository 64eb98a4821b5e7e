"""MT19937 + numpy-legacy masked-rejection `randint` (TEST ORACLE, pure Python).

Follows what `np.random.randint(0, high)` does for the reference's index draw
(`memory.py:51-56`): legacy RandomState -> `_rand_int64` -> masked rejection on
32-bit outputs when the range fits 32 bits (numpy `_bounded_integers.pyx`,
`buffered_bounded_masked_uint32`; `legacy` mode never buffers).  Algorithm is
the published Matsumoto-Nishimura generator (`init_genrand`, `genrand_int32`).
"""
from __future__ import annotations

N, M = 624, 397
MATRIX_A, UPPER, LOWER = 0x9908B0DF, 0x80000000, 0x7FFFFFFF


class MT19937:
  def __init__(self, seed: int):
    self.mt = [0] * N
    self.mt[0] = seed & 0xFFFFFFFF
    for i in range(1, N):  # init_genrand
      self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
    self.pos = N

  def _twist(self):
    mt = self.mt
    for k in range(N):
      y = (mt[k] & UPPER) | (mt[(k + 1) % N] & LOWER)
      mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (MATRIX_A if y & 1 else 0)
    self.pos = 0

  def next_uint32(self) -> int:
    if self.pos >= N:
      self._twist()
    y = self.mt[self.pos]
    self.pos += 1
    y ^= y >> 11
    y ^= (y << 7) & 0x9D2C5680
    y ^= (y << 15) & 0xEFC60000
    y ^= y >> 18
    return y & 0xFFFFFFFF

  def randint(self, high: int) -> int:
    """np.random.randint(0, high) for 0 < high <= 2**32."""
    rng = high - 1
    if rng == 0:
      return 0
    mask = rng
    for s in (1, 2, 4, 8, 16):
      mask |= mask >> s
    while True:
      v = self.next_uint32() & mask
      if v <= rng:
        return v


def sample_indices(gen: MT19937, n: int, size: int, idx: int, full: bool):
  """`[memory._sample_idx() for _ in range(n)]` (memory.py:51-59)."""
  out = []
  high = size if full else idx - 1
  excl = (idx - 1) % size
  for _ in range(n):
    while True:
      v = gen.randint(high)
      if v != excl:
        break
    out.append(v)
  return out
