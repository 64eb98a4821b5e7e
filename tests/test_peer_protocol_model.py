"""A model of the peer-window exchange protocol (imitation-learning_amd/csrc/peer.hip, include/il_hip.h il_peer_*) under adversarial interleavings, on the CPU.

What the kernel relies on, and what this checks with a randomised scheduler: every workgroup (rank, chunk) of an exchange pushes its chunk into slot [epoch parity][rank] of
EVERY rank's window, releases its epoch into the chunk's arrival line of every window, waits until all W arrival words of its own window have reached its epoch, reduces the W
slabs in rank order, then advances its per-chunk epoch. Launches of one rank are stream-ordered (exchange k + 1 starts when all workgroups of exchange k have finished on
that rank); nothing orders different ranks except the arrival words. Claims: (1) no deadlock - a workgroup pushes before it waits and never waits for its own GPU;
(2) two slot parities suffice - no rank can overwrite a slab another rank has not finished reading, however far a fast rank runs ahead; (3) every rank obtains the same
rank-ordered mean. A single slot parity must FAIL under the same scheduler (the model can tell the difference)."""
import random

import numpy as np
import pytest


class Window:
  def __init__(self, world, chunks, parities):
    self.slots = np.full((parities, world, chunks), np.nan)   # one float stands for a chunk's payload
    self.tag = np.full((parities, world, chunks), -1, dtype=np.int64)   # which exchange wrote it (model-only: detects overwrites of unread data)
    self.arrival = np.zeros((chunks, world), dtype=np.int64)


def workgroup(rank, chunk, world, windows, epochs, value_of, parities, results, violations):
  """One workgroup of one exchange as a generator: every `yield` is a point where the scheduler may run anything else."""
  e = epochs[rank][chunk] + 1
  par = e % parities
  v = value_of(rank, chunk, e)
  for i in range(1, world + 1):          # push: remote windows first, the own one last
    r = (rank + i) % world
    windows[r].slots[par, rank, chunk] = v
    windows[r].tag[par, rank, chunk] = e
    yield
  for r in range(world):                 # release the epoch into every window
    windows[r].arrival[chunk, rank] = e
    yield
  while not all(windows[rank].arrival[chunk, s] >= e for s in range(world)):   # wait (bounded in the kernel; here the scheduler guarantees progress)
    yield
  acc = 0.0
  for s in range(world):                 # reduce in rank order
    if windows[rank].tag[par, s, chunk] != e:
      violations.append((rank, chunk, e, s, int(windows[rank].tag[par, s, chunk])))
    acc += windows[rank].slots[par, s, chunk]
    yield
  results[(rank, chunk, e)] = acc / world
  epochs[rank][chunk] = e


def run(world, chunks, exchanges, parities, seed, bias=None):
  rng = random.Random(seed)
  windows = [Window(world, chunks, parities) for _ in range(world)]
  epochs = [[0] * chunks for _ in range(world)]
  value_of = lambda rank, chunk, e: float((rank + 1) * 1000 + chunk * 10 + e)
  results, violations = {}, []
  launched = [0] * world                     # exchanges launched so far per rank
  active = {r: [] for r in range(world)}     # live workgroup generators of the rank's current launch
  steps = 0
  while any(launched[r] < exchanges or active[r] for r in range(world)):
    for r in range(world):                   # stream order: the next launch starts when the previous one has drained
      if not active[r] and launched[r] < exchanges:
        active[r] = [workgroup(r, c, world, windows, epochs, value_of, parities, results, violations) for c in range(chunks)]
        launched[r] += 1
    ranks = [r for r in range(world) if active[r]]
    weights = [(bias[r] if bias else 1.0) for r in ranks]
    r = rng.choices(ranks, weights)[0]       # a biased scheduler lets one rank run far ahead of the others whenever the protocol allows it
    g = rng.choice(active[r])
    try:
      next(g)
    except StopIteration:
      active[r].remove(g)
    steps += 1
    assert steps < 2_000_000, 'no progress: deadlock in the model'
  return results, violations, value_of


@pytest.mark.parametrize('world,chunks', [(2, 1), (2, 3), (4, 2), (8, 3)])
def test_two_parities_are_enough_and_every_rank_gets_the_same_mean(world, chunks):
  for seed in range(12):
    bias = None if seed % 3 == 0 else [50.0 if r == seed % world else 1.0 for r in range(world)]   # one rank 50x more likely to be scheduled
    results, violations, value_of = run(world, chunks, exchanges=6, parities=2, seed=seed, bias=bias)
    assert not violations, f'a slab was overwritten before it was read: {violations[:3]}'
    for (rank, chunk, e), got in results.items():
      want = sum(value_of(s, chunk, e) for s in range(world)) / world
      assert got == want, (rank, chunk, e, got, want)
    assert len(results) == world * chunks * 6


def test_one_parity_is_not_enough():
  """The control: with a single slot per rank a fast rank's next push lands on a slab its peer is still reducing, and the scheduler finds it."""
  hit = 0
  for seed in range(40):
    bias = [50.0 if r == seed % 2 else 1.0 for r in range(2)]
    _, violations, _ = run(2, 2, exchanges=6, parities=1, seed=seed, bias=bias)
    hit += bool(violations)
  assert hit > 0, 'the model never exposed the single-buffer hazard: the scheduler is too tame to trust the positive result'
