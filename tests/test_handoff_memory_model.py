"""A model of the in-launch hand-off between two XCDs (imitation-learning_amd/csrc/il_common.hpp sync_signal / sync_wait, gail.hip k_gail_reduce -> sac.hip relabel role)
under an adversarial memory system, on the CPU - which form relies on which property of the hardware (profiles/r06_soak_under_load.md).

The machine of the model: one memory; a private write-back L2 per XCD (two XCDs' L2s are not coherent inside a launch: profiles/r06_l2_stale_probe.txt); stores, cache
invalidates and L2 write-backs are REQUESTS that an adversary delivers whenever it likes, except that (a) a wave that drains (s_waitcnt vmcnt(0)) waits for its own stores,
(b) an invalidate is ordered against the later loads of the wave that issued it, (c) a written-through store, once delivered, is in memory, (d) atomics on the counter act
on memory at once. `async_writeback`: whether the write-back behind a release-add may still be on its way when the add is visible (the property the measurements could not
rule out for a saturated memory system).

Forms: round 5 (plain stores, bare barrier, thread 0 releases; thread 0 acquires, plain loads) must FAIL; round 5's producer with round 6's consumer (loads below the
caches) must still FAIL (measured: 27 % of the runs either way); drain + release + every wave acquires is correct exactly as long as a write-back is complete at the add;
the committed form (written-through producer, drained, loads below the caches) needs none of that."""
import random

import pytest

LINES, OLD, NEW = 6, 0, 1


class Machine:
  def __init__(self, rng, async_writeback):
    self.rng, self.async_writeback = rng, async_writeback
    self.mem = {a: OLD for a in range(LINES)}; self.mem['ctr'] = 0
    self.l2 = [dict(), {a: OLD for a in range(LINES)}]   # XCD 0: producer, XCD 1: consumer - whose L2 still holds the lines the discriminator's own gradient launch read
    self.dirty = [set(), set()]
    self.requests = []   # [owner, kind, payload]: delivered by the adversary
    self.seq = 0

  def request(self, owner, kind, *payload):
    self.seq += 1
    r = [owner, kind, payload, self.seq]
    self.requests.append(r)
    return r

  def pending(self, owner):
    return any(r[0] == owner for r in self.requests)

  def deliver(self, r):
    self.requests.remove(r)
    _, kind, p, _ = r
    if kind == 'store': xcd, a, v = p; self.l2[xcd][a] = v; self.dirty[xcd].add(a)
    elif kind == 'through': xcd, a, v = p; self.mem[a] = v; self.l2[xcd][a] = v; self.dirty[xcd].discard(a)
    elif kind == 'invalidate': xcd, = p; self.l2[xcd] = {a: v for a, v in self.l2[xcd].items() if a in self.dirty[xcd]}
    elif kind == 'writeback': lines, = p; self.mem.update(lines)

  def writeback(self, owner, xcd):   # what the L2 holds dirty NOW; complete at once, or a request of its own
    lines = {a: self.l2[xcd][a] for a in self.dirty[xcd]}
    self.dirty[xcd].clear()
    if self.async_writeback: self.request(owner, 'writeback', lines)
    else: self.mem.update(lines)

  def load(self, xcd, a, coherent):
    if coherent: return self.mem[a]
    if a not in self.l2[xcd]: self.l2[xcd][a] = self.mem[a]
    return self.l2[xcd][a]


def producer_wave(m, w, waves, barrier, through, drain):
  me = ('p', w)
  for a in range(w, LINES, waves):
    m.request(me, 'through' if through else 'store', 0, a, NEW)
    yield
  if drain:
    while m.pending(me): yield
  barrier[0] += 1
  while barrier[0] < waves: yield
  if w == 0:   # the release-add of thread 0: write-back of what the L2 holds, its own outstanding requests, then the counter
    m.writeback(me, 0)
    while any(r[0] == me and r[1] != 'writeback' for r in m.requests): yield
    if not m.async_writeback:
      while m.pending(me): yield
    m.mem['ctr'] += 1


def consumer_wave(m, w, waves, barrier, acquire, coherent, got):
  me = ('c', w)
  if w == 0:
    while m.mem['ctr'] < 1: yield
    if acquire == 'leader': m.request(me, 'invalidate', 1)   # (ordered against THIS wave's later loads only)
  barrier[0] += 1
  while barrier[0] < waves: yield
  if acquire == 'all': m.request(me, 'invalidate', 1)
  while any(r[0] == me and r[1] == 'invalidate' for r in m.requests): yield   # a wave's own invalidate precedes its own loads
  for a in range(w, LINES, waves):
    got[a] = m.load(1, a, coherent)
    yield


def run(seed, through, drain, acquire, coherent, async_writeback):
  rng = random.Random(seed)
  m = Machine(rng, async_writeback)
  pb, cb, got = [0], [0], {}
  procs = [producer_wave(m, w, 3, pb, through, drain) for w in range(3)] + [consumer_wave(m, w, 2, cb, acquire, coherent, got) for w in range(2)]
  for _ in range(100000):
    if not procs and not m.requests: break
    if m.requests and (not procs or rng.random() < 0.35):
      m.deliver(rng.choice(m.requests))
      continue
    p = rng.choice(procs)
    try: next(p)
    except StopIteration: procs.remove(p)
  assert not procs, 'the model did not terminate'
  return all(got[a] == NEW for a in range(LINES))


FORMS = {
    'round 5': dict(through=False, drain=False, acquire='leader', coherent=False),
    'round-5 producer, loads below the caches': dict(through=False, drain=False, acquire='leader', coherent=True),
    'drain + release, every wave acquires': dict(through=False, drain=True, acquire='all', coherent=False),
    'committed: written through, drained, loads below the caches': dict(through=True, drain=True, acquire='leader', coherent=True),
}
SEEDS = range(400)


def failures(form, async_writeback):
  return sum(not run(seed, async_writeback=async_writeback, **FORMS[form]) for seed in SEEDS)


@pytest.mark.parametrize('async_writeback', [False, True])
def test_the_round_5_forms_fail_in_the_model(async_writeback):
  assert failures('round 5', async_writeback) > 0
  assert failures('round-5 producer, loads below the caches', async_writeback) > 0   # the consumer side alone does not fix it (measured: still 27 % of the runs)


def test_drain_and_release_is_correct_exactly_while_a_writeback_is_complete_at_the_add():
  assert failures('drain + release, every wave acquires', async_writeback=False) == 0
  assert failures('drain + release, every wave acquires', async_writeback=True) > 0   # (measured: 27 % -> ~2 % of the runs in a bare shell, 2 of 8 inside the test suite)


@pytest.mark.parametrize('async_writeback', [False, True])
def test_the_committed_form_needs_neither_property(async_writeback):
  assert failures('committed: written through, drained, loads below the caches', async_writeback) == 0


def test_a_leader_only_acquire_with_cached_loads_can_be_overtaken():
  """drain + release with thread 0's acquire alone: the other wave's plain loads may run ahead of the invalidate (why sync_wait_leader is reserved for data no cache of the
  XCD can hold a pre-write copy of)."""
  form = dict(through=False, drain=True, acquire='leader', coherent=False)
  assert sum(not run(seed, async_writeback=False, **form) for seed in SEEDS) > 0
