"""`-m gpu`: the headline schedule (device hand-offs between two unjoined streams, direct launches) next to a busy NEIGHBOUR process.

Round 6 found a silent race this way (profiles/r06_soak_under_load.md): with a second process saturating the memory system, about one update in 10^5 relabelled a 16-row
tile with discriminator parameters that were only partly this update's - k_gail_reduce's "every thread stores, barrier, thread 0 release-add" had become visible to the other
XCDs before all of its lines were in memory.  No wait expired, nothing raised; only the digest of the learner differed from the quiet run's.  The schedule must be a pure
function of its inputs whatever else the GPU is doing: 50 000 updates, four times, beside a process that copies 256 MiB buffers back to back, inside this pytest process's own
(idle) GPU context - the worst situation found (the round-5 hand-offs failed one such run in four)."""
import os
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UPDATES, RUNS = 50000, 4


def _learner_source():
  sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
  src = open(os.path.join(ROOT, 'profiles', 'tools', 'pair_soak.py')).read()
  return src.split('LEARNER = f"""')[1].split('"""')[0].replace('{N}', str(UPDATES))


COPIES = ("import torch, time\na = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)\nt = time.time()\n"
          "while time.time() - t < 120: b.copy_(a); torch.cuda.synchronize()\n")


def _digest(source):
  r = subprocess.run([sys.executable, '-c', source], env=dict(os.environ, IL_SOAK_LAUNCH='direct'), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]   # (the learner asserts sync_timeouts() == 0 itself)
  return [l for l in r.stdout.splitlines() if l.startswith('DIGEST')][-1].split()[1]


def _runs_beside_the_neighbour(source, runs):
  got = []
  for _ in range(runs):
    neighbour = subprocess.Popen([sys.executable, '-c', COPIES], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
      time.sleep(3)   # the neighbour's copies are running before the learner's first launch
      got.append(_digest(source))
    finally:
      neighbour.kill(); neighbour.wait()
  return got


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_updates_beside_a_copy_hammering_process_equal_the_quiet_run():
  """Round-5 hand-offs: one such run in four differed; the committed ones: 0 of 108. A single mismatch gets a second batch and the test fails only if that one mismatches too
  (the round-5 behaviour fails ~50 % of the time, a one-in-a-hundred residual ~0.2 %)."""
  source = _learner_source()
  quiet = _digest(source)
  got = _runs_beside_the_neighbour(source, RUNS)
  if got != [quiet] * RUNS:
    again = _runs_beside_the_neighbour(source, RUNS)
    assert again == [quiet] * RUNS, f'{UPDATES} updates beside a busy neighbour process differ from the quiet run in two batches: {quiet} vs {got} and {again}'
