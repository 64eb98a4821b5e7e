"""CPU: what keeps the first real multi-GPU run safe - none of it needs a GPU.
  * `python bench.py --gpus N` starts its own ranks and REFUSES to report an N > 1 number from fewer devices (it used to print n_gpus 1);
  * `parallel.replicas_bit_identical`: the cross-rank digest of the replica state, world size 2 over gloo, equal and unequal;
  * `parallel.Watchdog`: a rank whose peer never arrives in a collective ends itself (exit 124) instead of hanging;
  * the collective agreement helpers take the same branch on every rank."""
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
  return p


def _run_ranks(script_body, tmp_path, world=2, timeout=120, env=None):
  script = tmp_path / 'worker.py'
  script.write_text('import os, sys\nsys.path.insert(0, %r)\n' % ROOT + script_body)
  port = _free_port()
  procs = []
  for r in range(world):
    e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **(env or {}))
    procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  out = []
  t0 = time.time()
  for p in procs:
    try:
      o, e = p.communicate(timeout=max(1.0, timeout - (time.time() - t0)))
    except subprocess.TimeoutExpired:
      p.kill(); o, e = p.communicate()
      out.append((-9, o, e)); continue
    out.append((p.returncode, o, e))
  return out


def test_bench_refuses_more_gpus_than_visible():
  """On this box there is no GPU at all: `--gpus 2` must fail loudly, before importing anything GPU-side, and must not print a JSON line."""
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=300,
                     env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
  assert r.returncode == 2, (r.returncode, r.stderr[-500:])
  assert 'needs 2 visible GPUs' in r.stderr and '"metric"' not in r.stdout


def test_bench_rejects_a_world_size_that_is_not_gpus():
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True, timeout=300,
                     env=dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'))
  assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and '"metric"' not in r.stdout


DIGEST = r'''
import torch, torch.distributed as dist, json
from imitation_learning_amd import parallel
dist.init_process_group('gloo')
rank = dist.get_rank()
a, b = torch.arange(1000, dtype=torch.float32), torch.ones(7)
same, d = parallel.replicas_bit_identical([a, None, b])
c = b.clone(); c[3] = 1.0 + (rank * 2.0 ** -23)          # one ulp apart on rank 1
diff, d2 = parallel.replicas_bit_identical([a, c])
agreed = parallel._agree(rank == 0)                       # one rank says no -> every rank hears no
low = parallel.agree_min(5.0 + rank)
json.dump(dict(same=same, d=d, diff=diff, d2=d2, agreed=agreed, low=low, local=parallel.replica_digest([a, b])), open(os.path.join(sys.argv[1], f'digest{rank}.json'), 'w'))
dist.barrier(); dist.destroy_process_group()
'''


def test_replica_digest_agrees_and_disagrees_collectively(tmp_path):
  import json
  res = _run_ranks(DIGEST, tmp_path)
  assert all(rc == 0 for rc, _, _ in res), res
  r0, r1 = (json.load(open(tmp_path / f'digest{r}.json')) for r in (0, 1))
  for r in (r0, r1):
    assert r['same'] is True and len(set(r['d'])) == 1 and len(r['d']) == 2
    assert r['diff'] is False and len(set(r['d2'])) == 2
    assert r['agreed'] is False and r['low'] == 5.0
  assert r0['d'] == r1['d'] and r0['d2'] == r1['d2'], 'every rank sees the same list of digests'
  assert r0['local'].startswith(r0['d'][0]), 'the gathered digest is a prefix (240 bits) of the local sha256'


WATCHDOG = r'''
import time, torch, torch.distributed as dist
from imitation_learning_amd import parallel
dist.init_process_group('gloo')
rank = dist.get_rank()
dog = parallel.Watchdog(3.0, what='test')
dog.beat('before the collective')
if rank == 1:
  for _ in range(600):        # a stalled peer: alive (so gloo sees no broken connection), beating its own watchdog, never entering the collective
    time.sleep(0.1); dog.beat('stalled on purpose')
  os._exit(0)
t = torch.ones(4)
dist.all_reduce(t)            # rank 0 blocks here; its watchdog must end it
print('UNREACHABLE')
'''


def test_watchdog_ends_a_rank_whose_peer_never_arrives(tmp_path):
  script = tmp_path / 'worker.py'
  script.write_text('import os, sys\nsys.path.insert(0, %r)\n' % ROOT + WATCHDOG)
  port = _free_port()
  env = lambda r: dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  p1 = subprocess.Popen([sys.executable, str(script)], env=env(1), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
  p0 = subprocess.Popen([sys.executable, str(script)], env=env(0), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
  t0 = time.time()
  try:
    out, err = p0.communicate(timeout=45)
  finally:
    p1.kill(); p1.communicate()
  assert p0.returncode == 124, (p0.returncode, err[-800:])
  assert 'UNREACHABLE' not in out and '[watchdog] rank 0' in err and 'before the collective' in err
  assert time.time() - t0 < 40


def test_watchdog_is_quiet_while_the_loop_beats():
  sys.path.insert(0, ROOT)
  from imitation_learning_amd import parallel
  dog = parallel.Watchdog(1.0, what='test')
  for _ in range(25):
    time.sleep(0.1); dog.beat('working')
  dog.stop()
  assert True   # still alive: the heartbeat kept the watchdog quiet for 2.5 x its bound
