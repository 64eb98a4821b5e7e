"""Developer tool: soak of the pair-mode hops (plain-store L2 hops behind the XCC_ID check, write-through arrivals read below the L1) UNDER UNEVEN LOAD. The MI355X guide's
warning: idle chips, uniform load and L1-cold consumers hide hand-off failures. One learner, N captured replays back to back; in a second process a load generator
hammers the same GPU (large device-to-device copies, or a GEMM loop) so that the update's workgroups are delayed unevenly. The digest of every persistent tensor must
equal the digest of the 16-wave schedule (IL_PAIR=0) run quietly - the arithmetic is deterministic, so ANY stale or torn hop in 2 N hops x 96 tiles shows.
  python profiles/tools/pair_soak.py [replays]"""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000

LEARNER = f"""
import sys, hashlib, numpy as np, torch; sys.path[:0] = ['.', 'tests', 'tests/golden']
import bench
from imitation_learning_amd import training as T
plan, nets, _ = bench.build(torch.device('cuda', 0), 0, seed=11)
import os
if os.environ.get('IL_SOAK_LAUNCH') == 'direct':   # (round 6) the product's default launch path on one GPU: UpdatePlan.launch_direct
  for _ in range(3): plan.run()
  torch.cuda.synchronize(); plan.record_direct(); step = plan.launch_direct
elif os.environ.get('IL_SOAK_LAUNCH') == 'dp':   # one rank of the data-parallel fused schedule: the gradient exchanges inside the optimiser launches (peer windows), direct launches
  os.environ.setdefault('IL_PEER_EXCHANGE', 'force')   # (one rank: the windows are this GPU's own)
  from imitation_learning_amd.parallel import DataParallelUpdate
  runner = DataParallelUpdate(plan)
  for _ in range(3): runner.run()
  torch.cuda.synchronize(); assert runner.direct_launch_ok(); runner.record_direct(); step = runner.launch_direct
elif os.environ.get('IL_SOAK_LAUNCH') == 'thread':   # the launcher thread (UpdatePlan.launch_async)
  for _ in range(3): plan.run()
  torch.cuda.synchronize(); plan.record_direct(); step = plan.launch_async
else:
  plan.capture(warmup=2); plan.replay(); step = plan.replay   # (3 updates ahead of the loop in every mode)
for _ in range({N}): step()
plan.join()
torch.cuda.synchronize()
assert plan.sync_timeouts() == 0
h = hashlib.sha256()
for n in list(nets) + [plan.logp, plan.q, plan.rewards, plan.idx]: h.update(np.ascontiguousarray((n.flat if hasattr(n, 'flat') else n).detach().cpu().numpy()).tobytes())
import ctypes
from imitation_learning_amd import _lib as _L
chk = ''
if hasattr(_L.lib(), 'il_debug_check'):   # developer build -DIL_EXP_CHECK: consumed-vs-produced counters of the fence-free hand-offs
  buf = (ctypes.c_uint32 * 16)(); _L.lib().il_debug_check(buf); chk = ' CHECK ' + ','.join(str(int(v)) for v in buf[:9])
print('DIGEST', h.hexdigest()[:16] + chk.replace(' ', '_'))
"""
LOADS = {
    'copies': "import torch, time\na = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)\nt = time.time()\nwhile time.time() - t < 60: b.copy_(a); torch.cuda.synchronize()\n",
    'gemms': "import torch, time\na = torch.randn(4096, 4096, device='cuda'); t = time.time()\nwhile time.time() - t < 60: (a @ a).sum().item()\n",
    'bursts': "import torch, time\na = torch.empty(64 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)\nt = time.time()\nwhile time.time() - t < 60:\n  b.copy_(a); torch.cuda.synchronize(); time.sleep(0.0003)\n",
}


def learner(env):
  t = time.time()
  r = subprocess.run([sys.executable, '-c', LEARNER], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  return [l for l in r.stdout.splitlines() if l.startswith('DIGEST')][-1].split()[1], time.time() - t


want, dt = learner(dict(IL_PAIR='0'))
print(f'16-wave schedule, quiet: {want[:16]} ({N} replays, {dt:.1f} s)', flush=True)
ok = True
for name in [None] + list(LOADS):
  bg = subprocess.Popen([sys.executable, '-c', LOADS[name]], cwd=ROOT) if name else None
  if bg: time.sleep(3)
  got, dt = learner(dict(IL_PAIR='1'))
  if bg: bg.kill(); bg.wait()
  print(f'pair mode, load = {name}: {got[:16]} {"OK" if got == want else "MISMATCH"} ({dt:.1f} s)', flush=True)
  ok &= got == want
print(json.dumps(dict(replays=N, identical=ok)))
sys.exit(0 if ok else 1)
