"""Developer tool (round 6): WHICH tensor diverges first when N back-to-back updates run beside a copy-hammering process? Checkpoint digests of every persistent tensor and of
the per-update outputs every K updates, quiet run first (reference), then under load; prints the first checkpoint that differs and the tensors that differ there.
  python profiles/tools/soak_first_divergence.py [updates] [every] [runs]   (IL_SOAK_LAUNCH=direct|graph)"""
import hashlib, os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 250
RUNS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
LEARNER = f"""
import sys, hashlib, json, os, numpy as np, torch; sys.path[:0] = ['.', 'tests', 'tests/golden']
import bench
plan, nets, _ = bench.build(torch.device('cuda', 0), 0, seed=11)
for _ in range(3): plan.run()
torch.cuda.synchronize()
if os.environ.get('IL_SOAK_LAUNCH', 'direct') == 'direct': plan.record_direct(); step = plan.launch_direct
else: plan.capture(warmup=0); step = plan.replay
names = ['actor', 'critic', 'target', 'log_alpha', 'disc', 'sn', 'logp', 'q', 'rewards', 'idx', 'eidx']
out = []
for c in range({N} // {K}):
  for _ in range({K}): step()
  plan.join(); torch.cuda.synchronize()
  ts = [n.flat if hasattr(n, 'flat') else n for n in nets] + [nets[4].sn, plan.logp, plan.q, plan.rewards, plan.idx, plan.eidx]
  out.append([hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:12] for t in ts])
assert plan.sync_timeouts() == 0
print('DIGESTS', json.dumps(dict(names=names, d=out)))
"""
COPIES = "import torch, time\\na = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)\\nt = time.time()\\nwhile time.time() - t < 300: b.copy_(a); torch.cuda.synchronize()\\n"
def learner():
  r = subprocess.run([sys.executable, '-c', LEARNER], cwd=ROOT, capture_output=True, text=True, timeout=1200)
  assert r.returncode == 0, r.stderr[-1500:]
  return json.loads([l for l in r.stdout.splitlines() if l.startswith('DIGESTS')][-1][8:])
ref = learner()
print(f'quiet reference: {len(ref["d"])} checkpoints of {K} updates', flush=True)
for run in range(RUNS):
  bg = subprocess.Popen([sys.executable, '-c', COPIES.replace('\\n', chr(10))], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  time.sleep(3)
  got = learner()
  bg.kill(); bg.wait()
  first = next((i for i, (a, b) in enumerate(zip(ref['d'], got['d'])) if a != b), None)
  if first is None: print(f'run {run}: identical', flush=True); continue
  diff = [n for n, a, b in zip(ref['names'], ref['d'][first], got['d'][first]) if a != b]
  print(f'run {run}: first divergence at checkpoint {first} (updates {first * K + 1} .. {(first + 1) * K}): tensors that differ there: {diff}', flush=True)
