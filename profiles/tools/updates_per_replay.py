"""Developer tool: what does the graph-to-graph boundary cost? The headline plan captured with K consecutive updates per replay (UpdatePlan.capture(updates=K): the offline
regime, several updates per environment step), K = 1, 2, 4: updates/s and the implied cost of one replay boundary.   python profiles/tools/updates_per_replay.py"""
import sys, time
sys.path.insert(0, '.')
import torch, bench
from imitation_learning_amd import training as T

dev = torch.device('cuda', 0)
res = {}
for K in (1, 2, 4, 1, 2, 4):
  T._NOISE.clear(); T._WS.clear()
  plan, nets, _ = bench.build(dev, 0)
  plan.capture(warmup=3, updates=K)
  for _ in range(300 // K): plan.replay()
  torch.cuda.synchronize()
  n = 3000 // K
  t0 = time.perf_counter()
  for _ in range(n): plan.replay()
  torch.cuda.synchronize()
  us = 1e6 * (time.perf_counter() - t0) / (n * K)
  assert plan.sync_timeouts() == 0
  res.setdefault(K, []).append(us)
  print(f'updates per replay {K}: {us:.2f} us per update = {1e6 / us:.0f} updates/s', flush=True)
t1, t4 = min(res[1]), min(res[4])
print(f'implied cost of one replay boundary: {(t1 - t4) * 4 / 3:.2f} us')
