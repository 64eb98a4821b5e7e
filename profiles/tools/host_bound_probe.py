"""Developer tool: is the update loop GPU-bound or bound by the host's hipGraphLaunch rate? Times the enqueue loop and the final wait separately
(a GPU-bound loop spends its time blocked on queue back-pressure or in the final synchronize; a host-bound one finishes with nothing left to wait for),
and the host cost of one replay of each graph with an idle GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
for _ in range(5): plan.run()
torch.cuda.synchronize()
plan.capture(warmup=0)
for _ in range(300): plan.replay()
torch.cuda.synchronize()
N = 3000
t0 = time.perf_counter()
for _ in range(N): plan.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'loop {1e6 * (t1 - t0) / N:.2f} us/update enqueue, final wait {1e3 * (t2 - t1):.2f} ms, total {1e6 * (t2 - t0) / N:.2f} us/update')
# host cost of a replay when the GPU is idle (sync between iterations)
for name, fn in (('both graphs', plan.replay),):
  ts = []
  for _ in range(200):
    torch.cuda.synchronize()
    a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
  ts.sort()
  print(f'{name}: host time per replay, idle GPU: median {1e6 * ts[len(ts) // 2]:.2f} us, p10 {1e6 * ts[len(ts) // 10]:.2f} us')
print('sync timeouts', plan.sync_timeouts())
