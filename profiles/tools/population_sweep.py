import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
import imitation_learning_amd as il
import bench
dev = torch.device('cuda', 0)
L = int(sys.argv[1])
plans = [bench.build(dev, 0, seed=l, learner_id=l)[0] for l in range(L)]
pop = il.BatchedPopulationPlan(plans)
for _ in range(3): pop.run()
torch.cuda.synchronize()
pop.capture()
for _ in range(20): pop.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): pop.replay()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('batched L=%d: %.1f updates/s aggregate, %.1f us per replay' % (L, L * 300 / dt, dt / 300 * 1e6), flush=True)
