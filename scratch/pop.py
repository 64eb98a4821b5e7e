import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch, time
import imitation_learning_amd as il
import bench
dev = torch.device('cuda', 0)
L = int(sys.argv[1])
plans = [bench.build(dev, 0, seed=l, learner_id=l)[0] for l in range(L)]
print('built', flush=True)
pop = il.PopulationPlan(plans)
for _ in range(3): pop.run()
torch.cuda.synchronize(); print('eager ok', flush=True)
t0 = time.perf_counter()
for _ in range(200): pop.run()
torch.cuda.synchronize(); print('eager', L * 200 / (time.perf_counter() - t0), 'updates/s', flush=True)
pop.capture()
print('captured', flush=True)
for _ in range(20): pop.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): pop.replay()
torch.cuda.synchronize(); print('graph', L * 500 / (time.perf_counter() - t0), 'updates/s', flush=True)
