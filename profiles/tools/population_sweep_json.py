"""Population sweep for profiles/r02_population_sweep.json: aggregate updates/s of BatchedPopulationPlan (captured) at L = 8, 16, 32, 64 learners, with the whole-update
roofline fractions.  python profiles/tools/population_sweep_json.py > gpurun_out/pop_sweep.json"""
import json, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
import imitation_learning_amd as il
import bench
dev = torch.device('cuda', 0)
_, _, ub, uf = bench.algorithmic_model()
out = dict(what='BatchedPopulationPlan (il_*_population launches, hipGraph replay), SAC+GAIL, batch 256 per learner, HalfCheetah dims, own 1e6-row ring / index stream / Philox counter per learner',
           algorithmic_bytes_per_update=ub, algorithmic_flops_per_update=uf, sweep=[])
plans = []
for L, G in ((8, 1), (16, 1), (32, 1), (64, 1), (64, 2)):   # G: sub-populations as parallel graph branches (BatchedPopulationPlan(groups=))
  plans += [bench.build(dev, 0, seed=len(plans) + l, learner_id=len(plans) + l)[0] for l in range(L - len(plans))]
  pop = il.BatchedPopulationPlan(plans, groups=G)
  for _ in range(3): pop.run()
  torch.cuda.synchronize()
  pop.capture()
  for _ in range(20): pop.replay()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(200): pop.replay()
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
  rate = L / dt
  out['sweep'].append(dict(learners=L, groups=G, aggregate_updates_per_s=round(rate, 1), us_per_replay=round(dt * 1e6, 1), fp32_frac=round(rate * uf / 1e12 / bench.FP32_PEAK_TFLOPS, 4),
                           hbm_frac=round(rate * ub / 1e9 / bench.HBM_PEAK_GBS, 4)))
  print(out['sweep'][-1], file=sys.stderr, flush=True)
  del pop
print(json.dumps(out, indent=1))
