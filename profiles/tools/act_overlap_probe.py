"""Does the acting launch (own stream, parameter snapshot) really run beside the captured update?  Measures the host-visible latency of
worker.act() alone, right after an update-graph replay, and the update's duration, for a few stream configurations."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
import imitation_learning_amd as il

dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
actor, memory = nets[0], plan.memory
for _ in range(3): plan.run()
obs = np.zeros(memory.state_size, np.float32)
out = {}
for prio in (-1, 0):
  w = il.ActingWorker(actor, memory, mirror=True)
  w.act_stream = torch.cuda.Stream(device=dev, priority=prio)
  plan.graph = None; plan.pre_hooks.clear(); plan.post_hooks.clear()
  w.attach(plan); plan.capture(warmup=0)
  for _ in range(20): w.act(obs)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(500): w.act(obs)
  alone = (time.perf_counter() - t0) / 500
  lat, tot = [], []
  for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.replay()
    t1 = time.perf_counter()
    w.act(obs)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    lat.append(t2 - t1); tot.append(t3 - t0)
  out[f'prio{prio}'] = dict(act_alone_us=round(alone * 1e6, 1), act_after_replay_us=round(float(np.median(lat)) * 1e6, 1), replay_plus_act_total_us=round(float(np.median(tot)) * 1e6, 1))
print(json.dumps(out))
