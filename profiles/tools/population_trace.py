"""Developer tool: per-kernel HIP-event times of the batched population path.  python profiles/tools/population_trace.py <learners>"""
import ctypes as C
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch
import imitation_learning_amd as il
from imitation_learning_amd import _lib
import bench
dev = torch.device('cuda', 0)
Lp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pop = il.BatchedPopulationPlan([bench.build(dev, 0, seed=l, learner_id=l)[0] for l in range(Lp)])
for _ in range(5): pop.run()
torch.cuda.synchronize()
L = _lib.lib(); L.il_trace_enable(1)
for _ in range(20): pop.run()
buf = C.create_string_buffer(1 << 16); _lib.check(L.il_trace_report(buf, len(buf))); L.il_trace_enable(0)
bytes_k, flops_k, ub, uf = bench.algorithmic_model()
tot = 0
for line in buf.value.decode().strip().splitlines():
  name, cnt, ms = line.split(); us = float(ms) / int(cnt) * 1e3; tot += us
  extra = ''
  if name in flops_k: extra += f'  {Lp * flops_k[name] / us / 1e6:.2f} TFLOP/s'
  if name in bytes_k: extra += f'  {Lp * bytes_k[name] / us / 1e3:.1f} GB/s'
  print(f'{name:18s} {us:8.2f} us{extra}')
print('sum', round(tot, 1), 'us for', Lp, 'learners ->', round(Lp / tot * 1e6), 'updates/s if back to back')
