"""Developer tool: per-kernel HIP-event times of the data-parallel split path on ONE rank (eager launches), with the gradient exchange selected by the environment:
  IL_PEER_EXCHANGE=0|force  IL_PEER_FUSED=0|1  [IL_DP_HANDOFF=0|1]
Usage: IL_PEER_EXCHANGE=force IL_PEER_FUSED=1 python profiles/tools/dp_kernel_times.py"""
import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
import bench
from imitation_learning_amd import _lib
from imitation_learning_amd.parallel import DataParallelUpdate
dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
dp = DataParallelUpdate(plan)
for _ in range(20): dp.run()
torch.cuda.synchronize()
L = _lib.lib()
L.il_trace_enable(1)
N = 50
for _ in range(N): dp.run()
buf = C.create_string_buffer(1 << 16)
L.il_trace_report(buf, len(buf))
L.il_trace_enable(0)
tot = 0.0
for line in buf.value.decode().strip().splitlines():
  name, cnt, ms = line.split()
  print(f'{name:24s} {int(cnt) / N:5.1f} per update  {float(ms) / int(cnt) * 1e3:8.2f} us avg')
  tot += float(ms)
print(f'sum of kernel time per update: {tot / N * 1e3:.1f} us; exchange: peer={dp.peer is not None} fused={sorted(dp.peer.fused) if dp.peer else None} handoff={dp.handoff} timeouts={dp.exchange_timeouts()}')
