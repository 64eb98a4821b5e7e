"""Developer tool (round 6): does the priority of the stream that carries the SAC branch change the update period? (kernel-to-kernel boundaries are command-processor work; the
discriminator branch sits in another hardware queue)  python profiles/tools/stream_priority_probe.py"""
import sys, time
sys.path[:0] = ['.', 'tests', 'tests/golden']
import torch, bench
from imitation_learning_amd import training as il_training
dev = torch.device('cuda', 0)
def rate(prio, side_prio=None):
  il_training._NOISE.clear(); il_training._WS.clear()
  s = torch.cuda.Stream(priority=prio) if prio is not None else torch.cuda.current_stream()
  with torch.cuda.stream(s):
    plan, nets, _ = bench.build(dev, 0)
    if side_prio is not None: plan.side = torch.cuda.Stream(priority=side_prio)
    for _ in range(5): plan.run()
    torch.cuda.synchronize()
    plan.record_direct()
    for _ in range(300): plan.launch_direct()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3000): plan.launch_direct()
    torch.cuda.synchronize()
    r = 3000 / (time.perf_counter() - t0)
    assert plan.sync_timeouts() == 0
  return r
for rep in range(2):
  print(f'main on the default stream: {rate(None):.0f} | main priority -1 (high): {rate(-1):.0f} | main high, side low (0): {rate(-1, 0):.0f} | main default, side high: {rate(None, -1):.0f}', flush=True)
