"""Developer tool: per-kernel HIP-event times of the GMMIL reward at B = 1024, Ant dims (D = 120)."""
import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib
dev = torch.device('cuda', 0)
rs = np.random.RandomState(5)
Sg, Ag, Bg = 112, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mk = lambda shift: (torch.from_numpy((rs.standard_normal((Bg, Sg)) + shift).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (Bg, Ag)).astype(np.float32)).to(dev))
(xs, xa), (es, ea) = mk(0.0), mk(0.5)
w = torch.ones(Bg, device=dev)
gm = il.GMMILDiscriminator(Sg, Ag, bench.Cfg(state_only=False))
for _ in range(20): gm.predict_reward(xs, xa, es, ea, w, w)
torch.cuda.synchronize()
L = _lib.lib(); L.il_trace_enable(1)
for _ in range(100): gm.predict_reward(xs, xa, es, ea, w, w)
buf = C.create_string_buffer(1 << 14); L.il_trace_report(buf, len(buf)); L.il_trace_enable(0)
pf = 2 * Bg * Bg * (Sg + Ag)
for line in buf.value.decode().strip().splitlines():
  n, c, ms = line.split(); us = float(ms) / int(c) * 1e3
  print(f'{n:16s} {us:8.2f} us' + (f'  {3 * pf / us / 1e6:.1f} TFLOP/s (3 flop per pair-feature), {pf / us / 1e6:.2f} T pair-features/s' if 'tile' in n else ''))
