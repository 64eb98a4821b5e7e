"""Developer tool: phase stamps inside k_gmmil_sx (a -DIL_TIMELINE build), every workgroup of the LAST of a run of calls.
  IL_HIP_LIBRARY=variants/tl/libil_hip.so python profiles/tools/gmmil_sx_timeline.py"""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib
dev = torch.device('cuda', 0)
rs = np.random.RandomState(5)
Sg, Ag, Bg = 112, 8, 1024
mk = lambda shift: (torch.from_numpy((rs.standard_normal((Bg, Sg)) + shift).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (Bg, Ag)).astype(np.float32)).to(dev))
(xs, xa), (es, ea) = mk(0.0), mk(0.5)
w = torch.ones(Bg, device=dev)
gm = il.GMMILDiscriminator(Sg, Ag, bench.Cfg(state_only=False))
for _ in range(200): gm.predict_reward(xs, xa, es, ea, w, w)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
K, W, Sl = 12, 512, 8
buf = (C.c_ulonglong * (K * W * Sl))()
assert raw.il_debug_timeline_gmmil(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, Sl).astype(np.float64) / 100.0
g = t[0]
g = g[g[:, 0] > 0]
t0 = g[:, 0].min()
print(f'k_gmmil_sx, B = {Bg}, D = {Sg + Ag}: {len(g)} workgroups stamped (x + gx * y of z = 0 .. 1 < 512); us after the first workgroup started; min / p10 / median / p90 / max')
names = ('started', 'operands requested + stored to LDS (before the barrier)', 'barrier passed', 'feature loop done (wave 0)', 'exp epilogue done (wave 0), partials in LDS', 'every wave there (barrier)', 'partials handed over, drained, barrier', 'ticket taken')
for n, sl in zip(names, range(8)):
  a = g[:, sl] - t0
  a = a[g[:, sl] > 0]
  print(f'  {n:58s} {a.min():6.2f} {np.percentile(a, 10):6.2f} {np.median(a):6.2f} {np.percentile(a, 90):6.2f} {a.max():6.2f}')
d = g[:, 1:8] - g[:, 0:7]
for n, k in zip(('prologue (loads + LDS stores)', 'barrier wait', 'feature loop (wave 0)', 'exp epilogue (wave 0)', 'waiting for the other waves', 'hand-over + drain + barrier', 'ticket'), range(7)):
  v = d[:, k]
  print(f'  phase {n:32s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}')
c = t[1] * 100.0   # raw s_memtime values of slots 2, 3 (the table divides by 100: undo)
m = c[:, 2] > 0
dt_clk, dt_us = c[m, 3] - c[m, 2], (t[0][m, 3] - t[0][m, 2])
print(f'  shader clock during the feature loop (s_memtime delta / 100 MHz delta): median {np.median(dt_clk / dt_us):.0f} ticks per us')
