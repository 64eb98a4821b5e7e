"""Developer tool: phase stamps inside k_gmmil_mfma (a -DIL_TIMELINE build), every workgroup of the LAST of a run of calls.
  bash profiles/tools/build_variants.sh tl:"-DIL_TIMELINE -w"; IL_HIP_LIBRARY=variants/tl/libil_hip.so python profiles/tools/gmmil_mfma_timeline.py"""
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
g = t[2]
g = g[g[:, 0] > 0]
t0 = g[:, 0].min()
print(f'k_gmmil_mfma, B = {Bg}, D = {Sg + Ag}: {len(g)} workgroups stamped; us after the first workgroup started; min / p10 / median / p90 / max')
names = ('started', 'loads requested, centre partials + weight sums in LDS', 'centre ready (two barriers)', 'operands centred: columns in LDS, rows in registers', 'barrier passed', 'Gram tiles + exponentials done (wave 0)',
         'partials handed over, drained, barrier', 'ticket taken')
for n, sl in zip(names, range(8)):
  a = g[:, sl] - t0
  a = a[g[:, sl] > 0]
  print(f'  {n:58s} {a.min():6.2f} {np.percentile(a, 10):6.2f} {np.median(a):6.2f} {np.percentile(a, 90):6.2f} {a.max():6.2f}')
d = g[:, 1:8] - g[:, 0:7]
for n, k in zip(('loads + centre partials', 'centre (2 barriers)', 'centring + LDS stores', 'barrier wait', 'Gram tiles + exponentials', 'hand-over + drain + barrier', 'ticket'), range(7)):
  v = d[:, k]
  print(f'  phase {n:32s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}')
