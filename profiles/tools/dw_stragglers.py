"""Developer tool (-DIL_TIMELINE build): which block jobs of the two k_dw_adam launches finish last?  IL_HIP_LIBRARY=variants/tl/libil_hip.so python profiles/tools/dw_stragglers.py"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
from imitation_learning_amd import _lib
K, W, S = 12, 512, 8
plan, nets, _ = bench.build(torch.device('cuda', 0), 0)
plan.capture(warmup=3)
for _ in range(50): plan.replay()
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (K * W * S))(); assert raw.il_debug_timeline_sac(buf) == 0
sac = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S).astype(np.int64)
for kid, name, per_net, nets_ in ((1, 'critic', 80, 2), (2, 'actor', 80, 1)):
  a = sac[kid]
  live = a[:, 7] > 0
  t0 = a[live, 0].min()
  n = per_net * nets_
  start, prod, done = (a[:n, 0] - t0) / 100.0, (a[:n, 1] - t0) / 100.0, (a[:n, 7] - t0) / 100.0
  kind = np.array([('HxH' if j % per_net < 64 else ('W1' if j % per_net < 72 else 'W3+b')) for j in range(n)])
  print(f'k_dw_adam ({name}): {n} block jobs; us after the first workgroup started')
  for k in ('HxH', 'W1', 'W3+b'):
    m = kind == k
    print(f'  {k:5s} n={m.sum():3d} start {np.median(start[m]):5.2f}  products done {np.median(prod[m]):5.2f}  done median {np.median(done[m]):5.2f} max {done[m].max():5.2f}')
  order = np.argsort(-done)[:10]
  print('  slowest:', [(int(j), kind[j], round(float(done[j]), 2), 'xcd %d' % (j % 8)) for j in order])
  for x in range(8):
    m = (np.arange(n) % 8) == x
    print(f'  xcd {x}: done median {np.median(done[m]):5.2f} max {done[m].max():5.2f}', end=';')
  print()
