"""Developer probe: the tile phases of k_sac_chain's forward roles when the weight panels have NOT been rewritten since the previous launch (the forward-only chain of
il_sac_update, launched back to back) against the same phases inside a real update (profiles/tools/update_timeline.py, where the previous AdamW kernel has just rewritten
every panel): separates what a layer costs from what fetching freshly written weights costs. Needs a -DIL_TIMELINE build:
  IL_HIP_LIBRARY=variants/tl/libil_hip.so python profiles/tools/warm_forward_probe.py [launches] [rows per tile: 16]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
from imitation_learning_amd import _lib

K, W, S = 12, 512, 8
dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
plan.sample_all()
L, st = _lib.lib(), _lib.stream_ptr()
sync, plan.sac.sync = plan.sac.sync, None   # a plain stream-ordered launch: nothing to hand over
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
_lib.check(L.il_sac_prepare(C.byref(plan.sac), st))
for _ in range(n):
  _lib.check(L.il_sac_update(C.byref(plan.sac), C.byref(plan.pb), None, None, _lib.ptr(plan.logp), _lib.ptr(plan.q), _lib.IL_FLAG_SAC_FORWARD_ONLY | _lib.IL_FLAG_SAC_PREPARED, st))
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (K * W * S))()
assert raw.il_debug_timeline_sac(buf) == 0
sac = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S).astype(np.int64)
nt = plan.B // (int(sys.argv[2]) if len(sys.argv) > 2 else 16)   # tiles per network (the experiment branch with 8-row tiles: pass 8)
c = sac[0][:6 * nt]
t0 = c[:, 0].min()
print(f'forward-only chain, {n} launches back to back, weights untouched in between; us after the first workgroup started (median over workgroups)')
print(f'  launched {np.median((c[:, 0] - t0) / 100):.2f} | last workgroup done {((c[:, 7].max() - t0) / 100):.2f}')
for kid, name in ((5, "actor(s') tile"), (6, 'actor(s) tile')):
  a = sac[kid]
  a = a[(a[:, 0] >= t0) & (a[:, 5] >= t0)]
  d = (a[:, 1:6] - a[:, 0:5]) / 100.0
  print(f'  {name:16s} (n = {len(a)}) rows gathered {np.median(d[:, 0]):.2f} | layer 1 {np.median(d[:, 1]):.2f} | layer 2 {np.median(d[:, 2]):.2f} | head GEMM {np.median(d[:, 3]):.2f} | sample, log-prob, stores {np.median(d[:, 4]):.2f}')
a = sac[8]
a = a[(a[:, 0] >= t0) & (a[:, 3] >= t0)]
d = (a[:, 1:4] - a[:, 0:3]) / 100.0
print(f'  critic / target forward tiles (n = {len(a)}): rows (targets: incl. the wait for a\') median {np.median(d[:, 0]):.2f} | layer 1 {np.median(d[:, 1]):.2f} | layer 2 {np.median(d[:, 2]):.2f}')
