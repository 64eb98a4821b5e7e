"""Developer tool: per-workgroup timeline of k_gmmil_tile<0> from a debug build that stores s_memtime at phase boundaries for EVERY workgroup
(profiles/tools/gmmil_dbg_build.py generates and builds it from gmmil.hip; not part of the product library).
  IL_HIP_LIBRARY=imitation-learning_amd/csrc/build/ab/libil_hip_dbg.so python profiles/tools/gmmil_timeline.py"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib

dev = torch.device('cuda', 0)
L = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
rs = np.random.RandomState(5)
Sg, Ag, Bg = 112, 8, 1024
mk = lambda shift: (torch.from_numpy((rs.standard_normal((Bg, Sg)) + shift).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (Bg, Ag)).astype(np.float32)).to(dev))
(xs, xa), (es, ea) = mk(0.0), mk(0.5)
w = torch.ones(Bg, device=dev)
gm = il.GMMILDiscriminator(Sg, Ag, bench.Cfg(state_only=False))
for _ in range(30): gm.predict_reward(xs, xa, es, ea, w, w)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 4096))()
assert raw.il_debug_gmmil_timeline(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8)
n = 16 * 16 * 2
t = t[:n].astype(np.int64)
t0 = t[:, 0].min()
start, first, loop_end, part_end, ticket = (t[:, i] - t0 for i in range(5))
pct = lambda a: ' '.join(f'{int(np.percentile(a, q)):6d}' for q in (0, 10, 50, 90, 100))
print('all values in s_memtime ticks; percentiles 0 / 10 / 50 / 90 / 100 over', n, 'workgroups')
print('start offset          ', pct(start))
print('start -> first barrier', pct(first - start))
print('feature loop          ', pct(loop_end - first))
print('exp + partial sums    ', pct(part_end - loop_end))
print('ticket                ', pct(ticket - part_end))
print('end offset (ticket)   ', pct(ticket))
hw = t[:, 7]
wg = np.arange(n); itv, jtv, matv = wg % 16, (wg // 16) % 16, wg // 256
loop = loop_end - first
for name, grp in (('matrix', matv), ('xcc', (hw >> 32)), ('jt', jtv), ('it', itv)):
  print(f'feature loop by {name}:', {int(g): int(np.median(loop[grp == g])) for g in np.unique(grp)})
print('chunk 0 compute (first barrier -> end of chunk 0)', pct(t[:, 6] - t[:, 1]))
key = (hw >> 32) * 100000 + ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 0x1) + 32 * ((hw >> 13) & 0x7)   # (xcc, se, sh, cu) from HW_REG_XCC_ID / HW_REG_HW_ID
u, c = np.unique(key, return_counts=True)
print('distinct (xcc, se, sh, cu) keys:', len(u), ' workgroups per key histogram:', dict(zip(*np.unique(c, return_counts=True))))
# co-residency: for each key, do the workgroups overlap in time?
ov = 0
for k in u:
  idx = np.where(key == k)[0]
  iv = sorted((start[i], ticket[i]) for i in idx)
  ov += sum(1 for a, b in zip(iv, iv[1:]) if b[0] < a[1])
print('pairs of workgroups on the same CU that overlap in time:', ov)
last = t[:, 5] > t[:, 4]   # stale stamps of earlier launches are older than this launch's ticket
print('last-arriver tail (ticket -> rewards written), workgroups that ran it:', int(last.sum()), pct((t[:, 5] - t0 - ticket)[last]) if last.any() else '')
