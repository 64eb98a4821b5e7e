"""Developer tool (round 6): workgroup-level timeline of the LAST update of a run of back-to-back overlapped updates (UpdatePlan.launch_direct(join=False): the SAC branch's
four launches alternating over two streams, il_sac_update_gather_overlap), from the always-on launch stamps (il_kernel_stamp_rows: begin, end, placement, gate).

  python profiles/tools/overlap_timeline.py [updates]          (IL_MAIN_OVERLAP=0: the in-order schedule, for comparison)

Times in microseconds relative to the first workgroup of the update's forward / critic-loss launch; min / median / max over the workgroups of a group. `gate` = the moment a
workgroup's wait for the other stream's launch was satisfied (0 rows: the group has no such wait)."""
import ctypes as C
import functools
import sys
import time
sys.path.insert(0, '.')
import numpy as np, torch, bench
from imitation_learning_amd import _lib

dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
for _ in range(5): plan.run()
torch.cuda.synchronize()
plan.record_direct()
step = functools.partial(plan.launch_direct, join=False) if plan._direct_overlap else plan.launch_direct
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
period = (time.perf_counter() - t0) / n * 1e6
assert plan.sync_timeouts() == 0 and not plan.poisoned()
L = _lib.lib()
W = int(L.il_kernel_stamp_workgroups())


def rows(kernel):
  buf = (C.c_uint64 * (4 * W))()
  _lib.check(L.il_kernel_stamp_rows(_lib.STAMP_KERNELS.index(kernel), buf))
  a = np.frombuffer(buf, dtype=np.uint64).reshape(W, 4).astype(np.float64)
  live = (a[:, 0] > 0) & (a[:, 1] > 0)
  return a, live


nt = plan.B // 16
ch, chl = rows('k_sac_chain_pair')
t_ref = ch[chl, 0].min()
us = lambda x: (x - t_ref) / 100.0


def line(name, a, idx):
  idx = [i for i in idx if i < W and a[i, 0] > 0 and a[i, 1] > 0]
  if not idx:
    print(f'  {name:34s} (no workgroups)'); return
  b, e, g = us(a[idx, 0]), us(a[idx, 1]), a[idx, 3]
  gs = us(g[g > 0]) if (g > 0).any() else None
  f = lambda v: f'{v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}'
  print(f'  {name:34s} n={len(idx):3d} | begin {f(b)} | gate {f(gs) if gs is not None else "      -       -       -"} | end {f(e)}')


print(f'overlapped={bool(plan._direct_overlap)}  period {period:.2f} us = {1e6 / period:.0f} updates/s; times relative to the first workgroup of k_sac_chain_pair (min / median / max)')
nwg = int(chl.sum())
relabel = 1 if nwg >= 10 * nt else 0
o = 0
seg = {}
for name, cnt in (("chain actor(s') half 1", nt), ("chain actor(s') half 0", nt), ('chain targets half 1', 2 * nt), ('chain targets half 0', 2 * nt)) + ((('chain relabel', nt),) if relabel else ()) + (('chain critics', 2 * nt), ('chain actor(s)', nt)):
  seg[name] = range(o, o + cnt); o += cnt
seg['chain row copies'] = range(o, nwg)
for k, v in seg.items(): line(k, ch, v)
dc, dcl = rows('k_dw_adam_critic')
line('dW critic blocks', dc, range(int(dcl.sum())))
pc, pcl = rows('k_policy_critic_pair')
line('policy-critic critics half 1', pc, range(0, 2 * nt)); line('policy-critic critics half 0', pc, range(2 * nt, 4 * nt)); line('policy-critic helpers', pc, range(4 * nt, int(pcl.sum())))
da, dal = rows('k_dw_adam_actor')
nda = int(dal.sum())
line('dW actor blocks', da, range(0, nda - 69)); line('dW actor tail block 0', da, [nda - 69]); line('dW actor tail (target step)', da, range(nda - 68, nda))
gg, ggl = rows('k_gail_grad'); line('k_gail_grad', gg, range(int(ggl.sum())))
gr, grl = rows('k_gail_reduce'); line('k_gail_reduce', gr, range(int(grl.sum())))
ends = {k: us(a[l, 1].max()) for k, (a, l) in dict(chain=(ch, chl), dwc=(dc, dcl), pc=(pc, pcl), dwa=(da, dal)).items()}
print('last workgroup end:', {k: round(float(v), 2) for k, v in ends.items()}, '| next chain would begin its period at', round(period, 2))
cu = lambda a, l: len(set(a[l, 2].astype(np.int64).tolist()))
print('distinct CUs:', dict(chain=cu(ch, chl), dwc=cu(dc, dcl), pc=cu(pc, pcl), dwa=cu(da, dal), gail_grad=cu(gg, ggl)))
