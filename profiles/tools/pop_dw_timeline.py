"""Developer tool: the WHOLE k_dw_adam_pop launches of one population update, sampled (every IL_TL_STRIDE-th workgroup), from a -DIL_TIMELINE -DIL_TL_STRIDE=5 build:
per job class (64 x 64 block / 32 x 32 block / bias / tail) when the workgroups start, how long they run and when the last one ends - i.e. whether the launch is
bound by its rounds of block workgroups, by the small jobs or by its end.
  bash profiles/tools/build_variants.sh tl5:"-DIL_TIMELINE -DIL_TL_STRIDE=5 -w"
  IL_HIP_LIBRARY=$PWD/variants/tl5/libil_hip.so python profiles/tools/pop_dw_timeline.py [learners] [stride]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib

K, W, S = 12, 512, 8
dev = torch.device('cuda', 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pop = il.BatchedPopulationPlan([bench.build(dev, 0, seed=l, learner_id=l)[0] for l in range(L)])
for _ in range(3): pop.run()
pop.capture() if hasattr(pop, 'capture') else None
for _ in range(10): (pop.replay() if getattr(pop, 'graph', None) is not None else pop.run())
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (K * W * S))()
assert raw.il_debug_timeline_sac(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S)
H, nb64 = 256, {10: 32, 11: 16}
small, bias = {10: 32, 11: 16}, {10: 8, 11: 4}
print(f'{L} learners, every {stride}th workgroup of the launch sampled; us after the first sampled workgroup started')
for kid, name in ((10, 'critic'), (11, 'actor + tail')):
  a = t[kid]
  a = a[(a[:, 0] > 0) & (a[:, 7] > 0)]
  st, en = a[:, 0].astype(np.float64) / 100.0, a[:, 7].astype(np.float64) / 100.0
  t0 = st.min(); st -= t0; en -= t0
  bx = (a[:, 2] & 0xffffffff).astype(np.int64)
  cls = np.where(bx < nb64[kid], 0, np.where(bx < nb64[kid] + small[kid], 1, np.where(bx < nb64[kid] + small[kid] + bias[kid], 2, 3)))
  print(f'k_dw_adam_pop {name}: {len(a)} sampled workgroups (~{len(a) * stride} in the launch), first start -> last end {en.max():.2f} us')
  for c, cn in enumerate(('64 x 64 blocks', '32 x 32 block jobs (layers 1, 3 + biases)', 'bias 2 wave jobs', 'tail (Polyak, temperature, copies)')):
    m = cls == c
    if not m.any(): continue
    d = en[m] - st[m]
    extra = ''
    if c == 0:
      pr = a[m][:, 1].astype(np.float64) / 100.0 - t0 - st[m]
      extra = f' | products {np.median(pr):.2f}, epilogue {np.median(d - pr):.2f}'
    print(f'  {cn:42s} n {m.sum():4d} | start min {st[m].min():6.2f} med {np.median(st[m]):6.2f} max {st[m].max():6.2f} | duration med {np.median(d):6.2f} max {d.max():6.2f} | last end {en[m].max():6.2f}{extra}')
  edges = np.arange(0.0, en.max() + 4.0, 4.0)
  act = [(int(((st < hi) & (en > lo)).sum()) * stride) for lo, hi in zip(edges[:-1], edges[1:])]
  big = [(int(((st < hi) & (en > lo) & (cls == 0)).sum()) * stride) for lo, hi in zip(edges[:-1], edges[1:])]
  print('  workgroups alive per 4 us window (all / 64 x 64 blocks): ' + ' '.join(f'{x}/{y}' for x, y in zip(act, big)))
  ck = a[(a[:, 4] > 0) & (a[:, 5] > a[:, 4])]
  if len(ck):
    mhz = (ck[:, 5] - ck[:, 4]).astype(np.float64) / ((ck[:, 7] - ck[:, 0]).astype(np.float64) / 100.0)
    print(f'  shader clock over a workgroup\'s lifetime (s_memtime delta / 100 MHz counter delta): min {mhz.min():.0f} med {np.median(mhz):.0f} max {mhz.max():.0f} MHz (n = {len(ck)})')
  cu = a[:, 3]
  print(f'  distinct (XCD, SE/SH/CU) placements among the samples: {len(set(cu.tolist()))}')
