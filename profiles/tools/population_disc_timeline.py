"""Developer tool: phase timing inside the workgroups of the population path's DISCRIMINATOR launches (il_gail_step_population) from a -DIL_TIMELINE build
(bash profiles/tools/build_variants.sh tl:-DIL_TIMELINE): k_gail_grad - the 48 workgroups of learner 0 (blockIdx.z == 0) - and k_gail_reward - the first 512 workgroups.
  IL_HIP_LIBRARY=$PWD/variants/tl/libil_hip.so python profiles/tools/population_disc_timeline.py [learners]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib

K, W, S = 12, 512, 8
dev = torch.device('cuda', 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pop = il.BatchedPopulationPlan([bench.build(dev, 0, seed=l, learner_id=l)[0] for l in range(L)])
for _ in range(6): pop.run()
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (K * W * S))()
assert raw.il_debug_timeline_gail(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S).astype(np.float64) / 100.0   # us
med = lambda a: float(np.median(a))


def phases(kid, names, slots, rows=None):
  a = t[kid]
  ok = np.all(a[:, list(slots)] > 0, axis=1)
  if rows is not None: ok &= rows
  a = a[ok]
  d = [a[:, s1] - a[:, s0] for s0, s1 in zip(slots[:-1], slots[1:])]
  return (' | '.join(f'{n} {med(x):.2f}' for n, x in zip(names, d)) + f' || workgroup total median {med(a[:, slots[-1]] - a[:, slots[0]]):.2f} max {(a[:, slots[-1]] - a[:, slots[0]]).max():.2f}, '
          f'first start -> last end of the {len(a)} sampled workgroups {a[:, slots[-1]].max() - a[:, slots[0]].min():.2f}, start spread {a[:, slots[0]].max() - a[:, slots[0]].min():.2f}')


print(f'{L} learners, one stream, eager launches; us')
names = ('weights -> LDS', 'Gram', 'power iterations || rows', 'forward + loss / penalty products', 'gradient slab', 'exit')
for c, call in enumerate(('policy', 'expert', 'penalty mix')):
  sel = np.zeros(W, bool); sel[16 * c:16 * (c + 1)] = True
  print(f'k_gail_grad learner 0, {call:11s}:', phases(0, names, (0, 1, 3, 4, 5, 6, 7), sel))
print('k_gail_reward              :', phases(2, ('rows', 'parameters -> LDS, sigma, forward, head'), (0, 1, 7)))
print('k_gail_reduce              :', phases(1, ('whole',), (0, 7)))


def concurrency(kid, name):
  a = t[kid]; a = a[(a[:, 0] > 0) & (a[:, 7] > 0)]
  s, e = a[:, 0] - a[:, 0].min(), a[:, 7] - a[:, 0].min()
  grid = np.arange(0, e.max(), 0.5)
  live = [(int(((s <= x) & (e > x)).sum())) for x in grid]
  print(f'{name}: workgroups in flight every 0.5 us from the first start: {live}')
  print(f'{name}: start percentiles (us) 10/25/50/75/90/100: {[round(float(np.percentile(s, p)), 2) for p in (10, 25, 50, 75, 90, 100)]}')


concurrency(2, 'k_gail_reward')
concurrency(1, 'k_gail_reduce')


def by_id(kid, name):
  a = t[kid]; ok = (a[:, 0] > 0) & (a[:, 7] > 0)
  s = a[:, 0] - a[ok, 0].min(); dur = a[:, 7] - a[:, 0]
  print(f'{name}: start (us) / duration of workgroup ids 0, 8, 16, ... (XCD 0), then ids 1, 9, ... (XCD 1):')
  for x in (0, 1):
    ids = [i for i in range(x, W, 8) if ok[i]]
    print('  ' + ' '.join(f'{s[i]:.1f}/{dur[i]:.1f}' for i in ids))


by_id(2, 'k_gail_reward')
