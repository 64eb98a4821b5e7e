"""Developer tool: phase timing inside the workgroups of the batched population launches (il_*_population) from a -DIL_TIMELINE build (see update_timeline.py for the
build recipe): the first 512 workgroups (linear id x + gridDim.x * learner) of every instrumented kernel store s_memrealtime (100 MHz) at their phase boundaries.
  IL_HIP_LIBRARY=imitation-learning_amd/csrc/build/ab/libil_hip_tl.so python profiles/tools/population_timeline.py [learners]"""
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
for _ in range(3): pop.run()
pop.capture() if hasattr(pop, 'capture') else None
for _ in range(10): (pop.replay() if getattr(pop, 'graph', None) is not None else pop.run())
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (K * W * S))()
assert raw.il_debug_timeline_sac(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S).astype(np.float64) / 100.0   # us
med = lambda a: float(np.median(a))


def phases(kid, names, slots):
  a = t[kid]
  ok = np.all(a[:, list(slots)] > 0, axis=1)   # workgroups that passed every listed boundary (k_dw_adam_pop: only the 64 x 64 block workgroups stamp slot 1)
  a = a[ok]
  d = [a[:, s1] - a[:, s0] for s0, s1 in zip(slots[:-1], slots[1:])]
  span = a[:, slots[-1]].max() - a[:, slots[0]].min()
  return ' | '.join(f'{n} {med(x):.2f}' for n, x in zip(names, d)) + f' || workgroup total {med(a[:, slots[-1]] - a[:, slots[0]]):.2f}, first start -> last end of the {len(a)} sampled workgroups {span:.2f}, start spread {a[:, slots[0]].max() - a[:, slots[0]].min():.2f}'


print(f'{L} learners, B = {pop.B}; medians over the first 512 workgroups of each launch, us')
print("k_actor_fwd  s' tiles:", phases(5, ('rows', 'layer 1', 'layer 2', 'head GEMM', 'sample + stores', 'exit'), (0, 1, 2, 3, 4, 5, 7)))
print('k_actor_fwd  s  tiles:', phases(6, ('rows', 'layer 1', 'layer 2', 'head GEMM', 'sample + stores', 'exit'), (0, 1, 2, 3, 4, 5, 7)))
print('k_critic_fwd         :', phases(8, ('rows', 'layer 1', 'layer 2', 'Q'), (0, 1, 2, 3, 7)))
print('k_critic_bwd         :', phases(9, ('whole',), (0, 7)))
print('k_dw_adam_pop critic : 64 x 64 blocks of the H x H layers:', phases(10, ('products', 'AdamW epilogue'), (0, 1, 7)))
print('                       all workgroups:', phases(10, ('whole',), (0, 7)))
print('k_policy_critic      :', phases(7, ('rows', 'layer 1', 'layer 2', 'Q + mask', 'layer 2 backward', 'dQ/da columns'), (0, 1, 2, 3, 5, 6, 7)))
print('k_dw_adam_pop actor  : 64 x 64 blocks:', phases(11, ('products', 'AdamW epilogue'), (0, 1, 7)))
print('                       all workgroups:', phases(11, ('whole',), (0, 7)))
