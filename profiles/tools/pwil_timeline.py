"""Developer tool: phase stamps inside k_pwil_step (a -DIL_TIMELINE build): the selecting workgroups and the last arriver's merge, for the LAST of a run of steps.
  IL_HIP_LIBRARY=variants/tl/libil_hip.so python profiles/tools/pwil_timeline.py"""
import ctypes as C, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib
import inputs as gi
dev = torch.device('cuda', 0)
atoms, agent = gi.pwil_case(22, 25000, 24, 1100)
S, A, Nn = 18, 6, 25000
mem = il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn), next_states=torch.from_numpy(atoms[:, :S]),
                                                        terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn), num_trajectories=25), device=dev)
d = il.PWILDiscriminator(S, A, bench.Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, 1000)
ag = torch.from_numpy(agent).to(dev)
for k in range(300): d.compute_reward_async(ag[k:k + 1, :S], ag[k:k + 1, S:])
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
K, W, Sl = 12, 512, 8
buf = (C.c_ulonglong * (K * W * Sl))()
assert raw.il_debug_timeline_pwil(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(K, W, Sl).astype(np.float64) / 100.0
G = (Nn + 255) // 256
sel = t[0, :G]
t0 = sel[:, 0].min()
print(f'k_pwil_step, N = {Nn}, {G} workgroups; us after the first workgroup started; min / median / max over workgroups')
for name, s in (('started', 0), ('agent row scaled (z)', 1), ('distances in LDS', 2), ('ranked, candidates written', 3), ('ticket taken', 4)):
  a = sel[:, s] - t0
  print(f'  select: {name:28s} {a.min():6.2f} {np.median(a):6.2f} {a.max():6.2f}')
m = t[1]
wg = int(np.argmax(m[:, 7]))   # the last arriver of the last step
names = ('merge entered, loads issued', 'heads in LDS', 'heads sorted', 'threshold', 'survivors compacted', 'survivors ranked', 'greedy coupling done', 'merge done')
for n, s in zip(names, range(8)):
  print(f'  merge (workgroup {wg}): {n:30s} {m[wg, s] - t0:6.2f}')
g = t[2]
for n, sl in zip(('greedy: survivors in registers', 'greedy: loop done', 'greedy: consumed atoms marked', 'greedy: reward written'), range(4)):
  print(f'  merge (workgroup {wg}): {n:30s} {g[wg, sl] - t0:6.2f}')
