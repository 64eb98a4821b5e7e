import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import imitation_learning_amd as il
from imitation_learning_amd import training as il_training
from imitation_learning_amd.parallel import DataParallelUpdate
from test_gpu_parity import _make_plan
from gpu_util import N
names = ['actor', 'critic', 'target', 'log_alpha', 'disc']
for steps in (1, 2):
  outs = []
  for dp in (False, True):
    il.seed(21); il_training._NOISE.clear()
    plan, nets = _make_plan('GAIL', 13)
    runner = DataParallelUpdate(plan) if dp else plan
    for _ in range(steps): runner.run()
    torch.cuda.synchronize()
    ao, co = plan._keep[4], plan._keep[5]
    outs.append([N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [N(plan.logp), N(plan.q), N(plan.rewards), N(ao.exp_avg), N(co.exp_avg), N(ao.exp_avg_sq)])
  for nm, a, b in zip(names + ['logp', 'q', 'rewards', 'actor_m', 'critic_m', 'actor_v'], *outs):
    d = np.abs(a - b)
    print(steps, nm, 'ndiff', int((d > 0).sum()), 'max', float(d.max()))
