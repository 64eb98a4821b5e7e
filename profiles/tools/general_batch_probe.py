import sys, time
sys.path[:0] = ['.', 'tests', 'tests/golden']
import numpy as np, torch
import imitation_learning_amd as il
import inputs as gi, bench
from gpu_util import Cfg, tbatch
from imitation_learning_amd.memory import batch_views
dev = torch.device('cuda', 0)
S, A, B = 18, 6, 256
plan, nets, _ = bench.build(dev, 0)
def mk():
  cfg = Cfg(hidden_size=256, depth=3, activation='tanh')
  actor, critic = il.SoftActor(S, A, cfg, device=dev), il.TwinCritic(S, A, cfg, device=dev)
  target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=dev)
  ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
  return actor, critic, log_alpha, target, ao, co, to
def rate(b, reps=300, tag=''):
  a, c, la, t, ao, co, to = mk()
  for _ in range(30): il.sac_update(a, c, la, t, b, ao, co, to, 0.97, -3.0, 0.99)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): il.sac_update(a, c, la, t, b, ao, co, to, 0.97, -3.0, 0.99)
  torch.cuda.synchronize()
  r = reps / (time.perf_counter() - t0)
  # host-only cost: time the calls without the final sync
  t0 = time.perf_counter()
  for _ in range(reps): il.sac_update(a, c, la, t, b, ao, co, to, 0.97, -3.0, 0.99)
  h = (time.perf_counter() - t0) / reps * 1e6
  torch.cuda.synchronize()
  print(f'{tag}: {r:.1f} updates/s; host per call {h:.1f} us', flush=True)
rate(tbatch(gi.transitions(np.random.RandomState(0), B, S, A, weighted=True)), tag='tbatch (separate contiguous tensors)')
rate(batch_views(plan.memory.ring[:B].clone(), S, A, True), tag='batch_views of packed rows (bench.py)')
rate(tbatch(gi.transitions(np.random.RandomState(0), B, S, A, weighted=True)), tag='tbatch again')
# the same call inside bench.secondary()'s sequence, then once more afterwards
out = bench.secondary(dev, plan, nets)
print('bench.secondary:', out.get('sac_general_shape_depth3_tanh_updates_per_s'), out.get('sac_general_shape_depth3_tanh_captured_plan_updates_per_s'), flush=True)
rate(batch_views(plan.memory.ring[:B].clone(), S, A, True), tag='after bench.secondary, batch_views')
import os
print('IL_GENERAL_TILES', os.environ.get('IL_GENERAL_TILES'))
