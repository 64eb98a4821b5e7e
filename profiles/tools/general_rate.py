"""Developer tool: sac_update through the general-shape kernels (csrc/general.hip) - updates/s of the per-function call at HalfCheetah dims, batch 256:
the fused shape (depth 2, ReLU, 256) forced through the general path next to the fused per-function path, then depth 3 / tanh / 256 and depth 2 / ReLU / 512.
  python profiles/tools/general_rate.py"""
import sys, time
sys.path[:0] = ['.', 'tests', 'tests/golden']
import numpy as np, torch
import imitation_learning_amd as il
import inputs as gi
from gpu_util import Cfg, tbatch

dev = torch.device('cuda', 0)
S, A, B = 18, 6, 256


def rate(hidden, depth, act, force_general, reps=300):
  cfg = Cfg(hidden_size=hidden, depth=depth, activation=act)
  actor, critic = il.SoftActor(S, A, cfg, device=dev), il.TwinCritic(S, A, cfg, device=dev)
  if force_general: actor.general = critic.general = True
  target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=dev)
  ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
  b = tbatch(gi.transitions(np.random.RandomState(0), B, S, A, weighted=True))
  for _ in range(20): il.sac_update(actor, critic, log_alpha, target, b, ao, co, to, 0.97, -3.0, 0.99)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): il.sac_update(actor, critic, log_alpha, target, b, ao, co, to, 0.97, -3.0, 0.99)
  torch.cuda.synchronize()
  return reps / (time.perf_counter() - t0)


print(f'fused per-function path, depth 2 relu 256:          {rate(256, 2, "relu", False):9.1f} updates/s')
print(f'general kernels, the same shape (forced):            {rate(256, 2, "relu", True):9.1f} updates/s')
print(f'general kernels, depth 3 tanh 256:                   {rate(256, 3, "tanh", False):9.1f} updates/s')
print(f'general kernels, depth 2 relu 512:                   {rate(512, 2, "relu", False):9.1f} updates/s')
print(f'general kernels, depth 1 sigmoid 1024:               {rate(1024, 1, "sigmoid", False):9.1f} updates/s')
