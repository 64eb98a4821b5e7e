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


def plan_rate(hidden, depth, act, reps=300):
  """WHOLE updates (device index draws + gather + sac_update) of an UpdatePlan on this shape, captured as one hipGraph; then the per-kernel HIP-event times of eager launches."""
  import ctypes as C
  from imitation_learning_amd import _lib
  cfg = Cfg(hidden_size=hidden, depth=depth, activation=act)
  actor, critic = il.SoftActor(S, A, cfg, device=dev), il.TwinCritic(S, A, cfg, device=dev)
  target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=dev)
  ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
  mem = il.ReplayMemory(100_000, S, A, True, device=dev)
  tr = gi.transitions(np.random.RandomState(1), 50_000, S, A, weighted=True)
  for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'weights'):
    getattr(mem, k)[:50_000] = torch.from_numpy(np.asarray(tr[k], np.float32)).to(dev).reshape(getattr(mem, k)[:50_000].shape)
  mem.idx, mem.full = 50_000, False
  mem._sync_ring_state()
  plan = il.UpdatePlan('SAC', actor, critic, log_alpha, target, mem, ao, co, to, B, 0.97, -3.0, 0.99, learner_id=9100 + depth)
  plan.run(); plan.capture(warmup=2)
  for _ in range(20): plan.replay()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): plan.replay()
  torch.cuda.synchronize()
  r = reps / (time.perf_counter() - t0)
  L = _lib.lib()
  b = tbatch(gi.transitions(np.random.RandomState(0), B, S, A, weighted=True))
  L.il_trace_enable(1)
  for _ in range(20): il.sac_update(actor, critic, log_alpha, target, b, ao, co, to, 0.97, -3.0, 0.99)
  buf = C.create_string_buffer(1 << 16)
  _lib.check(L.il_trace_report(buf, len(buf)))
  L.il_trace_enable(0)
  rows = [(n, int(c) / 20, float(t) / int(c) * 1e3) for n, c, t in (ln.split() for ln in buf.value.decode().strip().splitlines())]
  return r, rows


for shape in ((256, 3, 'tanh'), (256, 2, 'relu')):
  if shape[1] == 2 and shape[2] == 'relu': continue   # (the fused kernels take that shape)
  r, rows = plan_rate(*shape)
  print(f'captured UpdatePlan, depth {shape[1]} {shape[2]} {shape[0]}:          {r:9.1f} updates/s')
  tot = 0.0
  for n, per, us in rows:
    print(f'    {n:16s} {per:5.1f} launches / update x {us:7.2f} us (HIP events, eager)'); tot += per * us
  print(f'    sum of kernel times {tot:.1f} us per update')
