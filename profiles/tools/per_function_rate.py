"""Updates/s of the PER-FUNCTION update block (what train.py runs for the algorithms without a captured plan): two memory.sample calls, the
algorithm's reward pass, sac_update - launched one by one from Python."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
import imitation_learning_amd as il
dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
actor, critic, target, log_alpha, disc = nets
mem, emem, keep = plan.memory, plan.expert_memory, plan._keep
B, S, A = bench.B, bench.S, bench.A
red = il.REDDiscriminator(S, A, bench.Cfg(state_only=False, reward_bandwidth_scale=1.0, discriminator=bench.Cfg(hidden_size=32, depth=1, activation='relu', input_dropout=0, dropout=0)))
gm = il.GMMILDiscriminator(S, A, bench.Cfg(state_only=False))
def step(kind):
  t, e = mem.sample(B), emem.sample(B)
  if kind == 'RED': t['rewards'].copy_(red.predict_reward(t['states'], t['actions']))
  elif kind == 'GMMIL': t['rewards'] = gm.predict_reward(t['states'], t['actions'], e['states'], e['actions'], t['weights'].contiguous(), e['weights'].contiguous())
  il.sac_update(actor, critic, log_alpha, target, t, keep[4], keep[5], keep[6], 0.97, -0.5 * A, 0.99)
for kind in ('SAC', 'RED', 'GMMIL'):
  for _ in range(30): step(kind)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(500): step(kind)
  torch.cuda.synchronize()
  print(kind, round(500 / (time.perf_counter() - t0), 1), 'updates/s (per-function)')
