"""Developer tool: a few hundred captured updates of a general-shape SAC UpdatePlan (depth 3 / tanh / 256 at HalfCheetah dims, batch 256) for `rocprofv3 --kernel-trace`."""
import sys
sys.path[:0] = ['.', 'tests', 'tests/golden']
import numpy as np, torch
import imitation_learning_amd as il
import inputs as gi
from gpu_util import Cfg

dev = torch.device('cuda', 0)
S, A, B = 18, 6, 256
hidden, depth, act = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 3, sys.argv[3] if len(sys.argv) > 3 else 'tanh'
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
plan = il.UpdatePlan('SAC', actor, critic, log_alpha, target, mem, ao, co, to, B, 0.97, -3.0, 0.99, learner_id=9200)
plan.run(); plan.capture(warmup=2)
for _ in range(300): plan.replay()
torch.cuda.synchronize()
