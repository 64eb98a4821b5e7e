"""Developer tool (round 6): BASELINE.json configs[3] as whole updates - algorithm=GMMIL env=ant, batch 1024 - replayed under rocprofv3 --kernel-trace (which kernels the 140 us go to).
  rocprofv3 --kernel-trace -d /tmp/gp -o gp -- python profiles/tools/gmmil_plan_trace.py; python profiles/summarize_rocpd.py /tmp/gp/*/*.db"""
import sys
sys.path[:0] = ['.', 'tests', 'tests/golden']
import numpy as np, torch, time
import imitation_learning_amd as il
import inputs as gi, bench
from bench import Cfg
dev = torch.device('cuda', 0)
Sg, Ag, Bg, H = 112, 8, 1024, 256
rs2 = np.random.RandomState(6)
cfgn = Cfg(hidden_size=H, depth=2, activation='relu')
a2, c2 = il.SoftActor(Sg, Ag, cfgn, device=dev), il.TwinCritic(Sg, Ag, cfgn, device=dev)
t2, la2 = il.create_target_network(c2), torch.zeros(1, device=dev)
o2 = (il.AdamW(a2, lr=3e-4, weight_decay=0), il.AdamW(c2, lr=3e-4, weight_decay=0), il.Adam(la2, lr=3e-4))
def ring(n, cap, shift):
  tr = gi.transitions(rs2, n, Sg, Ag, state_shift=shift, absorbing_frac=0.01, terminal_frac=0.001)
  m = il.ReplayMemory(cap, Sg, Ag, True, device=dev)
  for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
    getattr(m, k)[:n] = torch.from_numpy(tr[k]).to(dev)
  m.step[:n] = torch.arange(1, n + 1, dtype=torch.float32, device=dev)
  m.idx, m.full = n % cap, n == cap
  m._sync_ring_state()
  return m
gplan = il.UpdatePlan('GMMIL', a2, c2, la2, t2, ring(100_000, 1_000_000, 0.0), *o2, Bg, 0.99, -1.0 * Ag, 0.995, expert_memory=ring(25_000, 25_000, 0.5),
                      discriminator=il.GMMILDiscriminator(Sg, Ag, Cfg(state_only=False)), learner_id=9002)
gplan.run(); gplan.capture(warmup=0)
for _ in range(30): gplan.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): gplan.replay()
torch.cuda.synchronize()
print(f'GMMIL ant B=1024: {300 / (time.perf_counter() - t0):.1f} updates/s')
