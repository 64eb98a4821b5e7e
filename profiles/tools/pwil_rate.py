"""PWIL per-step reward rate at the timed size (N = 25,000 atoms, D = 24, T = 1000): steps/s and the HIP-event averages of k_pwil_select / k_pwil_merge.
  python profiles/tools/pwil_rate.py            (IL_PWIL_SERIAL_MERGE=1: the round-2 one-wave merge; IL_HIP_LIBRARY: another build)"""
import ctypes as C, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib
dev = torch.device('cuda', 0)
import inputs as gi
atoms, agent = gi.pwil_case(22, 25000, 24, 1100)
S, A, Nn = 18, 6, 25000
mem = il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn), next_states=torch.from_numpy(atoms[:, :S]),
                                                        terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn), num_trajectories=25), device=dev)
d = il.PWILDiscriminator(S, A, bench.Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, 1000)
ag = torch.from_numpy(agent).to(dev)
def episode(n=1000):
  for k in range(n):
    d.compute_reward_async(ag[k:k + 1, :S], ag[k:k + 1, S:])
  d.reset()
episode(200); torch.cuda.synchronize()
t0 = time.perf_counter(); episode(1000); episode(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
L = _lib.lib(); L.il_trace_enable(1); episode(300); buf = C.create_string_buffer(1 << 14); L.il_trace_report(buf, len(buf)); L.il_trace_enable(0)
k = {l.split()[0]: round(float(l.split()[2]) / int(l.split()[1]) * 1e3, 2) for l in buf.value.decode().strip().splitlines()}
print('pwil', round(2000 / dt, 1), 'steps/s', round(dt / 2000 * 1e6, 2), 'us/step', k, flush=True)
