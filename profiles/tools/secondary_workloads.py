"""Developer tool: runs the non-headline kernels for a rocprofv3 --kernel-trace pass:
  gmmil       GMMIL.predict_reward at B = 1024, Ant dims (k_gmmil_pack / k_gmmil_tile), 300 calls
  pwil        PWIL compute_reward against 25,000 atoms, D = 24, T = 1000 (k_pwil_select / k_pwil_merge), 1,100 steps incl. a reset
  population  BatchedPopulationPlan, L learners (the *_pop launches), 40 eager updates
Usage: rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x -o x -- python profiles/tools/secondary_workloads.py gmmil|pwil|population [L]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
import imitation_learning_amd as il
dev = torch.device('cuda', 0)
what = sys.argv[1]
if what == 'gmmil':
  rs = np.random.RandomState(5)
  Sg, Ag, Bg = 112, 8, 1024
  mk = lambda shift: (torch.from_numpy((rs.standard_normal((Bg, Sg)) + shift).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (Bg, Ag)).astype(np.float32)).to(dev))
  (xs, xa), (es, ea) = mk(0.0), mk(0.5)
  w = torch.ones(Bg, device=dev)
  gm = il.GMMILDiscriminator(Sg, Ag, bench.Cfg(state_only=False))
  for _ in range(300): gm.predict_reward(xs, xa, es, ea, w, w)
elif what == 'gmmil_rate':   # calls/s + the kernel's duration from its launch stamps (il_kernel_stamps): IL_GMMIL_RESIDENT / IL_GMMIL_DIRECT select the form
  import time, hashlib
  from imitation_learning_amd import _lib
  rs = np.random.RandomState(5)
  Sg, Ag, Bg = 112, 8, 1024
  mk = lambda shift: (torch.from_numpy((rs.standard_normal((Bg, Sg)) + shift).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (Bg, Ag)).astype(np.float32)).to(dev))
  (xs, xa), (es, ea) = mk(0.0), mk(0.5)
  w = torch.ones(Bg, device=dev)
  gm = il.GMMILDiscriminator(Sg, Ag, bench.Cfg(state_only=False))
  for _ in range(50): r = gm.predict_reward(xs, xa, es, ea, w, w)
  torch.cuda.synchronize()
  durs, rates = [], []
  for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(1000): r = gm.predict_reward(xs, xa, es, ea, w, w)
    torch.cuda.synchronize()
    rates.append(1000 / (time.perf_counter() - t0))
    st = _lib.kernel_stamps().get('k_gmmil_direct')
    if st: durs.append(st['duration_us'])
  print(f"gmmil B=1024 Ant: {np.median(rates):.0f} calls/s, kernel (device stamps) {np.median(durs) if durs else float('nan'):.2f} us, digest {hashlib.sha256(r.cpu().numpy().tobytes()).hexdigest()[:12]}")
elif what == 'pwil':
  import inputs as gi
  atoms, agent = gi.pwil_case(22, 25000, 24, 1100)
  S, A, Nn = 18, 6, 25000
  mem = il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn), next_states=torch.from_numpy(atoms[:, :S]),
                                                          terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn), num_trajectories=25), device=dev)
  d = il.PWILDiscriminator(S, A, bench.Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, 1000)
  ag = torch.from_numpy(agent).to(dev)
  for k in range(1100):
    d.compute_reward_async(ag[k:k + 1, :S], ag[k:k + 1, S:])
    if k % 1000 == 999: d.reset()
else:
  L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
  pop = il.BatchedPopulationPlan([bench.build(dev, 0, seed=l, learner_id=l)[0] for l in range(L)])
  for _ in range(40): pop.run()
torch.cuda.synchronize()
