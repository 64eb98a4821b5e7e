#!/usr/bin/env python3
"""The reference's own CPU path, timed (SURVEY.md §8d "CPU baseline"): the UNMODIFIED code of /root/reference
(`memory.py`, `models.py`, `training.py`; imported with the 12-line omegaconf stub of tests/golden/make_golden.py) executing the update block of
`train.py:173-203` for algorithm=GAIL at the BASELINE configuration - batch 256, HalfCheetah dims, ring 1e6 filled with 1e5 synthetic rows, 25,000 expert
rows, the same synthetic buffers bench.py uploads - on torch CPU fp32.  Measured with 1 thread and with all host cores, with and without the two
`memory.sample` calls, and written to profiles/cpu_reference.json together with the host it ran on.

/root/reference exists only in the build container (it cannot travel to the GPU box), so this is where the reference is timed; bench.py reports this file
as `cpu_baseline` (kind "reference", where "build container") next to the oracle port it times live on the GPU box's host cores (`cpu_port`).

  python profiles/tools/cpu_reference.py [--updates 200] [--warmup 20]
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as mg  # noqa: E402  (sets up the stub and imports the reference modules: mg.ref_memory / ref_models / ref_training)
import inputs as gi  # noqa: E402

S, A, B = 18, 6, 256


def build():
  """Reference objects as train.py:60-95 builds them for algorithm=GAIL env=halfcheetah (conf/algorithm/GAIL.yaml), on the synthetic buffers of bench.py."""
  DC = mg.DictConfig
  torch.manual_seed(0); np.random.seed(0)
  net = DC(hidden_size=256, depth=2, activation='relu')
  actor, critic = mg.ref_models.SoftActor(S, A, net), mg.ref_models.TwinCritic(S, A, net)
  target = mg.ref_models.create_target_network(critic)
  log_alpha = torch.zeros(1, requires_grad=True)
  icfg = DC(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=1.0, entropy_bonus=0.0, mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=float('inf'),
            discriminator=DC(hidden_size=64, depth=1, activation='relu', input_dropout=0.5, dropout=0.75, reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
  disc = mg.ref_models.GAILDiscriminator(S, A, icfg, 0.97)
  opts = (torch.optim.AdamW(actor.parameters(), lr=3e-4, weight_decay=0), torch.optim.AdamW(critic.parameters(), lr=3e-4, weight_decay=0), torch.optim.Adam([log_alpha], lr=3e-4),
          torch.optim.AdamW(disc.parameters(), lr=3e-5, weight_decay=10))
  tr = gi.transitions(np.random.RandomState(1000), 100_000, S, A, absorbing_frac=0.01, terminal_frac=0.001)
  memory = mg.ref_memory.ReplayMemory(1_000_000, S, A, True)
  n = 100_000
  for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
    getattr(memory, k)[:n] = torch.from_numpy(tr[k])
  memory.step[:n] = torch.arange(1, n + 1, dtype=torch.float32)
  memory.idx = n
  et = gi.transitions(np.random.RandomState(77), 25_000, S, A, state_shift=0.5, absorbing_frac=0.01, terminal_frac=0.001)
  expert = mg.ref_memory.ReplayMemory(25_000, S, A, True, transitions={**{k: torch.from_numpy(v) for k, v in et.items() if k != 'absorbing'}, 'num_trajectories': 25})
  return actor, critic, target, log_alpha, disc, icfg, opts, memory, expert


def update(objs, batches=None):
  """train.py:173-203 for algorithm=GAIL, verbatim in structure."""
  actor, critic, target, log_alpha, disc, icfg, (ao, co, to, do), memory, expert = objs
  transitions, expert_transitions = batches if batches is not None else (memory.sample(B), expert.sample(B))
  disc.train()
  mg.ref_training.adversarial_imitation_update(actor, disc, transitions, expert_transitions, do, icfg)
  disc.eval()
  with torch.inference_mode():
    rewards = disc.predict_reward(**mg.ref_models.make_gail_input(transitions['states'], transitions['actions'], transitions['next_states'], transitions['terminals'], actor, False, False))
  transitions = dict(transitions, rewards=rewards.clone())
  mg.ref_training.sac_update(actor, critic, log_alpha, target, transitions, ao, co, to, 0.97, -0.5 * A, 0.99)


def timed(objs, updates, warmup, with_sampling):
  fixed = None if with_sampling else (objs[7].sample(B), objs[8].sample(B))
  for _ in range(warmup):
    update(objs, fixed)
  t0 = time.perf_counter()
  for _ in range(updates):
    update(objs, fixed)
  return updates / (time.perf_counter() - t0)


def cpu_model():
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        return line.split(':', 1)[1].strip()
  except OSError:
    pass
  return platform.processor()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--updates', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  args = ap.parse_args()
  nproc = os.cpu_count()
  out = dict(what='reference training.py / models.py / memory.py (unmodified, imported from /root/reference) executing train.py:173-203, algorithm=GAIL, batch 256, HalfCheetah dims, '
                  'ring 1e6 / fill 1e5, 25,000 expert rows, torch CPU fp32', unit='updates/s', where='build container (the reference cannot travel to the GPU box)',
             nproc=nproc, cpu_model=cpu_model(), torch=torch.__version__, updates_timed=args.updates, warmup=args.warmup, results={})
  for threads in (1, nproc):
    torch.set_num_threads(threads)
    for with_sampling in (True, False):
      objs = build()
      rate = timed(objs, args.updates, args.warmup, with_sampling)
      out['results'][f'threads_{threads}_{"with" if with_sampling else "without"}_memory_sample'] = round(rate, 2)
      print(f'threads={threads} sampling={with_sampling}: {rate:.2f} updates/s', flush=True)
  path = os.path.join(ROOT, 'profiles', 'cpu_reference.json')
  json.dump(out, open(path, 'w'), indent=1)
  print('wrote', path)


if __name__ == '__main__':
  main()
