#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE, not product code: the reference's OWN update block timed on this host's cores (bench.py's `cpu_baseline`, kind "reference").

Loads the reference's byte-compiled hot-path modules from oracle/_ref/ (built by oracle/build_ref.py from /root/reference; absent -> exit code 3) and executes
train.py:173-203 for algorithm=GAIL at the BASELINE configuration - batch 256, HalfCheetah dims, ring capacity 1e6 filled with 1e5 synthetic rows, 25,000 expert rows,
the same synthetic buffers bench.py uploads (tests/golden/inputs.py) - on torch CPU fp32, with 1 thread and with all cores, with and without the two `memory.sample`
calls, inside a wall-clock budget. Run as a subprocess (bench.py does): CPU only, its torch thread settings stay its own. Prints one JSON line.

  python oracle/ref_cpu_baseline.py [--budget 20]
"""
import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, '_ref')
S, A, B = 18, 6, 256


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
  ap.add_argument('--budget', type=float, default=20.0, help='seconds of timed CPU work in total (split over the four measurements)')
  args = ap.parse_args()
  if not all(os.path.isfile(os.path.join(REF, m + '.pyc')) for m in ('memory', 'models', 'training')):
    print(json.dumps(dict(error='oracle/_ref is absent (build it in the container: python oracle/build_ref.py)')))
    sys.exit(3)
  sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
  sys.path.insert(0, REF)
  import numpy as np
  import torch
  from omegaconf import DictConfig as DC
  import memory as ref_memory
  import models as ref_models
  import training as ref_training
  import inputs as gi

  def build():
    """Reference objects as train.py:60-95 builds them for algorithm=GAIL env=halfcheetah (conf/algorithm/GAIL.yaml), on bench.py's synthetic buffers."""
    torch.manual_seed(0); np.random.seed(0)
    net = DC(hidden_size=256, depth=2, activation='relu')
    actor, critic = ref_models.SoftActor(S, A, net), ref_models.TwinCritic(S, A, net)
    target = ref_models.create_target_network(critic)
    log_alpha = torch.zeros(1, requires_grad=True)
    icfg = DC(state_only=False, spectral_norm=True, loss_function='BCE', grad_penalty=1.0, entropy_bonus=0.0, mixup_alpha=1, pos_class_prior=0.7, nonnegative_margin=float('inf'),
              discriminator=DC(hidden_size=64, depth=1, activation='relu', input_dropout=0.5, dropout=0.75, reward_shaping=False, subtract_log_policy=False, reward_function='AIRL'))
    disc = ref_models.GAILDiscriminator(S, A, icfg, 0.97)
    opts = (torch.optim.AdamW(actor.parameters(), lr=3e-4, weight_decay=0), torch.optim.AdamW(critic.parameters(), lr=3e-4, weight_decay=0), torch.optim.Adam([log_alpha], lr=3e-4),
            torch.optim.AdamW(disc.parameters(), lr=3e-5, weight_decay=10))
    n = 100_000
    tr = gi.transitions(np.random.RandomState(1000), n, S, A, absorbing_frac=0.01, terminal_frac=0.001)
    memory = ref_memory.ReplayMemory(1_000_000, S, A, True)
    for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
      getattr(memory, k)[:n] = torch.from_numpy(tr[k])
    memory.step[:n] = torch.arange(1, n + 1, dtype=torch.float32)
    memory.idx = n
    et = gi.transitions(np.random.RandomState(77), 25_000, S, A, state_shift=0.5, absorbing_frac=0.01, terminal_frac=0.001)
    expert = ref_memory.ReplayMemory(25_000, S, A, True, transitions={**{k: torch.from_numpy(v) for k, v in et.items() if k != 'absorbing'}, 'num_trajectories': 25})
    return actor, critic, target, log_alpha, disc, icfg, opts, memory, expert

  def update(objs, batches=None):
    """train.py:173-203 for algorithm=GAIL: the calls, in the reference's order."""
    actor, critic, target, log_alpha, disc, icfg, (ao, co, to, do), memory, expert = objs
    transitions, expert_transitions = batches if batches is not None else (memory.sample(B), expert.sample(B))
    disc.train()
    ref_training.adversarial_imitation_update(actor, disc, transitions, expert_transitions, do, icfg)
    disc.eval()
    with torch.inference_mode():
      rewards = disc.predict_reward(**ref_models.make_gail_input(transitions['states'], transitions['actions'], transitions['next_states'], transitions['terminals'], actor, False, False))
    transitions = dict(transitions, rewards=rewards.clone())
    ref_training.sac_update(actor, critic, log_alpha, target, transitions, ao, co, to, 0.97, -0.5 * A, 0.99)

  objs_cache = {}

  def timed(threads, with_sampling, seconds):
    """Strictly time-boxed (a 256-thread OpenMP pool makes the reference's ~2,800 tiny aten calls per update SLOWER than one thread: the loop must not insist on a count)."""
    torch.set_num_threads(threads)
    objs = objs_cache.setdefault('objs', build())   # one set of buffers / networks for every configuration (the 1e6-row ring is the expensive part)
    fixed = None if with_sampling else (objs[7].sample(B), objs[8].sample(B))
    t_w = time.perf_counter()
    update(objs, fixed)
    first = time.perf_counter() - t_w
    if first > seconds / 2:   # one update already eats half the slot (hundreds of threads fighting over 16-row GEMMs): that single update is the measurement
      return 1.0 / first, 1
    for _ in range(2):        # warm-up: at most a quarter of the slot
      update(objs, fixed)
      if time.perf_counter() - t_w > seconds / 4: break
    t0, k = time.perf_counter(), 0
    while k < 2 or time.perf_counter() - t0 < seconds * 0.75:
      update(objs, fixed); k += 1
    return k / (time.perf_counter() - t0), k

  nproc = os.cpu_count()
  usable = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else nproc
  # The largest pool is capped at 64 threads (round 6): at 256 OpenMP threads ONE update of the reference's ~2,800 tiny aten calls took ~20 s - a single sample, 55 s of the bench's
  # wall clock - and no pool beyond a socket's physical cores has ever been the best (`value` is the best of 1 / 8 / this many). IL_CPU_BASELINE_THREADS overrides.
  many = min(usable, int(os.environ.get('IL_CPU_BASELINE_THREADS', min(usable, 64))))
  configs = [(1, 'one_thread')] + ([(8, 'eight_threads')] if many > 8 else []) + [(many, 'all_cores')]
  res, counts = {}, {}
  t_all = time.perf_counter()
  for threads, label in configs:
    for with_sampling in (True, False):
      if not with_sampling and counts.get(f'{label}_with_memory_sample') == 1:
        # the thread count is hopeless for this workload (one update ate the whole slot: 256 OpenMP threads on 16-row GEMMs take ~20 s per update): one sample of it is enough
        res[f'{label}_without_memory_sample'], counts[f'{label}_without_memory_sample'] = None, 0
        continue
      rate, k = timed(threads, with_sampling, args.budget / (2 * len(configs)))
      key = f'{label}_{"with" if with_sampling else "without"}_memory_sample'
      res[key], counts[key] = round(rate, 3), k
  print(json.dumps(dict(unit='updates/s', nproc=nproc, usable_cores=usable, threads_all_cores=many, cpu_model=cpu_model(), torch=torch.__version__, results=res, updates_timed=counts,
                        seconds=round(time.perf_counter() - t_all, 1), manifest=json.load(open(os.path.join(REF, 'MANIFEST.json'))),
                        what='the reference\'s own training.py / models.py / memory.py (byte-compiled from /root/reference, unmodified) executing train.py:173-203, algorithm=GAIL, batch 256, '
                             'HalfCheetah dims, ring 1e6 / fill 1e5, 25,000 expert rows, torch CPU fp32')))


if __name__ == '__main__':
  os.environ.setdefault('HIP_VISIBLE_DEVICES', '')
  main()
