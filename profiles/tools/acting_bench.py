"""Env-steps/s of the acting path on the synthetic HalfCheetah stand-in: per-function calls vs il_act_step
(exact / overlap schedules), without updates and with one captured GAIL update per env step.  Usage: python profiles/tools/acting_bench.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
import imitation_learning_amd as il  # noqa: E402
from imitation_learning_amd.environments import make_env  # noqa: E402


def loop(schedule, plan, actor, memory, env, steps, update, early_act=True, direct=True, thread=False):
  worker = il.ActingWorker(actor, memory, mirror=schedule == 'overlap') if schedule != 'per_function' else None
  step_update = plan.replay
  if schedule == 'overlap' and update:   # the update with this worker's append before and its snapshot after: recorded as direct launches (round 6), or re-captured as graphs
    plan.graph = plan.graph_side = None; plan.pre_hooks.clear(); plan.post_hooks.clear()
    worker.attach(plan)
    if direct and plan.direct_launch_ok():
      plan.record_direct(); step_update = plan.launch_async if thread else plan.launch_direct
    else:
      plan.capture(warmup=0)
  state, t = env.reset(), 0
  action = worker.act(state) if schedule in ('fused', 'overlap') else None
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for step in range(1, steps + 1):
    if worker is None:
      action = actor(state).sample()
      nxt, r, term = env.step(action); t += 1
      memory.append(step, state, action, r, nxt, term and t != env.max_episode_steps, t == env.max_episode_steps)
      if term and t != env.max_episode_steps: memory.wrap_for_absorbing_states()
      state = env.reset() if term else nxt
    elif schedule == 'exact':
      action = worker.act(state)
      nxt, r, term = env.step(action); t += 1
      worker.append(step, nxt, r, term and t != env.max_episode_steps, t == env.max_episode_steps)
      state = env.reset() if term else nxt
    elif schedule == 'fused':
      nxt, r, term = env.step(action); t += 1
      action = worker.step(step, nxt, r, term and t != env.max_episode_steps, t == env.max_episode_steps, obs=env.reset() if term else None)
    else:
      nxt, r, term = env.step(action); t += 1
      worker.post(step, state, action, nxt, r, term and t != env.max_episode_steps, t == env.max_episode_steps)
      state = env.reset() if term else nxt
      if not update: worker.enqueue_append()
    if term: t = 0
    if schedule == 'overlap' and early_act: worker.act_begin(state)   # (round 6) the act launch ahead of the update's host work: its turn-around hides behind the replay
    if update: step_update()
    if schedule == 'overlap': action = worker.act_end() if early_act else worker.act(state)
  plan.launcher_wait()
  torch.cuda.synchronize()
  return steps / (time.perf_counter() - t0)


def main():
  dev = torch.device('cuda', 0)
  plan, nets, _ = bench.build(dev, 0)
  actor, memory = nets[0], plan.memory
  env = make_env('halfcheetah', True)
  env.seed(0)
  for _ in range(3): plan.run()
  plan.capture(warmup=0)
  # host-only cost of the environment itself, for reference
  s, t0 = env.reset(), time.perf_counter()
  a = torch.zeros(1, 6)
  for _ in range(2000): env.step(a)
  env_only = 2000 / (time.perf_counter() - t0)
  out = dict(env_only_steps_per_s=round(env_only, 1))
  for update in (False, True):
    for schedule in ('per_function', 'exact', 'fused', 'overlap'):
      loop(schedule, plan, actor, memory, env, 200, update)
      out[f'{schedule}{"+update" if update else ""}'] = round(loop(schedule, plan, actor, memory, env, 3000, update), 1)
  loop('overlap', plan, actor, memory, env, 200, True, thread=True)
  out['overlap+update (launcher thread: UpdatePlan.launch_async)'] = round(loop('overlap', plan, actor, memory, env, 3000, True, thread=True), 1)
  loop('overlap', plan, actor, memory, env, 200, True, direct=False)
  out['overlap+update (graph replays instead of direct launches)'] = round(loop('overlap', plan, actor, memory, env, 3000, True, direct=False), 1)
  loop('overlap', plan, actor, memory, env, 200, True, early_act=False, direct=False)
  out['overlap+update (graph replays, act launched behind the replay: round 5)'] = round(loop('overlap', plan, actor, memory, env, 3000, True, early_act=False, direct=False), 1)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
