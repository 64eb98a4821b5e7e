"""Developer tool (round 6): where the host's time goes in the `overlap` acting schedule with one captured update per environment step (profiles/tools/acting_bench.py's
loop with a clock around every phase). Usage: python profiles/tools/acting_host_profile.py [steps]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
import imitation_learning_amd as il
from imitation_learning_amd.environments import make_env

dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
actor, memory = nets[0], plan.memory
env = make_env('halfcheetah', True); env.seed(0)
for _ in range(3): plan.run()
worker = il.ActingWorker(actor, memory, mirror=True)
worker.attach(plan); plan.capture(warmup=0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
state, t = env.reset(), 0
action = worker.act(state)
torch.cuda.synchronize()
T = dict(env=0.0, post=0.0, replay=0.0, act_post_launch=0.0, act_wait=0.0, reset=0.0)
pc = time.perf_counter
box = worker._act_box
t_all = pc()
for step in range(1, steps + 1):
  a = pc(); nxt, r, term = env.step(action); t += 1; b = pc(); T['env'] += b - a
  worker.post(step, state, action, nxt, r, term and t != env.max_episode_steps, t == env.max_episode_steps); a = pc(); T['post'] += a - b
  state = env.reset() if term else nxt
  if term: t = 0
  b = pc(); T['reset'] += b - a
  plan.replay(); a = pc(); T['replay'] += a - b
  # worker.act(state), split: post + launch, then the wait for the device's echo
  from imitation_learning_amd.acting import _row
  seq = box.post(worker._next_seq(), 0, obs=_row(state))
  worker._launch(box, stream=worker.act_stream, snapshot=True); b = pc(); T['act_post_launch'] += b - a
  action = worker._collect(box, seq); a = pc(); T['act_wait'] += a - b
torch.cuda.synchronize()
tot = pc() - t_all
print(json.dumps(dict(env_steps_per_s=round(steps / tot, 1), us_per_step=round(tot / steps * 1e6, 2), host_us_per_step={k: round(v / steps * 1e6, 2) for k, v in T.items()},
                      unaccounted_us=round((tot - sum(T.values())) / steps * 1e6, 2))))
