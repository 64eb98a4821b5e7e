"""Greedy-policy evaluation: same call and return shape as the reference's evaluate_agent (evaluation.py:11-35).

The policy mean comes from `il_actor_act` (one launch per environment step, SoftActor.get_greedy_action); the environment and the
episode bookkeeping stay on the host, as in the reference. Episodes are recorded into preallocated host buffers (the horizon is
known from env.max_episode_steps) instead of python lists of 1-row tensors.
"""
import torch


class _EpisodeLog:
  """Host-side record of one evaluation episode."""

  def __init__(self, horizon: int, keep_steps: bool):
    self.keep_steps, self.length, self.total = keep_steps, 0, 0.0
    self.reward = torch.zeros(horizon, dtype=torch.float32)
    self.state = self.action = None

  def push(self, state, action, reward: float):
    t = self.length
    if t >= self.reward.numel():  # environments without a declared horizon: grow geometrically
      self.reward = torch.cat([self.reward, torch.zeros_like(self.reward)])
      if self.state is not None:
        self.state, self.action = torch.cat([self.state, torch.zeros_like(self.state)]), torch.cat([self.action, torch.zeros_like(self.action)])
    if self.keep_steps:
      if self.state is None:
        self.state = torch.zeros(self.reward.numel(), state.shape[-1]); self.action = torch.zeros(self.reward.numel(), action.shape[-1])
      self.state[t], self.action[t] = state.reshape(-1).cpu(), action.reshape(-1).cpu()
    self.reward[t] = reward
    self.total += float(reward)
    self.length = t + 1

  def as_trajectory(self):
    n = self.length
    done = torch.zeros(n); done[n - 1] = 1.0
    return dict(states=self.state[:n].clone(), actions=self.action[:n].clone(), rewards=self.reward[:n].clone(), terminals=done)


def evaluate_agent(actor, env, num_episodes: int, return_trajectories: bool = False, render: bool = False):
  horizon = int(getattr(env, 'max_episode_steps', 0) or 1024)
  logs = []
  with torch.inference_mode():
    for _ in range(num_episodes):
      log, obs, finished = _EpisodeLog(horizon, return_trajectories), env.reset(), False
      while not finished:
        act = actor.get_greedy_action(obs)
        nxt, r, finished = env.step(act)
        log.push(obs, act, r)
        obs = nxt
      logs.append(log)
  returns = [log.total for log in logs]
  if return_trajectories:
    return returns, [log.as_trajectory() for log in logs]
  return returns
