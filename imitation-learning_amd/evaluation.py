"""Greedy-policy evaluation (reference evaluation.py:11-35)."""
from typing import List

import torch


def evaluate_agent(actor, env, num_episodes: int, return_trajectories: bool = False, render: bool = False):
  returns, trajectories = [], []
  with torch.inference_mode():
    for _ in range(num_episodes):
      states, actions, rewards = [], [], []
      state, terminal = env.reset(), False
      while not terminal:
        action = actor.get_greedy_action(state)
        next_state, reward, terminal = env.step(action)
        if return_trajectories:
          states.append(state); actions.append(action.cpu())
        rewards.append(reward)
        state = next_state
      returns.append(sum(rewards))
      if return_trajectories:
        terminals = torch.cat([torch.zeros(len(rewards) - 1), torch.ones(1)])
        trajectories.append({'states': torch.cat(states), 'actions': torch.cat(actions), 'rewards': torch.tensor(rewards, dtype=torch.float32), 'terminals': terminals})
  return (returns, trajectories) if return_trajectories else returns
