"""AdRIL / SQIL batch mixing and reward relabelling (TEST ORACLE, numpy) -- restates reference `models.py:287-318`.

`mix` follows `mix_expert_agent_transitions` (models.py:287-290): the first B//2 entries of EVERY key are overwritten with the
expert batch's.  `RelabellerOracle.resample_and_relabel` follows `RewardRelabeller` (models.py:293-318): balanced mode alternates
all-expert / all-policy batches (stateful, starts with expert), unbalanced mode mixes halves; rewards are constants:
  AdRIL (update_freq > 0): expert +1/|expert trajectories|; policy -[round(step) > round(row.step)] / max(|policy trajectories|, 1)
                           with round(x) = ceil(x / update_freq) -- note -1 * 0 keeps the sign bit (-0.0), as in torch;
  SQIL  (update_freq == 0): expert 1, policy 0.
Pinned by tests/golden/adril.npz (generated from the reference classes by tests/golden/make_golden.py).
"""
from __future__ import annotations

from math import ceil

import numpy as np

f32 = np.float32


def mix(transitions: dict, expert_transitions: dict):
  half = transitions['rewards'].shape[0] // 2
  for k in transitions:
    transitions[k][:half] = expert_transitions[k][:half]


class RelabellerOracle:
  def __init__(self, update_freq: int, balanced: bool):
    self.update_freq, self.balanced, self.sample_expert = update_freq, balanced, True

  def resample_and_relabel(self, transitions: dict, expert_transitions: dict, step: int, num_trajectories: int, num_expert_trajectories: int):
    B = transitions['rewards'].shape[0]
    if self.balanced:
      if self.sample_expert:
        for k in transitions:
          transitions[k] = expert_transitions[k].copy()
        n_expert = B
      else:
        n_expert = 0
      self.sample_expert = not self.sample_expert
    else:
      mix(transitions, expert_transitions)
      n_expert = B // 2
    r = transitions['rewards']
    if self.update_freq > 0:
      r[:n_expert] = f32(1 / num_expert_trajectories)
      round_num = ceil(step / self.update_freq)
      row_round = np.ceil(transitions['step'][n_expert:].astype(f32) / f32(self.update_freq))
      r[n_expert:] = (f32(-1) * (f32(round_num) > row_round).astype(f32)) / f32(max(num_trajectories, 1))
    else:
      r[:n_expert] = 1
      r[n_expert:] = 0
    return n_expert
