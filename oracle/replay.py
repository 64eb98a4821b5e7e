"""Replay ring oracle (TEST ORACLE, numpy) -- restates reference `memory.py:12-68`.

Struct-of-arrays float32 ring with the reference's exact quirks:
  * `__len__` returns capacity (memory.py:37-38);
  * index draw = one masked-rejection randint per sample plus rejection of the
    slot just before the cursor (memory.py:51-56); not-full range is [0, idx-2];
  * `sample` adds `absorbing = states[:, -1]` (or zeros) (memory.py:62);
  * `wrap_for_absorbing_states` rewrites the last row and appends the
    absorbing->absorbing row (memory.py:65-68);
  * `transfer_transitions` re-appends without weights (memory.py:46-48).
"""
from __future__ import annotations

import numpy as np

from .mt19937 import MT19937, sample_indices

FIELDS = ('step', 'states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights')


class ReplayOracle:
  def __init__(self, size, state_size, action_size, absorbing, transitions=None):
    f32 = np.float32
    self.size, self.num_trajectories, self.idx, self.full = size, 0, 0, False
    self.absorbing = absorbing
    self.step = np.zeros(size, f32)
    self.states = np.zeros((size, state_size), f32)
    self.actions = np.zeros((size, action_size), f32)
    self.rewards = np.zeros(size, f32)
    self.next_states = np.zeros((size, state_size), f32)
    self.terminals = np.zeros(size, f32)
    self.timeouts = np.zeros(size, f32)
    self.weights = np.zeros(size, f32)
    if transitions is not None:  # memory.py:18-23
      n = min(transitions['states'].shape[0], size)
      self.step[:n] = np.arange(1, size + 1, dtype=f32)[:n] if n < size else np.arange(1, size + 1, dtype=f32)
      for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
        getattr(self, k)[:n] = np.asarray(transitions[k], f32)[:n]
      self.num_trajectories = transitions['num_trajectories']
      self.idx = n % self.size
      self.full = self.idx == 0 and n > 0

  def __len__(self):
    return self.terminals.shape[0]

  def row(self, i):
    return {k: getattr(self, k)[i].copy() for k in FIELDS}

  def append(self, step, state, action, reward, next_state, terminal, timeout):  # memory.py:40-44
    i = self.idx
    self.step[i], self.rewards[i], self.terminals[i], self.timeouts[i], self.weights[i] = step, reward, terminal, timeout, 1
    self.states[i], self.actions[i], self.next_states[i] = np.reshape(state, -1), np.reshape(action, -1), np.reshape(next_state, -1)
    self.idx = (self.idx + 1) % self.size
    self.full = self.full or self.idx == 0
    if terminal or timeout:
      self.num_trajectories += 1

  def transfer_transitions(self, other: 'ReplayOracle'):  # memory.py:46-48
    for i in range(len(other)):
      r = other.row(i)
      self.append(r['step'], r['states'], r['actions'], r['rewards'], r['next_states'], r['terminals'], r['timeouts'])

  def sample_idx(self, gen: MT19937, n: int):
    return sample_indices(gen, n, self.size, self.idx, self.full)

  def gather(self, idxs):  # memory.py:60-62
    idxs = np.asarray(idxs, np.int64)
    out = {k: getattr(self, k)[idxs].copy() for k in FIELDS}
    out['absorbing'] = out['states'][:, -1].copy() if self.absorbing else np.zeros_like(out['terminals'])
    return out

  def sample(self, gen: MT19937, n: int):
    return self.gather(self.sample_idx(gen, n))

  def wrap_for_absorbing_states(self):  # memory.py:65-68
    s = self.states.shape[1]
    absorbing_state = np.concatenate([np.zeros(s - 1, np.float32), np.ones(1, np.float32)])
    last = (self.idx - 1) % self.size
    self.next_states[last], self.terminals[last] = absorbing_state, 0
    self.append(self.step[last], absorbing_state, np.zeros(self.actions.shape[1], np.float32), 0, absorbing_state, False, False)
