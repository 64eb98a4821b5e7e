"""Environment + expert-data side of the loop (reference environments.py).

gym 0.23 + d4rl + MuJoCo are not installable in this environment (no network), and CPU physics is not the path being
accelerated (out of scope). `SyntheticD4RLEnv` reproduces what the update path depends on: observation /
action shapes of the four D4RL locomotion tasks, the absorbing-indicator bit, action clipping, `max_episode_steps`, early
termination for the tasks that have it, and `get_dataset()` with the reference's trajectory split, truncation, absorbing-state
wrapping, importance weights and sub-sampling semantics (environments.py:63-125) on synthetic expert rollouts.
If gym and d4rl ARE importable, `make_env` returns `GymD4RLEnv`, the real task behind the same interface (exercised in tests/test_config_cpu.py
against a stub gym; the real packages cannot be installed here).
"""
from __future__ import annotations

from typing import List, Tuple

import zlib

import numpy as np
import torch
from torch import Tensor

from .memory import ReplayMemory, index_stream

ENVS = ['ant', 'halfcheetah', 'hopper', 'walker2d']
_SPECS = {  # obs dim (without absorbing bit), action dim, can terminate early, (ref_min_score, ref_max_score) of the D4RL tasks
    'ant': (111, 8, True, (-325.6, 3879.7)), 'halfcheetah': (17, 6, False, (-280.178953, 12135.0)),
    'hopper': (11, 3, True, (-20.272305, 3234.3)), 'walker2d': (17, 6, True, (1.629008, 4592.3)),
}


class _Space:
  def __init__(self, dim):
    self.shape = (dim,)
    self.low, self.high = -torch.ones(dim), torch.ones(dim)


class SyntheticD4RLEnv:
  """Seeded linear-Gaussian locomotion stand-in with the interface of the reference's `D4RLEnv`."""

  def __init__(self, env_name: str, absorbing: bool, load_data: bool = False, dataset_trajectories: int = 30, max_episode_steps: int = 1000, dataset_path: str = None):
    assert env_name in ENVS
    self.name, self.absorbing = env_name, absorbing
    self.obs_dim, self.act_dim, self.can_terminate, (self.ref_min_score, self.ref_max_score) = _SPECS[env_name]
    self.max_episode_steps = max_episode_steps
    rs = np.random.RandomState(zlib.crc32(env_name.encode()))  # stable across processes (str hashes are salted)
    self._A = (np.eye(self.obs_dim) * 0.95 + rs.standard_normal((self.obs_dim, self.obs_dim)) * 0.02).astype(np.float32)
    self._Bm = (rs.standard_normal((self.obs_dim, self.act_dim)) * 0.3).astype(np.float32)
    self._K = (rs.standard_normal((self.act_dim, self.obs_dim)) * 0.2).astype(np.float32)  # the synthetic expert's linear policy
    self._w = (rs.standard_normal(self.obs_dim) * 0.1).astype(np.float32)
    self.observation_space, self.action_space = _Space(self.obs_dim + (1 if absorbing else 0)), _Space(self.act_dim)
    self.env = self  # `env.env.ref_max_score` (train.py:58)
    self._rs, self._t, self._x = np.random.RandomState(0), 0, None
    self._dataset_trajectories = dataset_trajectories
    self.dataset = (load_dataset_file(dataset_path) if dataset_path else self._make_dataset()) if load_data else None   # `+synthetic_env.dataset_path=<file>`: real D4RL arrays

  # --- gym-like API
  def seed(self, seed: int) -> List[int]:
    self._rs = np.random.RandomState(seed)
    return [seed]

  def _obs(self, x) -> Tensor:
    state = torch.tensor(x, dtype=torch.float32).unsqueeze(0)
    return torch.cat([state, torch.zeros(1, 1)], dim=1) if self.absorbing else state

  def reset(self) -> Tensor:
    self._x, self._t = (self._rs.standard_normal(self.obs_dim) * 0.1).astype(np.float32), 0
    return self._obs(self._x)

  def _dynamics(self, x, a, rs):
    x2 = self._A @ x + self._Bm @ a + (rs.standard_normal(self.obs_dim) * 0.01).astype(np.float32)
    reward = float(self._w @ x2 - 0.05 * float(a @ a) + 1.0)
    done = bool(self.can_terminate and abs(float(x2[0])) > 1.5)
    return x2.astype(np.float32), reward, done

  def step(self, action: Tensor) -> Tuple[Tensor, float, bool]:
    a = action.detach().to('cpu', torch.float32).clamp(-1, 1)[0].numpy()
    self._x, reward, done = self._dynamics(self._x, a, self._rs)
    self._t += 1
    return self._obs(self._x), reward, done or self._t >= self.max_episode_steps

  def render(self): pass
  def close(self): pass

  # --- synthetic "D4RL" dataset in the raw D4RL format (flat arrays + terminals/timeouts flags)
  def _make_dataset(self):
    rs = np.random.RandomState(12345)
    obs, act, nxt, term, tout = [], [], [], [], []
    for _ in range(self._dataset_trajectories):
      x = (rs.standard_normal(self.obs_dim) * 0.1).astype(np.float32)
      for t in range(self.max_episode_steps):
        a = np.tanh(self._K @ x + rs.standard_normal(self.act_dim).astype(np.float32) * 0.05).astype(np.float32)
        x2, _, done = self._dynamics(x, a, rs)
        last = t == self.max_episode_steps - 1
        obs.append(x); act.append(a); nxt.append(x2); term.append(float(done)); tout.append(float(last and not done))
        x = x2
        if done or last:
          break
    f = lambda v: torch.tensor(np.asarray(v), dtype=torch.float32)
    return dict(observations=f(obs), actions=f(act), next_observations=f(nxt), terminals=f(term), timeouts=f(tout))

  def get_dataset(self, trajectories: int = 0, subsample: int = 1, device=None) -> ReplayMemory:
    """Reference environments.py:63-125 on this environment's raw dataset."""
    return dataset_to_memory(self.dataset, self.absorbing, trajectories, subsample, device)


def load_dataset_file(path: str) -> dict:
  """Raw D4RL-format dataset (observations, actions, next_observations, terminals, timeouts) from disk: `.npz`, or the `.hdf5` files D4RL ships
  when h5py is importable (it is not in this image). `next_observations` is rebuilt from the observations when the file lacks it, as d4rl.qlearning_dataset does."""
  keys = ('observations', 'actions', 'next_observations', 'terminals', 'timeouts')
  if path.endswith('.npz'):
    with np.load(path) as f:
      raw = {k: np.asarray(f[k]) for k in keys if k in f}
  else:
    try:
      import h5py
    except ImportError as e:
      raise ImportError(f'{path}: reading HDF5 needs h5py (not installed here); convert the file to .npz with the five D4RL arrays') from e
    with h5py.File(path, 'r') as f:
      raw = {k: np.asarray(f[k]) for k in keys if k in f}
  if 'next_observations' not in raw:
    nxt = np.roll(raw['observations'], -1, axis=0)
    nxt[-1] = raw['observations'][-1]
    raw['next_observations'] = nxt
  missing = [k for k in keys if k not in raw]
  if missing:
    raise KeyError(f'{path}: missing dataset arrays {missing}')
  return {k: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32) for k, v in raw.items()}


def dataset_to_memory(dataset: dict, absorbing: bool, trajectories: int = 0, subsample: int = 1, device=None) -> ReplayMemory:
  """Expert-data ingest (reference environments.py:63-125): split the flat arrays into trajectories at the terminal / timeout flags (a dangling
  tail is dropped), keep the first `trajectories`, append the absorbing indicator and - for true terminations - rewrite the last next-state and add
  the absorbing -> absorbing transition with importance weight 1/subsample, sub-sample every `subsample`-th step from a random phase (same index
  stream as the reference's np.random.choice) while keeping the two absorbing rows, and upload the result as one packed ReplayMemory."""
  f = lambda v: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=torch.float32)
  states, actions, next_states, terminals, timeouts = (f(dataset[k]) for k in ('observations', 'actions', 'next_observations', 'terminals', 'timeouts'))
  S, A = states.size(1), actions.size(1)
  ends = torch.sort(torch.cat([terminals.nonzero().flatten(), timeouts.nonzero().flatten()]))[0].tolist()
  starts = [0] + [e + 1 for e in ends[:-1]]
  if trajectories > 0:
    starts, ends = starts[:trajectories], ends[:trajectories]
  out = {k: [] for k in ('states', 'actions', 'next_states', 'terminals', 'timeouts', 'weights')}
  for s0, e0 in zip(starts, ends):
    sl = slice(s0, e0 + 1)
    st, ac, nx, te, to = states[sl], actions[sl], next_states[sl], terminals[sl].clone(), timeouts[sl]
    w = torch.ones_like(te)
    if absorbing:
      st, nx = torch.cat([st, torch.zeros(st.size(0), 1)], 1), torch.cat([nx, torch.zeros(nx.size(0), 1)], 1)
      if not to[-1]:  # true termination: rewrite the last next-state to the absorbing state and append absorbing -> absorbing
        absorbing_state = torch.cat([torch.zeros(1, S), torch.ones(1, 1)], 1)
        nx[-1], te[-1], w[-1] = absorbing_state[0], 0, 1 / subsample
        st, ac, nx = torch.cat([st, absorbing_state]), torch.cat([ac, torch.zeros(1, A)]), torch.cat([nx, absorbing_state])
        te, to, w = torch.cat([te, torch.zeros(1)]), torch.cat([to, torch.zeros(1)]), torch.cat([w, torch.full((1,), 1 / subsample)])
    if subsample > 1:
      T = st.size(0)
      idxs = set(range(index_stream().randint(subsample), T, subsample))  # np.random.choice(subsample) in the reference: same stream
      if absorbing: idxs |= {T - 2, T - 1}
      idxs = sorted(idxs)
      st, ac, nx, te, to, w = st[idxs], ac[idxs], nx[idxs], te[idxs], to[idxs], w[idxs]
    for k, v in zip(out, (st, ac, nx, te, to, w)):
      out[k].append(v)
  tr = {k: torch.cat(v) for k, v in out.items()}
  tr['num_trajectories'], tr['rewards'] = len(starts), torch.zeros_like(tr['terminals'])
  return ReplayMemory(tr['states'].size(0), S + (1 if absorbing else 0), A, absorbing, transitions=tr, device=device)


class GymD4RLEnv:
  """The real D4RL locomotion task behind the interface the training loop uses (reference environments.py:20-61): `{env}-expert-v2` from gym + d4rl,
  observations / actions as [1, dim] float32 CPU tensors, the absorbing indicator bit appended to every observation when `absorbing`, actions clipped to
  the action space, `max_episode_steps` from the TimeLimit wrapper, D4RL's reference scores on `env.env`, and `get_dataset()` = the same ingest as the
  synthetic environment (`dataset_to_memory`) on the arrays d4rl ships."""

  def __init__(self, env_name: str, absorbing: bool, load_data: bool = False):
    import d4rl  # noqa: F401  (registers the `-expert-v2` ids)
    import gym
    assert env_name in ENVS
    self.env, self.absorbing = gym.make(f'{env_name}-expert-v2'), absorbing
    self.dataset = None
    if load_data:
      raw = self.env.get_dataset()
      if 'next_observations' not in raw:   # older d4rl files: what d4rl.qlearning_dataset does
        raw = dict(raw); raw['next_observations'] = np.concatenate([raw['observations'][1:], raw['observations'][-1:]])
      self.dataset = {k: torch.as_tensor(np.ascontiguousarray(raw[k]), dtype=torch.float32) for k in ('observations', 'actions', 'next_observations', 'terminals', 'timeouts')}
    space = self.env.action_space
    self._low, self._high = torch.as_tensor(space.low, dtype=torch.float32), torch.as_tensor(space.high, dtype=torch.float32)
    obs_dim = int(self.env.observation_space.shape[0])
    self.observation_space, self.action_space = _Space(obs_dim + (1 if absorbing else 0)), _Space(int(space.shape[0]))
    self.action_space.low, self.action_space.high = self._low, self._high

  def _obs(self, x) -> Tensor:
    state = torch.as_tensor(np.asarray(x), dtype=torch.float32).reshape(1, -1)
    return torch.cat([state, torch.zeros(1, 1)], dim=1) if self.absorbing else state   # the absorbing rewrite itself happens in the replay memory

  def reset(self) -> Tensor:
    out = self.env.reset()
    return self._obs(out[0] if isinstance(out, tuple) else out)   # gym >= 0.26 returns (obs, info)

  def step(self, action: Tensor) -> Tuple[Tensor, float, bool]:
    a = torch.max(torch.min(action.detach().to('cpu', torch.float32), self._high), self._low)[0].numpy()
    out = self.env.step(a)
    obs, reward, done = out[0], out[1], (bool(out[2]) if len(out) == 4 else bool(out[2]) or bool(out[3]))   # gym >= 0.26: (obs, r, terminated, truncated, info)
    return self._obs(obs), float(reward), done

  def seed(self, seed: int):
    return self.env.seed(seed) if hasattr(self.env, 'seed') else [seed]

  def render(self):
    return self.env.render()

  def close(self):
    self.env.close()

  @property
  def max_episode_steps(self) -> int:
    return int(self.env._max_episode_steps)

  def get_dataset(self, trajectories: int = 0, subsample: int = 1, device=None) -> ReplayMemory:
    return dataset_to_memory(self.dataset, self.absorbing, trajectories, subsample, device)


def make_env(env_name: str, absorbing: bool, load_data: bool = False, **kw):
  """The real D4RL task when gym + d4rl are importable (`GymD4RLEnv`), otherwise the synthetic stand-in - loudly, once: results on it say nothing about D4RL scores.
  `+synthetic_env.*` keys force the stand-in (they have no meaning for the real task)."""
  if not kw:
    try:
      import d4rl  # noqa: F401
      import gym  # noqa: F401
      return GymD4RLEnv(env_name, absorbing, load_data)
    except ImportError:
      pass
  global _WARNED
  if not _WARNED and not kw:
    import warnings
    warnings.warn('gym / d4rl are not importable: using SyntheticD4RLEnv (D4RL-shaped linear-Gaussian stand-in). Normalised scores are not D4RL scores.', RuntimeWarning, stacklevel=2)
    _WARNED = True
  return SyntheticD4RLEnv(env_name, absorbing, load_data, **kw)


_WARNED = False
D4RLEnv = make_env
