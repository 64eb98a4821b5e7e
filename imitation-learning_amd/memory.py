"""HBM-resident replay ring with the reference's `ReplayMemory` interface (reference memory.py:12-68).

Layout: one fp32 tensor [size, row] on the GPU, row = il_ring_row_floats(S, A) (packed fields, 16-byte multiple), so a sample
is ONE coalesced gather of whole rows (k_gather) instead of 8 x B tensor-index ops, and the returned dict holds strided views
into the packed batch.  Index draws reproduce numpy's legacy MT19937 `randint` stream bit for bit (`seed()` == np.random.seed):
host-side by default, or fully on the device (`sample_device`) so a captured update graph needs no H2D traffic.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Union

import torch
from torch import Tensor

from . import _lib

FIELDS = ('step', 'states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights')


class IndexStream:
  """The process-wide index RNG (the reference draws every index from numpy's global stream, memory.py:54)."""

  def __init__(self, seed: int = 0):
    self.state = (C.c_uint32 * 625)()
    self.seed(seed)
    self._dev = None

  def seed(self, seed: int):
    _lib.check(_lib.lib().il_mt19937_seed(self.state, seed & 0xFFFFFFFF))
    self._dev = None

  def draw(self, n: int, size: int, idx: int, full: bool) -> Tensor:
    out = torch.empty(n, dtype=torch.int32)
    _lib.check(_lib.lib().il_mt19937_sample_indices(self.state, n, size, idx, int(full), C.cast(out.data_ptr(), _lib.c_i32p)))
    return out

  def randint(self, high: int) -> int:
    """`np.random.randint(0, high)` / `np.random.choice(high)` from the same stream."""
    out = (C.c_int32 * 1)()
    _lib.check(_lib.lib().il_mt19937_randint(self.state, high, 1, out))
    return int(out[0])

  # device-resident copy of the same stream (moved once; afterwards the device state is the master)
  def device_state(self, device) -> Tensor:
    if self._dev is None:
      host = torch.frombuffer(bytearray(bytes(self.state)), dtype=torch.int32).clone()
      self._dev = host.to(device)
    return self._dev


_STREAM: Optional[IndexStream] = None


def index_stream() -> IndexStream:
  global _STREAM
  if _STREAM is None:
    _STREAM = IndexStream(0)
  return _STREAM


def seed(s: int):
  """Equivalent of `np.random.seed(s)` for replay index draws (reference train.py:51)."""
  index_stream().seed(s)


def row_layout(state_size: int, action_size: int):
  S, A = state_size, action_size
  return dict(states=(0, S), actions=(S, A), next_states=(S + A, S), rewards=(2 * S + A, 1), terminals=(2 * S + A + 1, 1), timeouts=(2 * S + A + 2, 1),
              weights=(2 * S + A + 3, 1), step=(2 * S + A + 4, 1))


def batch_views(rows: Tensor, state_size: int, action_size: int, absorbing: bool) -> Dict[str, Tensor]:
  """The reference's `transitions` dict as strided views into packed rows [n, row]."""
  out = {}
  for k, (o, n) in row_layout(state_size, action_size).items():
    out[k] = rows[:, o:o + n] if n > 1 or k in ('states', 'actions', 'next_states') else rows[:, o]
  out = {k: out[k] for k in FIELDS}
  out['absorbing'] = out['states'][:, -1] if absorbing else torch.zeros_like(out['terminals'])
  return out


def batch_desc(t: Dict[str, Tensor]) -> _lib.Batch:
  """il_batch from a transitions dict (any fp32 device tensors with unit inner stride)."""
  b = _lib.Batch()
  n = None
  for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'weights', 'absorbing'):
    v = t[k]
    if v.dtype != torch.float32 or not _lib.on_device(v):
      raise TypeError(f'transitions[{k!r}] must be a float32 CUDA tensor (got {v.dtype} on {v.device})')
    if v.dim() == 2 and v.stride(1) != 1:
      raise ValueError(f'transitions[{k!r}] must have unit inner stride')
    setattr(b, k, v.data_ptr()); setattr(b, 'ld_' + k, v.stride(0) if v.size(0) > 1 else (v.size(1) if v.dim() == 2 else 1))
    n = v.size(0) if n is None else n
    assert v.size(0) == n, 'ragged transitions dict'
  b.n = n
  return b


class ReplayMemory(torch.utils.data.Dataset):
  def __init__(self, size: int, state_size: int, action_size: int, absorbing: bool, transitions: Optional[Dict[str, Union[Tensor, int]]] = None, device=None):
    super().__init__()
    from .models import default_device
    self.device = torch.device(device) if device is not None else default_device()
    self.size, self.num_trajectories, self.idx, self.full = int(size), 0, 0, False
    self.absorbing, self.state_size, self.action_size = absorbing, state_size, action_size
    self.row = int(_lib.lib().il_ring_row_floats(state_size, action_size))
    self.layout = row_layout(state_size, action_size)
    self.ring = torch.zeros(self.size, self.row, dtype=torch.float32, device=self.device)
    self._stage = torch.zeros(2, self.row, dtype=torch.float32, pin_memory=self.device.type == 'cuda')
    self._ring_state = torch.zeros(3, dtype=torch.int64, device=self.device)
    self.index_rng: Optional[IndexStream] = None  # None = the process-wide stream (reference behaviour); set to give this memory its own
    if transitions is not None:
      n = min(transitions['states'].size(0), self.size)
      for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'timeouts', 'weights'):
        getattr(self, k)[:n] = transitions[k][:n].to(self.device, torch.float32)
      self.step[:n] = torch.arange(1, n + 1, dtype=torch.float32, device=self.device)
      self.num_trajectories = transitions['num_trajectories']
      self.idx = n % self.size
      self.full = self.idx == 0 and n > 0
    self._sync_ring_state()

  # --- field views (the reference exposes .states/.actions/... tensors; PWIL writes .rewards[i], train.py:139)
  def _field(self, k):
    o, n = self.layout[k]
    return self.ring[:, o:o + n] if k in ('states', 'actions', 'next_states') else self.ring[:, o]

  step = property(lambda self: self._field('step'))
  states = property(lambda self: self._field('states'))
  actions = property(lambda self: self._field('actions'))
  rewards = property(lambda self: self._field('rewards'))
  next_states = property(lambda self: self._field('next_states'))
  terminals = property(lambda self: self._field('terminals'))
  timeouts = property(lambda self: self._field('timeouts'))
  weights = property(lambda self: self._field('weights'))

  def _sync_ring_state(self):
    self._ring_state.copy_(torch.tensor([self.idx, int(self.full), self.size], dtype=torch.int64), non_blocking=True)

  def __getitem__(self, idx: Union[int, str]):
    if isinstance(idx, str):
      if idx in ('states', 'actions', 'terminals'):
        return getattr(self, idx)
      return None
    return {k: getattr(self, k)[idx] for k in FIELDS}

  def __len__(self) -> int:
    return self.size  # capacity, like the reference (memory.py:37-38)

  def _advance(self, terminal, timeout):
    self.idx = (self.idx + 1) % self.size
    self.full = self.full or self.idx == 0
    if terminal or timeout:
      self.num_trajectories += 1

  def append(self, step, state: Tensor, action: Tensor, reward, next_state: Tensor, terminal, timeout):
    S, A = self.state_size, self.action_size
    r = torch.empty(self.row, dtype=torch.float32, device=self.device) if any(torch.is_tensor(x) and x.is_cuda for x in (state, action, next_state, reward, step)) else self._stage[0]
    if r.is_cuda:  # inputs already on the device (acting path): assemble the row there, no host round trip
      r.zero_()
      r[:S] = state.reshape(-1); r[S:S + A] = action.reshape(-1); r[S + A:2 * S + A] = next_state.reshape(-1)
      r[2 * S + A] = reward; r[2 * S + A + 1] = float(terminal); r[2 * S + A + 2] = float(timeout); r[2 * S + A + 3] = 1.0; r[2 * S + A + 4] = step
      src = r
    else:
      r.zero_()
      r[:S] = torch.as_tensor(state, dtype=torch.float32).reshape(-1); r[S:S + A] = torch.as_tensor(action, dtype=torch.float32).reshape(-1)
      r[S + A:2 * S + A] = torch.as_tensor(next_state, dtype=torch.float32).reshape(-1)
      r[2 * S + A] = float(reward); r[2 * S + A + 1] = float(terminal); r[2 * S + A + 2] = float(timeout); r[2 * S + A + 3] = 1.0; r[2 * S + A + 4] = float(step)
      src = r.to(self.device, non_blocking=False)
    _lib.check(_lib.lib().il_replay_write_rows(_lib.ptr(self.ring), self.size, self.row, self.idx, _lib.ptr(src), 1, _lib.stream_ptr()))
    self._advance(bool(terminal), bool(timeout))
    self._sync_ring_state()

  def transfer_transitions(self, memory: 'ReplayMemory'):
    """memory.py:46-48: re-append every slot of `memory` (weights reset to 1). Done as one bulk device copy, cursor walked on the host. The reference appends one
    row at a time, so when the source holds more rows than this ring the LAST write to a slot wins: only the final `self.size` rows are copied (a parallel
    launch over all n rows would race on the slots that are written twice), to the slots the sequential appends would have left them in."""
    n = len(memory)
    flags = (memory.terminals[:n] + memory.timeouts[:n]).ne(0).cpu()
    skip = max(0, n - self.size)                      # rows that a later append overwrites anyway
    src = memory.ring[skip:n].clone()
    src[:, self.layout['weights'][0]] = 1.0
    _lib.check(_lib.lib().il_replay_write_rows(_lib.ptr(self.ring), self.size, self.row, (self.idx + skip) % self.size, _lib.ptr(src), n - skip, _lib.stream_ptr()))
    self.num_trajectories += int(flags.sum())         # every append counts its episode end, overwritten or not (memory.py:44)
    self.full = self.full or (self.idx + n >= self.size)
    self.idx = (self.idx + n) % self.size
    self._sync_ring_state()

  def stream(self) -> IndexStream:
    return self.index_rng if self.index_rng is not None else index_stream()

  def _sample_idx_tensor(self, n: int) -> Tensor:
    return self.stream().draw(n, self.size, self.idx, self.full)

  def gather(self, idx: Tensor) -> Tensor:
    idx = idx.to(self.device, torch.int32, non_blocking=True)
    out = torch.empty(idx.numel(), self.row, dtype=torch.float32, device=self.device)
    _lib.check(_lib.lib().il_replay_gather(_lib.ptr(self.ring), self.size, self.row, _lib.ptr(idx), idx.numel(), _lib.ptr(out), _lib.stream_ptr()))
    return out

  def sample(self, n: int) -> Dict[str, Tensor]:
    rows = self.gather(self._sample_idx_tensor(n))
    return batch_views(rows, self.state_size, self.action_size, self.absorbing)

  def sample_device(self, n: int, idx_out: Tensor, rows_out: Tensor) -> Dict[str, Tensor]:
    """Graph-capturable sample: MT19937 draw + rejection on the device (same stream), then the gather; no host involvement."""
    st = self.stream().device_state(self.device)
    _lib.check(_lib.lib().il_mt19937_sample_indices_device(_lib.ptr(st), _lib.ptr(self._ring_state), n, _lib.ptr(idx_out), _lib.stream_ptr()))
    _lib.check(_lib.lib().il_replay_gather(_lib.ptr(self.ring), self.size, self.row, _lib.ptr(idx_out), n, _lib.ptr(rows_out), _lib.stream_ptr()))
    return batch_views(rows_out, self.state_size, self.action_size, self.absorbing)

  def wrap_for_absorbing_states(self):
    last = (self.idx - 1) % self.size
    _lib.check(_lib.lib().il_replay_wrap_absorbing(_lib.ptr(self.ring), self.size, self.state_size, self.action_size, last, self.idx, _lib.stream_ptr()))
    self._advance(False, False)
    self._sync_ring_state()
