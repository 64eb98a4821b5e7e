"""Acting worker: the per-environment-step part of the reference loop (train.py:151-168) with the device work of one step in one launch.

The reference does, per env step, `actor(state).sample()` (models.py:90-94), `memory.append(...)` (memory.py:40-44) and, when an
episode ends by true termination with absorbing=true, `memory.wrap_for_absorbing_states()` (memory.py:65-68).  Here those three are
`il_act_step`: the host writes the observation / reward / flags into a pinned, device-mapped mailbox, launches ONE kernel, and spins
on the sequence-number echo the kernel stores (system-scope release) after the action — no stream synchronisation, no H2D/D2H copies,
no per-field device ops.  The ring cursor is advanced on the device; the host mirrors it arithmetically for the index draws.

Two schedules:
  exact   : act(obs) -> env.step -> append(transition) -> [update]           (reference order; 2 launches, 1 wait per env step)
  overlap : step(transition, obs) = append + act in one launch -> [update] runs on the GPU WHILE the host steps the environment;
            the action of step t+1 is sampled before update t, i.e. the behaviour policy lags by one update.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _lib

PENDING, WRAP_ABSORBING, GREEDY, NO_ACTION = 1, 2, 4, 8  # IL_ACT_* (include/il_hip.h)
_HEADER = 8
_SEQ_MOD = 1 << 20  # sequence numbers travel as fp32


class _Mailbox:
  """Host view of one il_act_step mailbox (layout documented in include/il_hip.h)."""

  def __init__(self, S: int, A: int):
    n = int(_lib.lib().il_act_mailbox_floats(S, A))
    self.tensor = torch.zeros(n, dtype=torch.float32, pin_memory=True)
    self.host = self.tensor.numpy()
    Sp, Ap = (S + 3) & ~3, (A + 3) & ~3
    self.S, self.A = S, A
    self.o_next, self.o_obs, self.o_act = _HEADER, _HEADER + Sp, _HEADER + 2 * Sp
    self.o_echo = self.o_act + Ap
    assert self.o_echo < n
    self.seq = 0
    self.host[self.o_echo] = -1.0

  def post(self, flags: int, reward: float = 0.0, terminal: float = 0.0, timeout: float = 0.0, step: float = 0.0, next_obs=None, obs=None) -> float:
    h = self.host
    self.seq = (self.seq + 1) % _SEQ_MOD
    h[0:6] = (self.seq, flags, reward, terminal, timeout, step)
    if next_obs is not None:
      h[self.o_next:self.o_next + self.S] = next_obs
    if obs is not None:
      h[self.o_obs:self.o_obs + self.S] = obs
    return float(self.seq)

  def wait(self, seq: float, what: str, timeout_s: float = 10.0):
    h, e = self.host, self.o_echo
    spins = 0
    while h[e] != seq:
      spins += 1
      if spins & 0xFFFF == 0:
        if time.perf_counter() - self._t0 > timeout_s:
          torch.cuda.synchronize()  # surfaces an asynchronous launch failure, if that is what happened
          raise RuntimeError(f'{what}: no echo from the device after {timeout_s:.0f} s (sequence {seq}, mailbox holds {h[e]})')
      elif spins == 1:
        self._t0 = time.perf_counter()


def _row(x) -> np.ndarray:
  if torch.is_tensor(x):
    x = x.detach().to('cpu', torch.float32).numpy()
  return np.asarray(x, dtype=np.float32).reshape(-1)


class ActingWorker:
  """One environment worker feeding one `ReplayMemory` from one `SoftActor` (train.py:151-168)."""

  def __init__(self, actor, memory):
    assert actor.flat.is_cuda and memory.ring.is_cuda, 'ActingWorker needs the actor and the ring on the GPU (there is no CPU path)'
    assert actor.state_size == memory.state_size and actor.action_size == memory.action_size
    self.actor, self.memory = actor, memory
    self.S, self.A = memory.state_size, memory.action_size
    self._act_box, self._append_box = _Mailbox(self.S, self.A), _Mailbox(self.S, self.A)
    self.carry = torch.zeros(self.S + self.A, dtype=torch.float32, device=memory.ring.device)
    self._seed = C.c_uint64(torch.initial_seed() & (2**64 - 1))
    self._fixed = {}

  def _launch(self, box: _Mailbox, acts: bool = True):
    a = self.actor
    a._act_calls += int(acts)  # the Philox offset is shared with SoftActor._act, so the two entry points never reuse noise
    fixed = self._fixed.get(id(box))
    if fixed is None or fixed[0] != a.flat.data_ptr():  # pointers are stable for the life of the worker; re-derive if the arena was re-homed
      fixed = self._fixed[id(box)] = (a.flat.data_ptr(), _lib.lib().il_act_step, _lib.ptr(a.flat), C.c_void_p(box.tensor.data_ptr()), _lib.ptr(self.carry),
                                      _lib.ptr(self.memory.ring), _lib.ptr(self.memory._ring_state))
    _, fn, p_actor, p_box, p_carry, p_ring, p_state = fixed
    rc = fn(p_actor, self.S, self.A, a.hidden, p_box, p_carry, p_ring, p_state, self._seed, a._act_calls & 0xFFFFFFFF, torch.cuda.current_stream().cuda_stream)
    if rc: _lib.check(rc)

  def _collect(self, box: _Mailbox, seq: float) -> torch.Tensor:
    box.wait(seq, 'il_act_step')
    return torch.from_numpy(box.host[box.o_act:box.o_act + self.A].copy()).unsqueeze(0)

  def _mirror_append(self, terminal: bool, timeout: bool, wrap: bool):
    m = self.memory
    m._advance(terminal, timeout)
    if wrap:
      m._advance(False, False)

  # --- exact schedule
  def act(self, obs, greedy: bool = False) -> torch.Tensor:
    """`actor(obs).sample()` (or the greedy action) as a [1, A] CPU tensor; remembers (obs, action) on the device for `append`."""
    box = self._act_box
    seq = box.post(GREEDY if greedy else 0, obs=_row(obs))
    self._launch(box)
    return self._collect(box, seq)

  def append(self, step, next_obs, reward, terminal: bool, timeout: bool):
    """`memory.append(step, state, action, reward, next_state, terminal, timeout)` for the (state, action) of the last `act`, plus the
    absorbing wrap when the episode ended by true termination (train.py:157,161). Asynchronous: nothing is waited for."""
    wrap = bool(self.memory.absorbing and terminal and not timeout)
    box = self._append_box
    if box.seq: box.wait(float(box.seq), 'il_act_step(append)')  # normally already echoed: the act in between ran after it on the same stream
    box.post(PENDING | NO_ACTION | (WRAP_ABSORBING if wrap else 0), float(reward), float(terminal), float(timeout), float(step), next_obs=_row(next_obs))
    self._launch(box, acts=False)
    self._mirror_append(bool(terminal), bool(timeout), wrap)

  # --- overlap schedule
  def step(self, step, next_obs, reward, terminal: bool, timeout: bool, obs=None, greedy: bool = False) -> torch.Tensor:
    """append(transition of the last action) + act(obs) in ONE launch. `obs` defaults to `next_obs`; pass the reset observation when
    the episode ended."""
    wrap = bool(self.memory.absorbing and terminal and not timeout)
    box = self._act_box
    nxt = _row(next_obs)
    seq = box.post(PENDING | (WRAP_ABSORBING if wrap else 0) | (GREEDY if greedy else 0), float(reward), float(terminal), float(timeout), float(step), next_obs=nxt,
                   obs=nxt if obs is None else _row(obs))
    self._launch(box)
    self._mirror_append(bool(terminal), bool(timeout), wrap)
    return self._collect(box, seq)
