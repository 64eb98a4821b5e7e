"""Acting worker: the per-environment-step part of the reference loop (train.py:151-168) with the device work of one step in one launch.

The reference does, per env step, `actor(state).sample()` (models.py:90-94), `memory.append(...)` (memory.py:40-44) and, when an
episode ends by true termination with absorbing=true, `memory.wrap_for_absorbing_states()` (memory.py:65-68).  Here those three are
`il_act_step`: the host writes the observation / reward / flags into a pinned, device-mapped mailbox, launches ONE kernel, and spins
on the sequence-number echo the kernel stores (system-scope release) after the action — no stream synchronisation, no H2D/D2H copies,
no per-field device ops.  The ring cursor is advanced on the device; the host mirrors it arithmetically for the index draws.

Schedules:
  exact   : act(obs) -> env.step -> append(transition) -> [update]           (reference order; 2 launches, 1 wait per env step)
  fused   : step(transition, obs) = append + act in ONE launch on the update stream (the action of step t+1 is sampled before update t)
  overlap : `ActingWorker(..., mirror=True)`: act runs on its OWN stream against a published snapshot of the actor (il_act_publish, three
            slots + a version word), so the host gets its action in ~15 us and steps the environment WHILE the GPU runs the update;
            the append and the publish ride in the update's stream / hipGraph (UpdatePlan.pre_hooks / post_hooks). The behaviour
            policy lags the learner by one to two updates (what a host-side actor mirror would do).
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _lib

PENDING, WRAP_ABSORBING, GREEDY, NO_ACTION, CARRY_FROM_MAILBOX = 1, 2, 4, 8, 16  # IL_ACT_* (include/il_hip.h)
_HEADER = 8
_SEQ_MOD = 1 << 17  # the commit word (sequence * 64 + flags) travels as fp32: < 2^23


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
    self.word = 0.0          # commit word of the last post
    self.host[self.o_echo] = -1.0

  def post(self, seq: int, flags: int, reward: float = 0.0, terminal: float = 0.0, timeout: float = 0.0, step: float = 0.0, next_obs=None, obs=None, action=None) -> float:
    """Payload first, commit word (sequence * 64 + flags) last: a launch that is already queued sees either the previous post or this one, whole."""
    h = self.host
    h[2:6] = (reward, terminal, timeout, step)
    if next_obs is not None:
      h[self.o_next:self.o_next + self.S] = next_obs
    if obs is not None:
      h[self.o_obs:self.o_obs + self.S] = obs
    if action is not None:
      h[self.o_act:self.o_act + self.A] = action
    self.word = float(seq * 64 + flags)
    h[0] = self.word
    return self.word

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

  def __init__(self, actor, memory, mirror: bool = False):
    assert _lib.on_device(actor.flat) and _lib.on_device(memory.ring), 'ActingWorker needs the actor and the ring on the GPU (there is no CPU path)'
    assert actor.state_size == memory.state_size and actor.action_size == memory.action_size
    if getattr(actor, 'general', False):
      raise NotImplementedError('ActingWorker: the one-launch acting step is built for the fused actor shape (depth 2, ReLU, hidden <= 256, action_size <= 8); a general-shape actor acts through '
                                'actor(state).sample() (csrc/general.hip)')
    self.actor, self.memory = actor, memory
    self.S, self.A = memory.state_size, memory.action_size
    self._act_box, self._append_box = _Mailbox(self.S, self.A), _Mailbox(self.S, self.A)
    dev = memory.ring.device
    self.carry = torch.zeros(self.S + self.A + 4, dtype=torch.float32, device=dev)
    self._seq = 0   # one sequence for both mailboxes: the device de-duplicates appends by commit word
    self._seed = C.c_uint64(torch.initial_seed() & (2**64 - 1))
    self._fixed = {}
    self._pending_seq = None
    self.mirror = None
    if mirror:
      self._stride = (actor.flat.numel() + 63) // 64 * 64
      self.mirror = torch.zeros(3, self._stride, dtype=torch.float32, device=dev)
      self._version = torch.zeros(2, dtype=torch.int32, device=dev)   # {snapshot version, completion counter of k_act_publish}
      self.act_stream = torch.cuda.Stream(device=dev, priority=-1)     # short, latency-critical launches next to the update graph
      self.enqueue_publish()
      torch.cuda.current_stream().synchronize()

  def _launch(self, box: _Mailbox, acts: bool = True, stream=None, snapshot: bool = False):
    a = self.actor
    a._act_calls += int(acts)  # the Philox offset is shared with SoftActor._act, so the two entry points never reuse noise
    key = (id(box), snapshot)
    fixed = self._fixed.get(key)
    if fixed is None or fixed[0] != a.flat.data_ptr():  # pointers are stable for the life of the worker; re-derive if the arena was re-homed
      params = self.mirror if snapshot else a.flat
      fixed = self._fixed[key] = (a.flat.data_ptr(), None, _lib.ptr(params), C.c_void_p(box.tensor.data_ptr()), _lib.ptr(self.carry),
                                  _lib.ptr(self.memory.ring), _lib.ptr(self.memory._ring_state), _lib.ptr(self._version) if snapshot else None,
                                  self._stride if snapshot else 0)
    _, _, p_actor, p_box, p_carry, p_ring, p_state, p_version, stride = fixed
    fn = _lib.lib().il_act_step   # (looked up per call: UpdatePlan.record_direct walks the hooks with a recording stand-in for the library)
    st = (stream or torch.cuda.current_stream()).cuda_stream
    rc = fn(p_actor, self.S, self.A, a.hidden, p_box, p_carry, p_ring, p_state, self._seed, a._act_calls & 0xFFFFFFFF, p_version, stride, st)
    if rc: _lib.check(rc)

  def _next_seq(self) -> int:
    self._seq = self._seq % (_SEQ_MOD - 1) + 1   # 1 .. 2^17-1: never 0, so a zeroed carry matches nothing
    return self._seq

  def _collect(self, box: _Mailbox, seq: float) -> torch.Tensor:
    box.wait(seq, 'il_act_step')
    return torch.from_numpy(box.host[box.o_act:box.o_act + self.A].copy()).unsqueeze(0)

  def _mirror_append(self, terminal: bool, timeout: bool, wrap: bool):
    m = self.memory
    m._advance(terminal, timeout)
    if wrap:
      m._advance(False, False)

  # --- exact schedule (and the act half of the overlap schedule)
  def act(self, obs, greedy: bool = False) -> torch.Tensor:
    """`actor(obs).sample()` (or the greedy action) as a [1, A] CPU tensor. Without a mirror: on the current stream with the live
    parameters, remembering (obs, action) on the device for `append`. With a mirror: on the worker's own stream from the latest snapshot."""
    box = self._act_box
    seq = box.post(self._next_seq(), GREEDY if greedy else 0, obs=_row(obs))
    if self.mirror is None:
      self._launch(box)
    else:
      self._launch(box, stream=self.act_stream, snapshot=True)
    return self._collect(box, seq)

  def act_begin(self, obs, greedy: bool = False):
    """The launch half of `act` (round 6): post the observation and enqueue the act launch, return at once; `act_end()` collects the action. With a mirror the launch runs
    on the worker's own stream from the latest snapshot, so a caller can put its update's host work (a graph replay or the direct launches: ~30 us) between the two calls -
    the act launch's ~29 us turn-around (dispatch, one workgroup through the actor, the echo into pinned memory) then hides behind it instead of following it
    (profiles/r06_acting_host_profile.json: 96 -> us per environment step with one update per step)."""
    box = self._act_box
    self._pending_seq = box.post(self._next_seq(), GREEDY if greedy else 0, obs=_row(obs))
    if self.mirror is None:
      self._launch(box)
    else:
      self._launch(box, stream=self.act_stream, snapshot=True)

  def act_end(self) -> torch.Tensor:
    seq, self._pending_seq = self._pending_seq, None
    assert seq is not None, 'act_end() without act_begin()'
    return self._collect(self._act_box, seq)

  def append(self, step, next_obs, reward, terminal: bool, timeout: bool):
    """`memory.append(step, state, action, reward, next_state, terminal, timeout)` for the (state, action) of the last `act`, plus the
    absorbing wrap when the episode ended by true termination (train.py:157,161). Asynchronous: nothing is waited for."""
    assert self.mirror is None, 'with a mirror the act launches run ahead of the appends: use post() + enqueue_append()'
    wrap = bool(self.memory.absorbing and terminal and not timeout)
    box = self._append_box
    if box.word: box.wait(box.word, 'il_act_step(append)')  # normally already echoed: the act in between ran after it on the same stream
    box.post(self._next_seq(), PENDING | NO_ACTION | (WRAP_ABSORBING if wrap else 0), float(reward), float(terminal), float(timeout), float(step), next_obs=_row(next_obs))
    self._launch(box, acts=False)
    self._mirror_append(bool(terminal), bool(timeout), wrap)

  # --- fused schedule
  def step(self, step, next_obs, reward, terminal: bool, timeout: bool, obs=None, greedy: bool = False) -> torch.Tensor:
    """append(transition of the last action) + act(obs) in ONE launch. `obs` defaults to `next_obs`; pass the reset observation when
    the episode ended."""
    assert self.mirror is None
    wrap = bool(self.memory.absorbing and terminal and not timeout)
    box = self._act_box
    nxt = _row(next_obs)
    seq = box.post(self._next_seq(), PENDING | (WRAP_ABSORBING if wrap else 0) | (GREEDY if greedy else 0), float(reward), float(terminal), float(timeout), float(step), next_obs=nxt,
                   obs=nxt if obs is None else _row(obs))
    self._launch(box)
    self._mirror_append(bool(terminal), bool(timeout), wrap)
    return self._collect(box, seq)

  # --- overlap schedule: the append and the publish ride in the update stream (UpdatePlan.pre_hooks / post_hooks), the act runs beside it
  def post(self, step, obs, action, next_obs, reward, terminal: bool, timeout: bool):
    """Host side of an append: fill the append mailbox with the whole transition. The next `enqueue_append` launch (direct, or the one
    captured in an update graph) consumes it exactly once. Blocks only if the previous post has not been consumed yet (back-pressure:
    the host can run at most one update ahead of the GPU)."""
    box = self._append_box
    if box.word: box.wait(box.word, 'il_act_step(append)', timeout_s=30.0)
    wrap = bool(self.memory.absorbing and terminal and not timeout)
    box.post(self._next_seq(), PENDING | NO_ACTION | CARRY_FROM_MAILBOX | (WRAP_ABSORBING if wrap else 0), float(reward), float(terminal), float(timeout), float(step), next_obs=_row(next_obs),
             obs=_row(obs), action=_row(action))
    self._mirror_append(bool(terminal), bool(timeout), wrap)

  def enqueue_append(self):
    """Launch the append kernel on the current stream (capturable: every argument is a fixed pointer; what to append is read from the mailbox)."""
    self._launch(self._append_box, acts=False)

  def enqueue_publish(self):
    """Snapshot the actor arena for the act stream; enqueue after anything that changes the actor (capturable)."""
    a = self.actor
    _lib.check(_lib.lib().il_act_publish(_lib.ptr(a.flat), a.flat.numel(), _lib.ptr(self.mirror), self._stride, _lib.ptr(self._version), _lib.stream_ptr()))

  enqueue_append._il_recordable = True    # UpdatePlan.record_direct: library calls only, every argument a fixed pointer (what to append / publish is read on the device)
  enqueue_publish._il_recordable = True

  def attach(self, plan):
    """Make `plan` (UpdatePlan) carry this worker's append before, and its parameter snapshot after, every update. Attach before capture."""
    assert self.mirror is not None and plan.graph is None
    plan.pre_hooks.append(self.enqueue_append)
    plan.post_hooks.append(self.enqueue_publish)
    return self
