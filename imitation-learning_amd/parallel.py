"""Data-parallel SAC/GAIL over RCCL (one process per GPU, torch.distributed backend "nccl" == RCCL on ROCm).

The path shards as plain data parallelism: every loss is a mean over samples, so the average of per-rank
gradients on rank-local batches equals the gradient on the concatenated batch.  Each rank owns an env worker + agent replay
shard (never exchanged) and a full replica of every network; there are exactly three exchange steps per update, in
dependency order, each ONE flat pre-allocated fp32 bucket:
    discriminator grad [P_d]  ->  critic grad [2*Ps]  ->  actor grad + log_alpha grad [Pa + 1 (+pad)]
Messages are 6.7 KB / 580 KB / 295 KB: latency-bound on xGMI, so one bucket per sync point, no splitting.
Replicas stay bit-identical because every rank applies the same averaged gradient with the same kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib


def all_reduce_mean_(bucket: torch.Tensor, group=None):
  """In-place mean over ranks. NCCL/RCCL has a native AVG; gloo (CPU tests) sums then scales."""
  if not dist.is_initialized() or (dist.get_world_size(group) == 1 and os.environ.get('IL_FORCE_ALLREDUCE') != '1'):
    return bucket  # IL_FORCE_ALLREDUCE=1 issues the collective even on one rank (exercises the RCCL + graph-capture path on a 1-GPU box)
  if dist.get_backend(group) == 'nccl':
    dist.all_reduce(bucket, op=dist.ReduceOp.AVG, group=group)
  else:
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    bucket.div_(dist.get_world_size(group))
  return bucket


def broadcast_parameters(tensors, src: int = 0, group=None):
  """One-time replica sync of flat arenas (params, SN buffers, log_alpha) from rank `src`."""
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    for t in tensors:
      dist.broadcast(t, src=src, group=group)


def rank_seed(base_seed: int) -> int:
  """Rank-offset seed for index draws / sampling noise (each rank must see different data)."""
  return base_seed + (dist.get_rank() if dist.is_initialized() else 0)


class GradBuckets:
  """The exchange buffers of one learner: ONE flat fp32 tensor per sync point.
      disc   [P_d]              discriminator gradient (GAIL only)
      critic [2 * Ps]           twin-critic gradient arena (net stride padding included: the pad floats stay zero on every rank)
      actor  [roundup4(Pa + 1)] actor gradient followed by the log-alpha gradient in slot Pa, zero-padded to a 16-byte multiple
  `actor_grad` / `alpha_grad` are VIEWS into the actor bucket: the grads-only kernels write straight into the message, nothing is packed or copied
  around a collective. Device-agnostic (the world_size-2 gloo test drives it with CPU tensors of the real sizes)."""

  def __init__(self, n_actor: int, critic_grad: torch.Tensor, disc_grad: Optional[torch.Tensor] = None, device=None):
    device = device if device is not None else critic_grad.device
    self.n_actor = int(n_actor)
    self.actor = torch.zeros((self.n_actor + 1 + 3) // 4 * 4, dtype=torch.float32, device=device)
    self.actor_grad, self.alpha_grad = self.actor[:self.n_actor], self.actor[self.n_actor:self.n_actor + 1]
    self.critic, self.disc = critic_grad, disc_grad

  def exchange(self, which: str, group=None) -> torch.Tensor:
    """In-place mean over ranks of one bucket ('disc' | 'critic' | 'actor')."""
    return all_reduce_mean_(getattr(self, which), group)


def replica_tensors(actor, critic, target_critic, log_alpha, discriminator=None):
  """Every tensor that must be bit-identical on all ranks before the first update: parameter arenas, log alpha, and the discriminator's parameters AND buffers
  (spectral-norm u / v: they depend only on the replicated weights afterwards, but their random initial values differ per rank)."""
  out = [actor.flat, critic.flat, target_critic.flat, log_alpha]
  if discriminator is not None and hasattr(discriminator, 'flat'):
    out.append(discriminator.flat)
    for name in ('sn', 'target_flat'):
      if getattr(discriminator, name, None) is not None: out.append(getattr(discriminator, name))
  return out


def broadcast_scalars(values, src: int = 0, group=None):
  """Host scalars that replicas must agree on (GMMIL bandwidths, RED sigma, DRIL threshold): rank `src`'s values everywhere."""
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    box = [list(values)]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]
  return list(values)


def init_from_env(world_size: int, backend: str = 'nccl'):
  """Joins the process group torch.distributed.run (or the bench driver) prepared: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, one process per
  GPU. Returns (rank, local_rank, device). backend 'nccl' is RCCL on ROCm; 'gloo' exists for tests (two ranks may then share one GPU)."""
  rank, local, world = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
  if world != world_size:
    raise RuntimeError(f'distributed.world_size={world_size} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nproc-per-node {world_size} --master-addr 127.0.0.1 train.py ...')
  n_dev = torch.cuda.device_count()
  device = torch.device('cuda', local if backend == 'nccl' else local % max(n_dev, 1))
  torch.cuda.set_device(device)
  if not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
    if backend == 'nccl': dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else: dist.init_process_group(backend, rank=rank, world_size=world)
  return rank, local, device


def _agree(ok: bool, group=None) -> bool:
  """Collective AND of a per-rank flag (every rank must take the same branch afterwards)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return bool(ok)
  t = torch.tensor([1.0 if ok else 0.0], device='cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
  dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
  return bool(t.item() > 0.5)


def agree_min(value: float, group=None) -> float:
  """Collective MIN of a per-rank number (host value; every rank gets the same result)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return float(value)
  t = torch.tensor([float(value)], dtype=torch.float64, device='cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
  dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
  return float(t.item())


def replica_digest(tensors) -> str:
  """sha256 over the bytes of every replica tensor (parameter / Adam arenas, log alpha, spectral-norm buffers) in the given order. A few MB copied to the host: an
  end-of-run / evaluation-time check, never inside the timed region."""
  import hashlib
  h = hashlib.sha256()
  for t in tensors:
    if t is None: continue
    h.update(t.detach().contiguous().cpu().numpy().tobytes())
  return h.hexdigest()


def replicas_bit_identical(tensors, group=None):
  """Collective. (True | False, [digest of every rank]): do all ranks hold the same bits? Data-parallel replicas apply the same averaged gradient with the same kernels,
  so anything but identical digests means an exchange delivered different (stale, torn) gradients to different ranks."""
  mine = replica_digest(tensors)
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return True, [mine]
  words = torch.tensor([int(mine[i:i + 15], 16) for i in range(0, 60, 15)], dtype=torch.int64)   # 240 of the 256 bits, as four int64: travels through any backend as a plain tensor
  dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
  words = words.to(dev)
  every = [torch.empty_like(words) for _ in range(dist.get_world_size(group))]
  dist.all_gather(every, words, group=group)
  digests = [''.join(f'{int(v):015x}' for v in w.cpu().tolist()) for w in every]
  return all(d == digests[0] for d in digests), digests


class Watchdog:
  """A rank that dies (or stalls) inside a collective must not hang the others past `timeout_s`: RCCL's kernels spin on the device for a peer that never arrives and the
  host then sits in a synchronise for ever. A daemon thread watches a heartbeat the main loop refreshes (`beat()`, a host-side timestamp store); when it goes stale the
  thread says which phase was running and ends THIS process with `os._exit` (exit code 124) - the launcher (torch.distributed.run, or bench.py's own) then tears the
  remaining ranks down. IL_WATCHDOG_S overrides the bound; 0 disables."""
  EXIT_CODE = 124

  def __init__(self, timeout_s: float = 300.0, what: str = 'run', armed: bool = True):
    """armed=False: the heartbeat is only watched from `arm()` on - a training run arms it at its first captured update, so that the single-rank phases before it whose
    length the bound knows nothing about (expert-data load, 100k pretraining iterations, PWIL relabelling, peer-window set-up and soak, graph capture) cannot trip it."""
    import threading
    import time
    self.timeout_s = float(os.environ.get('IL_WATCHDOG_S', timeout_s))
    self.armed = bool(armed)
    self.what, self.phase, self._time, self._last, self._stop = what, 'start', time, time.monotonic(), threading.Event()
    self._thread = None
    if self.timeout_s > 0:
      self._thread = threading.Thread(target=self._watch, name='il-watchdog', daemon=True)
      self._thread.start()

  def beat(self, phase: Optional[str] = None):
    self._last = max(self._last, self._time.monotonic())   # (never shortens a grace period)
    if phase is not None: self.phase = phase

  def arm(self, phase: Optional[str] = None):
    self._last, self.armed = self._time.monotonic(), True
    if phase is not None: self.phase = phase

  def grace(self, seconds: float, phase: Optional[str] = None):
    """A phase every rank knows to be long (rank 0's evaluation episodes, during which its peers wait inside the next collective): no alarm for `seconds` + the bound."""
    self._last = max(self._last, self._time.monotonic() + float(seconds))
    if phase is not None: self.phase = phase

  def stop(self):
    self._stop.set()

  def _watch(self):
    import sys
    while not self._stop.wait(min(1.0, self.timeout_s / 4)):
      idle = self._time.monotonic() - self._last
      if self.armed and idle > self.timeout_s:
        rank = os.environ.get('RANK', '0')
        print(f'[watchdog] rank {rank}: no progress for {idle:.0f} s in phase "{self.phase}" of {self.what} (bound {self.timeout_s:.0f} s): a peer rank died or stalled inside a '
              f'collective / device-side wait. Ending this rank (exit {self.EXIT_CODE}) so that the job fails instead of hanging.', file=sys.stderr, flush=True)
        os._exit(self.EXIT_CODE)


class PeerExchange:
  """The three gradient exchanges as ONE kernel each over peer-mapped windows (include/il_hip.h il_peer_*, csrc/peer.hip) instead of RCCL all-reduces.

  xGMI is a point-to-point mesh and the messages are 6.7 / 580 / 295 KB, i.e. latency-bound: every rank stores its bucket straight into slot [rank] of every rank's
  receive window (one fabric crossing, each peer over its own link), waits on the device for the others' arrival words and sums the slabs in rank order - the result is
  the same bits on every rank, so replicas stay identical exactly as with an all-reduce. Set-up is collective: one uncached / fine-grained window per rank holding a region
  per bucket, hipIpc handles all-gathered through the host-side process group, every rank maps the others' windows. `create()` returns None - on EVERY rank - when any rank
  cannot set it up or the self-test (known patterns through the real kernel, several epochs) fails anywhere; the caller then keeps the collectives."""

  def __init__(self, sizes: dict, device, group=None, spin_limit: int = 0, jobs: Optional[dict] = None):
    L = _lib.lib()
    self.group, self.device = group, device
    self.jobs = {k: int(v) for k, v in (jobs or {}).items() if v}   # buckets whose exchange rides in the kernel producing the gradients: arrival lines per producing workgroup
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1
    self.rank = dist.get_rank(group) if dist.is_initialized() else 0
    if self.world > _lib.IL_PEER_MAX_RANKS:
      raise RuntimeError(f'PeerExchange: {self.world} ranks > IL_PEER_MAX_RANKS = {_lib.IL_PEER_MAX_RANKS}')
    self.sizes = {k: int(n) for k, n in sizes.items() if n}
    # '_load': a bucket nobody trains on - the soak test (below) exchanges it on a third stream so that the real buckets are verified while the links carry traffic
    load = int(os.environ.get('IL_PEER_SOAK_LOAD_FLOATS', 1 << 20)) if self.world > 1 or os.environ.get('IL_PEER_SOAK_LOAD_FLOATS') else 0
    if load > 0: self.sizes['_load'] = load
    self.soak_report = None
    self.window, self.opened, self.desc, self._keep, self.uncached, self.form = None, [], {}, [], False, 0
    offsets, total = {}, 0
    for k, n in self.sizes.items():
      offsets[k] = total
      total += int(L.il_peer_job_region_bytes(self.world, n, self.jobs[k]) if k in self.jobs else L.il_peer_region_bytes(self.world, n))
    w, handle = C.c_void_p(), C.create_string_buffer(_lib.IL_PEER_HANDLE_BYTES)
    with torch.cuda.device(device):
      # Every rank takes part in the handle exchange whether or not its own allocation worked (a rank that raised before the collective would leave the others waiting
      # in it): a failed allocation travels as None and fails the set-up on every rank.
      kind = C.c_int32(0)
      rc = L.il_peer_window_alloc(total, C.byref(w), handle, C.byref(kind))
      mine = (handle.raw, int(kind.value)) if rc == 0 else None   # kind: 0 uncached, 1 fine-grained
      why = None if rc == 0 else L.il_last_error().decode()
      self.window = w.value if rc == 0 else None
      handles = [mine]
      if self.world > 1:
        handles = [None] * self.world
        dist.all_gather_object(handles, mine, group=group)
      try:
        if any(h is None for h in handles):
          raise RuntimeError('PeerExchange: no peer window on rank(s) ' + ', '.join(str(r) for r, h in enumerate(handles) if h is None) + (f' ({why})' if why else ''))
        self.uncached = all(h[1] == 0 for h in handles)   # the write-through form needs every window uncached
        windows = []
        for r, (h, _) in enumerate(handles):
          if r == self.rank:
            windows.append(self.window)
            continue
          o = C.c_void_p()
          _lib.check(L.il_peer_window_open(h, C.byref(o)))
          self.opened.append(o.value)
          windows.append(o.value)
      except Exception:
        self.close(collective=False)   # leave nothing mapped or allocated behind; create() turns this into a collective fall-back
        raise
    self.status = torch.zeros(2, dtype=torch.int64, device=device)
    for k, n in self.sizes.items():
      epoch = torch.zeros(self.jobs.get(k) or (n + _lib.IL_PEER_CHUNK_FLOATS - 1) // _lib.IL_PEER_CHUNK_FLOATS, dtype=torch.int32, device=device)
      d = _lib.PeerBucket(rank=self.rank, world=self.world, n=n, window_offset=offsets[k], epoch=epoch.data_ptr(), status=self.status.data_ptr(), spin_limit=int(spin_limit),
                          n_jobs=self.jobs.get(k, 0))
      for r, wp in enumerate(windows): d.windows[r] = wp
      self.desc[k] = d
      self._keep.append(epoch)
    torch.cuda.synchronize(device)

  @classmethod
  def create(cls, sizes: dict, device, group=None, verify_rounds: int = 3, soak_rounds: Optional[int] = None, jobs: Optional[dict] = None):
    """Collective. A PeerExchange that passed its self-test AND its soak test (`soak`) on every rank, or None on every rank. `jobs`: buckets laid out for the exchange inside
    the producing kernels (name -> arrival lines); the self-test and the soak drive them through the same device functions (k_peer_job_allreduce)."""
    x, err = None, None
    try:
      x = cls(sizes, device, group, jobs=jobs)
    except Exception as e:   # no IPC between these processes, no fine-grained memory, ...: fall back together
      err = e
    if dist.is_initialized() and dist.get_world_size(group) > 1:
      dist.barrier(group)   # every window is mapped everywhere before the first store into a peer
    ok = _agree(x is not None, group)
    if ok:
      # Two forms of the same kernel, tried in this order, each adopted only if the self-test passes on EVERY rank: write-through (payload through sc0 sc1 accesses +
      # drained stores, no fences: uncached windows only; IL_PEER_WRITE_THROUGH=0 skips it), then the compiler's system-scope release / acquire fences.
      forms = ([_lib.IL_PEER_WRITE_THROUGH] if _agree(x.uncached and os.environ.get('IL_PEER_WRITE_THROUGH', '1') != '0', group) else []) + [0]
      for form in forms:
        x.set_form(form)
        try:
          good = x.verify(verify_rounds) and x.soak(soak_rounds)
        except Exception as e:   # a HIP error in the self-test on this rank: still take part in the agreement
          good, err = False, e
        ok = _agree(good, group)
        if ok: break
        x.status.zero_()
    if not ok:
      if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group)   # EVERY rank, with or without an exchange of its own: no kernel of the self-test is still storing into a window that is about to go away
      if x is not None: x.close(collective=False)
      if err is not None and os.environ.get('IL_PEER_EXCHANGE') == 'require': raise err
      return None
    return x

  def set_form(self, flags: int):
    """IL_PEER_WRITE_THROUGH or 0 (fences) for every bucket; every rank must use the same form."""
    self.form = int(flags)
    for d in self.desc.values():
      d.flags = self.form

  def allreduce_mean(self, name: str, bucket: torch.Tensor):
    """bucket <- mean over ranks, in place, enqueued on the current stream (capturable). Every rank issues the same sequence of calls per name."""
    d = self.desc[name]
    assert bucket.numel() == d.n and bucket.dtype == torch.float32 and bucket.is_contiguous()
    _lib.check(_lib.lib().il_peer_allreduce_mean(C.byref(d), _lib.ptr(bucket), _lib.stream_ptr()))
    return bucket

  def timeouts(self) -> int:
    """Device-side waits for a peer that gave up (must stay 0; the affected update averaged stale slabs)."""
    return int(self.status[0].item())

  def verify(self, rounds: int = 3) -> bool:
    """Known rank- and round-dependent patterns through the real kernel; the mean is exact in fp32, so the comparison is bitwise. A stale line, a lost store or a
    window mapped to the wrong rank shows up as a mismatch; several rounds exercise both slot parities and the epoch logic."""
    ok = True
    # The ranks have just met at a barrier: a wait of the self-test that is going to be satisfied is satisfied within milliseconds. A shorter bound (a few seconds instead of
    # tens) keeps a set-up that does NOT work - stores that never become visible - from stalling the start of a run for minutes before it falls back.
    keep = {k: d.spin_limit for k, d in self.desc.items()}
    for d in self.desc.values():
      d.spin_limit = 1 << 21
    try:
      ok = self._verify_rounds(rounds)
    finally:
      for k, d in self.desc.items():
        d.spin_limit = keep[k]
    torch.cuda.synchronize(self.device)
    return ok and self.timeouts() == 0

  def _verify_rounds(self, rounds: int) -> bool:
    ok = True
    for k, n in self.sizes.items():
      i = torch.arange(n, device=self.device, dtype=torch.float32)
      for r in range(rounds):
        scratch = (i % 97) * 0.25 + float((self.rank + 1) * (r + 1))
        self.allreduce_mean(k, scratch)
        expect = (i % 97) * 0.25 + float((self.world + 1) * (r + 1)) / 2.0
        ok = ok and bool(torch.equal(scratch, expect))
    return ok

  def soak(self, rounds: Optional[int] = None) -> bool:
    """The set-up check a fabric deserves before gradients travel over it: `rounds` (IL_PEER_SOAK_ROUNDS, default 2,000) exchanges of EVERY bucket, interleaved the way the
    update issues them - the discriminator's on one stream, the critic's and the actor's on another, nothing ordering the two - while a third stream keeps exchanging the
    4 MB '_load' bucket so that the links are busy; the payload is rewritten on the device before every exchange with a pattern that depends on rank, round and position
    (rotating by one element per round, so a slab that is stale by two rounds - the same slot parity - or misplaced by a lane differs everywhere), and every result is
    compared BITWISE on the device (the pattern's sum over ranks is exact in fp32 and a multiple of W, so sum / W is exact for any W). No host synchronisation inside the loop.
    One mismatching element or one expired wait on any rank fails the soak (the caller, `create`, makes that decision collective): a flag overtaking its payload under load,
    a lost or torn store, a window mapped to the wrong rank all show up here rather than as silently averaged stale gradients. rounds <= 0 skips it."""
    rounds = int(os.environ.get('IL_PEER_SOAK_ROUNDS', 2000)) if rounds is None else int(rounds)
    if rounds <= 0 or not self.desc:
      return True
    import time
    dev, W, me = self.device, self.world, self.rank
    names = list(self.sizes)
    streams = {'side': torch.cuda.Stream(dev), 'main': torch.cuda.Stream(dev), 'load': torch.cuda.Stream(dev)}
    lane = {k: ('load' if k == '_load' else ('side' if k == 'disc' or (i % 2 == 0 and 'disc' not in self.sizes) else 'main')) for i, k in enumerate(names)}
    pos = {k: torch.arange(n, device=dev, dtype=torch.int32) for k, n in self.sizes.items()}
    bad = {k: torch.zeros((), dtype=torch.int64, device=dev) for k in names}
    keep = {k: d.spin_limit for k, d in self.desc.items()}
    for d in self.desc.values():
      d.spin_limit = 1 << 22   # the ranks met at a barrier moments ago; a wait that is going to be satisfied is satisfied within milliseconds
    before = self.timeouts()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    try:
      here = torch.cuda.current_stream(dev)
      for s in streams.values(): s.wait_stream(here)
      for r in range(rounds):
        k_r = float(r % 251 + 1)
        for k in names:
          with torch.cuda.stream(streams[lane[k]]):
            base = ((pos[k] + r) % 97).to(torch.float32) * 0.25
            buf = base + float(me + 1) * k_r                      # this rank's "gradient" of round r
            self.allreduce_mean(k, buf)
            bad[k] += (buf != base + k_r * (W + 1) / 2.0).sum()   # mean over ranks q of base + (q + 1) k_r, exact
      for s in streams.values(): here.wait_stream(s)
      torch.cuda.synchronize(dev)
    finally:
      for k, d in self.desc.items():
        d.spin_limit = keep[k]
    mismatches = {k: int(v.item()) for k, v in bad.items()}
    expired = self.timeouts() - before
    self.soak_report = dict(rounds=rounds, buckets={k: self.sizes[k] for k in names}, mismatching_elements=mismatches, expired_waits=expired,
                            seconds=round(time.perf_counter() - t0, 3), form='write-through' if self.form else 'fences')
    return expired == 0 and not any(mismatches.values())

  def close(self, collective: bool = True):
    """collective=True: every rank of the group calls close() together (a barrier makes sure nobody unmaps a window a peer's kernel may still store into)."""
    L = _lib.lib()
    torch.cuda.synchronize(self.device)
    if collective and dist.is_initialized() and dist.get_world_size(self.group) > 1:
      dist.barrier(self.group)
    for o in self.opened: L.il_peer_window_close(C.c_void_p(o))
    if self.window: L.il_peer_window_free(C.c_void_p(self.window))
    self.opened, self.window, self.desc = [], None, {}


class DataParallelUpdate:
  """`UpdatePlan` with the three gradient all-reduces between backward and optimiser (IL_FLAG_GRADS_ONLY entry points)."""

  def __init__(self, plan, group=None):
    self.plan, self.group = plan, group
    plan.data_parallel = True   # (the single-GPU extras that assume one branch pair per update - staged rows - stay off)
    # GAIL with a finite PUGAIL margin / subtract_log_policy / reward shaping / a depth-2 or tanh discriminator / Mixup with mixup_alpha != 1 (round 6): these plans run the
    # per-function entry points of their discriminator on the gathered rows, on stream dependencies; every one of those entry points takes IL_FLAG_GRADS_ONLY, so their
    # data-parallel form is one more bucket - the discriminator optimiser's gradient arena - averaged between the gradient launch and the AdamW step (run()).
    self.variant = bool(getattr(plan, '_variant', False) or getattr(plan, '_beta_alpha', None) is not None)
    if getattr(plan, 'general', False):
      raise NotImplementedError('DataParallelUpdate: actor / critic shapes outside depth 2 / ReLU / hidden <= 256 / action_size <= 8 (csrc/general.hip) have no gradient-only kernels '
                                'to all-reduce; run these configurations with distributed.world_size=1')
    # Device-side hand-off between the discriminator branch and the SAC branch, as on one GPU (UpdatePlan): no stream dependency between the two streams, the index
    # draw resident at the head of the discriminator branch, the reward relabel inline in the forward / critic-loss launch. The discriminator's all-reduce then sits
    # on a stream of its own and must not share a communicator with the critic / actor all-reduces of the main stream (two unordered streams could issue the
    # collectives of ONE communicator in different orders on different ranks): it gets its own process group. IL_DP_HANDOFF=0: stream dependencies, one communicator.
    # EVERY per-rank input of this decision (the stream / hardware-queue probe of UpdatePlan.__init__ is one) goes through a collective AND first: a rank that chose the
    # other schedule would create no side group and issue the discriminator's all-reduce on another communicator than its peers - a hang, not an error.
    mine = bool(plan.algorithm == 'GAIL' and plan.device_sync and plan.ring_mode and plan.inline_relabel and not plan.bc_aux and os.environ.get('IL_DP_HANDOFF', '1') != '0')
    self.handoff = _agree(mine, group)
    if not self.handoff:
      plan._set_device_sync(False)   # this schedule orders its two streams with events around the all-reduces
    self.side_group = group
    if plan.algorithm == 'GAIL' and dist.is_initialized():
      # unconditionally, on every rank (new_group is collective over the whole default group), whichever schedule was agreed on: it is only USED by the hand-off schedule
      self.side_group = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group), backend=dist.get_backend(group))
    self.graph = self.graph_side = None
    self._warm_collectives_pending = True
    # The exchanges INSIDE the optimiser launches (il_sac_update_gather_peer, il_gail_disc_step_draw_peer): the update is then the launch sequence of ONE GPU - two
    # launches on the discriminator's stream, four on the main one - instead of three + eight. Needs the hand-off schedule with the resident sampler, the block form of
    # the optimiser launches for this shape, and the peer windows (decided collectively in _setup_peer_exchange). IL_DP_FUSED=0: the exchange launches.
    L = _lib.lib()
    self._fused_jobs = dict(disc=int(L.il_gail_step_workgroups(C.byref(plan.disc))), critic=int(L.il_sac_peer_jobs(C.byref(plan.sac), 0)), actor=int(L.il_sac_peer_jobs(C.byref(plan.sac), 1))) \
        if plan.algorithm == 'GAIL' and plan.disc is not None else {}
    mine = bool(self.handoff and plan.resident_sampler and self._fused_jobs and all(self._fused_jobs.values()) and os.environ.get('IL_DP_FUSED', '1') != '0')
    self.fused = _agree(mine, group) if self.handoff else False
    self.peer = None   # PeerExchange once the first run() has set it up (collective); None = torch.distributed all-reduces
    self.peer_note = None   # why the peer-window exchange is not in use, when it was asked for
    ao, to = plan._keep[4], plan._keep[6]
    # actor grad and alpha grad travel in one bucket: the optimisers' gradient arenas become views into it
    self.buckets = GradBuckets(ao.grad.numel(), plan._keep[5].grad, plan._keep[8].grad if plan.algorithm == 'GAIL' else None)
    ao.grad, to.grad = self.buckets.actor_grad, self.buckets.alpha_grad
    plan.sac.actor_grad, plan.sac.alpha_grad = ao.grad.data_ptr(), to.grad.data_ptr()
    # The discriminator branch runs on the plan's own second stream: the one UpdatePlan's probes validated against the caller's stream (a fresh stream may be multiplexed
    # onto the caller's hardware queue, and the two branches would then serialise)
    self.side = (plan.side if plan.side is not None else torch.cuda.Stream()) if plan.algorithm == 'GAIL' else None
    self.actor_bucket, self.critic_bucket, self.disc_bucket = self.buckets.actor, self.buckets.critic, self.buckets.disc
    if dist.is_initialized() and dist.get_world_size(group) > 1 and plan.device_sync:
      # a rank waits for the all-reduced discriminator step INSIDE its SAC branch: that bound must not be shorter than the exchange's own (IL_PEER_SPIN_LIMIT), or ordinary
      # inter-rank skew (an environment reset, a GC pause) expires the inner wait first and the update trains on stale rewards
      plan.widen_handoff_bound()

  def _warm_collectives(self):
    """First use of a communicator sets up its connections (host-blocking, up to seconds) and ranks reach their first update seconds apart: neither may happen inside an
    update of the hand-off schedule, whose device-side waits are bounded. One all-reduce per (bucket, communicator) - the gradient arenas still hold zeros - then a
    barrier: afterwards the ranks are aligned and every later collective is a steady-state one."""
    self._warm_collectives_pending = False
    self._setup_peer_exchange()
    if not dist.is_initialized():
      return
    for bucket, group in ((self.disc_bucket, self.side_group), (self.critic_bucket, self.group), (self.actor_bucket, self.group)):
      if bucket is not None:
        all_reduce_mean_(bucket, group)
    torch.cuda.synchronize()
    if dist.get_world_size(self.group) > 1:
      dist.barrier(self.group)

  def _setup_peer_exchange(self):
    """IL_PEER_EXCHANGE: unset / '0' = torch.distributed all-reduces (RCCL: the exchange BASELINE.json's north_star names, and the default of every run); '1' = the
    one-kernel exchange over peer-mapped windows whenever more than one rank takes part (falls back, on every rank, if set-up or the self-test fails anywhere) - opt-in
    until a multi-GPU run has shown it to win (bench.py --gpus N times both in one job and reports the better VALID one); 'force' = also with a single rank (exercises the
    kernel on a 1-GPU box); 'require' = raise instead of falling back."""
    mode = os.environ.get('IL_PEER_EXCHANGE', '0')
    world = dist.get_world_size(self.group) if dist.is_initialized() else 1
    if mode == '0' or (world == 1 and mode not in ('force', 'require')):
      self.fused = False   # (the fused form lives on the peer windows)
      return
    sizes = dict(disc=self.disc_bucket.numel() if self.disc_bucket is not None else 0, critic=self.critic_bucket.numel(), actor=self.actor_bucket.numel())
    jobs = None
    if self.fused:   # the same three buckets laid out by PARAMETER offset with one arrival line per producing workgroup (the actor's carries log alpha's gradient in its last lane)
      L = _lib.lib()
      sizes = dict(disc=sizes['disc'], critic=int(L.il_sac_peer_bucket_floats(C.byref(self.plan.sac), 0)), actor=int(L.il_sac_peer_bucket_floats(C.byref(self.plan.sac), 1)))
      jobs = dict(self._fused_jobs)
    self.peer = PeerExchange.create(sizes, self.plan.rows.device, self.group, jobs=jobs)
    if self.peer is None:
      self.peer_note = 'set-up, self-test or soak test of the peer-window exchange failed on some rank: torch.distributed all-reduces'
    self.fused = bool(self.fused and self.peer is not None)
    self.plan.peer_desc = self.peer.desc if self.fused else None
    if self.peer is None and mode == 'require':
      raise RuntimeError('IL_PEER_EXCHANGE=require: the peer-window exchange could not be set up or failed its self-test on some rank')

  def _exchange(self, name: str, group=None):
    """One sync point: mean over ranks of bucket `name`, in place, on the current stream."""
    bucket = getattr(self, name + '_bucket')
    if self.peer is not None:
      return self.peer.allreduce_mean(name, bucket)
    return all_reduce_mean_(bucket, group)

  def _apply_phase(self, phase: int, group=None):
    """Sync point + il_sac_dp_phase(2 | 3): the exchange (il_peer_allreduce_mean as a launch of its own, or the all-reduce), then the plain phase. IL_PEER_APPLY=1: with the peer
    windows the exchange rides in the phase's apply launch instead (il_sac_dp_phase_peer: workgroup c exchanges chunk c and steps its parameters - same bits, one launch and one
    arena pass less; measured with one rank: 5 us less kernel time per update but 96.8 against 95.0 us per update, the 71-workgroup AdamW pass has a longer tail than the
    567-workgroup one, so it is not the default)."""
    p, L = self.plan, _lib.lib()
    name = 'critic' if phase == 2 else 'actor'
    logp, q = (_lib.ptr(p.logp), _lib.ptr(p.q)) if phase == 2 else (None, None)
    if self.peer is not None and os.environ.get('IL_PEER_APPLY', '0') == '1':
      _lib.check(L.il_sac_dp_phase_peer(C.byref(p.sac), C.byref(p.pb), phase, logp, q, 0, C.byref(self.peer.desc[name]), _lib.stream_ptr()))
      return
    self._exchange(name, group)
    _lib.check(L.il_sac_dp_phase(C.byref(p.sac), C.byref(p.pb), phase, logp, q, 0, _lib.stream_ptr()))

  def _bc_aux_step(self):
    """imitation.bc_aux_loss (train.py:201: `behavioural_cloning_update(actor, expert_transitions, actor_optimiser)` between the reward step and `sac_update`), data-parallel
    (round 6): the loss is a weighted mean over the expert batch (training.py:57-64), so the gradient of the ranks' concatenated expert batches is the mean of their
    gradients. il_bc_step(IL_FLAG_GRADS_ONLY) writes this rank's gradient into the actor bucket and ticks the actor's optimiser, the bucket is averaged (its last lane, log
    alpha's gradient, travels along and is not applied here), il_adam_step applies it with the step the gradient launch ticked - the arithmetic of the fused epilogue, bit
    for bit with one rank (test_data_parallel_bc_aux_equals_the_plain_plan_on_one_rank). It moves the actor: the lane-ordered copies are re-derived by phase 0."""
    p, L = self.plan, _lib.lib()
    a, ao = p._keep[0], p._keep[4]
    od = ao.desc()
    st = _lib.stream_ptr()
    _lib.check(L.il_bc_step(_lib.ptr(a.flat), _lib.ptr(ao.grad), C.byref(od), a.state_size, a.action_size, a.hidden, C.byref(p.eb), C.c_void_p(p.sac.workspace), p.sac.workspace_floats,
                            None, _lib.IL_FLAG_GRADS_ONLY, st))
    self._exchange('actor', self.group)
    _lib.check(L.il_adam_step(_lib.ptr(a.flat), _lib.ptr(ao.grad), C.byref(od), a.flat.numel(), 0, st))
    p._prepared = False

  def replica_state(self):
    """Every tensor the ranks must agree on bit for bit after any number of data-parallel updates: parameter arenas, log alpha, spectral-norm buffers, and each optimiser's
    moments and step counter (`replicas_bit_identical` hashes them)."""
    actor, critic, log_alpha, target, ao, co, to, disc, do = self.plan._keep
    out = replica_tensors(actor, critic, target, log_alpha, disc if self.plan.algorithm == 'GAIL' else None)
    for o in (ao, co, to) + ((do,) if self.plan.algorithm == 'GAIL' and do is not None else ()):
      out += [o.exp_avg, o.exp_avg_sq, o.step_count[:1]]
    return out

  def resync_replicas(self, src: int = 0):
    """Collective: rank `src`'s replica state everywhere (after a failed exchange left the replicas apart)."""
    torch.cuda.synchronize()
    broadcast_parameters(self.replica_state(), src=src, group=self.group)
    self.plan._prepared = False   # the lane-ordered weight copies are re-derived from the broadcast parameters by the next update
    torch.cuda.synchronize()

  def exchange_name(self) -> str:
    """What moves the gradients: 'peer write-through' | 'peer fences' (il_peer_allreduce_mean over peer-mapped windows) | 'rccl' | 'gloo' | 'none' (one rank, no exchange)."""
    if self.peer is not None:
      return ('peer write-through' if self.peer.form else 'peer fences') + (', inside the optimiser launches' if self.fused else '')
    if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and os.environ.get('IL_FORCE_ALLREDUCE') != '1'):
      return 'none'
    return 'rccl' if dist.get_backend(self.group) == 'nccl' else dist.get_backend(self.group)

  def use_collectives(self, why: str):
    """Collective: leave the peer-window exchange for torch.distributed all-reduces (RCCL) - after an expired exchange wait or diverged replicas. Captured graphs are dropped
    (they hold the peer launches): capture again. The caller re-broadcasts the replicas."""
    torch.cuda.synchronize()
    if self.peer is not None:
      self.peer.close(collective=True)
    self.peer, self.peer_note = None, why
    self.fused, self.plan.peer_desc = False, None
    self.graph = self.graph_side = None

  def exchange_timeouts(self) -> int:
    """Peer-window waits that gave up since set-up (0 with the collectives); a non-zero count means some update averaged stale gradients."""
    return self.peer.timeouts() if self.peer is not None else 0

  def agree_on_handoff(self) -> bool:
    """Call after a few eager updates: if ANY rank saw a bounded device-side wait expire (its two streams did not run concurrently, or a peer stalled for longer than the
    bound), EVERY rank leaves the hand-off schedule - the decision must be collective, the two schedules issue the discriminator's all-reduce on different communicators.
    Returns whether the hand-off schedule stays on. The updates that timed out used stale rewards: a training run should stop (train.py does); a benchmark carries on."""
    if not self.handoff:
      return False
    bad = torch.tensor([float(self.plan.sync_timeouts())], device=self.plan.rows.device)
    if dist.is_initialized() and dist.get_world_size(self.group) > 1:
      dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
    if float(bad.item()) > 0:
      torch.cuda.synchronize()
      self.handoff = False
      self.fused, self.plan.peer_desc = False, None   # the stream-dependency schedule keeps one exchange launch per sync point (the same peer regions serve it)
      keep = (self.plan.sync[self.plan._sync_spin].clone(), self.plan.sync[self.plan._sync_host_flag].clone())   # the widened bound and the host's time-out word survive the reset
      self.plan.sync.zero_()
      self.plan.sync[self.plan._sync_spin], self.plan.sync[self.plan._sync_host_flag] = keep
      self.plan._set_device_sync(False)
      # (round 6) an expired wait POISONS its learner: the optimiser launches of that update (and of every later one) skipped their stores on the affected rank only, so the
      # replicas are no longer identical - the zeroed sync buffer above cleared the poison; rank 0's state everywhere puts the ranks back on one learner (collective)
      self.resync_replicas()
    return self.handoff

  def run(self):
    """sample -> [side stream: discriminator grads -> all-reduce -> AdamW -> reward | main: SAC forward] -> critic grads -> all-reduce ->
    AdamW(critic) + actor grads -> all-reduce -> AdamW(actor) + Adam(alpha) + polyak.  The discriminator's all-reduce hides under the
    SAC forward kernels; the critic and actor all-reduces are on the critical path (the algorithm orders them)."""
    p, L = self.plan, _lib.lib()
    G = _lib.IL_FLAG_GRADS_ONLY
    main = torch.cuda.current_stream()
    if self._warm_collectives_pending:
      self._warm_collectives()
    if self.handoff:
      self._run_handoff(main)
      return
    if p.algorithm == 'GAIL' and self.variant:
      # a discriminator variant: its per-function entry points on the gathered rows (both batches packed), beside the reward-independent SAC forward
      p.sample_all()
      self.side.wait_stream(main)
      with torch.cuda.stream(self.side):
        p._disc_step_and_relabel_on_gathered_rows(_lib.stream_ptr(), exchange=lambda: self._exchange('disc', self.group))
    elif p.algorithm == 'GAIL' and p.device_index_draw:
      # The discriminator branch (gradients, all-reduce, AdamW, relabel: the longer one) reads its rows straight from the rings through the drawn indices
      # (il_batch.gather), so it forks right after the index draw; the gathers the SAC kernels need run on the main stream beside it.
      p.draw_all()
      self.side.wait_stream(main)
      rp, re_ = p._ring_batches()
      with torch.cuda.stream(self.side):
        _lib.check(L.il_gail_disc_step(C.byref(p.disc), C.byref(rp), C.byref(re_), None, None, G, _lib.stream_ptr()))
        self._exchange('disc', self.group)
        _lib.check(L.il_gail_apply_grads(C.byref(p.disc), _lib.stream_ptr()))
        _lib.check(L.il_gail_reward(C.byref(p.disc), C.byref(rp), _lib.ptr(p.rewards), None, None, _lib.stream_ptr()))
      p.gather_all(expert=p.bc_aux)   # the SAC kernels read the packed agent rows; the expert rows are only needed by the discriminator step (through the indices) and the BC auxiliary step
    else:
      p.sample_all()
      if p.algorithm not in ('SAC', 'PWIL', 'GAIL'):
        p._enqueue_reward_model(_lib.stream_ptr())   # GMMIL / RED / DRIL / AdRIL: no parameters are trained inside the update block, so nothing to all-reduce here
      if p.algorithm == 'GAIL':
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
          _lib.check(L.il_gail_disc_step(C.byref(p.disc), C.byref(p.pb), C.byref(p.eb), None, None, G, _lib.stream_ptr()))
          self._exchange('disc', self.group)
          _lib.check(L.il_gail_apply_grads(C.byref(p.disc), _lib.stream_ptr()))
          _lib.check(L.il_gail_reward(C.byref(p.disc), C.byref(p.pb), _lib.ptr(p.rewards), None, None, _lib.stream_ptr()))
    if p.bc_aux:
      self._bc_aux_step()
    _lib.check(L.il_sac_dp_phase(C.byref(p.sac), C.byref(p.pb), 0, None, None, p.prepared_flag(), _lib.stream_ptr()))
    if p.algorithm == 'GAIL':
      main.wait_stream(self.side)
    _lib.check(L.il_sac_dp_phase(C.byref(p.sac), C.byref(p.pb), 1, None, None, 0, _lib.stream_ptr()))
    self._apply_phase(2, self.group)
    self._apply_phase(3, self.group)
    p._prepared = True

  def _enqueue_side(self):
    """Discriminator branch: [resident index draw] -> gradients from the rings through the indices -> all-reduce (own communicator) -> AdamW, which signals [IL_SYNC_PARAMS]."""
    p, L = self.plan, _lib.lib()
    if self.fused:   # the exchange rides in the reduce + AdamW launch, which signals [IL_SYNC_PARAMS] as on one GPU
      p._disc_step(_lib.IL_FLAG_GAIL_CLOSE_EPOCH)
      return
    p._disc_step(_lib.IL_FLAG_GRADS_ONLY)   # the index draw rides in this launch when the sampler is resident
    self._exchange('disc', self.side_group)
    _lib.check(L.il_gail_apply_grads(C.byref(p.disc), _lib.stream_ptr()))

  def _enqueue_main(self):
    """SAC branch: [index draw] -> forward + critic loss chained per tile with the rewards relabelled inline (waits on the device for the discriminator's all-reduced step)
    -> critic gradients -> all-reduce -> AdamW(critic) + policy loss + actor gradients -> all-reduce -> AdamW(actor), Adam(alpha), polyak."""
    p, L = self.plan, _lib.lib()
    if self.fused:   # the single-GPU branch; its two optimiser launches carry the critic's and the actor's exchange (plan.peer_desc)
      p._enqueue_sac_branch()
      p._prepared = True
      return
    resident = p.resident_sampler
    if not resident:
      p.sample_all()
    flags = p.prepared_flag() | _lib.IL_FLAG_GRADS_ONLY | (_lib.IL_FLAG_SAC_WAIT_INDICES if resident else 0)
    _lib.check(L.il_sac_update_gather(C.byref(p.sac), C.byref(p.pb), C.byref(p._ring_batches()[0]), None, C.byref(p.disc), _lib.ptr(p.rewards), None, None, _lib.ptr(p.logp), _lib.ptr(p.q),
                                      flags, _lib.stream_ptr()))
    self._apply_phase(2, self.group)
    self._apply_phase(3, self.group)
    p._prepared = True

  def _run_handoff(self, main):
    if self.plan.resident_sampler:
      self.side.wait_stream(main)   # eager: appends the caller enqueued before this update precede the resident draw
    with torch.cuda.stream(self.side):
      self._enqueue_side()
    self._enqueue_main()
    main.wait_stream(self.side)     # eager: leave the caller's stream ordered after both branches

  def capture(self, warmup: int = 3):
    if self._warm_collectives_pending:
      self._warm_collectives()   # set-up (communicators, peer windows: allocations, host synchronisation) never happens inside a capture
    if self.handoff:   # two graphs, one per branch and per communicator, replayed on two streams with no edge between them (cf. UpdatePlan.capture)
      p = self.plan
      if not _agree(p._probe_device_sync(graph=True), self.group):   # e.g. a counter-collecting profiler serialises the two graphs on SOME rank: every rank takes stream dependencies, one graph (below)
        self.handoff = False
        self.fused, p.peer_desc = False, None
        p._set_device_sync(False)
        return self.capture(warmup)
      p.memory.stream().device_state(p.rows.device)
      for _ in range(warmup):
        self.run()
      torch.cuda.synchronize()
      self.graph_side = torch.cuda.CUDAGraph()
      with torch.cuda.graph(self.graph_side, stream=self.side):
        self._enqueue_side()
      self.graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(self.graph):
        self._enqueue_main()
      return self
    self.plan.memory.stream().device_state(self.plan.rows.device)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
      for _ in range(warmup):
        self.run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      self.run()
    return self

  # --- direct launches (round 6): with the exchanges INSIDE the optimiser launches (`fused`) a data-parallel update is the launch sequence of one GPU - the plan's own two
  # branches with the peer descriptors attached - so it can be issued the way the single-GPU plan is: two library calls per update, no hipGraph (a graph replay costs ~4.5 us
  # more between two updates than the launch boundary of the same kernels). The RCCL schedule keeps its graphs: its all-reduces are torch.distributed calls.
  def direct_launch_ok(self) -> bool:
    return bool(self.handoff and self.fused and self.peer is not None and self.plan.peer_desc is not None and self.plan.direct_launch_ok())

  def record_direct(self):
    if self._warm_collectives_pending:
      self._warm_collectives()
    assert self.direct_launch_ok(), 'DataParallelUpdate.record_direct: the fused peer-window schedule only (IL_PEER_EXCHANGE=1 with the device hand-off); capture() otherwise'
    self.plan.record_direct()
    return self

  def launch_direct(self):
    self.plan.launch_direct()

  def replay(self):
    if self.graph_side is not None:
      if self.plan.main_feeds_ring and self.plan.resident_sampler:
        self.side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(self.side):
        self.graph_side.replay()
    self.graph.replay()
