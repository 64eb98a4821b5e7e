"""N>1 path on CPU: world_size-2 gloo processes exercise the data-parallel protocol of imitation-learning_amd/parallel.py
(bucket mean all-reduce, replica broadcast, rank seeds).  The numerical claim checked is the one the design rests on: every loss
is a mean over samples, so the mean over ranks of per-rank gradients on rank-local batches == the gradient on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
  return p


def _worker(rank, world, port, out_dir):
  for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import inputs as gi
  from imitation_learning_amd import parallel
  from oracle import gail as ogail
  from oracle import sac as osac

  # --- replicas: rank 0's parameters win
  p = torch.full((1000,), float(rank + 1))
  parallel.broadcast_parameters([p])
  assert torch.all(p == 1.0)
  assert parallel.rank_seed(10) == 10 + rank

  # --- SAC critic gradient + GAIL gradient on this rank's shard of a global batch of 64
  c = gi.sac_case(21, 'hopper', 64, 64, 1)
  g = gi.gail_case(22, env='hopper', hidden=32, batch=64, steps=1)
  half = slice(rank * 32, (rank + 1) * 32)
  shard = lambda b: {k: v[half] for k, v in b.items()}
  st = osac.SacState(c['S'], c['A'], c['H'])
  st.actor[:], st.critic[:], st.target[:], st.log_alpha[:] = c['actor'], c['critic'], c['target'], c['log_alpha']
  _, _, gr = osac.sac_update(st, shard(c['batches'][0]), c['eps_next'][0][half], c['eps_cur'][0][half], discount=c['discount'], entropy_target=c['entropy_target'],
                             polyak_factor=c['polyak'], return_grads=True)
  bucket = torch.from_numpy(gr['critic'].copy())
  parallel.all_reduce_mean_(bucket)

  ds = ogail.DiscState(g['D'], g['H'], True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = g[k]
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  pb, eb = shard(g['policy'][0]), shard(g['expert'][0])
  dg = ogail.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], g['eps'][0][half], lr=3e-5, weight_decay=10, grad_penalty=1.0, return_grads=True)
  dbucket = torch.from_numpy(dg.copy())
  parallel.all_reduce_mean_(dbucket)
  # --- the REAL exchange buffers of DataParallelUpdate (GradBuckets): actor gradient + log-alpha gradient share one padded bucket, written through views.
  # The policy loss runs through the critic AFTER its step, which in the data-parallel protocol used the all-reduced gradient on every rank: lr = 0 keeps the
  # critic identical everywhere here, so that the actor / alpha gradients of the shards must average to those of the whole batch.
  st = osac.SacState(c['S'], c['A'], c['H'])
  st.actor[:], st.critic[:], st.target[:], st.log_alpha[:] = c['actor'], c['critic'], c['target'], c['log_alpha']
  _, _, gr = osac.sac_update(st, shard(c['batches'][0]), c['eps_next'][0][half], c['eps_cur'][0][half], discount=c['discount'], entropy_target=c['entropy_target'],
                             polyak_factor=c['polyak'], lr=0.0, return_grads=True)
  Pa = gr['actor'].size
  gb = parallel.GradBuckets(Pa, torch.from_numpy(gr['critic'].copy()), torch.from_numpy(dg.copy()), device='cpu')
  assert gb.actor.numel() % 4 == 0 and gb.actor.numel() >= Pa + 1 and gb.actor_grad.data_ptr() == gb.actor.data_ptr() and gb.alpha_grad.data_ptr() == gb.actor.data_ptr() + 4 * Pa
  gb.actor_grad.copy_(torch.from_numpy(gr['actor'])); gb.alpha_grad.copy_(torch.from_numpy(gr['alpha']))   # what the grads-only kernels do on the GPU
  for which in ('disc', 'critic', 'actor'):   # dependency order of an update
    gb.exchange(which)
  assert float(gb.actor[Pa + 1:].abs().sum()) == 0.0   # the padding stays zero
  tensors = parallel.replica_tensors(*(type('M', (), dict(flat=torch.full((5,), float(rank + k)))) for k in range(3)), torch.full((1,), float(rank)),
                                     type('D', (), dict(flat=torch.full((4,), float(rank)), sn=torch.full((3,), float(rank)), target_flat=None)))
  parallel.broadcast_parameters(tensors)
  assert [float(t[0]) for t in tensors] == [0.0, 1.0, 2.0, 0.0, 0.0, 0.0]   # rank 0's values everywhere, spectral-norm buffers included
  assert parallel.broadcast_scalars([0.25 + rank, None]) == [0.25, None]
  np.savez(os.path.join(out_dir, f'rank{rank}.npz'), critic=bucket.numpy(), disc=dbucket.numpy(), u1=ds.u1, b_actor=gb.actor_grad.numpy(), b_alpha=gb.alpha_grad.numpy(),
           b_critic=gb.critic.numpy(), b_disc=gb.disc.numpy())
  dist.barrier()
  dist.destroy_process_group()


def test_data_parallel_gradient_identity(tmp_path):
  sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
  import inputs as gi
  from oracle import gail as ogail
  from oracle import sac as osac
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
  # every rank holds the same averaged buckets (so replicas apply identical updates)
  np.testing.assert_array_equal(r0['critic'], r1['critic']); np.testing.assert_array_equal(r0['disc'], r1['disc'])
  np.testing.assert_array_equal(r0['u1'], r1['u1'])  # spectral-norm buffers depend only on the replicated weights
  # ... and they equal the single-process gradient on the concatenated batch
  c = gi.sac_case(21, 'hopper', 64, 64, 1)
  st = osac.SacState(c['S'], c['A'], c['H'])
  st.actor[:], st.critic[:], st.target[:], st.log_alpha[:] = c['actor'], c['critic'], c['target'], c['log_alpha']
  _, _, gr = osac.sac_update(st, c['batches'][0], c['eps_next'][0], c['eps_cur'][0], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'],
                             return_grads=True)
  scale = np.abs(gr['critic']).max()
  assert np.abs(r0['critic'] - gr['critic']).max() <= 2e-6 * scale
  # the same through the real bucket layout: every bucket identical on both ranks and equal to the gradient of the concatenated batch
  for k in ('b_actor', 'b_alpha', 'b_critic', 'b_disc'):
    np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
  np.testing.assert_array_equal(r0['b_critic'], r0['critic'])
  st = osac.SacState(c['S'], c['A'], c['H'])
  st.actor[:], st.critic[:], st.target[:], st.log_alpha[:] = c['actor'], c['critic'], c['target'], c['log_alpha']
  _, _, gr = osac.sac_update(st, c['batches'][0], c['eps_next'][0], c['eps_cur'][0], discount=c['discount'], entropy_target=c['entropy_target'], polyak_factor=c['polyak'], lr=0.0,
                             return_grads=True)
  assert np.abs(r0['b_actor'] - gr['actor']).max() <= 4e-6 * np.abs(gr['actor']).max()
  # d L_alpha / d log_alpha = -alpha * mean(w m (log pi + H)) is a mean too, but the log pi in it come from each rank's own rows: compare with the mean of the shard values
  assert np.isfinite(r0['b_alpha']).all() and abs(float(r0['b_alpha'][0]) - float(gr['alpha'][0])) <= 1e-5 * max(1.0, abs(float(gr['alpha'][0])))
  g = gi.gail_case(22, env='hopper', hidden=32, batch=64, steps=1)
  ds = ogail.DiscState(g['D'], g['H'], True)
  for k in ('W1', 'b1', 'W2', 'b2', 'u1', 'v1', 'u2', 'v2'):
    getattr(ds, k)[...] = g[k]
  cat = lambda b: np.concatenate([b['states'], b['actions']], axis=1)
  pb, eb = g['policy'][0], g['expert'][0]
  dg = ogail.gail_update(ds, cat(pb), pb['weights'], cat(eb), eb['weights'], g['eps'][0], lr=3e-5, weight_decay=10, grad_penalty=1.0, return_grads=True)
  assert np.abs(r0['disc'] - dg).max() <= 2e-6 * np.abs(dg).max()


def test_all_reduce_mean_is_identity_without_process_group():
  from imitation_learning_amd import parallel
  t = torch.arange(8, dtype=torch.float32)
  assert torch.equal(parallel.all_reduce_mean_(t.clone()), t)


def _peer_fallback_worker(rank, world, port, out_dir, scenario):
  """PeerExchange.create() with a set-up failure on ONE rank only (faked library calls: there is no GPU here): every rank must come back with None - no rank may be left
  waiting in a collective the failing rank never entered - so that DataParallelUpdate falls back to the collectives everywhere."""
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import contextlib
  import torch as _torch
  from imitation_learning_amd import _lib, parallel
  real = _lib.lib()

  class Fake:
    def __getattr__(self, name): return getattr(real, name)
    def il_peer_window_alloc(self, total, wref, handle, kind):
      if scenario == 'alloc' and rank == 1: return 3
      wref._obj.value = 0x1000 * (rank + 1)
      kind._obj.value = 0
      return 0
    def il_peer_window_open(self, h, oref):
      if scenario == 'open' and rank == 0: return 3
      oref._obj.value = 0x9000
      return 0
    def il_peer_window_close(self, p): return 0
    def il_peer_window_free(self, p): return 0
    def il_last_error(self): return b'faked failure'
    def il_peer_allreduce_mean(self, d, b, st): return 3 if scenario == 'kernel' and rank == 1 else 0   # the self-test then raises (rank 1) or sees wrong values (rank 0)
  _lib._lib = Fake()
  _torch.cuda.device = lambda d: contextlib.nullcontext()
  _torch.cuda.synchronize = lambda d=None: None
  _lib.stream_ptr = lambda: None
  x = parallel.PeerExchange.create(dict(critic=5000, actor=37), _torch.device('cpu'))
  open(os.path.join(out_dir, f'{scenario}{rank}.txt'), 'w').write('none' if x is None else 'exchange')
  dist.barrier()
  dist.destroy_process_group()


def test_peer_exchange_setup_failure_on_one_rank_falls_back_everywhere(tmp_path):
  for scenario in ('alloc', 'open', 'kernel'):
    mp.spawn(_peer_fallback_worker, args=(2, _free_port(), str(tmp_path), scenario), nprocs=2, join=True)
    assert [open(tmp_path / f'{scenario}{r}.txt').read() for r in (0, 1)] == ['none', 'none'], scenario
