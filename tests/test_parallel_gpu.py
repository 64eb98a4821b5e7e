"""`-m gpu`: the data-parallel path with world_size 2 on the real kernels.  The GPU box has one GPU, so the two ranks SHARE it and exchange through gloo
(RCCL refuses two ranks on one device); everything else is the production path: grads-only kernels writing into the GradBuckets views, all-reduce(mean) of
one bucket per sync point, apply kernels, rank-offset seeds, replica broadcast.  What must hold: the replicas stay BIT-identical on both ranks (they apply the
same averaged gradient with the same kernels) while their data differ, and `train.py distributed.world_size=2` runs end to end (rank 0 writes the checkpoint)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests')); sys.path.insert(0, os.path.join(sys.argv[1], 'tests', 'golden'))
import numpy as np, torch
import torch.distributed as dist
import imitation_learning_amd as il
from imitation_learning_amd import parallel
rank, _, dev = parallel.init_from_env(2, 'gloo')
import test_gpu_parity as tg
algorithm = sys.argv[3]
torch.manual_seed(100 + rank)                       # different initialisation per rank on purpose: the broadcast must fix it
il.seed(parallel.rank_seed(5))
plan, nets = tg._make_plan(algorithm, 40 + rank)    # different replay shards per rank
parallel.broadcast_parameters(parallel.replica_tensors(nets[0], nets[1], nets[2], nets[3], nets[4] if algorithm == 'GAIL' else None))
dp = parallel.DataParallelUpdate(plan)
idx0 = None
direct = os.environ.get('IL_TEST_DIRECT') == '1'
for k in range(4):
  if direct and k == 2:
    assert dp.direct_launch_ok()
    dp.record_direct()
    dist.barrier()
  (dp.launch_direct if direct and k >= 2 else dp.run)()
  torch.cuda.synchronize()
  if k == 0: idx0 = plan.idx.cpu().numpy().copy()
out = {f't{i}': (n.flat if hasattr(n, 'flat') else n).detach().cpu().numpy() for i, n in enumerate(nets)}
out['sn'] = nets[4].sn.cpu().numpy(); out['idx0'] = idx0; out['logp'] = plan.logp.cpu().numpy()
out['handoff'] = np.array([int(dp.handoff), plan.sync_timeouts()])
out['peer'] = np.array([int(dp.peer is not None), dp.exchange_timeouts(), int(dp.fused)])
np.savez(os.path.join(sys.argv[2], f'rank{rank}.npz'), **out)
dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
  return p


def _launch(args, cwd, timeout=600, **extra_env):
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + args
  r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
  assert r.returncode == 0, f'{" ".join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}'
  return r


@pytest.mark.parametrize('algorithm,handoff,peer', [('GAIL', '1', 'fused'), ('GAIL', '1', '1'), ('GAIL', '0', '1'), ('SAC', '0', '1'), ('GAIL', '0', 'in_apply')])
def test_two_ranks_keep_bit_identical_replicas(tmp_path, algorithm, handoff, peer):
  """handoff = '1': the device-side hand-off schedule of DataParallelUpdate (resident index draw, inline relabel, one communicator per branch); '0': stream dependencies.
  The gradient exchange runs over peer-mapped windows (csrc/peer.hip; the two processes map each other's window through hipIpc exactly as two GPUs would): peer = 'fused' inside
  the optimiser launches (il_sac_update_gather_peer / il_gail_disc_step_draw_peer: the launch sequence of one GPU - the default with the hand-off), '1' with one
  il_peer_allreduce_mean launch per sync point (IL_DP_FUSED=0), 'in_apply' with the critic / actor exchanges inside the apply launches (il_sac_dp_phase_peer, IL_PEER_APPLY=1);
  gloo all-reduces are covered by test_peer_exchange_equals_the_collective."""
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  _launch([str(script), ROOT, str(tmp_path), algorithm], str(tmp_path), IL_DP_HANDOFF=handoff, IL_PEER_EXCHANGE='require', IL_PEER_APPLY='1' if peer == 'in_apply' else '0',
          IL_DP_FUSED='1' if peer == 'fused' else '0')
  r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
  assert [int(r0['peer'][0]), int(r1['peer'][0])] == [1, 1]
  assert [int(r0['peer'][2]), int(r1['peer'][2])] == [int(peer == 'fused')] * 2
  assert int(r0['peer'][1]) == 0 and int(r1['peer'][1]) == 0, 'a device-side wait of the peer-window exchange expired'
  if algorithm == 'GAIL':
    assert [int(r0['handoff'][0]), int(r1['handoff'][0])] == [int(handoff)] * 2
    if handoff == '1' and (r0['handoff'][1] or r1['handoff'][1]):
      pytest.skip('bounded device-side waits expired: the two ranks of this test SHARE one GPU and were time-sliced against each other (a production rank owns its GPU)')
  for k in ('t0', 't1', 't2', 't3') + (('t4', 'sn') if algorithm == 'GAIL' else ()):
    assert np.isfinite(r0[k]).all()
    np.testing.assert_array_equal(r0[k], r1[k], err_msg=f'replica tensor {k} differs between the ranks')
  assert not np.array_equal(r0['idx0'], r1['idx0']) and not np.array_equal(r0['logp'], r1['logp']), 'the ranks must train on different data (rank-offset seeds, own shards)'


def test_exchange_inside_the_optimiser_launches_equals_the_exchange_launches(tmp_path):
  """Same seeds, same shards, four updates of the hand-off schedule: with the gradient exchange inside the producing kernels (block jobs of k_dw_adam, workgroups of
  k_gail_reduce) every replica tensor keeps the bits it has with one exchange launch per sync point - the same rank-ordered means of the same gradients."""
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  res = {}
  for fused in ('1', '0'):
    d = tmp_path / f'fused_{fused}'
    d.mkdir()
    _launch([str(script), ROOT, str(d), 'GAIL'], str(tmp_path), IL_DP_HANDOFF='1', IL_PEER_EXCHANGE='require', IL_DP_FUSED=fused)
    res[fused] = [np.load(d / 'rank0.npz'), np.load(d / 'rank1.npz')]
    for r in res[fused]:
      assert int(r['peer'][1]) == 0, 'a device-side wait of the peer-window exchange expired'
      if r['handoff'][1]:
        pytest.skip('bounded device-side waits expired: the two ranks of this test SHARE one GPU and were time-sliced against each other')
  assert int(res['1'][0]['peer'][2]) == 1 and int(res['0'][0]['peer'][2]) == 0
  for rank in (0, 1):
    for k in ('t0', 't1', 't2', 't3', 't4', 'sn', 'logp'):
      np.testing.assert_array_equal(res['1'][rank][k], res['0'][rank][k], err_msg=f'rank {rank}, {k}')


def test_data_parallel_direct_launches_equal_the_eager_updates(tmp_path):
  """Round 6: with the exchanges inside the optimiser launches a data-parallel update is the plan's own two branches, so `DataParallelUpdate.record_direct()` /
  `launch_direct()` issue it as direct launches (what bench.py --gpus N and train.py do on the fused schedule). Two ranks, two eager updates then two direct ones: every
  replica tensor keeps the bits of four eager updates, on both ranks."""
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  res = {}
  for direct in ('0', '1'):
    d = tmp_path / f'direct_{direct}'
    d.mkdir()
    _launch([str(script), ROOT, str(d), 'GAIL'], str(tmp_path), IL_DP_HANDOFF='1', IL_PEER_EXCHANGE='require', IL_DP_FUSED='1', IL_TEST_DIRECT=direct)
    res[direct] = [np.load(d / 'rank0.npz'), np.load(d / 'rank1.npz')]
    for r in res[direct]:
      assert int(r['peer'][1]) == 0, 'a device-side wait of the peer-window exchange expired'
      assert int(r['peer'][2]) == 1
      if r['handoff'][1]:
        pytest.skip('bounded device-side waits expired: the two ranks of this test SHARE one GPU and were time-sliced against each other')
  for rank in (0, 1):
    for k in ('t0', 't1', 't2', 't3', 't4', 'sn', 'logp'):
      np.testing.assert_array_equal(res['1'][rank][k], res['0'][rank][k], err_msg=f'rank {rank}, {k}')
  for k in ('t0', 't1', 't2', 't3', 't4', 'sn'):
    np.testing.assert_array_equal(res['1'][0][k], res['1'][1][k], err_msg=f'replica tensor {k} differs between the ranks')


def test_peer_exchange_equals_the_collective(tmp_path):
  """Same seeds, same shards, four updates: the peer-window exchange and gloo's all-reduce leave the SAME bits in every replica tensor (two ranks: (a + b) / 2 either way)."""
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  res = {}
  for peer in ('require', '0'):
    d = tmp_path / f'peer_{peer}'
    d.mkdir()
    _launch([str(script), ROOT, str(d), 'GAIL'], str(tmp_path), IL_DP_HANDOFF='0', IL_PEER_EXCHANGE=peer)
    res[peer] = np.load(d / 'rank0.npz')
  assert int(res['require']['peer'][0]) == 1 and int(res['0']['peer'][0]) == 0
  for k in ('t0', 't1', 't2', 't3', 't4', 'sn', 'logp'):
    np.testing.assert_array_equal(res['require'][k], res['0'][k], err_msg=k)


PEER_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
from imitation_learning_amd import parallel
rank, _, dev = parallel.init_from_env(2, 'gloo')
sizes = dict(a=1, b=5, c=1665, d=2048, e=2049, f=144904, ja=1, jb=7, jc=1665, jf=144904)
x = parallel.PeerExchange.create(sizes, dev, jobs=dict(ja=1, jb=3, jc=7, jf=160))   # j*: arrival lines per producing workgroup (the form the optimiser launches use), through k_peer_job_allreduce
assert x is not None, 'peer-window set-up or self-test failed'
g = torch.Generator(device='cpu'); g.manual_seed(7 + rank)
bad = 0
graph_bad = 0
form0 = x.form                                         # what the set-up ladder adopted: write-through (uncached windows), else fences
for rnd in range(8):                                   # odd and even epochs: both slot parities; rounds 5..7 with the OTHER form of the kernel (fences)
  if rnd == 5: x.set_form(0)
  for k, n in sizes.items():
    mine = torch.randn(n, generator=g)
    both = [torch.empty(n), torch.empty(n)]
    dist.all_gather(both, mine)
    t = mine.to(dev)
    x.allreduce_mean(k, t)
    want = ((both[0] + both[1]) / 2).to(dev)         # rank order, then the mean: what the kernel evaluates
    bad += int(not torch.equal(t, want))
# the same launch inside a captured graph: epochs are device counters, so replays keep working
buf = torch.zeros(sizes['f'], device=dev)
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
  gr = torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr, stream=s):
    x.allreduce_mean('f', buf)
  for rep in range(3):
    buf.fill_(float(rank + 1 + rep))
    gr.replay()
    s.synchronize()
    graph_bad += int(not torch.equal(buf, torch.full_like(buf, 1.5 + rep)))
torch.cuda.synchronize()
rep = x.soak_report
np.save(os.path.join(sys.argv[2], f'peer{rank}.npy'), np.array([bad, graph_bad, x.timeouts(), form0]))
np.save(os.path.join(sys.argv[2], f'soak{rank}.npy'), np.array([rep['rounds'], sum(rep['mismatching_elements'].values()), rep['expired_waits'], len(rep['buckets']), int('_load' in rep['buckets'])]))
x.close()
dist.barrier(); dist.destroy_process_group()
'''


def test_peer_exchange_kernel_two_ranks(tmp_path):
  """il_peer_allreduce_mean between two processes that map each other's window (hipIpc): bucket sizes that are not multiples of 4 floats nor of a chunk, several epochs,
  bitwise against (a + b) / 2 of the gathered inputs; then the launch captured in a hipGraph and replayed."""
  script = tmp_path / 'peer_worker.py'
  script.write_text(PEER_WORKER)
  _launch([str(script), ROOT, str(tmp_path)], str(tmp_path), IL_PEER_SOAK_ROUNDS='300')   # 2,000 by default; the two ranks of this test share one GPU
  for r in (0, 1):
    bad, graph_bad, timeouts, form0 = np.load(tmp_path / f'peer{r}.npy')
    assert (bad, graph_bad, timeouts) == (0, 0, 0), f'rank {r}: {bad} mismatching exchanges, {graph_bad} mismatching graph replays, {timeouts} expired waits'
    assert form0 == 1, 'the windows of an MI355X are uncached allocations: the set-up ladder should have adopted the write-through form'
    rounds, mism, expired, nb, has_load = np.load(tmp_path / f'soak{r}.npy')
    assert (rounds, mism, expired, nb, has_load) == (300, 0, 0, 11, 1), 'create() adopts a form only after the soak: every bucket + the 4 MB load bucket, interleaved on three streams, bitwise every round'


SOAK_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from imitation_learning_amd import parallel
rank, _, dev = parallel.init_from_env(2, 'gloo')
sizes = dict(disc=1665, critic=144904, actor=73744)                       # the three gradient buckets of the headline shape
x = parallel.PeerExchange.create(sizes, dev, jobs=dict(disc=7, critic=160, actor=149) if sys.argv[3] == 'jobs' else None)   # jobs: the form that rides in the optimiser launches
assert x is not None, 'peer-window set-up, self-test or soak failed'
rep = dict(x.soak_report, uncached=bool(x.uncached), form_flag=int(x.form), timeouts=x.timeouts())
json.dump(rep, open(os.path.join(sys.argv[2], f'soak{rank}.json'), 'w'))
x.close()
dist.barrier(); dist.destroy_process_group()
'''
HAMMER = "import torch, time\na = torch.empty(64 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)\nm = torch.randn(2048, 2048, device='cuda')\nt = time.time(); k = 0\nwhile time.time() - t < 120:\n  b.copy_(a); k += 1\n  if k % 7 == 0: (m @ m).sum().item()\n  torch.cuda.synchronize(); time.sleep(0.0003 * (k % 5))\n"


@pytest.mark.parametrize('kind,layout', [('uncached', 'chunks'), ('uncached', 'jobs'), ('finegrained', 'chunks'), ('finegrained', 'jobs')])
def test_peer_protocol_between_two_processes_soaks_under_uneven_load(tmp_path, kind, layout):
  """The peer-window protocol between two PROCESSES (own address spaces; each maps the other's window through hipIpc, as two GPUs would) under the set-up soak - every bucket
  of the headline shape, interleaved on three streams with the 4 MB load bucket, compared bitwise on the device every round - while a THIRD process hammers the same GPU with
  copy bursts and GEMMs at irregular intervals (uneven load: the MI355X guide's warning that idle chips and uniform load hide hand-off failures). Both window kinds forced
  (IL_PEER_WINDOW_KIND): uncached windows must adopt the write-through form (payload through sc0 sc1 stores, drained flag, no fences), fine-grained windows the
  system-scope release / acquire form. One GPU: what this cannot show is the xGMI path to a PEER's HBM."""
  import json
  script = tmp_path / 'soak_worker.py'
  script.write_text(SOAK_WORKER)
  bg = subprocess.Popen([sys.executable, '-c', HAMMER], cwd=ROOT)
  try:
    import time
    time.sleep(4)   # the hammer's first copies are under way
    _launch([str(script), ROOT, str(tmp_path), layout], str(tmp_path), IL_PEER_SOAK_ROUNDS='1500', IL_PEER_WINDOW_KIND=kind, IL_PEER_EXCHANGE='require')
  finally:
    bg.kill(); bg.wait()
  for r in (0, 1):
    rep = json.load(open(tmp_path / f'soak{r}.json'))
    assert rep['rounds'] == 1500 and rep['expired_waits'] == 0 and rep['timeouts'] == 0 and not any(rep['mismatching_elements'].values()), rep
    assert set(rep['buckets']) == {'disc', 'critic', 'actor', '_load'}
    assert rep['uncached'] == (kind == 'uncached') and rep['form'] == ('write-through' if kind == 'uncached' else 'fences'), rep


def test_peer_exchange_single_rank_is_identity():
  """World size 1 (IL_PEER_EXCHANGE=force, no process group): push to the own window, sum of one slab, mean = the input bits; the data-parallel split path through it equals
  the split path without an exchange."""
  from imitation_learning_amd import parallel
  dev = torch.device('cuda', 0)
  x = parallel.PeerExchange.create(dict(g=144904, h=7), dev)
  assert x is not None
  for n, k in ((144904, 'g'), (7, 'h')):
    for _ in range(3):
      t = torch.randn(n, device=dev)
      ref = t.clone()
      x.allreduce_mean(k, t)
      assert torch.equal(t, ref)
  assert x.timeouts() == 0
  x.close()


def test_train_py_runs_data_parallel(tmp_path):
  _launch([os.path.join(ROOT, 'train.py'), 'algorithm=GAIL', 'env=hopper', 'steps=400', 'training.start=200', 'training.batch_size=64', 'memory.size=2000', 'evaluation.interval=400',
           'evaluation.episodes=1', 'logging.interval=100', 'distributed.world_size=2', 'distributed.backend=gloo', '+synthetic_env.max_episode_steps=50',
           '+synthetic_env.dataset_trajectories=6'], str(tmp_path), IL_DEVICE_SYNC='0')   # stream dependencies: two processes time-slicing ONE GPU can starve a bounded device-side wait (train.py then raises, by design)
  runs = list((tmp_path / 'outputs' / 'GAIL_hopper').iterdir())
  assert len(runs) == 1, 'rank 0 alone owns the output directory'
  names = {p.name for p in runs[0].iterdir()}
  assert {'agent.pth', 'discriminator.pth', 'metrics.pth'} <= names
  m = torch.load(runs[0] / 'metrics.pth', weights_only=False)
  assert len(m['update_steps']) >= 2 and np.isfinite(np.asarray(m['Q_values'][-1])).all()


def test_soak_catches_a_corrupted_exchange():
  """The soak's comparison is live: one rank, one bucket, and a 'peer' that flips a single element of the result once, mid-run, must fail it."""
  from imitation_learning_amd import parallel
  dev = torch.device('cuda', 0)
  x = parallel.PeerExchange.create(dict(g=5000), dev, soak_rounds=50)
  assert x is not None and x.soak_report['rounds'] == 50 and not any(x.soak_report['mismatching_elements'].values())
  real, calls = x.allreduce_mean, [0]

  def tampering(name, bucket):
    out = real(name, bucket)
    calls[0] += 1
    if calls[0] == 17: bucket[1234] += 0.25
    return out
  x.allreduce_mean = tampering
  assert x.soak(40) is False and sum(x.soak_report['mismatching_elements'].values()) == 1
  x.allreduce_mean = real
  assert x.soak(40) is True
  x.close()


def test_expired_wait_raises_the_host_flag():
  """[IL_SYNC_HOST_FLAG]: a bounded device-side wait that gives up stores its count into a pinned host word as well (UpdatePlan.watch_timeouts / train.py's per-step check).
  The probe's waiter with no setter behind it is such a wait (~10 ms)."""
  import ctypes as C
  from imitation_learning_amd import _lib
  slots, timeouts, _, _, spin, host_flag = _lib.sync_layout()
  sync = torch.zeros(slots, dtype=torch.int64, device='cuda')
  word = torch.zeros(2, dtype=torch.int64).pin_memory()
  sync[host_flag] = word.data_ptr()
  _lib.check(_lib.lib().il_sync_probe(_lib.ptr(sync), 0, _lib.stream_ptr()))   # waiter only: expires
  torch.cuda.synchronize()
  assert int(sync[timeouts].item()) == 1 and int(word[0]) == 1
  _lib.check(_lib.lib().il_sync_probe(_lib.ptr(sync), 0, _lib.stream_ptr()))
  torch.cuda.synchronize()
  assert int(word[0]) == 2 and int(word[1]) == 0


def _bench(args, **env):
  e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
  e.update(HSA_ENABLE_IPC_MODE_LEGACY='0', **env)
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, capture_output=True, text=True, timeout=900)


def test_bench_gpus_2_refuses_on_a_one_gpu_box():
  if torch.cuda.device_count() >= 2:
    pytest.skip('this box has two GPUs')
  r = _bench(['--gpus', '2', '--steps', '5', '--warmup', '2'])
  assert r.returncode == 2 and 'needs 2 visible GPUs' in r.stderr and '"metric"' not in r.stdout


@pytest.mark.parametrize('fused', ['1', '0'])
def test_bench_runs_the_multi_gpu_default_schedule_with_one_rank(fused):
  """What every rank of `bench.py --gpus N` runs by default - the hand-off schedule with the resident sampler and the three exchanges INSIDE the optimiser launches over the
  peer windows - forced on with a world of one rank (IL_FORCE_DP=1, IL_PEER_EXCHANGE=force): the whole of bench.py must get through it, including the per-kernel section
  rank 0 runs after the timed region (round 4: it switched the plan to the stream-ordered draw while the fused exchange was still attached - the first real multi-GPU run
  would have died there after its measurement). IL_DP_FUSED=0: one exchange launch per sync point."""
  import json
  r = _bench(['--steps', '40', '--warmup', '10', '--trace-steps', '3', '--no-cpu-baseline', '--no-population', '--no-secondary'], IL_FORCE_DP='1', IL_PEER_EXCHANGE='force', IL_DP_FUSED=fused,
             IL_PEER_SOAK_ROUNDS='100')
  assert r.returncode == 0, r.stderr[-3000:]
  j = json.loads(r.stdout.strip().splitlines()[-1])
  c = j['config']
  assert j['n_gpus'] == 1 and j['value'] > 0 and c['replicas_bit_identical'] is True and c['exchange'].startswith('peer')
  assert ('inside the optimiser launches' in c['exchange']) == (fused == '1')
  assert any(k.startswith('k_sac_chain') for k in j['roofline']['kernels']) and c['finite'] is True


@pytest.mark.parametrize('peer', ['1', '0'])
def test_bench_starts_its_own_ranks_and_reports_the_replica_digest(peer):
  """`python bench.py --gpus 2` with no rank environment: bench.py launches the two ranks itself (here they share the box's GPU and meet over gloo: IL_BENCH_SHARE_GPU=1),
  prints n_gpus 2, the exchange actually used, the soak report and replicas_bit_identical."""
  import json
  r = _bench(['--gpus', '2', '--steps', '30', '--warmup', '5', '--trace-steps', '2'], IL_BENCH_SHARE_GPU='1', IL_PEER_EXCHANGE=peer, IL_PEER_SOAK_ROUNDS='200', IL_DEVICE_SYNC='0')
  assert r.returncode == 0, r.stderr[-3000:]
  j = json.loads(r.stdout.strip().splitlines()[-1])
  c = j['config']
  assert j['n_gpus'] == 2 and c['global_batch'] == 512 and 'bench.py itself' in c['launched_by']
  assert c['replicas_bit_identical'] is True and len(c['replica_digests']) == 2 and len(set(c['replica_digests'])) == 1
  if peer == '1':
    assert c['exchange_soak']['rounds'] == 200 and c['exchange_soak']['expired_waits'] == 0 and not any(c['exchange_soak']['mismatching_elements'].values())
    # one invocation times BOTH exchanges (an 8-GPU node may be there for one run only): the peer windows, then - from rank 0's replicas - the collectives
    ab = c['exchange_ab']
    assert ab['peer']['exchange'].startswith('peer') and ab['collectives']['exchange'] == 'gloo' and ab['peer']['valid'] and ab['collectives']['valid']
    for run in (ab['peer'], ab['collectives']):
      assert run['replicas_bit_identical'] is True and len(set(run['replica_digests'])) == 1 and run['updates_per_s'] > 0
    assert ab['chosen'] in ('peer', 'collectives') and j['value'] == ab[ab['chosen']]['updates_per_s']
    assert c['exchange'] == ab[ab['chosen']]['exchange']
  else:
    assert c['exchange'] == 'gloo' and c['exchange_soak'] is None and c['exchange_ab'] is None
  # the N = 1 schedule timed in the same job: data-parallel efficiency from this one line
  assert c['single_gpu_same_job']['updates_per_s'] > 0 and abs(c['dp_efficiency'] - j['value'] / 2 / c['single_gpu_same_job']['updates_per_s']) < 1e-3
  assert c['exchange_fallback'] is None and j['value'] > 0
