"""CPU-side checks: the C-ABI library loads and exports every symbol include/il_hip.h declares (no compute calls), host-side
index draws are bit-exact with numpy's legacy stream, and the product package never touches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'imitation-learning_amd')


def declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'il_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(il_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
  from imitation_learning_amd import _lib
  lib = C.CDLL(_lib.LIB_PATH)
  names = declared_symbols()
  assert len(names) >= 30
  for n in names:
    assert hasattr(lib, n), f'{n} is declared in include/il_hip.h but not exported by libil_hip.so'
  assert set(_lib._SIGNATURES) == set(names), set(_lib._SIGNATURES) ^ set(names)
  assert _lib.lib().il_abi_version() == 1


def test_struct_sizes_match_header():
  """ctypes mirrors vs the C structs (sizes computed from the header's field lists on an LP64 target)."""
  from imitation_learning_amd import _lib
  assert C.sizeof(_lib.Batch) == 7 * 8 + 8 * 4 + 8 + 8   # + gather, gather_capacity
  assert C.sizeof(_lib.Adam) == 3 * 8 + 5 * 8
  assert C.sizeof(_lib.Pwil) == 4 * 4 + 5 * 8 + 3 * 8
  assert C.sizeof(_lib.Sac) == 4 * 4 + 7 * 8 + 3 * C.sizeof(_lib.Adam) + 2 * 4 + 8 + 8 + 8 + 8 + 8 + 2 * 8 + 8 + 8   # + debug_masks
  assert _lib.lib().il_ring_row_floats(18, 6) == 48 and _lib.lib().il_ring_row_floats(112, 8) == 240
  assert _lib.lib().il_mlp_numel(18, 256, 12) == 73740 and _lib.lib().il_mlp_stride(24, 256, 1) == 72452


def test_product_package_never_imports_the_oracle():
  pat = re.compile(r'^\s*(from|import)\s+oracle\b|[\'"]oracle[/\'"]|_ref\b|ref_cpu_baseline|build_ref|/root/reference', re.M)   # neither the numpy oracle nor oracle/_ref (the reference's own modules, byte-compiled) nor the reference itself
  for dirpath, _, files in os.walk(PKG):
    for f in files:
      if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
        src = open(os.path.join(dirpath, f)).read()
        assert not pat.search(src), f'{os.path.join(dirpath, f)} references oracle/ (the oracle is test infrastructure only)'
  for f in ('train.py',):
    p = os.path.join(ROOT, f)
    if os.path.exists(p):
      assert not pat.search(open(p).read())


def test_reference_build_product_is_the_reference():
  """oracle/_ref (what bench.py's cpu_baseline leg times live): byte-compiled from the reference's own three hot-path modules - the MANIFEST's source hashes equal the files
  under /root/reference when that is present (this container), the modules import sourcelessly and expose the reference's entry points, and nothing of it is tracked by git."""
  import hashlib, json, subprocess, sys
  ref_dir = os.path.join(ROOT, 'oracle', '_ref')
  have_reference = os.path.isfile('/root/reference/training.py')
  if have_reference:
    subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'build_ref.py')], check=True, capture_output=True)
  if not os.path.isdir(ref_dir):
    pytest.skip('oracle/_ref is absent and there is no /root/reference to build it from')
  man = json.load(open(os.path.join(ref_dir, 'MANIFEST.json')))
  assert set(man['modules']) == {'memory', 'models', 'training'}
  if have_reference:
    for m, v in man['modules'].items():
      assert v['source_sha256'] == hashlib.sha256(open(f'/root/reference/{m}.py', 'rb').read()).hexdigest()
  assert not any(f.endswith('.py') for f in os.listdir(ref_dir)), 'no reference source text in the repo: byte code only (plus our omegaconf stand-in in its own directory)'
  code = ('import sys; sys.path.insert(0, %r); import memory, models, training; '
          'assert training.sac_update and training.adversarial_imitation_update and models.GAILDiscriminator and memory.ReplayMemory; print(memory.__file__)' % ref_dir)
  r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
  assert r.returncode == 0 and r.stdout.strip().endswith('memory.pyc'), r.stderr[-500:]
  tracked = subprocess.run(['git', '-C', ROOT, 'ls-files', 'oracle/_ref'], capture_output=True, text=True).stdout.strip()
  assert tracked == '', 'oracle/_ref must stay out of git history'


@pytest.mark.parametrize('seed,size,idx,full', [(0, 1000, 300, False), (1, 64, 22, True), (2, 500, 0, True), (3, 1_000_000, 100_000, False), (4, 25_000, 0, True), (5, 3, 0, True)])
def test_host_index_draws_bit_exact_with_numpy(seed, size, idx, full):
  """il_mt19937_sample_indices vs literally what memory.py:51-59 does with numpy's global legacy stream."""
  from imitation_learning_amd import _lib
  st = (C.c_uint32 * 625)()
  _lib.check(_lib.lib().il_mt19937_seed(st, seed))
  out = (C.c_int32 * 700)()
  got = []
  for _ in range(3):  # several calls: the state persists across calls (twist boundaries inside)
    _lib.check(_lib.lib().il_mt19937_sample_indices(st, 700, size, idx, int(full), out))
    got += list(out)
  np.random.seed(seed)
  want = []
  while len(want) < 2100:
    v = int(np.random.randint(0, size if full else idx - 1))
    if v != (idx - 1) % size:
      want.append(v)
  assert got == want


def test_host_index_draw_rejects_empty_ranges():
  from imitation_learning_amd import _lib
  st = (C.c_uint32 * 625)()
  _lib.lib().il_mt19937_seed(st, 0)
  out = (C.c_int32 * 4)()
  assert _lib.lib().il_mt19937_sample_indices(st, 4, 100, 1, 0, out) != 0   # not full, idx - 1 == 0 candidates
  assert b'range' in _lib.lib().il_last_error()


def test_general_shape_entry_points_size_and_reject_on_the_host():
  """csrc/general.hip without a GPU: the parameter counts of general shapes equal torch's (`parameters()` order: W [out][in], b [out] per layer), strides are 16-byte
  multiples, the workspace sizes grow with depth, and shapes / arguments outside the engine's range come back as error codes before any launch."""
  from imitation_learning_amd import _lib
  L = _lib.lib()
  for (i, h, d, o) in ((12, 48, 3, 6), (18, 80, 1, 12), (33, 320, 2, 1), (24, 72, 3, 1), (5, 7, 8, 3)):
    dims = [i] + [h] * d + [o]
    want = sum(dims[k + 1] * dims[k] + dims[k + 1] for k in range(len(dims) - 1))
    assert L.il_mlp_numel_general(i, h, d, o) == want
    s = L.il_mlp_stride_general(i, h, d, o)
    assert s % 4 == 0 and want <= s < want + 4
  assert L.il_mlp_numel_general(18, 256, 2, 12) == L.il_mlp_numel(18, 256, 12)   # the fused shape: the two layouts agree
  assert L.il_sac_workspace_floats_general(18, 6, 256, 3, 256, 3, 256) > L.il_sac_workspace_floats_general(18, 6, 256, 2, 256, 2, 256) > 0
  assert L.il_actor_workspace_floats_general(18, 6, 64, 2, 1) > 0
  ws = (C.c_float * 16)()
  out = (C.c_float * 16)()
  # depth 9 / an unknown activation / a workspace that is too small: refused before anything is launched
  assert L.il_actor_act_general(ws, 4, 2, 8, 9, 0, ws, 4, 1, None, 0, 0, 1, out, None, ws, 1 << 30, None) != 0 and b'depth' in L.il_last_error()
  assert L.il_actor_act_general(ws, 4, 2, 8, 2, 3, ws, 4, 1, None, 0, 0, 1, out, None, ws, 1 << 30, None) != 0
  assert L.il_actor_act_general(ws, 4, 2, 8, 2, 0, ws, 4, 1, None, 0, 0, 1, out, None, ws, 16, None) != 0 and b'workspace' in L.il_last_error()
  assert L.il_actor_log_prob_general(None, 4, 2, 8, 2, 0, ws, 4, ws, 2, 1, out, ws, 1 << 30, None) != 0
  assert L.il_sac_update_general(None, None, 2, 0, 8, 2, 0, None, None, None, None, 0, None) != 0 and b'il_sac_update_general' in L.il_last_error()
  assert L.il_noise_fill_beta(1, None, 0.0, 16, out, None) != 0 and b'alpha' in L.il_last_error()


def test_ctypes_structs_match_the_compiled_library():
  """sizeof() of every descriptor struct as compiled into libil_hip.so equals the ctypes mirror's (a stale binding would pass garbage)."""
  from imitation_learning_amd import _lib
  L = _lib.lib()
  for which, cls in enumerate((_lib.Batch, _lib.Adam, _lib.Sac, _lib.Disc, _lib.Pwil, _lib.SampleArgs, _lib.Red, _lib.Dril, _lib.DiscShaped, _lib.DiscDeep, _lib.PeerBucket, _lib.DiscShapedDeep)):
    assert L.il_struct_size(which) == C.sizeof(cls), cls.__name__
  assert L.il_struct_size(99) == -1


def test_peer_window_layout():
  """Host-side arithmetic of the peer-window exchange (include/il_hip.h il_peer_*): a bucket's region holds two parities x world slots of whole chunks plus one
  128-byte arrival line per chunk, rounded to 256 bytes, so that regions packed back to back keep every slot 16-byte aligned."""
  from imitation_learning_amd import _lib
  L = _lib.lib()
  CH = _lib.IL_PEER_CHUNK_FLOATS
  for world in (1, 2, 8, 16):
    for n in (1, 5, 1665, CH, CH + 1, 144904, 73744):
      nch = -(-n // CH)
      want = 2 * world * nch * CH * 4 + nch * 128
      got = L.il_peer_region_bytes(world, n)
      assert got % 256 == 0 and want <= got < want + 256, (world, n, got, want)
  assert L.il_peer_region_bytes(0, 10) == -1 and L.il_peer_region_bytes(17, 10) == -1 and L.il_peer_region_bytes(2, 0) == -1
  for world, n, jobs in ((2, 1665, 7), (8, 144904, 160), (16, 73748, 81)):   # the layout for exchanges inside the producing kernels: one arrival line per producing workgroup
    want = 2 * world * -(-n // CH) * CH * 4 + jobs * 128
    got = L.il_peer_job_region_bytes(world, n, jobs)
    assert got % 256 == 0 and want <= got < want + 256, (world, n, jobs, got, want)
  assert L.il_peer_job_region_bytes(2, 10, 0) == -1 and L.il_peer_job_region_bytes(0, 10, 1) == -1
  assert C.sizeof(_lib.PeerBucket) == 4 + 4 + 8 + 8 + 16 * 8 + 8 + 8 + 4 + 4 + 4 + 4


def test_reference_cpu_baseline_runner_reports_every_configuration():
  """oracle/ref_cpu_baseline.py (bench.py's cpu_baseline leg) on this host with a small budget: one JSON line with the reference's update rate at 1 thread and at all
  usable cores, with and without memory.sample, strictly time-boxed; needs oracle/_ref (built from /root/reference when that is present)."""
  import json, subprocess, sys, time
  if not os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'training.pyc')):
    if not os.path.isfile('/root/reference/training.py'):
      pytest.skip('no oracle/_ref and no /root/reference to build it from')
    subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'build_ref.py')], check=True, capture_output=True)
  t0 = time.time()
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py'), '--budget', '3'], capture_output=True, text=True, timeout=240, env=dict(os.environ, HIP_VISIBLE_DEVICES=''))
  assert r.returncode == 0, r.stderr[-800:]
  j = json.loads(r.stdout.strip().splitlines()[-1])
  res = j['results']
  assert {'one_thread_with_memory_sample', 'one_thread_without_memory_sample', 'all_cores_with_memory_sample'} <= set(res)
  assert res['one_thread_with_memory_sample'] > 1
  if res['one_thread_without_memory_sample'] is not None:   # (None: the 3 s budget of this test ran out on a loaded host before that configuration - the time box is the point)
    assert res['one_thread_without_memory_sample'] >= res['one_thread_with_memory_sample'] * 0.8
  assert j['nproc'] >= 1 and j['cpu_model'] and set(j['manifest']['modules']) == {'memory', 'models', 'training'}
  assert time.time() - t0 < 120, 'the runner must stay inside its time box'


def test_committed_pmc_file_lists_the_kernels_of_the_timed_schedule():
  """bench.py attaches roofline.traffic from profiles/pmc_latest.json and refuses (raises) when the file does not list the kernels of the schedule it has just timed: checked
  here, on the CPU, so that a stale counter file is caught before the round-end bench is."""
  import bench
  pmc = bench.load_pmc(list(bench.STAMP_MODEL))
  assert pmc.get('collected') and pmc.get('commit'), 'the PMC passes carry their collection date and commit (roofline.traffic_source)'
  for k in bench.STAMP_MODEL:
    assert pmc['kernels'][bench.PMC_NAMES[k]]['traffic_bytes'] > 0
  import pytest
  with pytest.raises(RuntimeError, match='predate the timed schedule'):
    bench.PMC_NAMES['k_new_kernel'] = 'k_new_kernel'
    try: bench.load_pmc(['k_new_kernel'])
    finally: bench.PMC_NAMES.pop('k_new_kernel')
