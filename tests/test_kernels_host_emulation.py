"""The HIP kernels executed on the HOST from their own source text (tests/host_emu: workgroups as fibers, MFMA / DPP / readlane / ballot on a wave-level exchange,
the `<<<>>>` launches rewritten - nothing else) against the reference fixtures. This is what `pytest -m "not gpu"` can say about kernel code where there is no GPU:
indexing, iteration orders, tile / slab / counter layouts, the reduce kernels' chain rules, the arithmetic. It says nothing about performance, memory ordering or
anything that needs two workgroups in flight; the `-m gpu` tests run the same comparisons on the real library.

Two layers: (1) the general GAIL discriminators through the raw C ABI with numpy pointers; (2) the BODIES of the `-m gpu` parity tests (tests/test_gpu_parity.py,
tests/test_timed_sizes.py) - the product's own models.py / training.py / memory.py in between, the GPU's tolerances - with the library handle swapped for the emulation."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'golden')]
import inputs as gi  # noqa: E402
from host_emu import build as emu_build  # noqa: E402
from imitation_learning_amd import _lib  # noqa: E402

f32 = np.float32
LOSS = {'BCE': 0, 'PUGAIL': 1, 'Mixup': 2}
REWARD = {'AIRL': 0, 'GAIL': 1, 'FAIRL': 2}
P = lambda a: None if a is None else C.c_void_p(a.ctypes.data)


@pytest.fixture(scope='module')
def golden_dir():
  return os.path.join(HERE, 'golden')


def emu(name=None):
  return emu_build.load()   # one library with every kernel file (built once per change of the sources, ~25 s)


class Keep(list):
  """Owns the numpy arrays a descriptor points into."""
  def arr(self, a, dtype=f32):
    a = np.ascontiguousarray(a, dtype=dtype)
    self.append(a)
    return a


def np_batch(keep, b):
  out = _lib.Batch()
  for k in ('states', 'actions', 'next_states', 'terminals', 'weights'):
    a = keep.arr(b[k])
    setattr(out, k, a.ctypes.data); setattr(out, 'ld_' + k, a.shape[1] if a.ndim == 2 else 1)
  out.n = b['states'].shape[0]
  return out


def np_adam(keep, n, lr, wd):
  o = _lib.Adam()
  m, v, step = keep.arr(np.zeros(n)), keep.arr(np.zeros(n)), keep.arr(np.zeros(16), np.int32)
  o.m, o.v, o.step, o.lr, o.beta1, o.beta2, o.eps, o.weight_decay = m.ctypes.data, v.ctypes.data, step.ctypes.data, lr, 0.9, 0.999, 1e-8, wd
  return o, step


def check(h, rc):
  if rc != 0:
    h.il_last_error.restype = C.c_char_p
    raise RuntimeError(f'emulated library error {rc}: {h.il_last_error().decode()}')


def close(a, b, what, rtol, atol_scale):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert a.shape == b.shape, what
  err = np.abs(a - b) - (atol_scale * np.abs(b).max() + rtol * np.abs(b))
  assert np.isfinite(a).all() and err.max() <= 0, f'{what}: worst excess {err.max():.3e} at {int(err.argmax())} ({a.ravel()[err.argmax()]} vs {b.ravel()[err.argmax()]})'


# ------------------------------------------------------------------------------------------------ the emulator's own checks
def test_emulator_mfma_is_the_16x16x4_instruction_the_kernels_assume():
  """v_mfma_f32_16x16x4_f32 in the lane layout of mlp_tile.hpp / DESIGN §3: against a float64 matmul, and bitwise against an fmaf chain over k in float32."""
  h = emu_build.load_selftest()
  rs = np.random.RandomState(0)
  K4 = 5
  A, B = rs.standard_normal((16, 4 * K4)).astype(f32), rs.standard_normal((4 * K4, 16)).astype(f32)
  D = np.zeros((16, 16), f32)
  h.selftest_mfma(P(A), P(B), P(D), K4)
  np.testing.assert_allclose(D, A.astype(np.float64) @ B.astype(np.float64), rtol=1e-5, atol=1e-5)
  chain = np.zeros((16, 16), f32)
  for k in range(4 * K4):   # fmaf(a, b, acc): one rounding per step; float64 holds the exact product and sum of two float32 terms up to the final rounding
    chain = (A[:, k:k + 1].astype(np.float64) * B[k:k + 1, :].astype(np.float64) + chain.astype(np.float64)).astype(f32)
  np.testing.assert_array_equal(D, chain)


def test_emulator_lane_instructions():
  h = emu_build.load_selftest()
  out = np.zeros((6, 64), np.int32)
  h.selftest_lanes(P(out))
  l = np.arange(64)
  np.testing.assert_array_equal(out[0], (l & ~15) | ((l - 1) & 15))                       # row_ror:1: lane i of a row reads lane i - 1 (mod 16)
  np.testing.assert_array_equal(out[1], np.where(l & 15, l - 1, -1))                      # row_shr:1: lane 0 of a row has no source and keeps `old`
  np.testing.assert_array_equal(out[2], np.full(64, 170))                                  # readlane(17)
  np.testing.assert_array_equal(out[3], np.full(64, 22))                                   # 22 multiples of 3 below 64
  np.testing.assert_array_equal(out[4], l ^ 5)
  np.testing.assert_array_equal(out[5], l ^ 1)                                             # quad_perm [1, 0, 3, 2]


def test_emulator_schedule_perturbation_exposes_a_missing_barrier():
  """IL_EMU_SCHEDULE=reverse|random:<seed> re-orders the waves between two barriers (and the lanes within a wave). A kernel that is missing a barrier gives a different
  answer under one of the orders; the whole file above passes under forward, reverse and random orders. Here: a deliberately racy kernel, in child processes
  (the schedule is read once per process)."""
  import subprocess
  code = ("import sys, numpy as np, ctypes as C; sys.path[:0] = ['tests']\n"
          "from host_emu import build\n"
          "h = build.load_selftest(); out = np.full(64, -1, np.float32)\n"
          "h.selftest_race(C.c_void_p(out.ctypes.data), int(sys.argv[1])); print(int(np.array_equal(out, np.arange(63, -1, -1, dtype=np.float32))))\n")
  run = lambda schedule, barrier: subprocess.run([sys.executable, '-c', code, str(barrier)], env=dict(os.environ, IL_EMU_SCHEDULE=schedule), cwd=os.path.dirname(HERE),
                                                 capture_output=True, text=True, timeout=300).stdout.strip()
  assert run('forward', 1) == '1' and run('reverse', 1) == '1' and run('random:3', 1) == '1'   # with the barrier: every order agrees
  assert run('forward', 0) == '1'                                                                # the race, hidden by the order a simple emulator would use
  assert run('reverse', 0) == '0'                                                                # ... and exposed


def test_emulator_sanitised_build_sees_one_element_past_a_buffer():
  import subprocess
  asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
  if not os.path.isabs(asan) or not os.path.exists(asan):
    pytest.skip('libasan.so not found next to gcc')
  code = ("import sys, numpy as np, ctypes as C; sys.path[:0] = ['tests']\n"
          "from host_emu import build\n"
          "h = build.load_selftest(); a = np.ones(40, np.float32); o = np.zeros(64, np.float32)\n"
          "h.selftest_overrun(C.c_void_p(a.ctypes.data), C.c_void_p(o.ctypes.data), int(sys.argv[1])); print('returned')\n")
  env = dict(os.environ, IL_EMU_ASAN='1', LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:detect_stack_use_after_return=0')
  ok = subprocess.run([sys.executable, '-c', code, '39'], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=600)    # reads in[0..39]: in bounds
  assert ok.returncode == 0 and 'returned' in ok.stdout, ok.stderr[-2000:]
  bad = subprocess.run([sys.executable, '-c', code, '40'], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=600)   # reads in[40]
  assert bad.returncode != 0 and 'heap-buffer-overflow' in bad.stderr, bad.stderr[-2000:]


# ------------------------------------------------------------------------------------------------ gail_deep.hip
@pytest.mark.parametrize('name', [n for n, *_ in gi.GAIL_DEEP_CASES])
def test_emulated_gail_deep_kernels_match_reference(golden_dir, name):
  h = emu('gail_deep')
  g = np.load(os.path.join(golden_dir, 'gail_deep.npz'))
  _, kw, loss, (lr, wd, gp, ent), rf = next(c for c in gi.GAIL_DEEP_CASES if c[0] == name)
  c = gi.gail_deep_case(**kw)
  from oracle import gail_deep as ogd
  ds = ogd.DeepDiscState(c['D'], c['H'], c['depth'], c['activation'], c['spectral_norm'])
  for l in range(c['depth'] + 1):
    ds.W[l][...] = c['W'][l]; ds.b[l][...] = c['b'][l]; ds.u[l][...] = c['u'][l]; ds.v[l][...] = c['v'][l]
  keep = Keep()
  h.il_disc_deep_numel.restype = h.il_disc_deep_workspace_floats.restype = C.c_int64
  Pn = int(h.il_disc_deep_numel(c['D'], c['H'], c['depth']))
  params, sn, grad = keep.arr(ds.pack()), keep.arr(ds.pack_sn()), keep.arr(np.zeros(Pn))
  assert params.size == Pn
  ws = keep.arr(np.full(int(h.il_disc_deep_workspace_floats(c['D'], c['H'], c['depth'], c['B'])), np.nan))
  d = _lib.DiscDeep()
  d.state_dim, d.action_dim, d.hidden, d.batch, d.spectral_norm, d.state_only = c['S'], c['A'], c['H'], c['B'], int(c['spectral_norm']), 0
  d.reward_function, d.loss_function, d.depth, d.activation = REWARD[rf], LOSS[loss], c['depth'], int(c['activation'] == 'tanh')
  d.params, d.sn, d.grad = params.ctypes.data, sn.ctypes.data, grad.ctypes.data
  d.opt, step = np_adam(keep, Pn, lr, wd)
  d.grad_penalty, d.entropy_bonus, d.pos_class_prior, d.workspace, d.workspace_floats = gp, ent, 0.7, ws.ctypes.data, ws.size
  for i in range(len(c['policy'])):
    if i:
      params[...] = g[f'{name}.p_{i}']
      if c['spectral_norm']: sn[...] = g[f'{name}.sn_{i}']
    pb, eb = np_batch(keep, c['policy'][i]), np_batch(keep, c['expert'][i])
    x = _lib.GailExtra(); em = keep.arr(c['eps_mix'][i]); x.eps_mix = em.ctypes.data
    check(h, h.il_gail_deep_step(C.byref(d), C.byref(pb), C.byref(eb), P(keep.arr(c['eps'][i])), C.byref(x), 0, None))
    close(grad, g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1}', rtol=2e-4, atol_scale=2e-6)
    if c['spectral_norm']: close(sn, g[f'{name}.sn_{i + 1}'], f'{name} u / v {i + 1}', rtol=1e-4, atol_scale=1e-6)
    params[...] = g[f'{name}.p_{i + 1}']
    r = keep.arr(np.zeros(c['B']))
    check(h, h.il_gail_deep_reward(C.byref(d), C.byref(pb), P(r), None, None, None))
    close(r, g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=1e-4, atol_scale=1e-5)
  assert int(step[0]) == len(c['policy'])


# ------------------------------------------------------------------------------------------------ gail_shaped.hip (depth-1 ReLU potential)
def _shaped_desc(h, keep, c, sn, loss, margin=float('inf')):
  h.il_disc_shaped_numel.restype = h.il_disc_shaped_workspace_floats.restype = C.c_int64
  from oracle import gail_shaped as ogs
  ods = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, sn)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2', 'ug', 'vg', 'u1', 'v1', 'u2', 'v2'):
    getattr(ods, k)[...] = c[k]
  Pn = int(h.il_disc_shaped_numel(c['S'], c['A'], c['H'], 0))
  params, grad = keep.arr(ods.pack()), keep.arr(np.zeros(Pn))
  bufs = {k: keep.arr(c[k]) for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2')}
  ws = keep.arr(np.full(int(h.il_disc_shaped_workspace_floats(c['S'], c['A'], c['H'], c['B'], 0)), np.nan))
  d = _lib.DiscShaped()
  d.state_dim, d.action_dim, d.hidden, d.batch, d.spectral_norm, d.state_only, d.reward_function, d.loss_function = c['S'], c['A'], c['H'], c['B'], int(sn), 0, 0, LOSS[loss]
  d.params, d.grad = params.ctypes.data, grad.ctypes.data
  for k, a in bufs.items(): setattr(d, k, a.ctypes.data)
  d.opt, step = np_adam(keep, Pn, 1e-3, 0.1)
  d.grad_penalty, d.entropy_bonus, d.pos_class_prior, d.discount, d.workspace, d.workspace_floats = 0.7, 0.01, 0.7, 0.97, ws.ctypes.data, ws.size
  d.pu_clamped, d.nonnegative_margin = int(margin != float('inf')), (0.0 if margin == float('inf') else margin)
  return d, params, grad, bufs, step


@pytest.mark.parametrize('name,sn,loss', [('sn_bce', True, 'BCE'), ('plain_pugail', False, 'PUGAIL')])
def test_emulated_gail_shaped_kernels_match_reference(golden_dir, name, sn, loss):
  h = emu('gail_shaped')
  g = np.load(os.path.join(golden_dir, 'gail_shaped.npz'))
  c = gi.gail_shaped_case(91, 'hopper', 32, 96, 2, sn)
  keep = Keep()
  d, params, grad, bufs, step = _shaped_desc(h, keep, c, sn, loss)
  for i in range(2):
    pb, eb = np_batch(keep, c['policy'][i]), np_batch(keep, c['expert'][i])
    check(h, h.il_gail_shaped_step(C.byref(d), C.byref(pb), C.byref(eb), P(keep.arr(c['eps'][i])), None, 0, None))
    close(grad, g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1}', rtol=1e-4, atol_scale=1e-5)
    if sn:
      for k in bufs: close(bufs[k], g[f'{name}.{k}_{i + 1}'], f'{name} {k} {i + 1}', rtol=1e-4, atol_scale=1e-5)
    params[...] = g[f'{name}.p_{i + 1}']
    r = keep.arr(np.zeros(c['B']))
    check(h, h.il_gail_shaped_reward(C.byref(d), C.byref(pb), P(r), None, None, None))
    close(r, g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=1e-4, atol_scale=1e-5)


# ------------------------------------------------------------------------------------------------ gail_shaped_deep.hip (any potential)
def _shaped_deep_desc(h, keep, c, loss, lr, wd, gp, ent, rf, margin):
  for fn in ('il_disc_shaped_deep_numel', 'il_disc_shaped_deep_sn_numel', 'il_disc_shaped_deep_workspace_floats', 'il_disc_shaped_deep_lds_bytes'):
    getattr(h, fn).restype = C.c_int64
  so = int(c['state_only'])
  Pn = int(h.il_disc_shaped_deep_numel(c['S'], c['A'], c['H'], c['depth'], so))
  assert int(h.il_disc_shaped_deep_lds_bytes(c['S'], c['A'], c['H'], c['depth'], so)) <= 160 * 1024
  ws = keep.arr(np.full(int(h.il_disc_shaped_deep_workspace_floats(c['S'], c['A'], c['H'], c['depth'], c['B'], so)), np.nan))
  params, grad = keep.arr(np.zeros(Pn)), keep.arr(np.zeros(Pn))
  sn = keep.arr(np.zeros(int(h.il_disc_shaped_deep_sn_numel(c['S'], c['A'], c['H'], c['depth'], so))))
  d = _lib.DiscShapedDeep()
  d.state_dim, d.action_dim, d.hidden, d.batch, d.spectral_norm, d.state_only = c['S'], c['A'], c['H'], c['B'], int(c['spectral_norm']), so
  d.reward_function, d.loss_function, d.depth, d.activation = REWARD[rf], LOSS[loss], c['depth'], int(c['activation'] == 'tanh')
  d.params, d.sn, d.grad = params.ctypes.data, sn.ctypes.data, grad.ctypes.data
  d.opt, step = np_adam(keep, Pn, lr, wd)
  d.grad_penalty, d.entropy_bonus, d.pos_class_prior, d.discount, d.workspace, d.workspace_floats = gp, ent, 0.7, 0.97, ws.ctypes.data, ws.size
  d.pu_clamped, d.nonnegative_margin = int(margin != float('inf')), (0.0 if margin == float('inf') else margin)
  return d, params, sn, grad, step


@pytest.mark.parametrize('name', [n for n, *_ in gi.GAIL_SHAPED_DEEP_CASES])
def test_emulated_gail_shaped_deep_kernels_match_reference(golden_dir, name):
  """k_gsd_grad / k_gsd_reduce / k_gsd_reward from their source text against adversarial_imitation_update + predict_reward of the reference (gail_shaped_deep.npz)."""
  from test_oracle_golden import _shaped_deep_state
  h = emu('gail_shaped_deep')
  g = np.load(os.path.join(golden_dir, 'gail_shaped_deep.npz'))
  _, kw, loss, (lr, wd, gp, ent), rf, margin = next(c for c in gi.GAIL_SHAPED_DEEP_CASES if c[0] == name)
  c = gi.gail_shaped_deep_case(**kw)
  ods = _shaped_deep_state(c)
  keep = Keep()
  d, params, sn, grad, step = _shaped_deep_desc(h, keep, c, loss, lr, wd, gp, ent, rf, margin)
  assert list(g[f'{name}.param_names'])[0] == ('g.bias' if c['spectral_norm'] else 'g.weight')
  params[...] = ods.pack(); sn[...] = ods.pack_sn()
  for i in range(len(c['policy'])):
    if i:
      params[...] = g[f'{name}.p_{i}']
      if c['spectral_norm']: sn[...] = g[f'{name}.sn_{i}']
    pb, eb = np_batch(keep, c['policy'][i]), np_batch(keep, c['expert'][i])
    x = _lib.GailExtra(); em = keep.arr(c['eps_mix'][i]); x.eps_mix = em.ctypes.data
    check(h, h.il_gail_shaped_deep_step(C.byref(d), C.byref(pb), C.byref(eb), P(keep.arr(c['eps'][i])), C.byref(x), 0, None))
    close(grad, g[f'{name}.g_{i + 1}'], f'{name} gradient {i + 1}', rtol=2e-4, atol_scale=2e-6)
    if c['spectral_norm']: close(sn, g[f'{name}.sn_{i + 1}'], f'{name} u / v {i + 1}', rtol=1e-4, atol_scale=1e-6)
    assert np.abs(params - g[f'{name}.p_{i + 1}']).max() <= 6e-6
    params[...] = g[f'{name}.p_{i + 1}']
    r, z = keep.arr(np.zeros(c['B'])), keep.arr(np.zeros(c['B']))
    check(h, h.il_gail_shaped_deep_reward(C.byref(d), C.byref(pb), P(r), P(z), None, None))
    close(r, g[f'{name}.reward_{i + 1}'], f'{name} reward {i + 1}', rtol=1e-4, atol_scale=1e-5)
    check(h, h.il_gail_shaped_deep_reward(C.byref(d), C.byref(pb), P(r), None, P(keep.arr(c['logp_policy'][i])), None))
    close(r, g[f'{name}.reward_logp_{i + 1}'], f'{name} reward with a log-policy offset {i + 1}', rtol=1e-4, atol_scale=1e-5)
  assert int(step[0]) == len(c['policy'])   # the PUGAIL value pass does not tick the optimiser


@pytest.mark.parametrize('fixture,prefix,sn,loss,margin_name', [('gail_shaped', 'sn_bce.', True, 'BCE', None), ('gail_shaped', 'plain_pugail.', False, 'PUGAIL', None),
                                                               ('gail_shaped_mixup', '', True, 'Mixup', None), ('gail_pu_margin_general', 'shaped.clamped.', True, 'PUGAIL', 'clamped'),
                                                               ('gail_pu_margin_general', 'shaped.open.', True, 'PUGAIL', 'open')])
def test_emulated_gail_shaped_deep_kernels_on_the_depth1_relu_fixtures(golden_dir, fixture, prefix, sn, loss, margin_name):
  """The general kernels on the depth-1 ReLU potential's reference fixtures (the shape gail_shaped.hip serves in the product): BCE, PUGAIL with and without a finite
  margin on both sides of the clamp, Mixup with fractional terminals."""
  h = emu('gail_shaped_deep')
  g = np.load(os.path.join(golden_dir, fixture + '.npz'))
  seed = {'gail_shaped': 91, 'gail_shaped_mixup': 95, 'gail_pu_margin_general': 93}[fixture]
  c = gi.gail_shaped_case(seed, 'hopper', 32, 96, 2, sn)
  c.update(depth=1, activation='relu', state_only=False)
  em = gi.mixup_draws(1095, 96, 2)
  margin = float(g[prefix + 'margin'][0]) if margin_name else float('inf')
  keep = Keep()
  d, params, snb, grad, step = _shaped_deep_desc(h, keep, c, loss, 1e-3, 0.1, 0.7, 0.01, 'AIRL', margin)
  from oracle import gail_shaped as ogs
  ods = ogs.ShapedState(c['S'], c['A'], c['H'], 0.97, sn)
  for k in ('Wg', 'bg', 'W1', 'b1', 'W2', 'b2'):
    getattr(ods, k)[...] = c[k]
  params[...] = ods.pack()
  snb[...] = np.concatenate([c[k] for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2')])
  for i in range(2):
    pb, eb = np_batch(keep, c['policy'][i]), np_batch(keep, c['expert'][i])
    x = _lib.GailExtra(); e = keep.arr(em[i]); x.eps_mix = e.ctypes.data
    check(h, h.il_gail_shaped_deep_step(C.byref(d), C.byref(pb), C.byref(eb), P(keep.arr(c['eps'][i])), C.byref(x), 0, None))
    close(grad, g[f'{prefix}g_{i + 1}'], f'{prefix} gradient {i + 1}', rtol=1e-4, atol_scale=1e-5)
    if sn:
      close(snb, np.concatenate([g[f'{prefix}{k}_{i + 1}'] for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2')]), f'{prefix} u / v {i + 1}', rtol=1e-4, atol_scale=1e-5)
    params[...] = g[f'{prefix}p_{i + 1}']
    if f'{prefix}reward_{i + 1}' in g:
      r = keep.arr(np.zeros(c['B']))
      check(h, h.il_gail_shaped_deep_reward(C.byref(d), C.byref(pb), P(r), None, None, None))
      close(r, g[f'{prefix}reward_{i + 1}'], f'{prefix} reward {i + 1}', rtol=1e-4, atol_scale=1e-5)


# ------------------------------------------------------------------------------------------------ the `-m gpu` test bodies over the emulated libraries
def _emulated_product(monkeypatch, streams=False):
  """Swaps the product's library handle for the host emulation (the entry points it exports; the real library's host-side functions - abi.hip - otherwise), lets CPU
  tensors through the product's one device guard (_lib.on_device) and hands back tests/test_gpu_parity.py with its GPU-only names bound to the CPU, so that the BODIES
  of the `-m gpu` parity tests - same inputs, same fixtures, same tolerances, the product's own host layer in between - run here on the kernel sources."""
  import torch
  import gpu_util
  import imitation_learning_amd as il
  import test_gpu_parity as tgp
  from imitation_learning_amd import memory as il_memory, training as il_training
  h, real = emu(), _lib.lib()

  class Facade:
    def __getattr__(self, fn_name):
      try:
        fn = getattr(h, fn_name)
      except AttributeError:
        return getattr(real, fn_name)
      fn.restype, fn.argtypes = _lib._SIGNATURES[fn_name]
      if os.environ.get('IL_EMU_LOG_CALLS'):   # which entry points does the emulated suite reach? (one name per line)
        with open(os.environ['IL_EMU_LOG_CALLS'], 'a') as f: f.write(fn_name + '\n')
      return fn

  monkeypatch.setattr(_lib, '_lib', Facade())
  monkeypatch.setattr(_lib, 'stream_ptr', lambda: None)
  monkeypatch.setattr(_lib, 'on_device', lambda t: True)
  monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
  monkeypatch.setattr(gpu_util, 'DEV', 'cpu')
  for k in ('DEV', 'N', 'T', 'Cfg', 'bracket', 'close', 'close_params', 'crit_from_flat', 'fill_memory', 'make_disc', 'make_sac', 'make_sac_oracle', 'tbatch'):
    monkeypatch.setattr(tgp, k, getattr(gpu_util, k), raising=False)
  for k, v in (('il', il), ('_lib', _lib), ('il_memory', il_memory), ('il_training', il_training)):
    monkeypatch.setattr(tgp, k, v, raising=False)
  monkeypatch.setattr(il_training, '_WS', {})      # CPU arenas of this test only
  monkeypatch.setattr(il_training, '_NOISE', {})
  if streams:
    # Two-stream schedules (UpdatePlan): torch's streams become emulated stream handles - launches on them are queued, run side by side with co-resident workgroups
    # (emu_hip.hpp) and finish at a synchronisation point. The per-function bodies above stay on the null stream: synchronous, as their host-side reads expect.
    import contextlib
    h.emu_stream_wait.argtypes = [C.c_size_t, C.c_size_t]

    class Stream:
      count = [0]
      def __init__(self, *a, priority=0, **k):
        Stream.count[0] += 1
        # (a priority stream - the acting worker's - is synchronous here: its host side polls a mailbox the launch writes, and nothing else would run the launch)
        self.cuda_stream = 0 if priority else 0x1000 * Stream.count[0]
      def wait_stream(self, other): h.emu_stream_wait(self.cuda_stream, other.cuda_stream)
      def synchronize(self): h.emu_drain()

    # The caller's stream is the null stream: a launch on it runs at once, together with whatever the other streams have queued (host code between two launches -
    # torch ops on CPU tensors standing in for stream-ordered device ops - then always sees finished results); only streams the product creates itself are asynchronous.
    main = Stream(); main.cuda_stream = 0
    current = [main]

    @contextlib.contextmanager
    def use(s):
      current.append(s)
      try: yield
      finally: current.pop()

    class Graph:   # torch.cuda.CUDAGraph: a recorded launch sequence (emu_hip.hpp: kernel arguments by value, forks / joins between the captured streams)
      count = [0]
      def __init__(self, *a, **k):
        Graph.count[0] += 1; self.id = Graph.count[0]
      def replay(self): h.emu_graph_launch(self.id, current[-1].cuda_stream)

    @contextlib.contextmanager
    def capture(g, pool=None, stream=None, **k):   # torch.cuda.graph(g, stream=...)
      h.emu_drain()
      s = stream if stream is not None else Stream()
      current.append(s)
      h.emu_capture_begin(g.id, s.cuda_stream)
      try: yield
      finally:
        h.emu_capture_end(g.id); current.pop()

    for fn, args in (('emu_capture_begin', [C.c_uint64, C.c_size_t]), ('emu_capture_end', [C.c_uint64]), ('emu_graph_launch', [C.c_uint64, C.c_size_t])):
      getattr(h, fn).argtypes = args
    monkeypatch.setattr(torch.cuda, 'CUDAGraph', Graph)
    monkeypatch.setattr(torch.cuda, 'graph', capture)

    monkeypatch.setattr(torch.Tensor, 'pin_memory', lambda self, *a, **k: self)   # pinned host words (time-out flags the host polls): plain host memory here
    zeros = torch.zeros
    monkeypatch.setattr(torch, 'zeros', lambda *a, pin_memory=False, **k: zeros(*a, **k))   # the acting mailbox is pinned host memory: plain host memory here

    class Props: multi_processor_count = 256
    monkeypatch.setattr(_lib, 'stream_ptr', lambda: C.c_void_p(current[-1].cuda_stream))
    monkeypatch.setattr(torch.cuda, 'Stream', Stream)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: current[-1])
    monkeypatch.setattr(torch.cuda, 'stream', use)
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: h.emu_drain())
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: bool(h.emu_is_capturing()))
    monkeypatch.setattr(torch.cuda, 'device', lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, 'get_device_properties', lambda *a, **k: Props())
    monkeypatch.setattr(il_training.UpdatePlan, '_probe_device_sync', lambda self, graph: True)   # "the two streams run concurrently": true of the emulated ones
    monkeypatch.setattr(il_training.UpdatePlan, '_probe_streams', lambda self, waiter, setter: True)
    host_copy = gpu_util.N
    monkeypatch.setattr(gpu_util, 'N', lambda t: (h.emu_drain(), host_copy(t))[1])
    monkeypatch.setattr(tgp, 'N', gpu_util.N, raising=False)
  return tgp


# the `-m gpu` parity tests whose bodies run here: everything that goes through the per-function entry points (the captured two-graph plans, the acting worker's
# mailbox and the multi-process paths need streams, graphs or a second process, i.e. a GPU)
GPU_BODIES = (
    'test_replay_matches_reference_bit_exact', 'test_transfer_transitions_matches_sequential_appends', 'test_replay_full_size_gather_property',
    'test_sac_update_matches_oracle_and_reference', 'test_sac_gradients_match_oracle', 'test_sac_update_other_shapes', 'test_general_shape_sac_matches_oracle_and_reference',
    'test_device_beta_draws_are_beta_distributed',
    'test_bc_update_matches_oracle_and_reference', 'test_actor_act_matches_oracle', 'test_adam_and_polyak_kernels',
    'test_gail_update_matches_oracle_and_reference', 'test_gail_loss_variants_match_reference', 'test_gail_ragged_batch_and_state_only',
    'test_gmmil_matches_oracle_and_reference', 'test_gmmil_full_size_properties', 'test_gmmil_centred_gram_form_is_as_close_to_float64_as_the_direct_form', 'test_pwil_matches_oracle_and_reference', 'test_pwil_every_launch_path_matches_oracle',
    'test_reward_relabeller_bit_exact', 'test_mix_expert_agent_transitions_bit_exact',
    'test_red_matches_reference', 'test_dril_matches_reference', 'test_every_shipped_red_dril_shape_runs_at_ant_dims', 'test_dril_onchip_masks_are_bernoulli_and_change_per_call',
    'test_general_shape_tile_engine_matches_oracle_at_block_batches',
    'test_gail_deep_discriminator_matches_reference', 'test_gail_deep_pugail_finite_margin_matches_reference',
    'test_gail_reward_shaping_matches_reference', 'test_gail_reward_shaping_mixup_matches_reference', 'test_gail_shaped_pugail_finite_margin_matches_reference',
    'test_gail_reward_shaping_general_potential_matches_reference', 'test_red_dril_shaped_at_ant_dims_match_oracle',
)


def _gpu_bodies():
  """(function name, keyword arguments) for every parametrisation the GPU test itself declares (its own pytest.mark.parametrize marks are read, not restated)."""
  import itertools
  import test_gpu_parity as tgp
  out = []
  for name in GPU_BODIES:
    fn = getattr(tgp, name)
    axes = []
    for m in getattr(fn, 'pytestmark', []):
      if m.name == 'parametrize':
        names = [n.strip() for n in m.args[0].split(',')]
        axes.append([dict(zip(names, v if len(names) > 1 else (v,))) for v in m.args[1]])
    for combo in itertools.product(*axes) if axes else [()]:
      kw = {}
      for part in combo: kw.update(part)
      out.append((name, kw))
  return out


def _body_id(name, kw):
  first = next(iter(kw.values()), '')
  return f'{name[5:]}-{first}' if kw else name[5:]


@pytest.mark.parametrize('body,kw', _gpu_bodies(), ids=[_body_id(n, kw) for n, kw in _gpu_bodies()])
def test_gpu_parity_bodies_on_the_emulated_kernels(golden_dir, monkeypatch, body, kw):
  """The replay ring and the device index draw, `sac_update` (incl. the headline shape: HalfCheetah, hidden 256, batch 256) and its gradients, behavioural cloning, the
  acting forward, AdamW / polyak, the GAIL discriminator with every loss variant, GMMIL, PWIL, the relabellers, RED, DRIL and the general discriminators: the parity
  tests the GPU runs, at the GPU's tolerances (rtol 1e-5 .. 2e-5, float64 brackets, bit equality for rows and indices), on the kernel sources executed by the emulator."""
  tgp = _emulated_product(monkeypatch)
  fn = getattr(tgp, body)
  if 'golden_dir' in fn.__code__.co_varnames[:fn.__code__.co_argcount]: kw = dict(kw, golden_dir=golden_dir)
  fn(**kw)


@pytest.mark.parametrize('body,kw', [('test_inline_relabel_heads_equal_the_reward_kernel', dict(reward_function='GAIL')), ('test_inline_relabel_heads_equal_the_reward_kernel', dict(reward_function='FAIRL')),
                                     ('test_batch_gather_is_rejected_where_it_is_not_honoured', {}),
                                     ('test_gail_pugail_finite_margin_matches_reference', dict(name='clamped')), ('test_gail_pugail_finite_margin_matches_reference', dict(name='open')),
                                     ('test_update_plan_mixup_with_beta_coefficients_drawn_on_the_device', dict(alpha=0.4)), ('test_update_plan_mixup_with_beta_coefficients_drawn_on_the_device', dict(alpha=2.5))],
                         ids=['inline_relabel-GAIL', 'inline_relabel-FAIRL', 'gather_rejected', 'pugail_margin-clamped', 'pugail_margin-open', 'mixup_beta-0.4', 'mixup_beta-2.5'])
def test_gpu_parity_bodies_that_build_a_plan_on_the_emulated_kernels(golden_dir, monkeypatch, body, kw):
  """More bodies of tests/test_gpu_parity.py: the ones that construct an UpdatePlan (its second stream, the device-sync probe) around what they check."""
  tgp = _emulated_product(monkeypatch, streams=True)
  fn = getattr(tgp, body)
  if 'golden_dir' in fn.__code__.co_varnames[:fn.__code__.co_argcount]: kw = dict(kw, golden_dir=golden_dir)
  fn(**kw)


SLOW = pytest.mark.skipif(os.environ.get('IL_EMU_SLOW', '0') != '1', reason='IL_EMU_SLOW=1: the heavier plan-level bodies (12 - 40 s each; all passed at the end of round 3)')


@SLOW
@pytest.mark.parametrize('body,args', [('test_update_plan_graph_replay_equals_eager', ('SAC',)), ('test_update_plan_graph_replay_equals_eager', ('GAIL',)),
                                       ('test_capture_warmup_runs_on_the_probed_stream_pair', (0,)), ('test_update_plan_host_and_device_index_draws_agree', ()),
                                       ('test_batched_population_equals_independent_learners', ()), ('test_data_parallel_path_equals_fused_path_on_one_rank', ('none',)),
                                       ('test_data_parallel_path_equals_fused_path_on_one_rank', ('peer_windows',)), ('test_data_parallel_path_equals_fused_path_on_one_rank', ('peer_windows_in_apply',))])
def test_plan_level_gpu_bodies_on_the_emulated_kernels_slow(monkeypatch, body, args):
  """Verbatim bodies of tests/test_gpu_parity.py that capture and replay graphs: graph replay == eager for SAC and GAIL, the capture warm-up, host vs device index draws,
  three learners batched vs independent, and DataParallelUpdate with one rank - eager AND as captured graphs - for the three exchange forms."""
  import inspect
  tgp = _emulated_product(monkeypatch, streams=True)
  monkeypatch.setenv('IL_PEER_SOAK_ROUNDS', '0')
  fn = getattr(tgp, body)
  fn(monkeypatch, *args) if 'monkeypatch' in inspect.signature(fn).parameters else fn(*args)


def _timed_path_modules(monkeypatch, tgp):
  import bench
  import gpu_util
  import test_timed_path_oracle as tt
  from imitation_learning_amd import training as il_training
  for k in ('DEV', 'N', 'Cfg', 'bracket', 'close', 'close_params', 'close_sparse', 'crit_from_flat'):
    monkeypatch.setattr(tt, k, getattr(gpu_util, k), raising=False)
  for k, v in (('il', tgp.il), ('_lib', _lib), ('il_training', il_training), ('bench', bench)):
    monkeypatch.setattr(tt, k, v, raising=False)
  return tt, bench


def test_timed_path_replays_through_the_oracle_on_the_emulated_kernels(monkeypatch):
  """What bench.py times (tests/test_timed_path_oracle.py on the GPU, fewer replays): the UpdatePlan at the BASELINE configuration - HalfCheetah dims, batch 256, ring 1e6 /
  fill 1e5, 25k expert rows, device MT19937 draws by the sampler workgroup riding in the discriminator launch, rows read from the rings through the indices, on-chip
  Philox noise, inline relabel - CAPTURED as its two graphs (emulated stream capture: launches recorded with their arguments by value) and replayed on two emulated
  streams, the branches' workgroups co-resident and handing over through device counters; recorded, and replayed through oracle.replay -> gail_update -> predict_reward
  -> sac_update: index draws bit-exact, rewards bracketed in float64, log pi / Q / every parameter and Adam moment at the GPU's bounds, no device-side wait expired."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import torch
  tt, bench = _timed_path_modules(monkeypatch, tgp)
  WARM, K, SEED = 1, 2, 3
  plan, nets, (tr, et) = bench.build(torch.device('cpu'), 0, seed=SEED)
  o = tt.OracleLearner(nets, plan, tr, et, index_seed=SEED)
  om = tt.OracleLearner(nets, plan, tr, et, index_seed=SEED)   # replays with the kernels' own ReLU decisions (il_sac.debug_masks)
  plan.record_relu_masks()
  for k in range(WARM):              # WARM eager updates (they count), then the two graphs
    plan.run(); torch.cuda.synchronize()
    o.update(k); om.update(k, masks=tt.relu_masks(plan))
  plan.capture(warmup=0)
  assert plan.device_sync and plan.ring_mode and plan.inline_relabel and plan.resident_sampler and plan.graph_side is not None, 'this must be the schedule bench.py times'
  tt.compare_learner(o, nets, plan, WARM, 'eager warm-up: ')
  _lib.check(_lib.lib().il_kernel_stamps_clear())   # (launches of earlier tests in this process: other grids)
  for k in range(WARM, WARM + K):
    plan.replay()
    torch.cuda.synchronize()
    got = tt.per_update_outputs(plan)
    tt.compare_outputs(got, o.update(k), k)
    om.update(k, masks=tt.relu_masks(plan))
    tt.reward_bracket(o, nets[4], got[2], k)
  assert plan.sync_timeouts() == 0
  # the launch stamps bench.py builds its roofline from (il_kernel_stamps): the six launches of the last replay, every workgroup stamped, begin before end
  st = _lib.kernel_stamps()
  assert set(st) >= {'k_gail_grad', 'k_gail_reduce', 'k_sac_chain_pair', 'k_dw_adam_critic', 'k_policy_critic_pair', 'k_dw_adam_actor'}, sorted(st)
  assert all(v['workgroups'] > 0 and v['begin_us'] < v['end_us'] and v['begin_us'] <= v['last_begin_us'] and v['first_end_us'] <= v['end_us'] for v in st.values()), st
  assert st['k_sac_chain_pair']['workgroups'] >= 10 * 16 and st['k_dw_adam_critic']['begin_us'] > st['k_sac_chain_pair']['begin_us']
  _lib.check(_lib.lib().il_kernel_stamps_clear()); assert _lib.kernel_stamps() == {}
  tt.compare_learner_masked(om, nets, plan, WARM + K)
  tt.compare_learner(o, nets, plan, WARM + K)


def test_direct_launches_equal_the_graph_replays_on_the_emulated_kernels(monkeypatch):
  tgp = _emulated_product(monkeypatch, streams=True)
  tt, bench = _timed_path_modules(monkeypatch, tgp)
  tt.test_direct_launches_equal_the_graph_replays(K=2, seed=9)


def test_overlapped_launches_equal_the_graph_replays_on_the_emulated_kernels(monkeypatch):
  """IL_MAIN_OVERLAP=1: il_sac_update_gather_overlap's four launches on two emulated streams (forward / critic loss and policy / critic on the caller's, the optimiser
  launches on another), co-resident with their predecessors and handing over through tickets, stage epochs and per-workgroup flag lines: the bits of the in-order replays."""
  tgp = _emulated_product(monkeypatch, streams=True)
  tt, bench = _timed_path_modules(monkeypatch, tgp)
  monkeypatch.setenv('IL_MAIN_OVERLAP', '1')
  tt.test_direct_launches_equal_the_graph_replays(K=3, seed=9, expect_overlap=True)


def test_expired_wait_poisons_the_learner_on_the_emulated_kernels(monkeypatch):
  """[IL_SYNC_POISON]: tests/test_timed_path_oracle.py's body - a branch launched without its partner under a four-poll bound - on the kernel sources: the optimiser
  epilogues of k_dw_adam / k_gail_reduce skip every store, the flags are raised, the next launch raises on the host."""
  tgp = _emulated_product(monkeypatch, streams=True)
  tt, bench = _timed_path_modules(monkeypatch, tgp)
  tt.test_expired_wait_poisons_the_learner_and_the_weights_stay()


def _update_plan_cases():
  """(The GAIL discriminator variants run torch operations - log-probability buffers, the reward copy - between their launches inside the plan; the emulator defers kernels on its
  streams but cannot defer torch's CPU operations with them, so those cases need a GPU.)"""
  import test_update_plans_gpu as tp
  return [c for c in tp.CASES if c[0] != 'GAIL']


@pytest.mark.parametrize('algorithm,mixed,bc_aux,kw', _update_plan_cases(), ids=[f'{c[0]}{"-mixed" if c[1] else ""}{"-bc_aux" if c[2] else ""}{"-" + "-".join(f"{k}={v}" for k, v in c[3].items()) if c[3] else ""}' for c in _update_plan_cases()])
def test_update_plan_of_every_algorithm_on_the_emulated_kernels(monkeypatch, algorithm, mixed, bc_aux, kw):
  """tests/test_update_plans_gpu.py's body: the captured UpdatePlan of GMMIL, RED, DRIL (on-chip dropout masks), AdRIL / SQIL (the relabeller's device-side round
  arithmetic), PWIL and SAC with mixed batches / the BC auxiliary step - one eager update, then graph replays - bit-identical to the per-function sequence of train.py."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import gpu_util
  import test_update_plans_gpu as tp
  from imitation_learning_amd import training as il_training
  for k in ('DEV', 'N', 'Cfg', 'fill_memory'):
    monkeypatch.setattr(tp, k, getattr(gpu_util, k), raising=False)
  for k, v in (('il', tgp.il), ('_lib', _lib), ('il_training', il_training)):
    monkeypatch.setattr(tp, k, v, raising=False)
  tp.test_plan_of_every_algorithm_equals_the_per_function_sequence(algorithm, mixed, bc_aux, kw)


def test_gmmil_centred_gram_kernel_on_random_shapes_on_the_emulated_kernels(monkeypatch):
  """k_gmmil_mfma over a sweep of shapes the fixed cases do not hit: row counts around the 64-row / 128-column block edges, every feature-group count (D <= 32, <= 64, <= 128),
  whole-lane and element-wise operand loads, state_only, zero weights, a constant offset - against float64 at the GMMIL bound, twice per shape (self-resetting tickets)."""
  tgp = _emulated_product(monkeypatch)
  import torch
  from imitation_learning_amd import training as il_training
  rs = np.random.RandomState(77)
  cases = [(1, 1, 4, 0), (63, 129, 8, 4), (65, 127, 33, 7), (128, 64, 60, 4), (70, 200, 100, 28), (64, 64, 124, 4), (33, 257, 17, 0), (130, 1, 3, 1)]
  for n1, n2, S, A in cases:
    D = S + A
    off = float(rs.choice([0.0, 30.0]))
    X = (rs.standard_normal((n1, D)) + off).astype(np.float32); E = (rs.standard_normal((n2, D)) * 0.7 + 0.3 + off).astype(np.float32)
    w = rs.uniform(0.2, 1.5, n1).astype(np.float32); we = rs.uniform(0.2, 1.5, n2).astype(np.float32)
    if n2 > 4: we[rs.randint(0, n2, 2)] = 0.0
    state_only = A == 0
    disc = tgp.il.GMMILDiscriminator(S, max(A, 1), tgp.Cfg(state_only=state_only))
    d64 = lambda a, b: ((a.astype(np.float64)[:, None, :] - b.astype(np.float64)[None, :, :]) ** 2).mean(2)
    dxe, dxx = d64(X, E), d64(X, X)
    g1, g2 = float(np.float32(1.0 / (np.median(dxe) + 1e-8))), float(np.float32(0.5 / (np.median(dxx) + 1e-3)))
    disc.gamma_1, disc.gamma_2 = g1, g2
    wn, wen = w.astype(np.float64) / w.astype(np.float64).sum(), we.astype(np.float64) / we.astype(np.float64).sum()
    sim64 = sum(wn * (np.exp(-gm * dxe) @ wen) for gm in (g1, g2)); self64 = sum(wn * (np.exp(-gm * dxx) @ wn) for gm in (g1, g2))
    Tn = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    act = lambda M: Tn(M[:, S:]) if A else Tn(np.zeros((M.shape[0], 1), np.float32))
    for _ in range(2):
      r, sim, self_sim = il_training.gmmil_predict_reward(disc, Tn(X[:, :S]), act(X), Tn(E[:, :S]), act(E), Tn(w), Tn(we), return_parts=True)
    bound = 1e-5 * max(np.abs(sim64).max(), np.abs(self64).max())
    got = [tgp.N(t) for t in (sim, self_sim, r)]
    for name, a, b in zip(('similarity', 'self similarity', 'reward'), got, (sim64, self64, sim64 - self64)):
      assert np.abs(a - b).max() <= bound, (n1, n2, S, A, off, name, np.abs(a - b).max() / bound)


@pytest.mark.parametrize('algorithm,mixed', [('SAC', False), ('DRIL', False)])
def test_data_parallel_bc_aux_on_the_emulated_kernels(monkeypatch, algorithm, mixed):
  """tests/test_update_plans_gpu.py::test_data_parallel_bc_aux_equals_the_plain_plan_on_one_rank on the emulated kernels: the BC auxiliary step of a data-parallel update
  (gradient launch -> bucket mean -> il_adam_step) against the plain plan's fused step, bit for bit, eager and captured."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import gpu_util
  import test_update_plans_gpu as tp
  from imitation_learning_amd import training as il_training
  for k in ('DEV', 'N', 'Cfg', 'fill_memory'):
    monkeypatch.setattr(tp, k, getattr(gpu_util, k), raising=False)
  for k, v in (('il', tgp.il), ('_lib', _lib), ('il_training', il_training)):
    monkeypatch.setattr(tp, k, v, raising=False)
  tp.test_data_parallel_bc_aux_equals_the_plain_plan_on_one_rank(algorithm, mixed)


@pytest.mark.parametrize('algorithm,mixed,bc_aux,kw', [('GMMIL', True, False, {}), ('RED', True, True, {}), ('DRIL', False, False, {}), ('AdRIL', False, False, dict(balanced=True)), ('SAC', False, True, {})])
def test_one_stream_plans_as_direct_launches_on_the_emulated_kernels(monkeypatch, algorithm, mixed, bc_aux, kw):
  """tests/test_update_plans_gpu.py::test_one_stream_plans_as_direct_launches_equal_the_per_function_sequence: the one-stream plans recorded once and re-issued as library calls
  (no hipGraph), bit-identical to the per-function sequence of train.py."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import gpu_util
  import test_update_plans_gpu as tp
  from imitation_learning_amd import training as il_training
  for k in ('DEV', 'N', 'Cfg', 'fill_memory'):
    monkeypatch.setattr(tp, k, getattr(gpu_util, k), raising=False)
  for k, v in (('il', tgp.il), ('_lib', _lib), ('il_training', il_training)):
    monkeypatch.setattr(tp, k, v, raising=False)
  tp.test_plan_of_every_algorithm_equals_the_per_function_sequence(algorithm, mixed, bc_aux, kw, launch='direct')


@pytest.mark.parametrize('algorithm,mixed,bc_aux,nets_name', [('SAC', False, True, 'd3_tanh'), ('GMMIL', True, False, 'mixed'), ('AdRIL', False, False, 'mixed')])
def test_general_shape_update_plan_on_the_emulated_kernels(monkeypatch, algorithm, mixed, bc_aux, nets_name):
  """UpdatePlan for actor / critic shapes outside the fused kernels (csrc/general.hip on one stream, captured as one graph): bit-identical to the per-function sequence."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import gpu_util
  import test_update_plans_gpu as tp
  from imitation_learning_amd import training as il_training
  for k in ('DEV', 'N', 'Cfg', 'fill_memory'):
    monkeypatch.setattr(tp, k, getattr(gpu_util, k), raising=False)
  for k, v in (('il', tgp.il), ('_lib', _lib), ('il_training', il_training)):
    monkeypatch.setattr(tp, k, v, raising=False)
  tp.test_general_shape_plan_equals_the_per_function_sequence(algorithm, mixed, bc_aux, nets_name)


@pytest.mark.parametrize('absorbing', [True, False])
@pytest.mark.parametrize('schedule', ['exact', 'fused', 'overlap'])
def test_acting_worker_on_the_emulated_kernels(monkeypatch, absorbing, schedule):
  """tests/test_gpu_parity.py::test_acting_worker_matches_separate_calls: the one-launch-per-env-step worker (mailbox, device-side cursor, absorbing wraps, ring
  wrap-around) against the per-function path - same actions, bit-identical ring. Round 3: the emulator, whose lanes do not run in lockstep, failed the append-only
  schedules here: k_act_step moved the ring cursor before every wave had read it (no barrier on that path; harmless within one wave on the device, a latent race for rows
  wider than a wave - Ant's 240 floats). The store now sits behind the kernel's barrier."""
  tgp = _emulated_product(monkeypatch, streams=True)
  tgp.test_acting_worker_matches_separate_calls(absorbing, schedule)


@pytest.mark.parametrize('schedule', ['exact', 'fused'])
def test_acting_launch_replays_through_the_oracle_on_the_emulated_kernels(monkeypatch, schedule):
  """tests/test_timed_path_oracle.py::test_acting_launch_replays_through_the_oracle: il_act_step against ReplayOracle + oracle.nets with the recorded Philox draws."""
  tgp = _emulated_product(monkeypatch, streams=True)
  tt, _ = _timed_path_modules(monkeypatch, tgp)
  tt.test_acting_launch_replays_through_the_oracle(schedule)


def test_acting_worker_greedy_and_loud_failure_on_the_emulated_kernels(monkeypatch):
  _emulated_product(monkeypatch, streams=True).test_acting_worker_greedy_and_loud_failure()


def test_population_launches_equal_independent_learners_on_the_emulated_kernels(monkeypatch):
  """tests/test_gpu_parity.py::test_batched_population_equals_independent_learners with two learners and two updates: the il_*_population launches (learner = a grid
  dimension, the 512-thread `_pop` tile kernels, k_dw_adam_pop) against the same learners advanced one by one - bit-identical, and the learners differ."""
  tgp = _emulated_product(monkeypatch, streams=True)
  results = []
  for batched in (False, True):
    plans, nets_all = tgp._population_learners(2)
    if batched:
      pop = tgp.il.BatchedPopulationPlan(plans)
      for _ in range(2): pop.run()
    else:
      for _ in range(2):
        for p in plans: p.run()
    results.append(tgp._population_state(plans, nets_all))
  for l, (a_l, b_l) in enumerate(zip(*results)):
    for i, (a, b) in enumerate(zip(a_l, b_l)):
      assert np.isfinite(a).all()
      np.testing.assert_array_equal(a, b, err_msg=f'learner {l}, tensor {i}')
  assert not np.array_equal(results[0][0][0], results[0][1][0])


@pytest.mark.parametrize('hidden,env_name', [(128, 'hopper'), (64, 'ant'), (192, 'walker2d')])
def test_population_launches_at_other_shapes_on_the_emulated_kernels(monkeypatch, hidden, env_name):
  """The population launches away from the headline shape (hidden 64 / 128 / 192, Hopper / Ant / Walker2d dims): two SAC learners, two updates, batched against one by one,
  every network incl. the target bit-identical. The optimiser launches derive the twin critic's layout (network stride, the H x H layers' ranges the folded target step
  owns) from these dims: a wrong range shows up as a target that differs."""
  tgp = _emulated_product(monkeypatch, streams=True)
  gi, il, torch = tgp.gi, tgp.il, tgp.torch

  def learners(n):
    tgp.il_training._NOISE.clear(); tgp.il_training._WS.clear()
    plans, nets_all = [], []
    for l in range(n):
      S, A = gi.DIMS[env_name]
      torch.manual_seed(30 + l)
      cfg = tgp.Cfg(hidden_size=hidden, depth=2, activation='relu')
      actor, critic = il.SoftActor(S, A, cfg, device=tgp.DEV), il.TwinCritic(S, A, cfg, device=tgp.DEV)
      target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=tgp.DEV)
      ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
      mem = il.ReplayMemory(20000, S, A, True, device=tgp.DEV); tgp.fill_memory(mem, gi.transitions(np.random.RandomState(30 + l), 5000, S, A), 5000)
      mem.index_rng = il.IndexStream(100 + l)
      plans.append(il.UpdatePlan('SAC', actor, critic, log_alpha, target, mem, ao, co, to, 256, 0.97, -0.5 * A, 0.99, overlap=False, learner_id=l))
      nets_all.append((actor, critic, target, log_alpha))
    return plans, nets_all

  results = []
  for batched in (False, True):
    plans, nets_all = learners(2)
    if batched:
      pop = il.BatchedPopulationPlan(plans)
      for _ in range(2): pop.run()
    else:
      for _ in range(2):
        for p in plans: p.run()
    results.append([[tgp.N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [tgp.N(p.idx), tgp.N(p.logp)] for nets, p in zip(nets_all, plans)])
  for l, (a_l, b_l) in enumerate(zip(*results)):
    for i, (a, b) in enumerate(zip(a_l, b_l)):
      assert np.isfinite(a).all()
      np.testing.assert_array_equal(a, b, err_msg=f'learner {l}, tensor {i}')
  assert not np.array_equal(results[0][0][0], results[0][1][0])


@pytest.mark.parametrize('env', [dict(IL_POP_DW_LDS='0'), dict(IL_POP_FUSE_POLYAK='0')], ids=['no 64 x 64 blocks', 'target step in the tail'])
def test_population_dw_switches_on_the_emulated_kernels(env):
  """The population's optimiser launches under their two process-wide switches (read once per process: a child pytest each): without the 64 x 64 blocks (IL_POP_DW_LDS=0)
  the target step must stay whole in the actor launch's tail - round 5 folded the H x H layers' share into those blocks, and a tail that skipped them while no block had
  stepped them left the target's hidden layers frozen (caught on the GPU by test_population_launch_switches_are_bit_identical; this is the no-GPU guard) - and with the
  folding switched off the launches must still equal independent learners."""
  import subprocess, sys
  r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k', 'test_population_launches_equal_independent_learners_on_the_emulated_kernels'],
                     env=dict(os.environ, **env), cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
  assert r.returncode == 0 and '1 passed' in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


_PLAIN_PLAN_RESULT = []


@pytest.mark.parametrize('name,env', [
                                      ('exchange launches', dict(IL_PEER_EXCHANGE='require', IL_PEER_APPLY='0', IL_DP_FUSED='0')),
                                      ('exchange in the optimiser launches', dict(IL_PEER_EXCHANGE='require', IL_DP_FUSED='1'))])
def test_data_parallel_schedules_equal_the_plain_plan_on_the_emulated_kernels(monkeypatch, name, env):
  """parallel.DataParallelUpdate with a world of one rank (the eager half of tests/test_gpu_parity.py::test_data_parallel_path_equals_fused_path_on_one_rank and of
  tests/test_parallel_gpu.py's fused variant): the IL_FLAG_GRADS_ONLY kernels, the peer window set up and soak-tested through the emulated library, the gradients
  travelling through the window's slot and back - as exchange launches, or inside the launches that produce them (k_dw_adam_peer's block jobs,
  k_gail_reduce: the schedule a multi-GPU run uses by default) - must evolve the learner bit for bit like the plain UpdatePlan."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import torch
  from imitation_learning_amd import training as il_training
  from imitation_learning_amd.parallel import DataParallelUpdate
  # (the soak interleaves torch ops with the exchange launches on three streams and relies on the device ordering them per stream; CPU torch ops run at once, so here it
  # would compare before the exchange ran - vacuous with one rank - and free its temporaries under the queued launches: off. `verify` runs, on the synchronous stream.)
  monkeypatch.setenv('IL_PEER_SOAK_ROUNDS', '0')
  outs = [_PLAIN_PLAN_RESULT[0]] if _PLAIN_PLAN_RESULT else []
  for dp in ((True,) if outs else (False, True)):
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    tgp.il.seed(21); il_training._NOISE.clear()
    plan, nets = tgp._make_plan('GAIL', 13)
    runner = DataParallelUpdate(plan) if dp else plan
    for _ in range(2):
      runner.run()
    torch.cuda.synchronize()
    assert (runner.exchange_timeouts() if dp else plan.sync_timeouts()) == 0
    if dp:
      assert (runner.peer is not None) == (env['IL_PEER_EXCHANGE'] != '0') and bool(runner.fused) == (env.get('IL_DP_FUSED') == '1'), (runner.exchange_name(), runner.fused)
    outs.append([tgp.N(n.flat if hasattr(n, 'flat') else n) for n in nets] + [tgp.N(plan.logp), tgp.N(plan.q), tgp.N(plan.rewards)])
    if not dp: _PLAIN_PLAN_RESULT.append(outs[0])   # the plain plan does not depend on the exchange settings: run once per session
  for i, (a, b) in enumerate(zip(*outs)):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b, err_msg=f'{name}: tensor {i}')


def _two_data_parallel_ranks(monkeypatch, fused, updates=2, world=2):
  """Two data-parallel ranks in ONE process: two learners built like bench.py's (the same networks, their own replay shard, index stream, noise stream and scratch), a
  DataParallelUpdate each, every bucket of their peer exchanges re-homed into windows both ranks point at (what PeerExchange does with hipIpc handles between processes), and
  each rank's update enqueued on emulated streams of its own: the four branches' workgroups are co-resident and hand over through the windows' arrival words."""
  tgp = _emulated_product(monkeypatch, streams=True)
  import bench
  import torch
  from imitation_learning_amd.parallel import DataParallelUpdate
  monkeypatch.setenv('IL_PEER_SOAK_ROUNDS', '0'); monkeypatch.setenv('IL_PEER_EXCHANGE', 'require'); monkeypatch.setenv('IL_DP_FUSED', '1' if fused else '0')
  L, W, keep = _lib.lib(), world, []
  ranks = []
  for r in range(W):
    plan, nets, _ = bench.build(torch.device('cpu'), r, seed=7, learner_id=r)
    runner = DataParallelUpdate(plan)
    runner._warm_collectives()          # sets the exchange up (a world of one) without running an update
    ranks.append((plan, nets, runner))
  assert all(bool(runner.fused) == fused for _, _, runner in ranks)
  names = [k for k in ranks[0][2].peer.desc if not k.startswith('_')]
  d0, offs, total = ranks[0][2].peer.desc, {}, 0
  for k in names:
    offs[k] = total
    total += int(L.il_peer_job_region_bytes(W, d0[k].n, d0[k].n_jobs) if d0[k].n_jobs else L.il_peer_region_bytes(W, d0[k].n))
  windows = [np.zeros(total // 4 + 64, f32) for _ in range(W)]
  base = [w.ctypes.data + (-w.ctypes.data) % 256 for w in windows]
  status = []
  for r, (plan, nets, runner) in enumerate(ranks):
    for k in names:
      d = runner.peer.desc[k]
      ep, st = np.zeros(d.n_jobs if d.n_jobs else (d.n + 2047) // 2048, np.uint32), np.zeros(2, np.int64)
      keep += [ep, st]; status.append(st)
      d.rank, d.world, d.window_offset, d.epoch, d.status = r, W, offs[k], ep.ctypes.data, st.ctypes.data
      for q in range(W): d.windows[q] = base[q]
  mains = [torch.cuda.Stream() for _ in range(W)]
  for _ in range(updates):
    for r, (plan, nets, runner) in enumerate(ranks):
      with torch.cuda.stream(mains[r]):
        runner.run()
    torch.cuda.synchronize()
  assert all(int(st[0]) == 0 for st in status) and all(plan.sync_timeouts() == 0 for plan, _, _ in ranks), 'a device-side wait expired'
  return [[tgp.N(n.flat if hasattr(n, 'flat') else n) for n in nets] for _, nets, _ in ranks], (windows, keep)


def test_two_data_parallel_ranks_on_the_emulated_kernels(monkeypatch):
  """The multi-GPU default - the three gradient exchanges INSIDE the launches that produce the gradients (k_dw_adam_peer's block jobs, k_gail_reduce) - between two ranks
  with different data, at the BASELINE configuration: the replicas stay bit-identical, and they equal, bit for bit, the same two ranks run with one exchange launch per
  sync point (tests/test_parallel_gpu.py checks this between two processes sharing one GPU; no run on two GPUs exists)."""
  fused, _k1 = _two_data_parallel_ranks(monkeypatch, True)
  for a, b in zip(*fused):
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a, b, err_msg='replicas differ (exchange inside the optimiser launches)')
  launches, _k2 = _two_data_parallel_ranks(monkeypatch, False)
  for a, b in zip(*launches):
    np.testing.assert_array_equal(a, b, err_msg='replicas differ (exchange launches)')
  for a, b in zip(fused[0], launches[0]):
    np.testing.assert_array_equal(a, b, err_msg='the two schedules differ')


@SLOW
def test_eight_data_parallel_ranks_on_the_emulated_kernels_slow(monkeypatch):
  """BASELINE config 5's shape of parallelism (8 ranks) with the fused exchange: eight learners in one process, replicas bit-identical after two updates (~70 s)."""
  state, _keep = _two_data_parallel_ranks(monkeypatch, True, world=8)
  for r in range(1, 8):
    for a, b in zip(state[0], state[r]):
      np.testing.assert_array_equal(a, b, err_msg=f'rank {r} differs from rank 0')


# ------------------------------------------------------------------------------------------------ the peer-window gradient exchange between emulated ranks
def _peer_exchange_rounds(world, n, n_jobs, write_through, rounds=6, depth=1, seed=0):
  """`world` ranks in one process: rank r's window is a host buffer every rank's descriptor points at (on the GPUs: a peer-mapped uncached allocation), its launch goes
  to its own emulated stream, the ranks' workgroups are co-resident and a rank that polls a peer's arrival word steps aside. The DEVICE code of the exchange
  (csrc/peer_device.hpp: push to every rank's slot, drain, arrival words, bounded wait, rank-ordered sum, parity double buffering) runs as it is."""
  h = emu()
  for name in ('il_peer_region_bytes', 'il_peer_job_region_bytes', 'il_peer_allreduce_mean'):
    getattr(h, name).restype, getattr(h, name).argtypes = _lib._SIGNATURES[name]
  rs = np.random.RandomState(seed)
  region = int(h.il_peer_job_region_bytes(world, n, n_jobs) if n_jobs else h.il_peer_region_bytes(world, n))
  assert region > 0
  windows = [np.zeros(region // 4 + 64, f32) for _ in range(world)]
  base = [w.ctypes.data + (-w.ctypes.data) % 256 for w in windows]
  lines = n_jobs if n_jobs else (n + 2047) // 2048
  epochs, status = [np.zeros(lines, np.uint32) for _ in range(world)], [np.zeros(2, np.int64) for _ in range(world)]
  buckets = []
  for r in range(world):
    b = _lib.PeerBucket()
    b.rank, b.world, b.n, b.window_offset = r, world, n, 0
    for q in range(world): b.windows[q] = base[q]
    b.epoch, b.status, b.spin_limit, b.flags, b.n_jobs = epochs[r].ctypes.data, status[r].ctypes.data, 1 << 16, int(write_through), n_jobs
    buckets.append(b)
  for it in range(0, rounds, depth):
    # `depth` exchanges are queued on every rank's stream before anything runs: a rank that the scheduler favours is then a whole exchange ahead of the others and
    # must be held back by the protocol (a slot of parity p is only rewritten once every rank has read the exchange before last)
    batch = []
    for _ in range(depth):
      xs = [rs.standard_normal(n).astype(f32) for _ in range(world)]
      want = xs[0].copy()
      for q in range(1, world): want = want + xs[q]          # the sum in rank order, then one division: what every rank must hold, bit for bit
      batch.append((xs, want / f32(world)))
    order = list(range(world)); rs.shuffle(order)            # the ranks reach the exchange in any order
    for r in order:
      for xs, _ in batch:
        assert h.il_peer_allreduce_mean(C.byref(buckets[r]), P(xs[r]), C.c_void_p(0x100 * (r + 1))) == 0
    h.emu_drain()
    for k, (xs, want) in enumerate(batch):
      for r in range(world):
        assert status[r][0] == 0, f'rank {r}: {status[r][0]} waits expired by round {it + k}'
        np.testing.assert_array_equal(xs[r], want, err_msg=f'rank {r}, round {it + k}')


@pytest.mark.parametrize('world,n,n_jobs', [(2, 5000, 0), (4, 1665, 0), (8, 18113, 0), (2, 1665, 7), (4, 36226, 20), (8, 9062, 5)])
@pytest.mark.parametrize('write_through', [0, 1])
def test_peer_exchange_between_emulated_ranks(world, n, n_jobs, write_through):
  """The gradient exchange that has never crossed a fabric link: here at least its LOGIC runs between 2 / 4 / 8 ranks - chunk mode (il_peer_allreduce_mean) and
  job mode (the form that rides in k_dw_adam_peer / k_gail_reduce), fence and write-through variants, six rounds (both parities, epochs wrapping the double buffer), ranks
  arriving in random order: every rank ends with the rank-ordered mean, bit-identical, and no wait expired. What this cannot show is the memory model - whether a
  remote uncached store is visible before the flag that follows it; that needs two GPUs."""
  _peer_exchange_rounds(world, n, n_jobs, write_through)
  _peer_exchange_rounds(world, n, n_jobs, write_through, rounds=9, depth=3, seed=1)   # three exchanges in flight per rank: the ranks drift apart as far as the protocol lets them


@pytest.mark.parametrize('body', ['test_gmmil_b1024_full_reward_vector_matches_reference', 'test_gail_b1024_mixup_update_matches_reference',
                                  pytest.param('test_pwil_25k_atoms_matches_reference', marks=pytest.mark.skipif(os.environ.get('IL_EMU_SLOW', '0') != '1', reason='1,100 emulated steps against 25,000 atoms take ~5 min (passes; IL_EMU_SLOW=1 runs it)'))])
def test_timed_sizes_on_the_emulated_kernels(monkeypatch, body):
  """The sizes the benchmarks run (tests/test_timed_sizes.py): BASELINE config 4's GMMIL reward pass as a full reward vector (B = 1024 against 1024 expert rows, Ant
  dims), one adversarial_imitation_update at B = 1024 with the tuned GAIL_5 hyper-parameters, PWIL against 25,000 atoms over 1,100 steps incl. a reset()."""
  tgp = _emulated_product(monkeypatch)
  import test_timed_sizes as tts
  for k in ('il', 'il_training', 'il_memory', '_lib', 'T', 'N', 'Cfg', 'DEV', 'bracket', 'close', 'close_params', 'make_disc', 'tbatch'):
    monkeypatch.setattr(tts, k, getattr(tgp, k), raising=False)
  getattr(tts, body)()


def test_sac_update_at_random_shapes_on_the_emulated_kernels(monkeypatch):
  """One sac_update against the oracle at ten shapes nobody picked by hand: state width 2 .. 40, action width 1 .. 8, hidden 64 / 128 / 192 / 256, 1 .. 7 row tiles
  (tests/test_gpu_parity.py::test_sac_update_other_shapes with the dims drawn from a seeded generator). 35 shapes were run this way at the end of round 3; the one
  mismatch was a single ReLU pre-activation within rounding of zero (one row of one W2 and what back-propagates through it), the conditioning tests/gpu_util.py
  documents - not a shape bug."""
  tgp = _emulated_product(monkeypatch)
  rs = np.random.RandomState(0)
  for _ in range(10):
    S, A, H, B = int(rs.randint(2, 41)), int(rs.randint(1, 9)), int(rs.choice([64, 128, 192, 256])), int(rs.choice([16, 32, 48, 80, 112]))
    monkeypatch.setitem(gi.DIMS, 'fuzz', (S, A))
    tgp.test_sac_update_other_shapes('fuzz', H, B)


@pytest.mark.skipif(os.environ.get('IL_EMU_ASAN', '0') == '1', reason='a preloaded AddressSanitizer cannot intercept the C++ exceptions torch / matplotlib throw and catch internally on this path (CHECK real___cxa_throw)')
def _train_cases():
  """Two configurations in every run; with IL_EMU_SLOW=1 every configuration of tests/test_train_gpu.py (26, ~15 s each; all passed at the end of round 3)."""
  import re
  import test_train_gpu as ttg
  slow = pytest.mark.skipif(os.environ.get('IL_EMU_SLOW', '0') != '1', reason='IL_EMU_SLOW=1 sweeps every train.py configuration of the GPU suite (~7 min)')
  fast = [['algorithm=GAIL', 'env=hopper'], ['algorithm=SAC', 'env=hopper', '+acting.schedule=overlap']]
  out = [pytest.param(a, id='-'.join(x.split('=')[-1] for x in a)) for a in fast]
  in_plan_variant = ('subtract_log_policy=true', 'reward_shaping=true', 'imitation.discriminator.depth=2', 'nonnegative_margin=')   # torch operations inside the plan: GPU only (_update_plan_cases)
  for args in [m for m in ttg.test_train_runs.pytestmark if m.name == 'parametrize'][0].args[1]:
    if args[0] == 'algorithm=GAIL' and any(v in x for x in args for v in in_plan_variant) and not any('mix_expert_data' in x or 'bc_aux' in x for x in args): continue
    if args not in fast:
      out.append(pytest.param([re.sub(r'iterations=\d+', 'iterations=8', x) for x in args], id='-'.join(x.split('=')[-1] for x in args)[:60], marks=slow))
  return out


@pytest.mark.parametrize('args', _train_cases())
def test_train_py_end_to_end_on_the_emulated_kernels(monkeypatch, tmp_path, args):
  """The entry point itself (tests/test_train_gpu.py on the GPU, shortened: 140 environment steps, 20 updates, one evaluation): configuration, expert-data ingest, the
  acting worker feeding the ring, the captured UpdatePlan, the time-out watch, evaluation, checkpoints - every line of train.py a GPU run executes, on CPU tensors and the
  emulated library. Also run by hand: AdRIL, PWIL, reward shaping with a depth-2 tanh potential (15 s each)."""
  _emulated_product(monkeypatch, streams=True)
  import torch
  sys.path.insert(0, os.path.dirname(HERE))
  import train
  from imitation_learning_amd import config
  monkeypatch.chdir(tmp_path)
  cfg = config.compose(args + ['steps=140', 'training.start=120', 'evaluation.interval=70', 'evaluation.episodes=1', 'logging.interval=10', '+synthetic_env.max_episode_steps=60',
                               '+synthetic_env.dataset_trajectories=6', 'training.batch_size=64'])
  score = train.train(cfg)
  assert np.isfinite(score)
  agent = torch.load(tmp_path / 'agent.pth', weights_only=False)
  assert all(torch.isfinite(v).all() for v in agent['actor'].values())
  if cfg.algorithm != 'BC':
    metrics = torch.load(tmp_path / 'metrics.pth', weights_only=False)
    assert 'critic_1.critic.0.weight' in agent['critic'] and len(metrics['update_steps']) >= 1 and all(np.isfinite(q).all() for q in metrics['Q_values'])
  if cfg.algorithm == 'GAIL' and not cfg.imitation.discriminator.reward_shaping:
    assert any(k.startswith('g.0.') for k in torch.load(tmp_path / 'discriminator.pth', weights_only=False))


def test_pwil_many_candidate_lists_on_the_emulated_kernels(monkeypatch):
  """PWIL against 20,000 atoms for 40 steps incl. a reset(): enough atom chunks (79 > 64) that the merging workgroup owns more than one candidate list per lane and stages
  candidates in LDS - the paths the 400-atom fixture never reaches and the 25,000-atom fixture (IL_EMU_SLOW=1, 5 min) does - against the oracle, which is pinned to the
  reference at both sizes. Rewards at the bound of those tests (rtol 2e-5), the remaining atoms exactly."""
  tgp = _emulated_product(monkeypatch)
  import torch
  from oracle import pwil as opwil
  Nn, D, steps, Th, A = 20000, 24, 40, 1000, 6   # (horizon 1000 as in the timed configuration: ~20 atoms consumed per step)
  S = D - A
  atoms, agent = gi.pwil_case(23, Nn, D, steps)
  mem = tgp.il.ReplayMemory(Nn, S, A, False, transitions=dict(states=torch.from_numpy(atoms[:, :S]), actions=torch.from_numpy(atoms[:, S:]), rewards=torch.zeros(Nn),
                                                             next_states=torch.from_numpy(atoms[:, :S]), terminals=torch.zeros(Nn), timeouts=torch.zeros(Nn), weights=torch.ones(Nn),
                                                             num_trajectories=20), device='cpu')
  d = tgp.il.PWILDiscriminator(S, A, tgp.Cfg(state_only=False, reward_scale=5, reward_bandwidth_scale=5), mem, Th)
  o = opwil.PwilOracle(atoms, Th, 5, 5)
  got, want = [], []
  for k in range(steps):
    got.append(float(d.compute_reward(tgp.T(agent[k:k + 1, :S]), tgp.T(agent[k:k + 1, S:]))))
    want.append(o.compute_reward(agent[k]))
    if k == 24:
      d.reset(); o.reset()   # train.py resets at the end of an episode, whatever its length
  np.testing.assert_allclose(got, want, rtol=2e-5)
  assert int((d.expert_weights >= 0).sum()) == len(o.weights)


def test_emulated_product_refuses_nothing_silently(monkeypatch):
  """The swap above is test scaffolding: outside it the product still refuses CPU tensors."""
  import torch
  from imitation_learning_amd import memory as il_memory
  z = torch.zeros(4)
  with pytest.raises(TypeError, match='CUDA'):
    il_memory.batch_desc(dict(states=torch.zeros(4, 3), actions=torch.zeros(4, 2), rewards=z, next_states=torch.zeros(4, 3), terminals=z, weights=z, absorbing=z))


SCHEDULE_SUBSET_EXTRA = ' or peer_exchange_between or timed_path_replays'   # the hand-offs between ranks and between the two streams of the timed plan: the perturbation also shuffles workgroup dispatch and which stream runs next   # the device hand-off between two streams: the schedule perturbation also shuffles which stream's workgroup runs next
ASAN_SUBSET = ('sac_update_matches_oracle_and_reference-sac_hopper_h64 or sac_gradients_match_oracle-sac_hopper_h64 or gail_update_matches_oracle_and_reference-gail_default or '
               'gail_loss_variants_match_reference-mixup_sublogp or gmmil_matches_oracle_and_reference-small or pwil_matches_oracle or replay_matches_reference_bit_exact-wrapped or '
               'red_matches_reference-hopper_d2_tanh_drop or dril_matches_reference-hopper_d2_relu or gail_deep_discriminator_matches_reference-hopper_d2_tanh_sn or '
               'general_potential_matches_reference-hopper_d2_relu_sn_margin or gail_reward_shaping_mixup or bc_update or actor_act or reward_relabeller')


@pytest.mark.skipif(os.environ.get('IL_EMU_ASAN_RUN', '0') != '1' and os.environ.get('IL_EMU_ASAN_ALL', '0') != '1',
                    reason='the sanitised build of the whole library takes ~45 s to compile: IL_EMU_ASAN_RUN=1 (18 bodies, ~1.5 min) or IL_EMU_ASAN_ALL=1 (everything, ~5 min)')
def test_emulated_kernels_are_address_sanitizer_clean():
  """The same emulation compiled with -fsanitize=address,undefined (IL_EMU_ASAN=1), a subset of the bodies above in a child process: every load and store of those
  kernels - the global buffers (numpy / torch allocations go through the intercepted malloc), the workgroup's LDS (allocated to the byte) - is bounds- and lifetime-
  checked, every 16-byte vector access alignment-checked, signed overflow and shifts checked. A GPU run cannot say this: an access a few words past a tensor lands in
  the caching allocator's pool and is silent. The whole file is clean under it (IL_EMU_ASAN_ALL=1 runs all of it, ~5 min; last run: end of round 3, 143 passed, the two train.py cases excluded); round 3 found one use-after-free this way -
  a temporary `torch.ones` whose pointer sat in an il_batch after the tensor had died (training.py)."""
  import subprocess
  asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
  if not os.path.isabs(asan) or not os.path.exists(asan):
    pytest.skip('libasan.so not found next to gcc')
  env = dict(os.environ, IL_EMU_ASAN='1', LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:detect_stack_use_after_return=0')
  sel = [] if os.environ.get('IL_EMU_ASAN_ALL', '0') == '1' else ['-k', ASAN_SUBSET]
  r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-s', '-p', 'no:cacheprovider', '--deselect',
                      os.path.abspath(__file__) + '::test_emulated_kernels_are_address_sanitizer_clean', *sel], env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=1500)
  out = r.stdout + r.stderr
  assert 'AddressSanitizer' not in out, out[out.index('AddressSanitizer') - 200:][:6000]
  assert r.returncode == 0 and ' passed' in out, out[-3000:]


@pytest.mark.parametrize('schedule', ['reverse', 'random:5'])
def test_emulated_kernels_do_not_depend_on_the_wave_schedule(schedule):
  """The same subset with the waves of every workgroup (and the lanes of every wave) scheduled in reverse / in a random order between two barriers: the parity bounds
  still hold, i.e. no result depends on which wave reaches a barrier-free stretch first (a missing __syncthreads() would - see the emulator's self-test above).
  IL_EMU_SCHEDULE=... with the whole file: 90 passed under reverse, random:1, random:7."""
  import subprocess
  r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-p', 'no:cacheprovider', '-k', ASAN_SUBSET + SCHEDULE_SUBSET_EXTRA], env=dict(os.environ, IL_EMU_SCHEDULE=schedule),
                     cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=1500)
  assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout + r.stderr)[-3000:]
