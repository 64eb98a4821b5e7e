"""The plain-VALU HIP kernels of the general GAIL discriminators, executed on the HOST from their own source text (tests/host_emu: workgroups as fibers, the `<<<>>>`
launches rewritten, nothing else) against the reference fixtures. This is what `pytest -m "not gpu"` can say about kernel code where there is no GPU: indexing, the
order of power iterations, slab / context layouts, the reduce kernel's chain rule. It says nothing about performance or about anything wave-level; the `-m gpu`
tests run the same comparisons on the real library. gail_deep.hip and gail_shaped.hip are also the emulator's own check: both are green on the GPU against these
very fixtures, so a disagreement here would be the emulator's."""
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


_HANDLES = {}


def emu(name):
  if name not in _HANDLES:
    _HANDLES[name] = emu_build.load(name)
  return _HANDLES[name]


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
    h.emu_il_last_error.restype = C.c_char_p
    raise RuntimeError(f'emulated library error {rc}: {h.emu_il_last_error().decode()}')


def close(a, b, what, rtol, atol_scale):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert a.shape == b.shape, what
  err = np.abs(a - b) - (atol_scale * np.abs(b).max() + rtol * np.abs(b))
  assert np.isfinite(a).all() and err.max() <= 0, f'{what}: worst excess {err.max():.3e} at {int(err.argmax())} ({a.ravel()[err.argmax()]} vs {b.ravel()[err.argmax()]})'


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


# ------------------------------------------------------------------------------------------------ the Python entry points over the emulated library
@pytest.mark.parametrize('name', ['hopper_d2_tanh_sn', 'walker2d_d1_tanh_sn', 'hopper_d2_relu_sn_margin'])
def test_python_entry_points_over_the_emulated_library(golden_dir, monkeypatch, name):
  """`GAILDiscriminator(...)` -> ShapedDeepGAILDiscriminator -> `adversarial_imitation_update` / `predict_reward` (models.py / training.py of the product package) with the
  library handle swapped for the host emulation of gail_shaped_deep.hip and CPU tensors: the descriptor the glue fills, the parameter order, the buffers' layout and
  the dispatch are what the GPU run uses. (The product refuses CPU tensors - memory.batch_desc - so the test substitutes a lenient batch descriptor.)"""
  import torch
  import imitation_learning_amd as il
  from imitation_learning_amd import training as il_training
  from gpu_util import Cfg
  from test_oracle_golden import _shaped_deep_state
  h, real = emu('gail_shaped_deep'), _lib.lib()

  class Facade:   # the emulated entry points where they exist, the real (host-side, size-query) ones otherwise
    def __getattr__(self, fn_name):
      try:
        fn = getattr(h, fn_name)
      except AttributeError:
        return getattr(real, fn_name)
      fn.restype, fn.argtypes = _lib._SIGNATURES[fn_name]
      return fn

  def lenient_batch_desc(t):
    b, n = _lib.Batch(), None
    for k in ('states', 'actions', 'rewards', 'next_states', 'terminals', 'weights', 'absorbing'):
      v = t[k]
      setattr(b, k, v.data_ptr()); setattr(b, 'ld_' + k, v.stride(0) if v.size(0) > 1 else (v.size(1) if v.dim() == 2 else 1)); n = v.size(0)
    b.n = n
    return b

  monkeypatch.setattr(_lib, '_lib', Facade())
  monkeypatch.setattr(_lib, 'stream_ptr', lambda: None)
  monkeypatch.setattr(il_training, 'batch_desc', lenient_batch_desc)
  il_training._WS.clear(); il_training._NOISE.clear()
  g = np.load(os.path.join(golden_dir, 'gail_shaped_deep.npz'))
  _, kw, loss, (lr, wd, gp, ent), rf, margin = next(c for c in gi.GAIL_SHAPED_DEEP_CASES if c[0] == name)
  c = gi.gail_shaped_deep_case(**kw)
  icfg = Cfg(state_only=c['state_only'], spectral_norm=c['spectral_norm'], loss_function=loss, grad_penalty=gp, mixup_alpha=0.7, entropy_bonus=ent, pos_class_prior=0.7, nonnegative_margin=margin,
             discriminator=Cfg(hidden_size=c['H'], depth=c['depth'], activation=c['activation'], reward_shaping=True, subtract_log_policy=False, reward_function=rf))
  d = il.GAILDiscriminator(c['S'], c['A'], icfg, 0.97, device='cpu')
  assert type(d).__name__ == 'ShapedDeepGAILDiscriminator' and [n for n, _ in d.named_parameters()] == list(g[f'{name}.param_names'])
  ods = _shaped_deep_state(c)
  TT = lambda a: torch.from_numpy(np.ascontiguousarray(a, f32))
  tb = lambda b: dict({k: TT(v) for k, v in b.items()}, absorbing=torch.zeros(len(b['weights'])))
  d.flat.copy_(TT(ods.pack())); d.sn.copy_(TT(ods.pack_sn()))
  opt = il.AdamW(d, lr=lr, weight_decay=wd)
  try:
    il.adversarial_imitation_update(None, d, tb(c['policy'][0]), tb(c['expert'][0]), opt, icfg, eps_gp=TT(c['eps'][0]), eps_mix=TT(c['eps_mix'][0]))
    close(opt.grad.numpy(), g[f'{name}.g_1'], f'{name} gradient', rtol=2e-4, atol_scale=2e-6)
    assert np.abs(d.flat.detach().numpy() - g[f'{name}.p_1']).max() <= 6e-6
    close(d.sn.numpy(), g[f'{name}.sn_1'], f'{name} u / v', rtol=1e-4, atol_scale=1e-6)
    p = tb(c['policy'][0])
    d.flat.copy_(TT(g[f'{name}.p_1']))
    r = d.predict_reward(**il.make_gail_input(p['states'], p['actions'], p['next_states'], p['terminals'], None, True, False))
    close(r.numpy(), g[f'{name}.reward_1'], f'{name} reward', rtol=1e-4, atol_scale=1e-5)
  finally:
    il_training._WS.clear(); il_training._NOISE.clear()   # CPU arenas must not outlive the test (a GPU test in the same process would find them)
