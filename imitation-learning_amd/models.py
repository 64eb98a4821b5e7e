"""Host-side mirror of the reference's model classes (reference models.py) over flat HBM arenas.

Same constructors, method names and state_dict keys as the reference (`actor.0.weight`, `critic_1.critic.0.weight`,
`g.0.parametrizations.weight.original`, `...weight.0._u`), so checkpoints stay interchangeable -- but every network is
ONE flat fp32 tensor on the GPU (torch `parameters()` order) that the HIP kernels update in place; the `nn.Module`
parameters are views into it.  Nothing here does arithmetic on the update path: that is libil_hip.so (see training.py).
"""
from __future__ import annotations

import copy
import ctypes as C
import os
from math import sqrt
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn
from torch.nn.utils import parametrizations

from . import _lib

REWARD_FUNCTIONS = {'AIRL': 0, 'GAIL': 1, 'FAIRL': 2}
LOSS_FUNCTIONS = {'BCE': 0, 'PUGAIL': 1, 'Mixup': 2}   # IL_LOSS_* (include/il_hip.h)


def default_device() -> torch.device:
  return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def _cfg_get(cfg, key, default=None):
  return cfg.get(key, default) if hasattr(cfg, 'get') else getattr(cfg, key, default)


ACTIVATION_IDS = {'relu': 0, 'tanh': 1, 'sigmoid': 2}   # models.py:16 ACTIVATION_FUNCTIONS; the ids of csrc/general.hip
_ACTIVATION_MODULES = {'relu': nn.ReLU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid}


def _require_fused_mlp(model_cfg, what):
  depth, act = _cfg_get(model_cfg, 'depth'), _cfg_get(model_cfg, 'activation')
  hidden = _cfg_get(model_cfg, 'hidden_size')
  if depth != 2 or act != 'relu' or hidden % 64 != 0 or not 64 <= hidden <= 256:
    raise NotImplementedError(f'{what}: the HIP path implements depth=2, activation=relu, hidden_size in {{64,128,192,256}} '
                              f'(got depth={depth}, activation={act}, hidden_size={hidden}); there is no CPU/torch fallback on this path')
  if _cfg_get(model_cfg, 'input_dropout', 0) or _cfg_get(model_cfg, 'dropout', 0):
    raise NotImplementedError(f'{what}: dropout networks (DRIL) are outside the HIP hot path')
  return hidden


def _mlp_shape(model_cfg, what):
  """(hidden, depth, activation, general) of an actor / critic network (models.py:48-69 `_create_fcnn` builds any depth with relu / tanh / sigmoid). general = False: the
  shape of every shipped configuration (depth 2, ReLU, hidden 64 .. 256), which the fused kernels of csrc/sac.hip run; True: any other shape, composed a layer at a time
  by csrc/general.hip (per-function path: sac_update, acting, log_prob, behavioural cloning - no captured plan, population or data-parallel form)."""
  depth, act, hidden = int(_cfg_get(model_cfg, 'depth')), str(_cfg_get(model_cfg, 'activation')), int(_cfg_get(model_cfg, 'hidden_size'))
  if _cfg_get(model_cfg, 'input_dropout', 0) or _cfg_get(model_cfg, 'dropout', 0):
    raise NotImplementedError(f'{what}: dropout networks (DRIL) are outside the HIP hot path')
  if act not in ACTIVATION_IDS:
    raise ValueError(f'{what}: activation must be one of {sorted(ACTIVATION_IDS)} (got {act})')
  fused = depth == 2 and act == 'relu' and hidden % 64 == 0 and 64 <= hidden <= 256
  if not fused and not (1 <= depth <= 8 and 1 <= hidden <= 2048):
    raise NotImplementedError(f'{what}: the general HIP path covers depth 1-8 and hidden_size <= 2048 (got depth={depth}, hidden_size={hidden}); there is no CPU/torch fallback')
  return hidden, depth, act, not fused


def _mlp(in_dim: int, hidden: int, out_dim: int, final_gain: float = 1.0, depth: int = 2, activation: str = 'relu') -> nn.Sequential:
  """`depth` x (Linear - activation) - Linear with the reference's init (orthogonal, gain calculate_gain(activation) hidden / final_gain last, zero bias; models.py:48-66)."""
  dims, layers = [in_dim] + [hidden] * depth + [out_dim], []
  for i in range(depth + 1):
    lin = nn.Linear(dims[i], dims[i + 1])
    nn.init.orthogonal_(lin.weight, gain=nn.init.calculate_gain(activation) if i < depth else final_gain)
    nn.init.constant_(lin.bias, 0)
    layers.append(lin)
    if i < depth:
      layers.append(_ACTIVATION_MODULES[activation]())
  return nn.Sequential(*layers)


def _flatten_into(modules_params, flat: Tensor, offsets):
  """Copies each parameter into its slot of `flat` and re-points `.data` at the slot (a view)."""
  for p, o in zip(modules_params, offsets):
    n = p.numel()
    view = flat[o:o + n].view(p.shape)
    view.copy_(p.data)
    p.data = view


class _FlatModule(nn.Module):
  """nn.Module whose parameters are views into one flat device tensor `self.flat`."""

  flat: Tensor

  def _adopt(self, numel_total: int, offsets, device):
    params = list(self.parameters())
    flat = torch.zeros(numel_total, dtype=torch.float32, device=device)
    with torch.no_grad():
      _flatten_into(params, flat, offsets)
    self.flat = flat
    for p in params:
      p.requires_grad_(False)  # gradients are produced by the HIP kernels, never by autograd

  @property
  def device(self):
    return self.flat.device


class SoftActor(_FlatModule):
  """Tanh-Gaussian policy (reference models.py:84-120). `actor(state).sample()` / `get_greedy_action` run k_act on the GPU."""

  def __new__(cls, state_size=None, action_size=None, model_cfg=None, device=None):
    # train.py:73 builds the DRIL "discriminator" with this same class from the imitation.discriminator config (conf/algorithm/DRIL.yaml: depth 1, tanh,
    # dropout 0.1; conf/optimised_hyperparameters/DRIL_*.yaml: also depth 2 / relu): a config WITH dropout is the policy ensemble. A dropout-free depth-1 tanh
    # network is an ordinary actor of the general-shape engine (reinforcement.actor may ask for it); a dropout-free ensemble is built as DropoutSoftActor(...) directly.
    if cls is SoftActor and model_cfg is not None and (float(_cfg_get(model_cfg, 'dropout', 0) or 0) > 0 or float(_cfg_get(model_cfg, 'input_dropout', 0) or 0) > 0):
      return super().__new__(DropoutSoftActor)
    return super().__new__(cls)

  def __init__(self, state_size: int, action_size: int, model_cfg, device=None):
    super().__init__()
    self.state_size, self.action_size = state_size, action_size
    self.hidden, self.depth, self.activation, self.general = _mlp_shape(model_cfg, 'SoftActor')
    if 2 * action_size > 16:
      self.general = True   # the fused head holds 2A <= 16 outputs: wider action spaces take the general kernels
    self.log_std_dev_min, self.log_std_dev_max = -20, 2
    self.actor = _mlp(state_size, self.hidden, 2 * action_size, depth=self.depth, activation=self.activation)
    offs, o = [], 0
    for p in self.parameters():
      offs.append(o); o += p.numel()
    self._adopt(o, offs, device or default_device())
    self._act_calls = 0

  class _Policy:
    def __init__(self, actor, state):
      self.actor, self.state = actor, state

    def sample(self, eps: Optional[Tensor] = None) -> Tensor:
      return self.actor._act(self.state, greedy=False, eps=eps)[0]

    rsample = sample

    def sample_with_log_prob(self, eps: Optional[Tensor] = None):
      return self.actor._act(self.state, greedy=False, eps=eps, want_logp=True)

  def forward(self, state: Tensor) -> 'SoftActor._Policy':
    return SoftActor._Policy(self, state)

  def _act(self, state: Tensor, greedy: bool, eps: Optional[Tensor] = None, want_logp: bool = False):
    state = state.to(self.flat.device, torch.float32)
    if state.dim() == 1:
      state = state.unsqueeze(0)
    assert state.stride(1) == 1
    n = state.size(0)
    out = torch.empty(n, self.action_size, device=self.flat.device)
    logp = torch.empty(n, device=self.flat.device) if want_logp else None
    if eps is not None:
      eps = eps.to(self.flat.device, torch.float32).contiguous()
    self._act_calls += 1
    if self.general:
      ws = self._general_workspace(n)
      _lib.check(_lib.lib().il_actor_act_general(_lib.ptr(self.flat), self.state_size, self.action_size, self.hidden, self.depth, ACTIVATION_IDS[self.activation], _lib.ptr(state), state.stride(0), n,
                                                 _lib.ptr(eps), C.c_uint64(torch.initial_seed() & (2**64 - 1)), self._act_calls & 0xFFFFFFFF, int(greedy), _lib.ptr(out), _lib.ptr(logp),
                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
      return out, logp
    _lib.check(_lib.lib().il_actor_act(_lib.ptr(self.flat), self.state_size, self.action_size, self.hidden, _lib.ptr(state), state.stride(0), n, _lib.ptr(eps),
                                       C.c_uint64(torch.initial_seed() & (2**64 - 1)), self._act_calls & 0xFFFFFFFF, int(greedy), _lib.ptr(out), _lib.ptr(logp), _lib.stream_ptr()))
    return out, logp

  def get_greedy_action(self, state: Tensor) -> Tensor:
    return self._act(state, greedy=True)[0]

  def _general_workspace(self, n: int) -> Tensor:
    """Scratch of the general-shape entry points (csrc/general.hip), grown on demand and kept with the module."""
    need = int(_lib.lib().il_actor_workspace_floats_general(self.state_size, self.action_size, self.hidden, self.depth, n))
    ws = getattr(self, '_gws', None)
    if ws is None or ws.numel() < need:
      ws = self._gws = torch.zeros(need, dtype=torch.float32, device=self.flat.device)
    return ws

  def log_prob(self, state: Tensor, action: Tensor) -> Tensor:
    """models.py:97-99: log pi(a|s) with the action clamped to +-(1 - 1e-6) (k_actor_logp)."""
    dev = self.flat.device
    state, action = state.to(dev, torch.float32), action.to(dev, torch.float32)
    if state.dim() == 1: state, action = state.unsqueeze(0), action.unsqueeze(0)
    if state.stride(1) != 1: state = state.contiguous()
    if action.stride(1) != 1: action = action.contiguous()
    n = state.size(0)
    out = torch.empty(n, device=dev)
    if self.general:
      ws = self._general_workspace(n)
      _lib.check(_lib.lib().il_actor_log_prob_general(_lib.ptr(self.flat), self.state_size, self.action_size, self.hidden, self.depth, ACTIVATION_IDS[self.activation], _lib.ptr(state),
                                                      state.stride(0) if n > 1 else state.size(1), _lib.ptr(action), action.stride(0) if n > 1 else action.size(1), n, _lib.ptr(out),
                                                      _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
      return out
    _lib.check(_lib.lib().il_actor_log_prob(_lib.ptr(self.flat), self.state_size, self.action_size, self.hidden, _lib.ptr(state), state.stride(0) if n > 1 else state.size(1),
                                            _lib.ptr(action), action.stride(0) if n > 1 else action.size(1), n, _lib.ptr(out), _lib.stream_ptr()))
    return out


class DropoutSoftActor(SoftActor):
  """The DRIL policy ensemble (reference models.py:84-120 built by `_create_fcnn` from the imitation.discriminator config): Dropout(p_in)-Linear(S,H)-
  Dropout(p)-act(-Linear(H,H)-Dropout(p)-act)-Linear(H,2A), depth 1-2, act in {tanh, relu} (conf/algorithm/DRIL.yaml: depth 1 tanh;
  conf/optimised_hyperparameters/DRIL_{10,25}_trajectories.yaml: depth 2 relu).

  Created through `SoftActor(state_size, action_size, cfg.imitation.discriminator)` like in the reference (train.py:73). It stays in train
  mode (dropout active) for its whole life, as the reference's does: `behavioural_cloning_update` on it runs k_dril_grad / k_dril_apply,
  the Monte-Carlo-dropout uncertainty runs k_dril_unc.  Pass `masks=(mask_in, mask_hidden[, mask_hidden2])` to reproduce given dropout draws; otherwise
  the masks come from the on-chip Philox stream.  state_dict keys: actor.N.* with the Dropout / activation modules occupying Sequential slots too."""
  ENSEMBLE = 5
  general = False   # (its kernels are k_dril_*: never the general-shape engine of csrc/general.hip)

  def __init__(self, state_size: int, action_size: int, model_cfg, device=None):
    nn.Module.__init__(self)
    self.state_size, self.action_size, self.hidden = state_size, action_size, int(_cfg_get(model_cfg, 'hidden_size'))
    self.depth, self.activation = int(_cfg_get(model_cfg, 'depth')), str(_cfg_get(model_cfg, 'activation'))
    self.p_in, self.p = float(_cfg_get(model_cfg, 'input_dropout', 0) or 0), float(_cfg_get(model_cfg, 'dropout', 0) or 0)
    if self.depth not in (1, 2) or self.activation not in ('tanh', 'relu'):
      raise NotImplementedError(f'DRIL policy: the HIP path implements depth 1-2 with tanh / relu (got depth={self.depth}, activation={self.activation}); no torch fallback')
    if state_size > 128 or 2 * action_size > 16 or self.hidden > 256 or self.hidden % 2:
      raise NotImplementedError(f'DRIL policy: state {state_size} (<= 128), action {action_size} (<= 8), hidden {self.hidden} (even, <= 256) outside the kernel limits')
    self.log_std_dev_min, self.log_std_dev_max = -20, 2
    act = nn.Tanh if self.activation == 'tanh' else nn.ReLU
    dims, layers = [state_size] + [self.hidden] * self.depth, []
    if self.p_in > 0: layers.append(nn.Dropout(self.p_in))
    for a, b in zip(dims[:-1], dims[1:]):   # module order (hence state_dict keys) and RNG consumption as in the reference's _create_fcnn
      lin = nn.Linear(a, b)
      nn.init.orthogonal_(lin.weight, gain=nn.init.calculate_gain(self.activation)); nn.init.constant_(lin.bias, 0)
      layers.append(lin)
      if self.p > 0: layers.append(nn.Dropout(self.p))
      layers.append(act())
    last = nn.Linear(self.hidden, 2 * action_size)
    nn.init.orthogonal_(last.weight, gain=1.0); nn.init.constant_(last.bias, 0)
    self.actor = nn.Sequential(*layers, last)
    offs, o = [], 0
    for p in self.parameters():
      offs.append(o); o += p.numel()
    assert o == int(_lib.lib().il_dril_numel(state_size, action_size, self.hidden, self.depth))
    self._adopt(o, offs, device or default_device())
    self._act_calls, self.q = 0, None

  def _desc(self, batch_size: int, opt=None) -> '_lib.Dril':
    d = _lib.Dril()
    d.state_dim, d.action_dim, d.hidden, d.batch, d.p_in, d.p = self.state_size, self.action_size, self.hidden, batch_size, self.p_in, self.p
    d.depth, d.activation = self.depth, int(self.activation == 'relu')
    d.params, d.noise_seed, d.q = self.flat.data_ptr(), torch.initial_seed() & (2**64 - 1), float(self.q) if self.q is not None else 0.0
    if opt is not None:
      from .training import _workspace
      ws = _workspace('dril', int(_lib.lib().il_dril_workspace_floats(self.state_size, self.action_size, self.hidden, batch_size, self.depth)), self.flat.device)
      d.grad, d.opt, d.workspace = opt.grad.data_ptr(), opt.desc(), ws.data_ptr()
    return d

  def _masks(self, masks, rows):
    """(mask_in, mask_hidden, mask_hidden2) tensors (None = drawn on chip); with masks given, one per dropout layer in module order."""
    if masks is None:
      return None, None, None
    dev = self.flat.device
    masks = [x.to(dev, torch.float32).contiguous() for x in masks]
    assert len(masks) == 1 + self.depth, f'dropout keep-masks: expected {1 + self.depth} (input + one per hidden layer)'
    assert masks[0].shape == (rows, self.state_size) and all(m.shape == (rows, self.hidden) for m in masks[1:]), 'dropout keep-masks must be [rows, S] and [rows, H]'
    return tuple(masks) + (None,) * (3 - len(masks))

  def _next_offset(self) -> int:
    self._act_calls += 1
    return self._act_calls & 0xFFFFFFFF

  def bc_update(self, expert_transition, optimiser, masks=None, want_loss: bool = False):
    from .memory import batch_desc
    t = dict(expert_transition)
    for k in ('rewards', 'terminals', 'absorbing', 'next_states'):
      t.setdefault(k, t['weights'] if k != 'next_states' else t['states'])
    b = batch_desc(t)
    m0, m1, m2 = self._masks(masks, b.n)
    loss = torch.empty(1, device=self.flat.device) if want_loss else None
    d = self._desc(b.n, optimiser)
    _lib.check(_lib.lib().il_dril_bc_step(C.byref(d), C.byref(b), _lib.ptr(m0), _lib.ptr(m1), _lib.ptr(m2), self._next_offset(), _lib.ptr(loss), 0, _lib.stream_ptr()))
    return loss

  def _uncertainty(self, state: Tensor, action: Tensor, masks=None, want_reward: bool = False) -> Tensor:
    from .training import _sa_batch
    dev = self.flat.device
    state, action = state.to(dev, torch.float32), action.to(dev, torch.float32)
    if state.stride(-1) != 1: state = state.contiguous()
    if action.stride(-1) != 1: action = action.contiguous()
    n = state.size(0)
    ones = torch.ones(n, device=dev)   # named: an il_batch holds raw pointers
    b = _sa_batch(state, action, ones)
    m0, m1, m2 = self._masks(masks, n * self.ENSEMBLE)
    out = torch.empty(n, device=dev)
    d = self._desc(n)
    _lib.check(_lib.lib().il_dril_uncertainty(C.byref(d), C.byref(b), _lib.ptr(m0), _lib.ptr(m1), _lib.ptr(m2), self._next_offset(), None if want_reward else _lib.ptr(out),
                                              _lib.ptr(out) if want_reward else None, _lib.stream_ptr()))
    return out

  def _get_action_uncertainty(self, state: Tensor, action: Tensor, masks=None) -> Tensor:
    """models.py:104-107: variance over 5 dropout masks of exp(log_prob(state, action))."""
    return self._uncertainty(state, action, masks)

  def set_uncertainty_threshold(self, expert_state: Tensor, expert_action: Tensor, quantile_cutoff: float, masks=None):
    self.q = torch.quantile(self._uncertainty(expert_state, expert_action, masks), quantile_cutoff).item()   # models.py:110-111

  def predict_reward(self, state: Tensor, action: Tensor, masks=None) -> Tensor:
    assert self.q is not None, 'predict_reward before set_uncertainty_threshold (train.py:126)'
    return self._uncertainty(state, action, masks, want_reward=True)

  def forward(self, state):
    raise NotImplementedError('the DRIL policy ensemble is only evaluated through log-probabilities of given actions (uncertainty / BC) on the HIP path')

  get_greedy_action = log_prob = forward


class Critic(nn.Module):
  def __init__(self, state_size: int, action_size: int, hidden: int, depth: int = 2, activation: str = 'relu'):
    super().__init__()
    self.critic = _mlp(state_size + action_size, hidden, 1, depth=depth, activation=activation)

  def forward(self, state: Tensor, action: Tensor) -> Tensor:
    return self.critic(torch.cat([state, action], dim=1)).squeeze(dim=1)


class TwinCritic(_FlatModule):
  """Two independent Q networks (reference models.py:123-141); flat layout critic_1 | critic_2 at stride il_mlp_stride."""

  def __init__(self, state_size: int, action_size: int, model_cfg, device=None):
    super().__init__()
    self.state_size, self.action_size = state_size, action_size
    self.hidden, self.depth, self.activation, self.general = _mlp_shape(model_cfg, 'TwinCritic')
    self.critic_1, self.critic_2 = (Critic(state_size, action_size, self.hidden, self.depth, self.activation) for _ in range(2))
    numel = sum(p.numel() for p in self.critic_1.parameters())
    self.net_stride = (numel + 3) // 4 * 4
    offs = []
    for k, net in enumerate((self.critic_1, self.critic_2)):
      o = k * self.net_stride
      for p in net.parameters():
        offs.append(o); o += p.numel()
    self._adopt(2 * self.net_stride, offs, device or default_device())

  def forward(self, state: Tensor, action: Tensor) -> Tuple[Tensor, Tensor]:
    return self.critic_1(state, action), self.critic_2(state, action)


def create_target_network(network: _FlatModule) -> _FlatModule:
  """Reference models.py:72-76. The copy gets its own flat arena with the same layout."""
  target = copy.deepcopy(network)
  target.flat = network.flat.clone()
  with torch.no_grad():
    for p, q in zip(target.parameters(), network.parameters()):
      off = (q.data_ptr() - network.flat.data_ptr()) // 4
      p.data = target.flat[off:off + q.numel()].view(q.shape)
      p.requires_grad = False
  return target


def update_target_network(network: _FlatModule, target_network: _FlatModule, polyak_factor: float):
  """Reference models.py:79-81 as one streaming kernel over the arena (sac_update fuses it; this is the stand-alone form)."""
  _lib.check(_lib.lib().il_polyak(_lib.ptr(target_network.flat), _lib.ptr(network.flat), network.flat.numel(), float(polyak_factor), _lib.stream_ptr()))


def make_gail_input(state, action, next_state, terminal, actor, reward_shaping: bool, subtract_log_policy: bool) -> Dict[str, Tensor]:
  """models.py:139-144."""
  out = {'state': state, 'action': action}
  if reward_shaping:
    out.update(next_state=next_state, terminal=terminal)
  if subtract_log_policy:
    out['log_policy'] = actor.log_prob(state, action)
  return out


class GAILDiscriminator(_FlatModule):
  """Depth-1 ReLU discriminator with optional spectral norm (reference models.py:152-180).

  Flat arena = `parameters()` order (spectral norm: g.0.bias, g.0...original, g.2.bias, g.2...original). The torch
  parametrization modules are kept only for initialisation (same RNG consumption as the reference: normal_ u, v + 15+1
  power iterations) and for the state_dict keys; their buffers are re-pointed at `self.sn` so the kernels own u, v.
  """

  def __new__(cls, state_size=None, action_size=None, imitation_cfg=None, discount=None, device=None):
    if cls is GAILDiscriminator and imitation_cfg is not None and imitation_cfg.discriminator.reward_shaping:
      # f = g(s, a) + (1 - t)(discount h(s') - h(s)): its own kernels - gail_shaped.hip for the depth-1 ReLU potential of the default configuration, gail_shaped_deep.hip for
      # depth 2 and / or tanh (IL_SHAPED_GENERAL=1 sends the default shape through the general kernels too: a cross-check of the two implementations)
      general = (imitation_cfg.discriminator.depth, imitation_cfg.discriminator.activation) != (1, 'relu') or os.environ.get('IL_SHAPED_GENERAL', '0') == '1'
      return super().__new__(ShapedDeepGAILDiscriminator if general else ShapedGAILDiscriminator)
    if cls is GAILDiscriminator and imitation_cfg is not None and (imitation_cfg.discriminator.depth, imitation_cfg.discriminator.activation) != (1, 'relu'):
      return super().__new__(DeepGAILDiscriminator)     # depth 2 and / or tanh: the general kernels (gail_deep.hip)
    return super().__new__(cls)

  def __init__(self, state_size: int, action_size: int, imitation_cfg, discount: float, device=None):
    super().__init__()
    model_cfg = imitation_cfg.discriminator
    self.discount, self.state_only = discount, bool(imitation_cfg.state_only)
    self.reward_shaping, self.subtract_log_policy, self.reward_function = model_cfg.reward_shaping, model_cfg.subtract_log_policy, model_cfg.reward_function
    self.spectral_norm = bool(imitation_cfg.spectral_norm)
    if self.reward_shaping or model_cfg.depth != 1 or model_cfg.activation != 'relu':
      raise NotImplementedError('GAILDiscriminator: the HIP path implements depth=1, activation=relu without reward shaping '
                                '(the closed-form gradient-penalty backward assumes it); no torch fallback on this path')
    self.state_size, self.action_size, self.hidden = state_size, action_size, model_cfg.hidden_size
    self.in_dim = state_size if self.state_only else state_size + action_size
    l1, l2 = nn.Linear(self.in_dim, self.hidden), nn.Linear(self.hidden, 1)
    nn.init.orthogonal_(l1.weight, gain=sqrt(2.0)); nn.init.constant_(l1.bias, 0)
    if self.spectral_norm: l1 = parametrizations.spectral_norm(l1)
    nn.init.orthogonal_(l2.weight, gain=1.0); nn.init.constant_(l2.bias, 0)
    if self.spectral_norm: l2 = parametrizations.spectral_norm(l2)
    self.g = nn.Sequential(l1, nn.ReLU(), l2)
    offs, o = [], 0
    for p in self.parameters():
      offs.append(o); o += p.numel()
    dev = device or default_device()
    self._adopt(o, offs, dev)
    H, D = self.hidden, self.in_dim
    self.sn = torch.zeros(2 * H + D + 1, device=dev)  # u1[H] | v1[D] | u2[1] | v2[H]
    if self.spectral_norm:
      with torch.no_grad():
        for mod, (ou, nu, ov, nv) in ((self.g[0].parametrizations.weight[0], (0, H, H, D)), (self.g[2].parametrizations.weight[0], (H + D, 1, H + D + 1, H))):
          u, v = self.sn[ou:ou + nu], self.sn[ov:ov + nv]
          u.copy_(mod._u); v.copy_(mod._v)
          mod._buffers['_u'], mod._buffers['_v'] = u, v
    self.eval()

  def views(self):
    H, D = self.hidden, self.in_dim
    return dict(u1=self.sn[:H], v1=self.sn[H:H + D], u2=self.sn[H + D:H + D + 1], v2=self.sn[H + D + 1:])

  def predict_reward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import gail_predict_reward
    assert (log_policy is not None) == bool(self.subtract_log_policy), 'pass log_policy exactly when subtract_log_policy is set (make_gail_input does)'
    return gail_predict_reward(self, state, action, log_policy=log_policy)

  def forward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import gail_predict_reward
    return gail_predict_reward(self, state, action, want_logits=True, log_policy=log_policy)[1]


class ShapedGAILDiscriminator(GAILDiscriminator):
  """GAIL discriminator with reward shaping (reference models.py:152-180, reward_shaping=True): g = Linear(Dg, 1) is the reward, h = Linear(S, H)-ReLU-
  Linear(H, 1) the potential, f = g(s, a) + (1 - terminal)(discount h(s') - h(s)); every weight optionally under spectral norm.  Created through
  `GAILDiscriminator(...)` when `imitation.discriminator.reward_shaping` is set.  Flat arena = parameters() order, buffers `self.sn` =
  ug[1] | vg[Dg] | u1[H] | v1[S] | u2[1] | v2[H]; same RNG consumption at construction as the reference (default-initialised g, orthogonal h)."""

  def __init__(self, state_size: int, action_size: int, imitation_cfg, discount: float, device=None):
    nn.Module.__init__(self)
    model_cfg = imitation_cfg.discriminator
    self.discount, self.state_only = discount, bool(imitation_cfg.state_only)
    self.reward_shaping, self.subtract_log_policy, self.reward_function = True, model_cfg.subtract_log_policy, model_cfg.reward_function
    self.spectral_norm = bool(imitation_cfg.spectral_norm)
    if model_cfg.depth != 1 or model_cfg.activation != 'relu' or model_cfg.hidden_size > 256:   # (other depths / activations are ShapedDeepGAILDiscriminator's)
      raise NotImplementedError('GAILDiscriminator (reward shaping, depth-1 ReLU potential): hidden_size <= 256; no torch fallback')
    self.state_size, self.action_size, self.hidden = state_size, action_size, model_cfg.hidden_size
    self.in_dim = state_size if self.state_only else state_size + action_size
    sn = parametrizations.spectral_norm if self.spectral_norm else (lambda layer: layer)
    self.g = sn(nn.Linear(self.in_dim, 1))   # default nn.Linear init, like the reference (models.py:158)
    l1 = nn.Linear(state_size, self.hidden); nn.init.orthogonal_(l1.weight, gain=sqrt(2.0)); nn.init.constant_(l1.bias, 0); l1 = sn(l1)
    l2 = nn.Linear(self.hidden, 1); nn.init.orthogonal_(l2.weight, gain=1.0); nn.init.constant_(l2.bias, 0); l2 = sn(l2)
    self.h = nn.Sequential(l1, nn.ReLU(), l2)
    offs, o = [], 0
    for p in self.parameters():
      offs.append(o); o += p.numel()
    assert o == int(_lib.lib().il_disc_shaped_numel(state_size, action_size, self.hidden, int(self.state_only)))
    dev = device or default_device()
    self._adopt(o, offs, dev)
    H, D, S = self.hidden, self.in_dim, state_size
    self.sn = torch.zeros(2 + D + 2 * H + S, device=dev)
    self._sn_slices = dict(ug=(0, 1), vg=(1, D), u1=(1 + D, H), v1=(1 + D + H, S), u2=(1 + D + H + S, 1), v2=(2 + D + H + S, H))
    if self.spectral_norm:
      with torch.no_grad():
        for mod, (ku, kv) in ((self.g.parametrizations.weight[0], ('ug', 'vg')), (self.h[0].parametrizations.weight[0], ('u1', 'v1')), (self.h[2].parametrizations.weight[0], ('u2', 'v2'))):
          u, v = self.views()[ku], self.views()[kv]
          u.copy_(mod._u); v.copy_(mod._v)
          mod._buffers['_u'], mod._buffers['_v'] = u, v
    self.eval()

  def views(self):
    return {k: self.sn[o:o + n] for k, (o, n) in self._sn_slices.items()}

  def predict_reward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import shaped_predict_reward
    assert next_state is not None and terminal is not None, 'reward shaping: pass next_state and terminal (make_gail_input does)'
    assert (log_policy is not None) == bool(self.subtract_log_policy)
    return shaped_predict_reward(self, state, action, next_state, terminal, log_policy=log_policy)

  def forward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import shaped_predict_reward
    return shaped_predict_reward(self, state, action, next_state, terminal, log_policy=log_policy, want_logits=True)[1]


class ShapedDeepGAILDiscriminator(GAILDiscriminator):
  """GAIL discriminator with reward shaping whose potential is any `_create_fcnn` shape (reference models.py:152-180 with reward_shaping=True and
  discriminator.depth in {1, 2}, activation in {relu, tanh}): g = Linear(Dg, 1), h = [Linear - act] x depth - Linear(H, 1) on the state, every Linear optionally under
  spectral norm. Created through `GAILDiscriminator(...)` when the potential is not the depth-1 ReLU shape ShapedGAILDiscriminator serves. Flat arena = parameters()
  order; `self.sn` = ug[1] | vg[Dg] | per layer of h [u | v]; same module order, state_dict keys and RNG consumption at construction as the reference."""

  def __init__(self, state_size: int, action_size: int, imitation_cfg, discount: float, device=None):
    nn.Module.__init__(self)
    model_cfg = imitation_cfg.discriminator
    self.discount, self.state_only = discount, bool(imitation_cfg.state_only)
    self.reward_shaping, self.subtract_log_policy, self.reward_function = True, model_cfg.subtract_log_policy, model_cfg.reward_function
    self.spectral_norm = bool(imitation_cfg.spectral_norm)
    self.depth, self.activation = int(model_cfg.depth), str(model_cfg.activation)
    self.state_size, self.action_size, self.hidden = state_size, action_size, int(model_cfg.hidden_size)
    self.in_dim = state_size if self.state_only else state_size + action_size
    if self.depth not in (1, 2) or self.activation not in ('relu', 'tanh') or self.hidden > 128 or self.hidden < 2 or state_size > 128 or self.in_dim > 256:
      raise NotImplementedError(f'GAILDiscriminator (reward shaping): the HIP path implements a potential of depth 1-2 with relu / tanh, hidden_size <= 128, state <= 128 '
                                f'(got depth={self.depth}, activation={self.activation}, hidden_size={self.hidden}, state={state_size}); no torch fallback')
    L = _lib.lib()
    lds = int(L.il_disc_shaped_deep_lds_bytes(state_size, action_size, self.hidden, self.depth, int(self.state_only)))
    if lds > 160 * 1024:
      raise NotImplementedError(f'GAILDiscriminator (reward shaping): state {state_size} x hidden {self.hidden} x depth {self.depth} needs {lds} bytes of LDS per workgroup (> 160 KiB)')
    sn = parametrizations.spectral_norm if self.spectral_norm else (lambda layer: layer)
    self.g = sn(nn.Linear(self.in_dim, 1))   # default nn.Linear init, like the reference (models.py:158)
    act = nn.ReLU if self.activation == 'relu' else nn.Tanh
    dims, layers = [state_size] + [self.hidden] * self.depth, []
    for a, b in zip(dims[:-1], dims[1:]):   # models.py:49-70 `_create_fcnn`
      lin = nn.Linear(a, b)
      nn.init.orthogonal_(lin.weight, gain=nn.init.calculate_gain(self.activation)); nn.init.constant_(lin.bias, 0)
      layers += [sn(lin), act()]
    last = nn.Linear(self.hidden, 1)
    nn.init.orthogonal_(last.weight, gain=1.0); nn.init.constant_(last.bias, 0)
    self.h = nn.Sequential(*layers, sn(last))
    offs, o = [], 0
    for p in self.parameters():
      offs.append(o); o += p.numel()
    assert o == int(L.il_disc_shaped_deep_numel(state_size, action_size, self.hidden, self.depth, int(self.state_only)))
    dev = device or default_device()
    self._adopt(o, offs, dev)
    self.sn = torch.zeros(int(L.il_disc_shaped_deep_sn_numel(state_size, action_size, self.hidden, self.depth, int(self.state_only))), device=dev)
    if self.spectral_norm:
      mods, o = [self.g.parametrizations.weight[0]] + [self.h[2 * l].parametrizations.weight[0] for l in range(self.depth + 1)], 0
      with torch.no_grad():
        for mod in mods:
          nu, nv = mod._u.numel(), mod._v.numel()
          u, v = self.sn[o:o + nu], self.sn[o + nu:o + nu + nv]
          u.copy_(mod._u); v.copy_(mod._v)
          mod._buffers['_u'], mod._buffers['_v'] = u, v
          o += nu + nv
      assert o == self.sn.numel()
    self.eval()

  def predict_reward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import shaped_predict_reward
    assert next_state is not None and terminal is not None, 'reward shaping: pass next_state and terminal (make_gail_input does)'
    assert (log_policy is not None) == bool(self.subtract_log_policy)
    return shaped_predict_reward(self, state, action, next_state, terminal, log_policy=log_policy)

  def forward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import shaped_predict_reward
    return shaped_predict_reward(self, state, action, next_state, terminal, log_policy=log_policy, want_logits=True)[1]


class DeepGAILDiscriminator(GAILDiscriminator):
  """GAIL discriminator of any `_create_fcnn` shape without reward shaping (reference models.py:152-162): depth 1-2, relu / tanh, every Linear optionally
  under spectral norm.  Created through `GAILDiscriminator(...)` for configurations other than depth 1 / relu (which keep the fast path).  Flat arena =
  parameters() order, `self.sn` = per layer [u | v]; same module order, state_dict keys and RNG consumption at construction as the reference."""

  def __init__(self, state_size: int, action_size: int, imitation_cfg, discount: float, device=None):
    nn.Module.__init__(self)
    model_cfg = imitation_cfg.discriminator
    self.discount, self.state_only = discount, bool(imitation_cfg.state_only)
    self.reward_shaping, self.subtract_log_policy, self.reward_function = False, model_cfg.subtract_log_policy, model_cfg.reward_function
    self.spectral_norm = bool(imitation_cfg.spectral_norm)
    self.depth, self.activation = int(model_cfg.depth), str(model_cfg.activation)
    self.state_size, self.action_size, self.hidden = state_size, action_size, int(model_cfg.hidden_size)
    self.in_dim = state_size if self.state_only else state_size + action_size
    if self.depth not in (1, 2) or self.activation not in ('relu', 'tanh') or self.hidden > 128 or self.in_dim > 128:
      raise NotImplementedError(f'GAILDiscriminator: the HIP path implements depth 1-2 with relu / tanh, hidden_size <= 128, input <= 128 '
                                f'(got depth={self.depth}, activation={self.activation}, hidden_size={self.hidden}); no torch fallback')
    lds = int(_lib.lib().il_disc_deep_lds_bytes(self.in_dim, self.hidden, self.depth))
    if lds > 160 * 1024:
      raise NotImplementedError(f'GAILDiscriminator: input {self.in_dim} x hidden {self.hidden} x depth {self.depth} needs {lds} bytes of LDS per workgroup (> 160 KiB)')
    sn = parametrizations.spectral_norm if self.spectral_norm else (lambda layer: layer)
    act = nn.ReLU if self.activation == 'relu' else nn.Tanh
    dims, layers = [self.in_dim] + [self.hidden] * self.depth, []
    for a, b in zip(dims[:-1], dims[1:]):   # models.py:49-70 `_create_fcnn`
      lin = nn.Linear(a, b)
      nn.init.orthogonal_(lin.weight, gain=nn.init.calculate_gain(self.activation)); nn.init.constant_(lin.bias, 0)
      layers += [sn(lin), act()]
    last = nn.Linear(self.hidden, 1)
    nn.init.orthogonal_(last.weight, gain=1.0); nn.init.constant_(last.bias, 0)
    self.g = nn.Sequential(*layers, sn(last))
    offs, o = [], 0
    for p in self.parameters():
      offs.append(o); o += p.numel()
    assert o == int(_lib.lib().il_disc_deep_numel(self.in_dim, self.hidden, self.depth))
    dev = device or default_device()
    self._adopt(o, offs, dev)
    self.sn = torch.zeros(int(_lib.lib().il_disc_deep_sn_numel(self.in_dim, self.hidden, self.depth)), device=dev)
    self._sn_slices, o = [], 0
    for l in range(self.depth + 1):
      n_out, n_in = (1 if l == self.depth else self.hidden), (self.in_dim if l == 0 else self.hidden)
      self._sn_slices.append(((o, n_out), (o + n_out, n_in))); o += n_out + n_in
    if self.spectral_norm:
      with torch.no_grad():
        for l, ((ou, nu), (ov, nv)) in enumerate(self._sn_slices):
          mod = self.g[2 * l].parametrizations.weight[0]
          u, v = self.sn[ou:ou + nu], self.sn[ov:ov + nv]
          u.copy_(mod._u); v.copy_(mod._v)
          mod._buffers['_u'], mod._buffers['_v'] = u, v
    self.eval()

  def predict_reward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import deep_predict_reward
    assert (log_policy is not None) == bool(self.subtract_log_policy)
    return deep_predict_reward(self, state, action, log_policy=log_policy)

  def forward(self, state: Tensor, action: Tensor, next_state=None, terminal=None, log_policy=None) -> Tensor:
    from .training import deep_predict_reward
    return deep_predict_reward(self, state, action, log_policy=log_policy, want_logits=True)[1]


class GMMILDiscriminator(nn.Module):
  """Kernel-mean-embedding reward (reference models.py:183-201); O(B^2 D) pair work runs in k_gmmil_tile."""

  def __init__(self, state_size: int, action_size: int, imitation_cfg):
    super().__init__()
    self.state_size, self.action_size = state_size, action_size
    self.state_only = bool(imitation_cfg.state_only)
    self.gamma_1, self.gamma_2 = None, None
    self._ws = None

  def predict_reward(self, state, action, expert_state, expert_action, weight, expert_weight) -> Tensor:
    from .training import gmmil_predict_reward
    return gmmil_predict_reward(self, state, action, expert_state, expert_action, weight, expert_weight)


def _calculate_normalisation_scale_offset(data: Tensor) -> Tuple[Tensor, Tensor]:
  """Reference models.py:204-207. torch's CPU reductions accumulate float32 inputs in double (acc_type) and round once; a float32 tree reduction over
  25,000 rows is off by ~4e-6, which PWIL's exp(-beta T / sqrt(D) * cost) turns into 1e-3 of reward. One-off statistics: accumulate in float64 too."""
  d64 = data.double()
  inv_scale, offset = d64.std(dim=0, keepdim=True).float(), (-d64.mean(dim=0, keepdim=True)).float()
  inv_scale[inv_scale == 0] = 1
  return 1 / inv_scale, offset


class PWILDiscriminator(nn.Module):
  """Greedy Wasserstein coupling reward (reference models.py:216-249) on device-resident standardised expert atoms."""

  def __init__(self, state_size: int, action_size: int, imitation_cfg, expert_memory, time_horizon: int):
    super().__init__()
    self.state_only = bool(imitation_cfg.state_only)
    self.state_size, self.action_size = state_size, action_size
    self.expert_memory, self.time_horizon = expert_memory, time_horizon
    raw = self._get_expert_atoms().contiguous()
    self.data_scale, self.data_offset = _calculate_normalisation_scale_offset(raw)
    dim = state_size if self.state_only else state_size + action_size
    self.reward_scale, self.reward_bandwidth = imitation_cfg.reward_scale, imitation_cfg.reward_bandwidth_scale * time_horizon / sqrt(dim)
    self.expert_atoms = (self.data_scale * (raw + self.data_offset)).contiguous()
    n = self.expert_atoms.size(0)
    scratch = int(_lib.lib().il_pwil_scratch_floats(n, 1 / time_horizon - 1e-6))
    self.expert_weights, self._dists, self._out = torch.empty(n, device=raw.device), torch.empty(scratch, device=raw.device), torch.empty(1, device=raw.device)
    self._scale, self._offset = self.data_scale.flatten().contiguous(), self.data_offset.flatten().contiguous()
    self._desc = _lib.Pwil(n, dim, state_size, action_size, self.expert_atoms.data_ptr(), self.expert_weights.data_ptr(), self._dists.data_ptr(), self._scale.data_ptr(),
                           self._offset.data_ptr(), float(self.reward_scale), float(self.reward_bandwidth), 1 / time_horizon - 1e-6)
    self.reset()

  def _get_expert_atoms(self) -> Tensor:
    return self.expert_memory['states'] if self.state_only else torch.cat([self.expert_memory['states'], self.expert_memory['actions']], dim=1)

  def reset(self):
    _lib.check(_lib.lib().il_pwil_reset(C.byref(self._desc), _lib.stream_ptr()))

  def compute_reward_async(self, state: Tensor, action: Tensor) -> Tensor:
    """Enqueues the greedy coupling for one (state, action); returns the 1-element device tensor (no host sync)."""
    dev = self.expert_atoms.device
    state, action = state.to(dev, torch.float32).contiguous(), action.to(dev, torch.float32).contiguous()
    _lib.check(_lib.lib().il_pwil_reward(C.byref(self._desc), _lib.ptr(state), _lib.ptr(action), _lib.ptr(self._out), _lib.stream_ptr()))
    return self._out

  def compute_reward(self, state: Tensor, action: Tensor) -> float:
    return float(self.compute_reward_async(state, action).item())


class REDDiscriminator(_FlatModule):
  """Random Expert Distillation (reference models.py:252-284): a predictor regressed onto a frozen random target network, both `_create_fcnn`
  MLPs D -> H (-> H) -> D with ReLU / Tanh; the predictor additionally carries the config's input_dropout / dropout layers (active in train mode:
  target_estimation_update and set_sigma, train.py:115-128; predict_reward runs after `discriminator.eval()`, train.py:147).
  reward = exp(-sigma_1 * mean (pred - target)^2).  `self.flat` is the predictor arena (what the optimiser owns), `self.target_flat` the frozen one;
  state_dict keys are the reference's (predictor.embedding.N.*, target.embedding.N.*: Dropout / activation modules occupy Sequential slots too).
  Pass `masks=(mask_in, mask_h1[, mask_h2])` (keep-masks, 0/1) to reproduce given dropout draws; otherwise they come from the on-chip Philox stream."""

  def __init__(self, state_size: int, action_size: int, imitation_cfg, device=None):
    super().__init__()
    model_cfg = imitation_cfg.discriminator
    self.depth, self.activation = int(model_cfg.depth), str(model_cfg.activation)
    self.p_in, self.p = float(_cfg_get(model_cfg, 'input_dropout', 0) or 0), float(_cfg_get(model_cfg, 'dropout', 0) or 0)
    if self.depth not in (1, 2) or self.activation not in ('relu', 'tanh'):
      raise NotImplementedError(f'REDDiscriminator: the HIP path implements depth 1-2 with relu / tanh (got depth={self.depth}, activation={self.activation}); no torch fallback')
    self.state_size, self.action_size, self.hidden, self.state_only = state_size, action_size, int(model_cfg.hidden_size), bool(imitation_cfg.state_only)
    self.in_dim = state_size if self.state_only else state_size + action_size
    if self.in_dim > 128 or self.hidden > 256 or self.hidden % 2:
      raise NotImplementedError(f'REDDiscriminator: input {self.in_dim} (<= 128) / hidden {self.hidden} (even, <= 256) outside the kernel limits')
    act, gain = (nn.ReLU, sqrt(2.0)) if self.activation == 'relu' else (nn.Tanh, 5.0 / 3.0)

    def embedding(p_in, p):   # models.py:49-70 `_create_fcnn`, same module order (hence the same state_dict keys) and RNG consumption
      dims, layers = [self.in_dim] + [self.hidden] * self.depth, []
      if p_in > 0: layers.append(nn.Dropout(p_in))
      for a, b in zip(dims[:-1], dims[1:]):
        lin = nn.Linear(a, b)
        nn.init.orthogonal_(lin.weight, gain=gain); nn.init.constant_(lin.bias, 0)
        layers.append(lin)
        if p > 0: layers.append(nn.Dropout(p))
        layers.append(act())
      last = nn.Linear(dims[-1], self.in_dim)
      nn.init.orthogonal_(last.weight, gain=1.0); nn.init.constant_(last.bias, 0)
      holder = nn.Module()
      holder.embedding = nn.Sequential(*layers, last)
      return holder
    self.predictor, self.target = embedding(self.p_in, self.p), embedding(0, 0)   # same construction order as the reference: identical RNG consumption
    dev = device or default_device()
    P = int(_lib.lib().il_red_numel(self.in_dim, self.hidden, self.depth))
    flats = []
    for net in (self.predictor, self.target):
      offs, o = [], 0
      for p in net.parameters():
        offs.append(o); o += p.numel()
      assert o == P
      flat = torch.zeros(P, dtype=torch.float32, device=dev)
      with torch.no_grad():
        _flatten_into(list(net.parameters()), flat, offs)
      flats.append(flat)
    self.flat, self.target_flat = flats
    for p in self.parameters():
      p.requires_grad_(False)
    self.sigma_1 = imitation_cfg.reward_bandwidth_scale
    self._noise_calls = 0

  def _desc(self, batch_size: int, opt=None) -> '_lib.Red':
    d = _lib.Red()
    d.state_dim, d.action_dim, d.hidden, d.batch, d.state_only = self.state_size, self.action_size, self.hidden, batch_size, int(self.state_only)
    d.depth, d.activation, d.p_in, d.p = self.depth, int(self.activation == 'tanh'), self.p_in, self.p
    d.noise_seed = torch.initial_seed() & (2**64 - 1)
    d.predictor, d.target = self.flat.data_ptr(), self.target_flat.data_ptr()
    d.sigma_1 = float(self.sigma_1) if self.sigma_1 else 0.0
    if opt is not None:
      from .training import _workspace
      ws = _workspace('red', int(_lib.lib().il_red_workspace_floats(self.in_dim, self.hidden, batch_size, self.depth)), self.flat.device)
      d.grad, d.opt, d.workspace = opt.grad.data_ptr(), opt.desc(), ws.data_ptr()
    return d

  def _masks(self, masks, rows: int):
    """(mask_in, mask_h1, mask_h2) pointers for the C ABI (None = on-chip Philox) + the tensors to keep alive, and this call's Philox counter."""
    self._noise_calls += 1
    if masks is None:
      return (None, None, None), (), self._noise_calls & 0xFFFFFFFF
    dev, keep = self.flat.device, []
    want = [(rows, self.in_dim)] + [(rows, self.hidden)] * self.depth
    assert len(masks) == len(want), f'REDDiscriminator: expected {len(want)} masks (input + one per hidden layer)'
    for m, shape in zip(masks, want):
      m = m.to(dev, torch.float32).contiguous()
      assert tuple(m.shape) == shape, f'mask shape {tuple(m.shape)} != {shape}'
      keep.append(m)
    ptrs = [_lib.ptr(m) for m in keep] + [None] * (3 - len(keep))
    return tuple(ptrs), tuple(keep), self._noise_calls & 0xFFFFFFFF

  def _batch(self, state: Tensor, action: Tensor):
    from .training import _sa_batch
    dev = self.flat.device
    state = state.to(dev, torch.float32)
    action = action.to(dev, torch.float32)
    if state.stride(-1) != 1: state = state.contiguous()
    if action.stride(-1) != 1: action = action.contiguous()
    ones = torch.ones(state.size(0), device=dev)
    return _sa_batch(state, action, ones), state.size(0), (state, action, ones)   # the il_batch holds raw pointers: the caller keeps the tensors alive

  def forward(self, state: Tensor, action: Tensor, masks=None) -> Tuple[Tensor, Tensor]:
    b, n, keep = self._batch(state, action)
    pred, targ = torch.empty(n, self.in_dim, device=self.flat.device), torch.empty(n, self.in_dim, device=self.flat.device)
    d = self._desc(n)
    (m0, m1, m2), alive, ctr = self._masks(masks, n) if self.training else ((None, None, None), (), 0)
    _lib.check(_lib.lib().il_red_forward(C.byref(d), C.byref(b), int(self.training), m0, m1, m2, ctr, None, _lib.ptr(pred), _lib.ptr(targ), _lib.stream_ptr()))
    return pred, targ

  def set_sigma(self, expert_state: Tensor, expert_action: Tensor, masks=None):
    """models.py:274-277: kernel median heuristic on one expert minibatch unless reward_bandwidth_scale was configured (train mode: dropout is active)."""
    if not self.sigma_1:
      from .training import embedding_sqdist
      pred, targ = self.forward(expert_state, expert_action, masks=masks)
      self.sigma_1 = 1 / embedding_sqdist(pred, targ).flatten().median().item()

  def predict_reward(self, state: Tensor, action: Tensor, masks=None) -> Tensor:
    assert self.sigma_1, 'REDDiscriminator.predict_reward before set_sigma (train.py:128)'
    b, n, keep = self._batch(state, action)
    out = torch.empty(n, device=self.flat.device)
    d = self._desc(n)
    (m0, m1, m2), alive, ctr = self._masks(masks, n) if self.training else ((None, None, None), (), 0)
    _lib.check(_lib.lib().il_red_forward(C.byref(d), C.byref(b), int(self.training), m0, m1, m2, ctr, _lib.ptr(out), None, None, _lib.stream_ptr()))
    return out


def _packed_rows(t: Dict[str, Tensor]) -> Tuple[Tensor, int, int]:
  """The packed [n, row] tensor a `ReplayMemory.sample` dict views into (plus state / action size), or TypeError."""
  st, ac, rw = t['states'], t['actions'], t['rewards']
  base = st._base if st._base is not None else None
  S, A = st.size(1), ac.size(1)
  row = int(_lib.lib().il_ring_row_floats(S, A))
  ok = (base is not None and _lib.on_device(base) and base.dim() == 2 and base.size(1) == row and base.is_contiguous() and st.data_ptr() == base.data_ptr()
        and ac.data_ptr() == base.data_ptr() + 4 * S and rw.data_ptr() == base.data_ptr() + 4 * (2 * S + A) and st.size(0) == base.size(0))
  if not ok:
    raise TypeError('expected a transitions dict produced by ReplayMemory.sample / batch_views (fields are views into packed device rows)')
  return base, S, A


def _mix_relabel(transitions, expert_transitions, n_expert: int, label: int = 0, update_freq: int = 0, round_num: int = 0, reward_expert: float = 0.0, policy_trajectories: int = 1):
  rows, S, A = _packed_rows(transitions)
  erows = _packed_rows(expert_transitions)[0] if n_expert else None
  assert erows is None or erows.size(0) >= n_expert
  _lib.check(_lib.lib().il_batch_mix_relabel(_lib.ptr(rows), _lib.ptr(erows), rows.size(0), S, A, n_expert, label, update_freq, round_num, reward_expert, policy_trajectories, _lib.stream_ptr()))


def mix_expert_agent_transitions(transitions: Dict[str, Tensor], expert_transitions: Dict[str, Tensor]):
  """Reference models.py:287-290: first half of EVERY key is overwritten with expert rows (in place); one k_mix_relabel launch."""
  _mix_relabel(transitions, expert_transitions, transitions['rewards'].size(0) // 2)


class RewardRelabeller:
  """AdRIL / SQIL constant-reward relabelling (reference models.py:293-318) as one k_mix_relabel launch per update.

  balanced: alternate all-expert and all-policy batches (stateful, expert first); otherwise the first half of the batch is expert data.
  update_freq > 0 (AdRIL): expert +1/|expert trajectories|, policy 0 for the current round / -1/|policy trajectories| for older rounds;
  update_freq == 0 (SQIL): expert 1, policy 0.  The batch is rewritten in place (the reference rebinds the dict entries in the
  all-expert case; here the expert rows are copied over the policy rows, which yields the same values)."""

  def __init__(self, update_freq: int, balanced: bool):
    self.update_freq, self.balanced, self.sample_expert = int(update_freq), bool(balanced), True

  def resample_and_relabel(self, transitions: Dict[str, Tensor], expert_transitions: Dict[str, Tensor], step: int, num_trajectories: int, num_expert_trajectories: int):
    B = transitions['rewards'].size(0)
    if self.balanced:
      n_expert = B if self.sample_expert else 0
      self.sample_expert = not self.sample_expert
    else:
      n_expert = B // 2
    if self.update_freq > 0:
      import numpy as np
      _mix_relabel(transitions, expert_transitions, n_expert, 2, self.update_freq, -(-int(step) // self.update_freq), float(np.float32(1 / num_expert_trajectories)), int(num_trajectories))
    else:
      _mix_relabel(transitions, expert_transitions, n_expert, 1)
