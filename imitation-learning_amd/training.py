"""Update functions with the reference's signatures (reference training.py) forwarding to libil_hip.so.

`sac_update`, `adversarial_imitation_update`, `behavioural_cloning_update` keep the argument lists of the reference (plus
optional keyword-only noise tensors used by the parity tests); every FLOP happens in the HIP kernels.  `UpdatePlan` is the
same sequence (train.py:173-203 for algorithm=SAC/GAIL) with persistent buffers so that it can be captured in a hipGraph.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .memory import ReplayMemory, batch_desc, batch_views
from .models import LOSS_FUNCTIONS, REWARD_FUNCTIONS, DropoutSoftActor, GAILDiscriminator, GMMILDiscriminator, SoftActor, TwinCritic
from .optim import Adam, AdamW

_WS: Dict[tuple, Tensor] = {}
_NOISE: Dict[torch.device, Tensor] = {}


def _workspace(kind: str, floats: int, device, tag=None, zero: bool = False) -> Tensor:
  """Scratch arena shared by every call of one kind on a device; `tag` gives a learner its own (learners that run concurrently must not share).
  zero=True: zero-filled at creation and never shared between sizes (kernels that keep self-resetting arrival counters in it: il_gmmil_reward)."""
  key = (kind, str(device), tag)
  ws = _WS.get(key)
  if ws is None or ws.numel() < floats or (zero and ws.numel() != floats):
    ws = (torch.zeros if zero else torch.empty)(int(floats), dtype=torch.float32, device=device)
    _WS[key] = ws
  return ws


def _noise_counter(device, tag=None) -> Tensor:
  if (device, tag) not in _NOISE:
    _NOISE[(device, tag)] = torch.zeros(1, dtype=torch.int32, device=device)
  return _NOISE[(device, tag)]


def _noise_seed() -> int:
  return torch.initial_seed() & (2**64 - 1)


def _cfg_value(cfg, key, default=None):
  return cfg.get(key, default) if hasattr(cfg, 'get') else getattr(cfg, key, default)


def _f32(t: Optional[Tensor], device) -> Optional[Tensor]:
  return None if t is None else t.to(device, torch.float32).contiguous()


def sac_descriptor(actor: SoftActor, critic: TwinCritic, log_alpha: Tensor, target_critic: TwinCritic, batch_size: int, actor_optimiser: AdamW, critic_optimiser: AdamW,
                   temperature_optimiser: Adam, discount: float, entropy_target: float, polyak_factor: float, tag=None, seed_offset: int = 0) -> _lib.Sac:
  S, A, H, dev = actor.state_size, actor.action_size, actor.hidden, actor.flat.device
  assert _lib.on_device(log_alpha) and log_alpha.dtype == torch.float32
  if isinstance(actor, DropoutSoftActor):
    raise NotImplementedError('sac_update: a dropout policy ensemble (DRIL discriminator) is not a reinforcement-learning actor of the HIP path')
  if _general_shape(actor, critic):   # csrc/general.hip: its own (larger) scratch layout; reinforcement.actor and reinforcement.critic may differ in every dimension
    floats = int(_lib.lib().il_sac_workspace_floats_general(S, A, H, actor.depth, critic.hidden, critic.depth, batch_size))
  else:
    assert critic.hidden == H
    # the fused kernels index the arenas by the depth-2 layout: refuse anything whose arena is not exactly that long (an out-of-bounds optimiser step otherwise)
    want_a, want_c = H * S + H + H * H + H + 2 * A * H + 2 * A, 2 * int(_lib.lib().il_mlp_stride(S + A, H, 1))
    if actor.flat.numel() != want_a or critic.flat.numel() != want_c or target_critic.flat.numel() != want_c:
      raise ValueError(f'sac_update: actor / critic arenas of {actor.flat.numel()} / {critic.flat.numel()} floats do not have the fused depth-2 layout ({want_a} / {want_c})')
    floats = int(_lib.lib().il_sac_workspace_floats(S, A, H, batch_size))
  for opt, net in ((actor_optimiser, actor), (critic_optimiser, critic)):
    if opt.grad.numel() < net.flat.numel():
      raise ValueError('sac_update: optimiser state arena shorter than its network')
  ws = _workspace('sac', floats, dev, tag)
  d = _lib.Sac()
  d.state_dim, d.action_dim, d.hidden, d.batch = S, A, H, batch_size
  d.actor, d.critic, d.target, d.log_alpha = actor.flat.data_ptr(), critic.flat.data_ptr(), target_critic.flat.data_ptr(), log_alpha.data_ptr()
  d.actor_grad, d.critic_grad, d.alpha_grad = actor_optimiser.grad.data_ptr(), critic_optimiser.grad.data_ptr(), temperature_optimiser.grad.data_ptr()
  d.actor_opt, d.critic_opt, d.alpha_opt = actor_optimiser.desc(), critic_optimiser.desc(), temperature_optimiser.desc()
  d.discount, d.entropy_target, d.polyak = float(discount), float(entropy_target), float(polyak_factor)
  d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
  d._ws = ws   # the descriptor holds a raw pointer: keep the arena alive with it (a later, larger request under the same key replaces the cached tensor)
  d.noise_seed, d.noise_counter = (_noise_seed() + seed_offset) & (2**64 - 1), _noise_counter(dev, tag).data_ptr()
  return d


def _general_shape(actor, critic=None) -> bool:
  """True when the networks are outside the fused kernels' shape (models._mlp_shape) and run through csrc/general.hip."""
  return bool(getattr(actor, 'general', False) or (critic is not None and getattr(critic, 'general', False)))


def sac_update(actor: SoftActor, critic: TwinCritic, log_alpha: Tensor, target_critic: TwinCritic, transitions: Dict[str, Tensor], actor_optimiser: AdamW,
               critic_optimiser: AdamW, temperature_optimiser: Adam, discount: float, entropy_target: float, polyak_factor: float, *,
               eps_next: Optional[Tensor] = None, eps_cur: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
  """One SAC update (reference training.py:14-54). Returns (log_probs[B], min(Q1,Q2)[B]) like the reference."""
  dev = actor.flat.device
  B = transitions['states'].size(0)
  d = sac_descriptor(actor, critic, log_alpha, target_critic, B, actor_optimiser, critic_optimiser, temperature_optimiser, discount, entropy_target, polyak_factor)
  b = batch_desc(transitions)
  logp, q = torch.empty(B, device=dev), torch.empty(B, device=dev)
  e1, e2 = _f32(eps_next, dev), _f32(eps_cur, dev)
  if _general_shape(actor, critic):
    from .models import ACTIVATION_IDS
    _lib.check(_lib.lib().il_sac_update_general(C.byref(d), C.byref(b), actor.depth, ACTIVATION_IDS[actor.activation], critic.hidden, critic.depth, ACTIVATION_IDS[critic.activation],
                                                _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(logp), _lib.ptr(q), 0, _lib.stream_ptr()))
    return logp, q
  _lib.check(_lib.lib().il_sac_update(C.byref(d), C.byref(b), _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(logp), _lib.ptr(q), 0, _lib.stream_ptr()))
  return logp, q


def behavioural_cloning_update(actor: SoftActor, expert_transition: Dict[str, Tensor], actor_optimiser: AdamW, *, masks=None) -> Tensor:
  """Reference training.py:57-64. Returns the (device) loss for logging; the reference returns None.
  The DRIL policy ensemble (depth-1 tanh dropout network, train.py:119) takes the k_dril_* path; `masks` = its dropout keep-masks, if given."""
  if hasattr(actor, 'bc_update'):
    return actor.bc_update(expert_transition, actor_optimiser, masks=masks, want_loss=True)
  dev = actor.flat.device
  t = {k: v for k, v in expert_transition.items()}
  for k in ('rewards', 'next_states', 'terminals', 'absorbing'):  # not read by k_bc_tile, but il_batch wants valid pointers
    t.setdefault(k, t['weights'] if k != 'next_states' else t['states'])
  b = batch_desc(t)
  S, A, H, B = actor.state_size, actor.action_size, actor.hidden, b.n
  if _general_shape(actor):
    from .models import ACTIVATION_IDS
    ws, loss, od = actor._general_workspace(B), torch.empty(1, device=dev), actor_optimiser.desc()
    _lib.check(_lib.lib().il_bc_step_general(_lib.ptr(actor.flat), _lib.ptr(actor_optimiser.grad), C.byref(od), S, A, H, actor.depth, ACTIVATION_IDS[actor.activation], C.byref(b), _lib.ptr(ws),
                                             ws.numel(), _lib.ptr(loss), 0, _lib.stream_ptr()))
    return loss[0]
  floats = int(_lib.lib().il_sac_workspace_floats(S, A, H, B))
  ws = _workspace('sac', floats, dev)
  parts = torch.empty(B // 16, device=dev)
  od = actor_optimiser.desc()
  _lib.check(_lib.lib().il_bc_step(_lib.ptr(actor.flat), _lib.ptr(actor_optimiser.grad), C.byref(od), S, A, H, C.byref(b), _lib.ptr(ws), ws.numel(), _lib.ptr(parts), 0,
                                   _lib.stream_ptr()))
  return parts.sum() / B


def target_estimation_update(discriminator, expert_transition: Dict[str, Tensor], discriminator_optimiser: AdamW, *, want_loss: bool = False, masks=None):
  """Reference training.py:68-75: regress the RED predictor onto the frozen target on one weighted expert batch (k_red_grad + k_red_apply), train mode:
  `masks` = the predictor's dropout keep-masks (input, hidden 1[, hidden 2]) when they must be reproduced, else drawn on chip."""
  t = dict(expert_transition)
  for k in ('rewards', 'terminals', 'absorbing', 'next_states'):
    t.setdefault(k, t['weights'] if k != 'next_states' else t['states'])
  b = batch_desc(t)
  d = discriminator._desc(b.n, discriminator_optimiser)
  loss = torch.empty(1, device=discriminator.flat.device) if want_loss else None
  (m0, m1, m2), alive, ctr = discriminator._masks(masks, b.n)
  _lib.check(_lib.lib().il_red_step(C.byref(d), C.byref(b), m0, m1, m2, ctr, _lib.ptr(loss), 0, _lib.stream_ptr()))
  return loss


# ----------------------------------------------------------------------------------------------- GAIL
def disc_descriptor(disc: GAILDiscriminator, batch_size: int, opt: AdamW, imitation_cfg=None, grad_penalty: float = 0.0, entropy_bonus: float = 0.0, tag=None,
                    seed_offset: int = 0) -> _lib.Disc:
  dev = disc.flat.device
  loss_function, prior, margin = 'BCE', 0.0, float('inf')
  if imitation_cfg is not None:
    loss_function, prior = imitation_cfg.loss_function, float(_cfg_value(imitation_cfg, 'pos_class_prior', 0.0) or 0.0)
    if loss_function not in LOSS_FUNCTIONS:
      raise ValueError(f'adversarial_imitation_update: unknown loss_function={loss_function}')
    margin = float(_cfg_value(imitation_cfg, 'nonnegative_margin', float('inf')))
    grad_penalty, entropy_bonus = float(imitation_cfg.grad_penalty), float(imitation_cfg.entropy_bonus)
  floats = int(_lib.lib().il_disc_workspace_floats(disc.in_dim, disc.hidden, batch_size))
  ws = _workspace('disc', floats, dev, tag)
  v = disc.views()
  d = _lib.Disc()
  d.state_dim, d.action_dim, d.hidden, d.batch = disc.state_size, disc.action_size, disc.hidden, batch_size
  d.spectral_norm, d.state_only, d.reward_function = int(disc.spectral_norm), int(disc.state_only), REWARD_FUNCTIONS[disc.reward_function]
  d.params, d.u1, d.v1, d.u2, d.v2 = disc.flat.data_ptr(), v['u1'].data_ptr(), v['v1'].data_ptr(), v['u2'].data_ptr(), v['v2'].data_ptr()
  if opt is not None:
    d.grad, d.opt = opt.grad.data_ptr(), opt.desc()
  else:
    d.grad = ws.data_ptr()  # reward only: never written
  d.grad_penalty, d.entropy_bonus = grad_penalty, entropy_bonus
  d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
  d._ws = ws   # the descriptor holds a raw pointer: keep the arena alive with it (a later, larger request under the same key replaces the cached tensor)
  d.noise_seed, d.noise_counter = (_noise_seed() + seed_offset) & (2**64 - 1), _noise_counter(dev, tag).data_ptr()
  d.loss_function, d.pos_class_prior = LOSS_FUNCTIONS[loss_function], prior
  if loss_function == 'PUGAIL' and margin != float('inf'):   # training.py:102: torch.clamp(..., min=-nonnegative_margin) on the batch-wide value (a value pass precedes the gradients)
    d.pu_clamped, d.nonnegative_margin = 1, margin
  return d


def _shaped_general(disc) -> bool:
  return type(disc).__name__ == 'ShapedDeepGAILDiscriminator'   # a depth-2 and / or tanh potential: gail_shaped_deep.hip, same arguments


def shaped_descriptor(disc, batch_size: int, opt, imitation_cfg=None):
  """il_disc_shaped (depth-1 ReLU potential, gail_shaped.hip) or il_disc_shaped_deep (any potential, gail_shaped_deep.hip) for a reward-shaping discriminator."""
  dev = disc.flat.device
  loss_function, prior, grad_penalty, entropy_bonus, margin = 'BCE', 0.0, 0.0, 0.0, float('inf')
  if imitation_cfg is not None:
    loss_function, prior = imitation_cfg.loss_function, float(_cfg_value(imitation_cfg, 'pos_class_prior', 0.0) or 0.0)
    if loss_function not in LOSS_FUNCTIONS:
      raise ValueError(f'adversarial_imitation_update: unknown loss_function={loss_function}')
    margin = float(_cfg_value(imitation_cfg, 'nonnegative_margin', float('inf')))
    grad_penalty, entropy_bonus = float(imitation_cfg.grad_penalty), float(imitation_cfg.entropy_bonus)
  if _shaped_general(disc):
    ws = _workspace('disc_shaped', int(_lib.lib().il_disc_shaped_deep_workspace_floats(disc.state_size, disc.action_size, disc.hidden, disc.depth, batch_size, int(disc.state_only))), dev)
    d = _lib.DiscShapedDeep()
    d.depth, d.activation, d.sn = disc.depth, int(disc.activation == 'tanh'), disc.sn.data_ptr()
  else:
    ws = _workspace('disc_shaped', int(_lib.lib().il_disc_shaped_workspace_floats(disc.state_size, disc.action_size, disc.hidden, batch_size, int(disc.state_only))), dev)
    v = disc.views()
    d = _lib.DiscShaped()
    d.ug, d.vg, d.u1, d.v1, d.u2, d.v2 = (v[k].data_ptr() for k in ('ug', 'vg', 'u1', 'v1', 'u2', 'v2'))
  d.state_dim, d.action_dim, d.hidden, d.batch = disc.state_size, disc.action_size, disc.hidden, batch_size
  d.spectral_norm, d.state_only, d.reward_function, d.loss_function = int(disc.spectral_norm), int(disc.state_only), REWARD_FUNCTIONS[disc.reward_function], LOSS_FUNCTIONS[loss_function]
  d.params = disc.flat.data_ptr()
  if opt is not None:
    d.grad, d.opt = opt.grad.data_ptr(), opt.desc()
  d.grad_penalty, d.entropy_bonus, d.pos_class_prior, d.discount = grad_penalty, entropy_bonus, prior, float(disc.discount)
  d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
  d._ws = ws   # the descriptor holds a raw pointer: keep the arena alive with it (a later, larger request under the same key replaces the cached tensor)
  d.noise_seed, d.noise_counter = _noise_seed() & (2**64 - 1), _noise_counter(dev, None).data_ptr()   # the learner's ONE update counter (advanced by the actor step of sac_update): fresh GP / Mixup draws per update
  if loss_function == 'PUGAIL' and margin != float('inf'):   # training.py:102, as in disc_descriptor
    d.pu_clamped, d.nonnegative_margin = 1, margin
  return d


def deep_descriptor(disc, batch_size: int, opt, imitation_cfg=None) -> _lib.DiscDeep:
  """il_disc_deep for a DeepGAILDiscriminator (depth 1-2, relu / tanh; gail_deep.hip)."""
  dev = disc.flat.device
  loss_function, prior, grad_penalty, entropy_bonus, margin = 'BCE', 0.0, 0.0, 0.0, float('inf')
  if imitation_cfg is not None:
    loss_function, prior = imitation_cfg.loss_function, float(_cfg_value(imitation_cfg, 'pos_class_prior', 0.0) or 0.0)
    if loss_function not in LOSS_FUNCTIONS:
      raise ValueError(f'adversarial_imitation_update: unknown loss_function={loss_function}')
    margin = float(_cfg_value(imitation_cfg, 'nonnegative_margin', float('inf')))
    grad_penalty, entropy_bonus = float(imitation_cfg.grad_penalty), float(imitation_cfg.entropy_bonus)
  L = _lib.lib()
  ws = _workspace('disc_deep', int(L.il_disc_deep_workspace_floats(disc.in_dim, disc.hidden, disc.depth, batch_size)), dev)
  d = _lib.DiscDeep()
  d.state_dim, d.action_dim, d.hidden, d.batch = disc.state_size, disc.action_size, disc.hidden, batch_size
  d.spectral_norm, d.state_only, d.reward_function, d.loss_function = int(disc.spectral_norm), int(disc.state_only), REWARD_FUNCTIONS[disc.reward_function], LOSS_FUNCTIONS[loss_function]
  d.depth, d.activation = disc.depth, int(disc.activation == 'tanh')
  d.params, d.sn = disc.flat.data_ptr(), disc.sn.data_ptr()
  if opt is not None:
    d.grad, d.opt = opt.grad.data_ptr(), opt.desc()
  d.grad_penalty, d.entropy_bonus, d.pos_class_prior = grad_penalty, entropy_bonus, prior
  d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
  d._ws = ws   # the descriptor holds a raw pointer: keep the arena alive with it (a later, larger request under the same key replaces the cached tensor)
  d.noise_seed, d.noise_counter = _noise_seed() & (2**64 - 1), _noise_counter(dev, None).data_ptr()   # the learner's ONE update counter (advanced by the actor step of sac_update): fresh GP / Mixup draws per update
  if loss_function == 'PUGAIL' and margin != float('inf'):   # training.py:102, as in disc_descriptor
    d.pu_clamped, d.nonnegative_margin = 1, margin
  return d


def deep_predict_reward(disc, state: Tensor, action: Tensor, log_policy: Optional[Tensor] = None, want_logits: bool = False):
  dev = disc.flat.device
  state, action = _f32(state, dev), _f32(action, dev)
  n = state.size(0)
  d = deep_descriptor(disc, n, None)
  dummy = torch.zeros(n, device=dev)
  b = batch_desc(dict(states=state, actions=action, rewards=dummy, next_states=state, terminals=dummy, weights=dummy, absorbing=dummy))
  out, logits = torch.empty(n, device=dev), (torch.empty(n, device=dev) if want_logits else None)
  off = _f32(log_policy, dev)
  _lib.check(_lib.lib().il_gail_deep_reward(C.byref(d), C.byref(b), _lib.ptr(out), _lib.ptr(logits), _lib.ptr(off), _lib.stream_ptr()))
  return (out, logits) if want_logits else out


def _shaped_batch(state, action, next_state, terminal, weight=None):
  w = weight if weight is not None else terminal
  return batch_desc(dict(states=state, actions=action, rewards=w, next_states=next_state, terminals=terminal, weights=w, absorbing=w))


def shaped_predict_reward(disc, state: Tensor, action: Tensor, next_state: Tensor, terminal: Tensor, log_policy: Optional[Tensor] = None, want_logits: bool = False):
  """models.py:173-180 for the reward-shaping discriminator (eval mode)."""
  dev = disc.flat.device
  n = state.size(0)
  d = shaped_descriptor(disc, n, None)
  terminal = terminal.to(dev, torch.float32).contiguous()   # named: the il_batch below holds raw pointers
  b = _shaped_batch(state, action, next_state, terminal)
  out, logits, off = torch.empty(n, device=dev), (torch.empty(n, device=dev) if want_logits else None), _f32(log_policy, dev)
  reward = _lib.lib().il_gail_shaped_deep_reward if _shaped_general(disc) else _lib.lib().il_gail_shaped_reward
  _lib.check(reward(C.byref(d), C.byref(b), _lib.ptr(out), _lib.ptr(logits), _lib.ptr(off), _lib.stream_ptr()))
  return (out, logits) if want_logits else out


def adversarial_imitation_update(actor, discriminator: GAILDiscriminator, transitions: Dict[str, Tensor], expert_transitions: Dict[str, Tensor], discriminator_optimiser: AdamW,
                                 imitation_cfg, *, eps_gp: Optional[Tensor] = None, eps_mix: Optional[Tensor] = None, flags: int = 0):
  """Reference training.py:85-134: loss_function BCE / PUGAIL (any nonnegative_margin) / Mixup, + gradient penalty, spectral norm, entropy bonus,
  subtract_log_policy.  `eps_gp` / `eps_mix`: the U(0,1) and Beta(alpha, alpha) draws (None: drawn here). `flags` = IL_FLAG_GRADS_ONLY: the gradient is left in
  `discriminator_optimiser.grad` (the optimiser is ticked, the spectral-norm buffers advance) and the caller applies it after averaging it over ranks (parallel.DataParallelUpdate)."""
  dev = discriminator.flat.device
  B = transitions['states'].size(0)
  pb, eb = batch_desc(transitions), batch_desc(expert_transitions)
  e = _f32(eps_gp, dev)
  x, keep = _lib.GailExtra(), []
  if getattr(discriminator, 'reward_shaping', False):   # models.py:157-160: its own kernels (k_gs_* for the depth-1 ReLU potential, k_gsd_* for the others)
    d = shaped_descriptor(discriminator, B, discriminator_optimiser, imitation_cfg)
    if imitation_cfg.loss_function == 'Mixup':   # training.py:104-113 on every field of the transition (the kernel mixes next_states and terminals too)
      alpha = float(_cfg_value(imitation_cfg, 'mixup_alpha', 1.0))
      if eps_mix is None and (alpha != 1.0 or discriminator.subtract_log_policy):
        eps_mix = torch.distributions.Beta(torch.full((B,), alpha), torch.full((B,), alpha)).sample()
      if eps_mix is not None:
        keep.append(_f32(eps_mix, dev)); x.eps_mix = keep[-1].data_ptr()
      if discriminator.subtract_log_policy:
        e2 = keep[-1].unsqueeze(1)
        mix = lambda k: e2 * _f32(expert_transitions[k], dev) + (1 - e2) * _f32(transitions[k], dev)
        keep.append(actor.log_prob(mix('states'), mix('actions')))
        x.logit_offset_mix = keep[-1].data_ptr()
    elif discriminator.subtract_log_policy:
      keep += [actor.log_prob(transitions['states'], transitions['actions']), actor.log_prob(expert_transitions['states'], expert_transitions['actions'])]
      x.logit_offset_policy, x.logit_offset_expert = keep[-2].data_ptr(), keep[-1].data_ptr()
    step = _lib.lib().il_gail_shaped_deep_step if _shaped_general(discriminator) else _lib.lib().il_gail_shaped_step
    _lib.check(step(C.byref(d), C.byref(pb), C.byref(eb), _lib.ptr(e), C.byref(x), int(flags), _lib.stream_ptr()))
    return
  deep = type(discriminator).__name__ == 'DeepGAILDiscriminator'   # depth 2 and / or tanh: the general kernels, same arguments
  d = (deep_descriptor if deep else disc_descriptor)(discriminator, B, discriminator_optimiser, imitation_cfg)
  if imitation_cfg.loss_function == 'Mixup':
    alpha = float(_cfg_value(imitation_cfg, 'mixup_alpha', 1.0))
    if eps_mix is None and (alpha != 1.0 or discriminator.subtract_log_policy):   # Beta(1, 1) = U(0, 1) normally comes from the on-chip Philox stream; other alphas,
      eps_mix = torch.distributions.Beta(torch.full((B,), alpha), torch.full((B,), alpha)).sample()   # and draws this function needs itself, are made here like the reference's
    if eps_mix is not None:
      keep.append(_f32(eps_mix, dev)); x.eps_mix = keep[-1].data_ptr()
    if discriminator.subtract_log_policy:   # training.py:108 + models.py:144: log pi of the MIXED inputs (no graph), a per-row shift of the mixed logits
      e2 = keep[-1].unsqueeze(1)
      mix = lambda k: e2 * _f32(expert_transitions[k], dev) + (1 - e2) * _f32(transitions[k], dev)
      keep.append(actor.log_prob(mix('states'), mix('actions')))
      x.logit_offset_mix = keep[-1].data_ptr()
  elif discriminator.subtract_log_policy:   # models.py:144: log pi(a|s) of both batches, no graph
    keep += [actor.log_prob(transitions['states'], transitions['actions']), actor.log_prob(expert_transitions['states'], expert_transitions['actions'])]
    x.logit_offset_policy, x.logit_offset_expert = keep[-2].data_ptr(), keep[-1].data_ptr()
  step = _lib.lib().il_gail_deep_step if deep else _lib.lib().il_gail_disc_step
  _lib.check(step(C.byref(d), C.byref(pb), C.byref(eb), _lib.ptr(e), C.byref(x), int(flags), _lib.stream_ptr()))


def gail_predict_reward(disc: GAILDiscriminator, state: Tensor, action: Tensor, want_logits: bool = False, log_policy: Optional[Tensor] = None):
  dev = disc.flat.device
  n = state.size(0)
  d = disc_descriptor(disc, n, None)
  dummy = torch.zeros(n, device=dev)
  b = batch_desc(dict(states=state, actions=action, rewards=dummy, next_states=state, terminals=dummy, weights=dummy, absorbing=dummy))
  out, logits = torch.empty(n, device=dev), (torch.empty(n, device=dev) if want_logits else None)
  off = _f32(log_policy, dev)
  _lib.check(_lib.lib().il_gail_reward(C.byref(d), C.byref(b), _lib.ptr(out), _lib.ptr(logits), _lib.ptr(off), _lib.stream_ptr()))
  return (out, logits) if want_logits else out


# ----------------------------------------------------------------------------------------------- GMMIL
_GMMIL_WS: Dict[tuple, Tensor] = {}   # insertion-ordered: the oldest shape is evicted beyond _GMMIL_WS_MAX entries


_GMMIL_WS_MAX = 16


def _gmmil_workspace(n1: int, n2: int, D: int, device, tag=None, stream: Optional[int] = None) -> Tensor:
  """il_gmmil_reward / il_gmmil_sqdist scratch (include/il_hip.h): partial row sums + self-resetting arrival counters, zero-filled at creation. One per (shape, learner
  `tag`, stream): two learners or two streams with the same shape must not share the counters; a bounded cache (callers with ever-changing batch sizes evict the oldest).
  A launch that was aborted mid-way leaves its counters non-zero: `_gmmil_workspace_reset()` after handling such an error."""
  if stream is None:
    try:
      stream = int(torch.cuda.current_stream().cuda_stream)
    except Exception:   # (no HIP device: the host emulation of tests/host_emu drives the library with CPU tensors)
      stream = 0
  key = (n1, n2, D, device, tag, stream)
  ws = _GMMIL_WS.pop(key, None)
  if ws is None:
    ws = torch.zeros(int(_lib.lib().il_gmmil_workspace_floats(n1, n2, D)), dtype=torch.float32, device=device)
    while len(_GMMIL_WS) >= _GMMIL_WS_MAX:
      _GMMIL_WS.pop(next(iter(_GMMIL_WS)))
  _GMMIL_WS[key] = ws   # (re-inserted: most recently used last)
  return ws


def _gmmil_workspace_reset():
  _GMMIL_WS.clear()


def _weighted_median(x: Tensor, weights: Tensor) -> Tensor:
  """Reference models.py:40-44 (first call only; a device sort of B^2 values)."""
  x_sorted, indices = torch.sort(x.flatten())
  w = (weights.flatten() / weights.sum())[indices]
  return x_sorted[torch.min((torch.cumsum(w, dim=0) >= 0.5).nonzero())]


def _sa_batch(state, action, weight):
  """il_batch of (state, action, weight) rows for the kernels that read nothing else (GMMIL / embedding distances): the per-row scalar fields alias
  `weight`. Same checks as memory.batch_desc, once per distinct tensor - this sits on the per-call path of `predict_reward`."""
  n = state.size(0)
  for k, v in (('states', state), ('actions', action), ('weights', weight)):
    if v.dtype != torch.float32 or not _lib.on_device(v):
      raise TypeError(f'transitions[{k!r}] must be a float32 CUDA tensor (got {v.dtype} on {v.device})')
    if v.dim() == 2 and v.stride(1) != 1:
      raise ValueError(f'transitions[{k!r}] must have unit inner stride')
    assert v.size(0) == n, 'ragged transitions dict'
  b = _lib.Batch()
  ps, pa, pw = state.data_ptr(), action.data_ptr(), weight.data_ptr()
  ls, la, lw = (state.stride(0), action.stride(0), weight.stride(0)) if n > 1 else (state.size(1), action.size(1), 1)
  b.states = b.next_states = ps; b.ld_states = b.ld_next_states = ls
  b.actions = pa; b.ld_actions = la
  b.rewards = b.terminals = b.weights = b.absorbing = pw
  b.ld_rewards = b.ld_terminals = b.ld_weights = b.ld_absorbing = lw
  b.n = n
  return b


def embedding_sqdist(x: Tensor, y: Tensor) -> Tensor:
  """models.py:25-29 `_squared_distance` between two sets of feature rows [n1, D], [n2, D] -> [n1, n2] (k_gmmil_tile, direct form)."""
  n1, n2, D = x.size(0), y.size(0), x.size(1)
  dev = x.device
  st = _lib.stream_ptr()   # (the workspace is keyed by the stream the launch goes to, not by torch's current stream: they differ when the caller passes a stream of its own)
  ws = _gmmil_workspace(n1, n2, D, dev, stream=getattr(st, 'value', None) or 0)
  out = torch.empty(n1, n2, device=dev)
  w1, w2 = torch.ones(n1, device=dev), torch.ones(n2, device=dev)   # named: an il_batch holds raw pointers, the tensors must outlive the launch
  ba, bb = _sa_batch(x, x, w1), _sa_batch(y, y, w2)
  _lib.check(_lib.lib().il_gmmil_sqdist(C.byref(ba), C.byref(bb), D, 0, 1, _lib.ptr(out), _lib.ptr(ws), ws.numel(), st))
  return out


def gmmil_sqdist(disc: GMMILDiscriminator, a_state, a_action, b_state, b_action) -> Tensor:
  dev = a_state.device
  na, nb = a_state.size(0), b_state.size(0)
  D = disc.state_size + (0 if disc.state_only else disc.action_size)
  st = _lib.stream_ptr()
  ws = _gmmil_workspace(na, nb, D, dev, tag=getattr(disc, '_ws_tag', None), stream=getattr(st, 'value', None) or 0)
  out = torch.empty(na, nb, device=dev)
  wa, wb = torch.ones(na, device=dev), torch.ones(nb, device=dev)
  ba, bb = _sa_batch(a_state, a_action, wa), _sa_batch(b_state, b_action, wb)
  _lib.check(_lib.lib().il_gmmil_sqdist(C.byref(ba), C.byref(bb), disc.state_size, disc.action_size, int(disc.state_only), _lib.ptr(out), _lib.ptr(ws), ws.numel(), st))
  return out


def gmmil_predict_reward(disc: GMMILDiscriminator, state, action, expert_state, expert_action, weight, expert_weight, return_parts: bool = False):
  """Reference models.py:189-201."""
  dev = state.device
  if disc.gamma_1 is None:  # median heuristic, frozen after the first batch (models.py:193-195)
    disc.gamma_1 = 1 / (_weighted_median(gmmil_sqdist(disc, state, action, expert_state, expert_action), torch.outer(weight, expert_weight)).item() + 1e-8)
    disc.gamma_2 = 1 / (_weighted_median(gmmil_sqdist(disc, expert_state, expert_action, expert_state, expert_action), torch.outer(expert_weight, expert_weight)).item() + 1e-8)
  n1, n2 = state.size(0), expert_state.size(0)
  D = disc.state_size + (0 if disc.state_only else disc.action_size)
  stream = _lib.stream_ptr()
  ws = _gmmil_workspace(n1, n2, D, dev, tag=getattr(disc, '_ws_tag', None), stream=getattr(stream, 'value', None) or 0)   # (arrival counters: zero at creation, left at zero by every call)
  out = torch.empty(n1, device=dev)
  sim, self_sim = (torch.empty(n1, device=dev), torch.empty(n1, device=dev)) if return_parts else (None, None)
  pb, eb = _sa_batch(state, action, weight), _sa_batch(expert_state, expert_action, expert_weight)
  _lib.check(_lib.lib().il_gmmil_reward(C.byref(pb), C.byref(eb), disc.state_size, disc.action_size, int(disc.state_only), float(disc.gamma_1), float(disc.gamma_2),
                                        _lib.ptr(out), _lib.ptr(sim), _lib.ptr(self_sim), _lib.ptr(ws), ws.numel(), stream))
  return (out, sim, self_sim) if return_parts else out


# ----------------------------------------------------------------------------------------------- captured update
class UpdatePlan:
  """The per-step update block of the reference loop (train.py:173-203) with persistent buffers, for every algorithm= of the reference:
  device-side index draws (agent batch, then expert batch: the order and the stream of train.py:173) -> row gathers -> the algorithm's reward step ->
  [behavioural-cloning auxiliary step] -> SAC update.  The reward step is
    SAC / PWIL  none (PWIL stores its rewards online, train.py:156),
    GAIL        discriminator step + reward relabel (two streams, device-side hand-off; see `_run_update`),
    GMMIL       k_gmmil_tile on the two batches (bandwidths frozen by the first, eager, update: models.py:193-195),
    RED         predictor / target forward (eval mode),   DRIL  5-member Monte-Carlo-dropout uncertainty (Philox counter on the device),
    AdRIL/SQIL  k_mix_relabel with the per-update scalars (round, trajectory count, balanced alternation) read from a device buffer (`relabel_args`),
  preceded by `mix_expert_agent_transitions` when imitation.mix_expert_data = mixed_batch (train.py:183; GAIL with mixing keeps the per-function path).
  `run()` enqueues it eagerly; `capture()` records it into a hipGraph (torch.cuda.CUDAGraph) that `replay()` launches with one host call."""
  ALGORITHMS = ('SAC', 'GAIL', 'GMMIL', 'PWIL', 'RED', 'DRIL', 'AdRIL')

  def __init__(self, algorithm: str, actor, critic, log_alpha, target_critic, memory: ReplayMemory, actor_optimiser, critic_optimiser, temperature_optimiser,
               batch_size: int, discount: float, entropy_target: float, polyak_factor: float, expert_memory: Optional[ReplayMemory] = None, discriminator=None,
               discriminator_optimiser=None, imitation_cfg=None, device_index_draw: bool = True, overlap: bool = True, learner_id=None, mix_expert: bool = False,
               bc_aux: bool = False):
    """`learner_id`: give this plan private scratch / noise / index-stream state so that several plans can run concurrently (`PopulationPlan`).
    `mix_expert`: imitation.mix_expert_data == 'mixed_batch' (DRIL / GMMIL / RED); `bc_aux`: imitation.bc_aux_loss (train.py:201)."""
    assert algorithm in self.ALGORITHMS, f'UpdatePlan: unknown algorithm {algorithm}'
    # reinforcement.actor / critic outside the fused kernels' shape (models._mlp_shape: any depth 1-8, relu / tanh / sigmoid, hidden <= 2048, wide action spaces): the same plan
    # on ONE stream - device draws + gather, the algorithm's reward step, il_sac_update_general's layer-at-a-time launches (csrc/general.hip) - capturable as one hipGraph.
    # No device hand-off, population or data-parallel form for these shapes.
    self.general = _general_shape(actor, critic)
    if self.general:
      overlap = False
    assert expert_memory is not None or (algorithm in ('SAC', 'PWIL') and not mix_expert and not bc_aux), f'UpdatePlan({algorithm}): needs the expert memory'
    assert not (mix_expert and algorithm in ('GAIL', 'SAC', 'PWIL', 'AdRIL')), 'mixed batches: DRIL / GMMIL / RED plans (train.py:175,183); GAIL with mixing runs the per-function path'
    # GAIL: discriminator branch || SAC branch. SAC / PWIL (no reward step, nothing host-side inside): the second stream only hosts the resident index draw.
    self._two_stream = bool(overlap and (algorithm == 'GAIL' or (algorithm in ('SAC', 'PWIL') and not mix_expert and not bc_aux)))
    self.overlap, self.side = overlap, (torch.cuda.Stream() if self._two_stream else None)
    self.algorithm, self.B, dev = algorithm, batch_size, actor.flat.device
    self.memory, self.expert_memory, self.device_index_draw = memory, expert_memory, device_index_draw
    self.peer_desc = None   # parallel.DataParallelUpdate (fused form): il_peer_bucket descriptors of the exchanges that ride in this plan's optimiser launches
    self.has_expert, self.mix_expert, self.bc_aux = expert_memory is not None, bool(mix_expert), bool(bc_aux)
    self.discriminator = discriminator
    self.rows = torch.empty(batch_size, memory.row, device=dev); self.idx = torch.empty(batch_size, dtype=torch.int32, device=dev)
    self.transitions = batch_views(self.rows, memory.state_size, memory.action_size, memory.absorbing)
    self.logp, self.q = torch.empty(batch_size, device=dev), torch.empty(batch_size, device=dev)
    tag, off = learner_id, (0 if learner_id is None else 7919 * (int(learner_id) + 1))
    self.sac = sac_descriptor(actor, critic, log_alpha, target_critic, batch_size, actor_optimiser, critic_optimiser, temperature_optimiser, discount, entropy_target, polyak_factor,
                              tag=tag, seed_offset=off)
    self.sac.out_logp, self.sac.out_q = self.logp.data_ptr(), self.q.data_ptr()
    self._keep = (actor, critic, log_alpha, target_critic, actor_optimiser, critic_optimiser, temperature_optimiser, discriminator, discriminator_optimiser)
    if self.has_expert:   # the expert batch is drawn for every algorithm (train.py:173), which keeps the index stream in step with the reference's
      self.erows = torch.empty(batch_size, expert_memory.row, device=dev); self.eidx = torch.empty(batch_size, dtype=torch.int32, device=dev)
      self.expert_transitions = batch_views(self.erows, expert_memory.state_size, expert_memory.action_size, expert_memory.absorbing)
      self.eb = batch_desc(self.expert_transitions)
    if algorithm in ('GMMIL', 'RED', 'DRIL'):
      self.rewards = torch.empty(batch_size, device=dev)
      self.transitions['rewards'] = self.rewards   # train.py:190-198: rewards replaced by the reward model's prediction
      if algorithm == 'RED':
        assert discriminator.sigma_1, 'UpdatePlan(RED): set_sigma first (train.py:128)'
        self.red = discriminator._desc(batch_size)
      elif algorithm == 'DRIL':
        assert discriminator.q is not None, 'UpdatePlan(DRIL): set_uncertainty_threshold first (train.py:126)'
        self.dril = discriminator._desc(batch_size)
        self.dril.noise_counter = self.sac.noise_counter   # the per-update part of the dropout masks' Philox counter: advanced on the device by the actor step
    if algorithm == 'AdRIL':
      self.dyn = torch.zeros(3, dtype=torch.int64, device=dev)   # {n_expert, round, policy trajectories} of the next update (il_batch_mix_relabel_dyn)
      self._dyn_set = False
    if algorithm == 'GAIL':
      host_mixup = imitation_cfg is not None and imitation_cfg.loss_function == 'Mixup' and float(_cfg_value(imitation_cfg, 'mixup_alpha', 1.0)) != 1.0
      deep = type(discriminator).__name__ == 'DeepGAILDiscriminator'   # depth 2 / tanh: the general kernels, per-function path
      pu_margin = imitation_cfg is not None and imitation_cfg.loss_function == 'PUGAIL' and float(_cfg_value(imitation_cfg, 'nonnegative_margin', float('inf'))) != float('inf')
      # The discriminator variants outside the fused depth-1 kernels' two-stream schedule - a finite PUGAIL margin (a value pass ahead of the gradients), subtract_log_policy
      # (two actor passes), reward shaping, depth-2 / tanh discriminators - run their per-function entry points (adversarial_imitation_update + predict_reward, the calls of
      # train.py:178-194) INSIDE the plan, on the gathered rows: every input is device-resident, so the same launches are captured with the rest of the update. They keep
      # plain stream dependencies (no device-side hand-off, no data-parallel or population form).
      sub = bool(imitation_cfg is not None and discriminator.subtract_log_policy)
      self._variant = bool(imitation_cfg is not None and (deep or pu_margin or sub or getattr(discriminator, 'reward_shaping', False)))
      if self._variant:
        if learner_id is not None:
          raise NotImplementedError('UpdatePlan: a population of GAIL learners with a finite PUGAIL margin / subtract_log_policy / reward shaping / a depth-2 or tanh discriminator')
        self._gail_parts, self.disc = (actor, discriminator, discriminator_optimiser, imitation_cfg), None
        host_mixup = host_mixup or (imitation_cfg.loss_function == 'Mixup' and sub)   # (that combination needs the coefficients on the host side of the call as well)
      else:
        self.disc = disc_descriptor(discriminator, batch_size, discriminator_optimiser, imitation_cfg, tag=tag, seed_offset=off)
      # Mixup (training.py:105-107): Beta(1, 1) = U(0, 1) coefficients come from the on-chip Philox stream inside the discriminator kernel, like the gradient penalty's. For
      # alpha != 1 (the reference draws them with torch's CPU Beta sampler: a host input per update) a launch captured ahead of the discriminator step draws them on the
      # device (il_noise_fill_beta: same distribution, Philox bits) into `eps_mix`, which that step takes as il_gail_extra.eps_mix. The launch reads the update counter on the
      # device, so it has to be stream-ordered behind the previous update's actor step: such a plan keeps plain stream dependencies (no device-side hand-off).
      self._beta_alpha = float(_cfg_value(imitation_cfg, 'mixup_alpha', 1.0)) if host_mixup else None
      if self._beta_alpha is not None:
        self.eps_mix = torch.empty(batch_size, device=dev)
        self._mix_extra = _lib.GailExtra(); self._mix_extra.eps_mix = self.eps_mix.data_ptr()
      self.rewards = torch.empty(batch_size, device=dev)
      self.transitions['rewards'] = self.rewards  # train.py:194: rewards replaced by the discriminator's prediction
    self.pb = batch_desc(self.transitions)
    # Device-side hand-off between the two branches (include/il_hip.h `il_sync`): the discriminator branch and the SAC forward then share no stream
    # dependency between the gather and the critic loss. Validated by `capture()`; IL_DEVICE_SYNC=0 keeps plain stream dependencies.
    self._sync_slots, self._sync_timeouts, self._sync_gather_wgs, _, self._sync_spin, self._sync_host_flag = _lib.sync_layout()
    self._sync_poison = _lib.sync_layout_ex()[6]
    self.sync = torch.zeros(self._sync_slots, dtype=torch.int64, device=dev)   # IL_SYNC_SLOTS: every counter on its own 128-byte line
    self.device_sync = False
    self._chain_fits = None
    self.pre_hooks, self.post_hooks = [], []   # callables enqueuing extra work on the update's stream before / after it (captured with it), e.g. ActingWorker
    self.stream_ordered_draw = False   # True: the index draw stays the first kernel of the SAC branch (bit-identical; what per-kernel timing wants: see bench.py roofline())
    if self._two_stream and device_index_draw and os.environ.get('IL_DEVICE_SYNC', '1') != '0' and getattr(self, '_beta_alpha', None) is None and not getattr(self, '_variant', False):
      # HIP multiplexes streams onto a few hardware queues (round-robin at creation): a side stream that landed on the caller's queue runs serialised with it and
      # fails the probe. Another stream usually lands elsewhere: try a few (the rejected ones stay alive meanwhile, so that the next one gets a different queue).
      ok, rejected = self._probe_device_sync(graph=False), []
      while not ok and len(rejected) < 8:
        rejected.append(self.side)
        self.side = torch.cuda.Stream()
        ok = self._probe_device_sync(graph=False)
      self._set_device_sync(ok)
    self.graph = self.graph_side = None
    self.main_feeds_ring = False   # set True when work enqueued on the caller's stream BETWEEN updates moves the agent ring's cursor (train.py: ActingWorker / memory.append)
    self._captured_resident = False
    self._ring_desc = None
    self._capturing = None   # 'main' / 'side' while one branch of the device-synchronised update is being captured
    self._prepared = False   # True once an update of THIS plan has left the lane-ordered weight copies in step with the parameters
    # (round 6) the SAC branch's four launches alternating over two streams (il_sac_update_gather_overlap): `ov_stream` hosts the two optimiser launches; `_ov_active`:
    # the last SAC-branch enqueue of this plan was an overlapped one (the stage epochs are in step with [IL_SYNC_MAIN_EPOCH]; the caller's stream is NOT ordered behind
    # `ov_stream` until `join()`); `_ov_probe`: None = not probed yet
    self.ov_stream, self._ov_active, self._ov_probe, self._recording, self._direct_overlap = None, False, None, False, False
    self._watch_np = None

  def record_relu_masks(self) -> Tensor:
    """Tests only (il_sac.debug_masks, include/il_hip.h): from now on every update ALSO writes, for its three back-propagated passes, which hidden pre-activations
    it found > 0 - [10, B, H] floats: actor(s) layers 1, 2; critic_k(s, a) at 2 + 2k, 3 + 2k; the updated critic_k(s, a~) at 6 + 2k, 7 + 2k. Call before capture()
    (a captured launch carries the descriptor by value). The oracle replays an update with these decisions (oracle.nets.mlp_forward(masks=...))."""
    assert self.graph is None, 'record_relu_masks(): before capture()'
    assert not self.general, 'record_relu_masks(): the fused kernels only'
    self.relu_masks = torch.zeros(10, self.B, self.sac.hidden, device=self.rows.device)
    self.sac.debug_masks = self.relu_masks.data_ptr()
    return self.relu_masks

  def invalidate(self):
    """Call after changing actor / critic / target parameters from outside the plan (load_state_dict, manual edits): the next update
    re-derives the lane-ordered weight copies (k_repack) instead of trusting the ones its own Adam / polyak epilogues maintain."""
    self._prepared = False
    if self.graph is not None:
      raise RuntimeError('UpdatePlan.invalidate(): re-capture the plan after changing parameters externally')

  def _set_device_sync(self, on: bool):
    self.device_sync = bool(on)
    if self._two_stream:   # [IL_SYNC_GATHER_WGS]: who signals IL_SYNC_ROWS, and how many times per update
      L = _lib.lib()
      self.sync[self._sync_gather_wgs] = int(L.il_sac_chain_gather_workgroups(self.B, self.memory.row, self.sac.hidden) if self.ring_mode
                         else L.il_replay_gather_workgroups(self.B, self.memory.row, self.expert_memory.row if self.has_expert else 0))
    ptr = self.sync.data_ptr() if on else None
    self.sac.sync = ptr
    if self.algorithm == 'GAIL' and self.disc is not None:
      self.disc.sync = ptr

  def _probe_device_sync(self, graph: bool) -> bool:
    """True if a kernel on the side stream and a kernel on the main stream really run concurrently (eagerly, or as two branches of a
    captured graph): a waiter enqueued first on the side stream must see the setter enqueued after it on the main stream."""
    L, main = _lib.lib(), torch.cuda.current_stream()

    def enqueue():
      self.side.wait_stream(main)
      with torch.cuda.stream(self.side):
        _lib.check(L.il_sync_probe(_lib.ptr(self.sync), 0, _lib.stream_ptr()))
      _lib.check(L.il_sync_probe(_lib.ptr(self.sync), 1, _lib.stream_ptr()))
      main.wait_stream(self.side)
    before = self.sync_timeouts()
    if not graph:
      _lib.check(L.il_sync_probe(_lib.ptr(self.sync), 1, _lib.stream_ptr()))   # code object loaded, counters one ahead: the first timed waiter cannot be late because of a cold launch
      with torch.cuda.stream(self.side):
        self.side.wait_stream(main)
        _lib.check(L.il_sync_probe(_lib.ptr(self.sync), 0, _lib.stream_ptr()))
      torch.cuda.synchronize()
      for _ in range(3): enqueue()
    else:   # the shape capture() uses: one graph per stream, the waiter's launched first
      gw, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
      with torch.cuda.graph(gw, stream=self.side):
        _lib.check(L.il_sync_probe(_lib.ptr(self.sync), 0, _lib.stream_ptr()))
      with torch.cuda.graph(gs):
        _lib.check(L.il_sync_probe(_lib.ptr(self.sync), 1, _lib.stream_ptr()))
      # first launches of freshly instantiated graphs can take > 10 ms (the probe's bound) to reach the device: launch each once in the order that cannot wait
      # (setter, then waiter), and only then run the real test with the waiter first
      gs.replay()
      torch.cuda.synchronize()
      with torch.cuda.stream(self.side):
        gw.replay()
      torch.cuda.synchronize()
      for _ in range(3):
        with torch.cuda.stream(self.side):
          gw.replay()
        gs.replay()
    torch.cuda.synchronize()
    ok = self.sync_timeouts() == before
    self.sync[self._sync_timeouts] = 0
    self.sync[self._sync_poison] = 0   # (an expired PROBE wait is not an expired wait of an update: the optimiser launches must not skip their stores for it)
    w = getattr(self, '_watch_host', None)
    if w is not None: w[0] = 0   # an expired PROBE wait also reached the pinned host word (watch_timeouts): it is not an expired wait of an update
    return ok

  def _probe_streams(self, waiter, setter) -> bool:
    """True if a kernel on `waiter` that polls a counter is overtaken by a kernel enqueued AFTER it on `setter` - i.e. the two streams sit on different hardware queues."""
    L = _lib.lib()

    def on(stream, which):
      with torch.cuda.stream(stream):
        _lib.check(L.il_sync_probe(_lib.ptr(self.sync), which, _lib.stream_ptr()))
    torch.cuda.synchronize()
    before = self.sync_timeouts()
    on(setter, 1)   # code object loaded, counters one ahead: the first timed waiter cannot be late because of a cold launch
    torch.cuda.synchronize()
    on(waiter, 0)
    torch.cuda.synchronize()
    for _ in range(3):
      on(waiter, 0)
      on(setter, 1)
      torch.cuda.synchronize()
    ok = self.sync_timeouts() == before
    self.sync[self._sync_timeouts] = 0
    self.sync[self._sync_poison] = 0
    w = getattr(self, '_watch_host', None)
    if w is not None: w[0] = 0
    return ok

  @property
  def main_overlap(self) -> bool:
    """(round 6) Ring mode on one GPU, pair-mode shape: the SAC branch's four launches ALTERNATE over two streams - forward / critic loss and policy / critic on the caller's,
    the two optimiser launches on `ov_stream` - and hand over through stage epochs on the device (il_sac_update_gather_overlap, include/il_hip.h): every launch is dispatched
    while its predecessor still runs, does what does not depend on it (row gathers, the optimiser's p / m / v streams, the target step, the critics' forward) and waits behind
    that. Bit-identical to the in-order schedule (`test_schedule_switches_are_bit_identical[IL_MAIN_OVERLAP-*]`). Needs three streams on three hardware queues (probed once:
    `ov_stream` against the caller's stream both ways, and against the discriminator branch's).
    OFF by default (IL_MAIN_OVERLAP=1 switches it on): measured SLOWER than in-order launches on one stream, 15.5k against 17.8k updates/s (profiles/r06_overlap_ab.txt,
    r06_overlap_timeline.txt): a device-side hand-off (drain, L2 write-back, ticket; poll, flag, poll; L2 invalidate) takes 3 - 4.5 us where the queue's own
    launch boundary takes 1.5 - 2.5, and that eats the 2 - 3 us of prologue each launch moves ahead of its wait."""
    if os.environ.get('IL_MAIN_OVERLAP', '0') != '1' or not self.device_sync or not self._prepared or self.general:
      return False
    if self.peer_desc is not None or getattr(self, 'data_parallel', False) or self.pre_hooks or self.post_hooks or not self.ring_mode:
      return False
    if self.sac.hidden != 256 or self.B % 128 != 0 or (self.sac.state_dim + self.sac.action_dim + 15) // 16 * 16 > 64:
      return False
    if self._ov_probe is None:
      main, rejected = torch.cuda.current_stream(), []
      self._ov_probe = False
      for _ in range(8):   # (streams are dealt to the hardware queues round-robin at creation: a rejected one stays alive so that the next lands elsewhere)
        cand = torch.cuda.Stream()
        if self._probe_streams(cand, main) and self._probe_streams(main, cand) and (self.side is None or self._probe_streams(self.side, cand)):
          self.ov_stream, self._ov_probe = cand, True
          break
        rejected.append(cand)
    return self._ov_probe

  def _ov_enter(self):
    """Before the first overlapped update that follows anything else on this learner: the stage epochs := [IL_SYNC_MAIN_EPOCH] (one tiny launch on the caller's stream), and
    `ov_stream` ordered behind it."""
    _lib.check(_lib.lib().il_sac_overlap_enter(C.byref(self.sac), _lib.stream_ptr()))
    self.ov_stream.wait_stream(torch.cuda.current_stream())
    self._ov_active = True

  def _ov_leave(self):
    """Before anything else is enqueued on the caller's stream for this learner: order it behind the optimiser launches on `ov_stream`."""
    if self._ov_active:
      torch.cuda.current_stream().wait_stream(self.ov_stream)
      self._ov_active = False

  def _sac_gather(self, ring, rewards, relabel, rewards_out, flags):
    """il_sac_update_gather, or its overlapped form when `main_overlap` applies and no hipGraph is being captured."""
    L = _lib.lib()
    if self.main_overlap and (self._capturing is None or self._recording):
      if not self._ov_active:
        assert not self._recording, 'record_direct() enters the overlapped schedule before it starts recording'
        self._ov_enter()
      _lib.check(L.il_sac_update_gather_overlap(C.byref(self.sac), C.byref(self.pb), C.byref(ring), rewards, relabel, rewards_out, None, None, _lib.ptr(self.logp), _lib.ptr(self.q),
                                                flags, _lib.stream_ptr(), C.c_void_p(self.ov_stream.cuda_stream)))
      return
    self._ov_leave()
    _lib.check(L.il_sac_update_gather(C.byref(self.sac), C.byref(self.pb), C.byref(ring), rewards, relabel, rewards_out, None, None, _lib.ptr(self.logp), _lib.ptr(self.q), flags, _lib.stream_ptr()))

  def poisoned(self) -> bool:
    """[IL_SYNC_POISON]: a bounded device-side wait of this learner has expired; from that launch on the optimiser launches skip their stores (the weights are those of the
    last complete update). Synchronous read."""
    return bool(int(self.sync[self._sync_poison].item()))

  def clear_poison(self):
    _lib.check(_lib.lib().il_sync_clear_poison(_lib.ptr(self.sync), _lib.stream_ptr()))
    w = getattr(self, '_watch_host', None)
    if w is not None: w[0] = 0

  def widen_handoff_bound(self, polls: Optional[int] = None):
    """[IL_SYNC_SPIN]: the bound of this learner's device-side waits, in polls. A data-parallel rank waits for the all-reduced discriminator step inside its SAC branch and
    for the previous update's end (which contains two gradient exchanges) in its resident index draw: those waits must outlast the exchange's own bound (IL_PEER_SPIN_LIMIT),
    or ordinary inter-rank skew expires the inner wait first. Default: twice the exchange's."""
    self.sync[self._sync_spin] = int(polls) if polls else 2 * _lib.IL_PEER_SPIN_LIMIT

  def watch_timeouts(self, peer_status: Optional[Tensor] = None):
    """Gives the device a pinned host word per time-out counter ([IL_SYNC_HOST_FLAG]; word [1] of the peer exchange's status): a device-side wait that gives up stores its
    count there as well, so the training loop reads plain host memory every step (`timeouts_seen()`) - no synchronisation, nothing added to the captured update - and an
    expired wait is noticed within the pipeline depth (one or two updates) instead of at the next logging interval. Works for captured graphs (the address is read on the
    device when a wait expires, not baked into a launch)."""
    if getattr(self, '_watch_host', None) is None:
      self._watch_host = torch.zeros(2, dtype=torch.int64).pin_memory()
      self._watch_np = self._watch_host.numpy()   # (the same pinned words as a numpy view: a ~100 ns host read per launch_direct / replay)
    self.sync[self._sync_host_flag] = self._watch_host.data_ptr()
    if peer_status is not None:
      peer_status[1] = self._watch_host[1:].data_ptr()
    return self

  def timeouts_seen(self):
    """(hand-off waits, exchange waits) that have expired so far, as far as the device has reported them. Host read of pinned memory: no synchronisation."""
    w = getattr(self, '_watch_host', None)
    return (0, 0) if w is None else (int(w[0]), int(w[1]))

  def sync_timeouts(self) -> int:
    """Bounded waits that gave up (device counter). Non-zero means the two branches did not run concurrently (e.g. a counter-collecting
    profiler serialises kernels): results of those updates are invalid; `capture()` checks this once and falls back to stream dependencies."""
    handoff = C.c_uint32(0)
    if getattr(self, '_prepared', False) and not self.general:   # (general shapes: another workspace layout, no in-launch waits)   # the in-launch waits of the chained kernels (counted since the first update's k_repack); with il_sync counters they are in sync[IL_SYNC_TIMEOUTS] too
      _lib.check(_lib.lib().il_sac_handoff_timeouts(C.byref(self.sac), C.byref(handoff)))
    return max(int(self.sync[self._sync_timeouts].item()), int(handoff.value))

  def prepared_flag(self) -> int:
    return _lib.IL_FLAG_SAC_PREPARED if (self._prepared and os.environ.get('IL_ALWAYS_REPACK') != '1') else 0

  def _sample(self, mem: ReplayMemory, idx: Tensor, rows: Tensor):
    if self.device_index_draw:
      mem.sample_device(self.B, idx, rows)
    else:
      idx.copy_(mem._sample_idx_tensor(self.B))
      _lib.check(_lib.lib().il_replay_gather(_lib.ptr(mem.ring), mem.size, mem.row, _lib.ptr(idx), self.B, _lib.ptr(rows), _lib.stream_ptr()))

  def draw_all(self):
    """The index draws of `sample_all` alone (device draw): consumers that read the rings through il_batch.gather (`_ring_batches`) can start from here,
    before `gather_all` has produced the packed rows."""
    assert self.device_index_draw
    m, e = self.memory, (self.expert_memory if self.has_expert else None)
    st = m.stream().device_state(m.device)
    _lib.check(_lib.lib().il_replay_sample_device(
        _lib.ptr(st), self.B, _lib.ptr(m._ring_state), _lib.ptr(m.ring), m.size, m.row, _lib.ptr(self.idx), None,
        _lib.ptr(e._ring_state) if e else None, _lib.ptr(e.ring) if e else None, e.size if e else 0, e.row if e else 0, _lib.ptr(self.eidx) if e else None, None, None, _lib.stream_ptr()))

  def gather_all(self, expert: bool = True):
    """The row gathers of `sample_all` for indices drawn by `draw_all` (same values as `sample_all`); expert=False when the expert rows are only read
    through the ring (nothing else consumes the packed expert batch)."""
    L, st = _lib.lib(), _lib.stream_ptr()
    m = self.memory
    _lib.check(L.il_replay_gather(_lib.ptr(m.ring), m.size, m.row, _lib.ptr(self.idx), self.B, _lib.ptr(self.rows), st))
    if self.has_expert and expert:
      e = self.expert_memory
      _lib.check(L.il_replay_gather(_lib.ptr(e.ring), e.size, e.row, _lib.ptr(self.eidx), self.B, _lib.ptr(self.erows), st))

  def sample_all(self):
    """Agent batch then expert batch (the order train.py:173 consumes the index stream); one launch when drawn on the device."""
    if not self.device_index_draw:
      self._sample(self.memory, self.idx, self.rows)
      if self.has_expert:
        self._sample(self.expert_memory, self.eidx, self.erows)
      return
    m, e = self.memory, (self.expert_memory if self.has_expert else None)
    st = m.stream().device_state(m.device)   # the agent memory's index stream feeds both draws of an update (one stream, agent then expert)
    _lib.check(_lib.lib().il_replay_sample_device(
        _lib.ptr(st), self.B, _lib.ptr(m._ring_state), _lib.ptr(m.ring), m.size, m.row, _lib.ptr(self.idx), None if self.ring_mode else _lib.ptr(self.rows),
        _lib.ptr(e._ring_state) if e else None, _lib.ptr(e.ring) if e else None, e.size if e else 0, e.row if e else 0, _lib.ptr(self.eidx) if e else None,
        _lib.ptr(self.erows) if e and not self.ring_mode else None, _lib.ptr(self.sync) if self.device_sync and self.algorithm == 'GAIL' else None, _lib.stream_ptr()))   # ring mode: the draw only (SAC / PWIL reach this call on their one-stream schedule only: no counters move)

  def run(self):
    self.launcher_wait()   # (round 6: updates handed to the launcher thread are issued first)
    for hook in self.pre_hooks:
      hook()
    self._run_update()
    for hook in self.post_hooks:
      hook()

  @property
  def _sampler_ok(self) -> bool:
    return not self.pre_hooks and not self.stream_ordered_draw and os.environ.get('IL_RESIDENT_SAMPLER', '1') != '0'

  @property
  def ring_mode(self) -> bool:
    """Device-side hand-off only: an update is DRAWN but never gathered by a kernel of its own. The discriminator step and the forward / critic-loss
    launch read their rows straight from the rings through the indices (il_batch.gather), so both branches start right after the index draw;
    extra workgroups of k_sac_chain write the gathered agent rows for the later kernels (il_sac_update_gather). IL_RING_GATHER=0: gather first.
    SAC / PWIL plans: only together with the resident sampler (their second stream has nothing else to do)."""
    if not self.device_sync or os.environ.get('IL_RING_GATHER', '1') == '0':
      return False
    if self.algorithm != 'GAIL' and not self._sampler_ok:
      return False
    # il_sac_update_gather's launch (6 workgroups per 16-row tile + the gather workgroups) must be co-resident: batch sizes beyond that gather first
    # (il_sac_update then also keeps its forward / critic-loss kernels separate)
    if self._chain_fits is None:
      cus = torch.cuda.get_device_properties(self.rows.device).multi_processor_count
      self._chain_fits = 6 * (self.B // 16) + int(_lib.lib().il_sac_chain_gather_workgroups(self.B, self.memory.row, self.sac.hidden)) <= cus
    return self._chain_fits

  def _ring_batches(self):
    if self._ring_desc is None:
      def ring_desc(mem, idx):
        t = batch_views(mem.ring, mem.state_size, mem.action_size, True)
        if not mem.absorbing: t['absorbing'] = self._zero_column(mem)
        b = batch_desc(t); b.n = self.B
        b.gather, b.gather_capacity = idx.data_ptr(), mem.size
        return b
      self._ring_desc = (ring_desc(self.memory, self.idx), ring_desc(self.expert_memory, self.eidx) if self.has_expert else None)
    return self._ring_desc

  @property
  def staged_rows(self) -> bool:
    """GAIL with the resident sampler on one GPU: the sampler workgroup also copies the drawn agent rows into a dense slab (`stage`) before it signals
    (il_gail_disc_step_draw_staged) - with the early draw that is ~30 us before the update starts - and the forward / critic-loss launch reads them from there
    (IL_FLAG_SAC_STAGED_ROWS): one global trip in its prologues instead of index -> row. The discriminator step keeps reading through the indices. Measured NEUTRAL on an
    interleaved A/B (17.55-17.63k against 17.47-17.64k updates/s, k_sac_chain_pair 22.1-22.3 us either way, profiles/r05_stage_rows_ab.txt): with the indices drawn ~30 us
    ahead the index -> row trip of the prologues is not what the forward / critic-loss launch waits for. Kept bit-identical and switchable, OFF by default (IL_STAGE_ROWS=1)."""
    return bool(self.algorithm == 'GAIL' and self.resident_sampler and self.peer_desc is None and not getattr(self, 'data_parallel', False) and os.environ.get('IL_STAGE_ROWS', '0') == '1')

  def _staged_batch(self):
    if getattr(self, '_stage_desc', None) is None:
      m = self.memory
      self.stage = torch.empty(self.B, m.row, device=self.rows.device)
      t = batch_views(self.stage, m.state_size, m.action_size, True)
      if not m.absorbing:
        self._keep_zero_stage = torch.zeros(1, dtype=torch.float32, device=self.rows.device)
        t['absorbing'] = self._keep_zero_stage.expand(self.B)
      self._stage_desc = batch_desc(t)
      self._stage_desc.n = self.B
    return self._stage_desc

  def _zero_column(self, mem):
    """A ring-strided view of zeros for the `absorbing` field of a ring without absorbing states (one float per ring row would be wasteful: every
    row reads the same zero through stride 0)."""
    z = torch.zeros(1, dtype=torch.float32, device=mem.ring.device)
    self._keep_zero = z
    return z.expand(mem.size)

  @property
  def resident_sampler(self) -> bool:
    """Ring mode only: the index draw is RESIDENT - one extra workgroup of the discriminator branch's first launch (il_gail_disc_step_draw) - instead of the first
    kernel of the SAC branch. It starts while the previous update is still running, waits on the device for that update's end ([IL_SYNC_MAIN_EPOCH]), draws and
    signals [IL_SYNC_INDICES]; the discriminator workgroups beside it and the forward / critic-loss launch on the other stream wait for that signal. The sampling
    launch (~7 us + a kernel boundary) leaves the update's critical path, and the discriminator kernel keeps preparing (weights, power iterations) ahead of the rows.
    Not with `pre_hooks` (an append captured at the head of the main branch must precede the draw in stream order). IL_RESIDENT_SAMPLER=0: draw on the main stream."""
    return self.ring_mode and self._sampler_ok

  @property
  def inline_relabel(self) -> bool:
    """Ring mode with a discriminator on (s, a) small enough for the critic-loss workgroups' spare LDS: the reward relabel runs INSIDE k_sac_chain as soon as
    the discriminator's AdamW step has signalled (il_sac_update_gather `relabel`), so the side branch ends with that step. IL_INLINE_RELABEL=0: separate kernel."""
    if not self.ring_mode or os.environ.get('IL_INLINE_RELABEL', '1') == '0' or self.disc.state_only:
      return False
    D, Hd = self.disc.state_dim + self.disc.action_dim, self.disc.hidden
    Dp = (D + 3) // 4 * 4
    return Hd * (Dp + 4) + 4 * Hd + Dp + 8 <= max(4, self.sac.hidden // 16) * 256 + 256 - 32

  def _disc_step(self, flags: int):
    """The discriminator step of the ring schedule; with the resident sampler the index draw rides in the same launch (il_gail_disc_step_draw)."""
    L, st = _lib.lib(), _lib.stream_ptr()
    rp, re_ = self._ring_batches()
    self._fused_exchange_needs_the_resident_sampler()
    if self.resident_sampler:
      m, e = self.memory, self.expert_memory
      mt = m.stream().device_state(m.device)
      if self.peer_desc is not None:   # data-parallel: the gradient exchange rides in the reduce + AdamW launch (parallel.DataParallelUpdate, fused form)
        _lib.check(L.il_gail_disc_step_draw_peer(C.byref(self.disc), C.byref(rp), C.byref(re_), _lib.ptr(mt), _lib.ptr(m._ring_state), _lib.ptr(self.idx), _lib.ptr(e._ring_state),
                                                 _lib.ptr(self.eidx), flags, C.byref(self.peer_desc['disc']), st))
        return
      if self.staged_rows:
        self._staged_batch()
        _lib.check(L.il_gail_disc_step_draw_staged(C.byref(self.disc), C.byref(rp), C.byref(re_), _lib.ptr(mt), _lib.ptr(m._ring_state), _lib.ptr(self.idx), _lib.ptr(e._ring_state),
                                                   _lib.ptr(self.eidx), _lib.ptr(self.stage), flags, st))
        return
      _lib.check(L.il_gail_disc_step_draw(C.byref(self.disc), C.byref(rp), C.byref(re_), _lib.ptr(mt), _lib.ptr(m._ring_state), _lib.ptr(self.idx), _lib.ptr(e._ring_state), _lib.ptr(self.eidx),
                                          flags, st))
    else:
      _lib.check(L.il_gail_disc_step(C.byref(self.disc), C.byref(rp), C.byref(re_), None, None, flags, st))

  def _enqueue_discriminator_branch(self):
    L, st = _lib.lib(), _lib.stream_ptr()
    if self.ring_mode:
      rp, re_ = self._ring_batches()
      if self.inline_relabel:
        self._disc_step(_lib.IL_FLAG_GAIL_CLOSE_EPOCH)
        return
      self._disc_step(0)
      _lib.check(L.il_gail_reward(C.byref(self.disc), C.byref(rp), _lib.ptr(self.rewards), None, None, st))
      return
    self._disc_step_and_relabel_on_gathered_rows(st)

  def _disc_step_and_relabel_on_gathered_rows(self, st, exchange=None):
    """train.py:178-194 on the gathered batches (the stream-dependency schedules): discriminator step, then the relabel kernel. `exchange` (parallel.DataParallelUpdate,
    round 6: every discriminator variant has a data-parallel form): the step leaves its gradient in the optimiser's arena (IL_FLAG_GRADS_ONLY; the optimiser is ticked and the
    spectral-norm buffers, which depend on the replicated weights only, advance), `exchange()` averages that arena over the ranks on this stream, and the AdamW step is applied
    from it - il_gail_apply_grads for the fused depth-1 kernels, il_adam_step for the variants' (the same adam_update on the same constants as their reduce launch's epilogue)."""
    L, extra = _lib.lib(), None
    G = _lib.IL_FLAG_GRADS_ONLY if exchange is not None else 0
    if getattr(self, '_beta_alpha', None) is not None:   # this update's Beta(alpha, alpha) coefficients, drawn on the device (see __init__); the learner's one Philox key / counter
      _lib.check(L.il_noise_fill_beta(C.c_uint64(self.sac.noise_seed), self.sac.noise_counter, self._beta_alpha, self.B, _lib.ptr(self.eps_mix), st))
      extra = C.byref(self._mix_extra)
    if getattr(self, '_variant', False):
      from .models import make_gail_input
      actor, disc, opt, icfg = self._gail_parts
      t, e = self.transitions, self.expert_transitions
      adversarial_imitation_update(actor, disc, t, e, opt, icfg, eps_mix=self.eps_mix if self._beta_alpha is not None else None, flags=G)
      if exchange is not None:
        exchange()
        od = opt.desc()
        _lib.check(L.il_adam_step(_lib.ptr(disc.flat), _lib.ptr(opt.grad), C.byref(od), min(disc.flat.numel(), opt.grad.numel()), 0, st))
      self.rewards.copy_(disc.predict_reward(**make_gail_input(t['states'], t['actions'], t['next_states'], t['terminals'], actor, bool(getattr(disc, 'reward_shaping', False)),
                                                               bool(disc.subtract_log_policy))))
      return
    _lib.check(L.il_gail_disc_step(C.byref(self.disc), C.byref(self.pb), C.byref(self.eb), None, extra, G, st))
    if exchange is not None:
      exchange()
      _lib.check(L.il_gail_apply_grads(C.byref(self.disc), st))
    _lib.check(L.il_gail_reward(C.byref(self.disc), C.byref(self.pb), _lib.ptr(self.rewards), None, None, st))

  def _fused_exchange_needs_the_resident_sampler(self):
    """parallel.DataParallelUpdate decides ONCE (collectively) that the gradient exchanges ride in the optimiser launches (`peer_desc`); `resident_sampler` is re-evaluated
    on every call and depends on `pre_hooks` / `stream_ordered_draw` / IL_RESIDENT_SAMPLER. A hook attached AFTER that decision (ActingWorker.attach for
    +acting.schedule=overlap) would send the discriminator step down the non-resident branch, which has no exchange at all: the replicas would diverge silently until the
    next replica check. Fail at the first such update instead."""
    if self.peer_desc is not None and not self.resident_sampler:
      raise RuntimeError('UpdatePlan: the data-parallel gradient exchange rides in the optimiser launches of the resident-sampler schedule, but that schedule is no longer '
                         'active (a pre-hook was attached, stream_ordered_draw was set or IL_RESIDENT_SAMPLER changed after DataParallelUpdate was built). Build the '
                         'DataParallelUpdate after attaching hooks, or call DataParallelUpdate.use_collectives() on every rank.')

  def _enqueue_sac_branch(self):
    self._fused_exchange_needs_the_resident_sampler()
    resident = self.resident_sampler
    if not resident:
      self.sample_all()
    if self.ring_mode:
      inline = self.inline_relabel
      flags = self.prepared_flag() | (_lib.IL_FLAG_SAC_WAIT_INDICES if resident else 0)
      if self.peer_desc is not None:   # data-parallel: the critic's and the actor's exchange ride in the two optimiser launches
        self._ov_leave()
        _lib.check(_lib.lib().il_sac_update_gather_peer(C.byref(self.sac), C.byref(self.pb), C.byref(self._ring_batches()[0]), None if inline else _lib.ptr(self.rewards),
                                                        C.byref(self.disc) if inline else None, _lib.ptr(self.rewards) if inline else None, None, None,
                                                        _lib.ptr(self.logp), _lib.ptr(self.q), flags, C.byref(self.peer_desc['critic']), C.byref(self.peer_desc['actor']), _lib.stream_ptr()))
        return
      staged = resident and self.staged_rows
      self._sac_gather(self._staged_batch() if staged else self._ring_batches()[0], None if inline else _lib.ptr(self.rewards), C.byref(self.disc) if inline else None,
                       _lib.ptr(self.rewards) if inline else None, flags | (_lib.IL_FLAG_SAC_STAGED_ROWS if staged else 0))
      return
    self._ov_leave()
    _lib.check(_lib.lib().il_sac_update(C.byref(self.sac), C.byref(self.pb), None, None, _lib.ptr(self.logp), _lib.ptr(self.q), self.prepared_flag(), _lib.stream_ptr()))

  def _run_update(self):
    L = _lib.lib()
    if self.algorithm == 'GAIL' and self.overlap:
      # Two streams: the discriminator step + reward relabel run next to the reward-independent SAC forward kernels and the critic loss needs both.
      main = torch.cuda.current_stream()
      fwd = _lib.IL_FLAG_SAC_FORWARD_ONLY | self.prepared_flag()   # (building the lane-ordered copies on the side stream first was measured: 7.2k vs 8.0k updates/s, the extra edge costs more than the 3 us kernel)
      if self.device_sync:
        # No stream dependency between the branches at all: they hand over on the device (k_gather2 -> k_gail_grad, k_gail_reward ->
        # k_critic_bwd). Captured, they are TWO graphs replayed on two streams (a fork inside one hipGraph delays one branch by 16-20 us).
        if self._capturing != 'main':
          if self._capturing is None and self.resident_sampler:
            self.side.wait_stream(main)   # eager: whatever the caller enqueued before this update (appends moving the ring cursor) precedes the resident draw
          with torch.cuda.stream(self.side):
            self._enqueue_discriminator_branch()
        if self._capturing != 'side':
          self._enqueue_sac_branch()
        if self._capturing is None:
          main.wait_stream(self.side)   # eager: leave the caller's stream ordered after both branches
          if self._ov_active: main.wait_stream(self.ov_stream)   # (... and after the optimiser launches of an overlapped update; the stage epochs stay in step)
        if self._capturing != 'side':
          self._prepared = True         # the SAC branch has (or, once replayed, will have) left the lane-ordered weight copies in step
        return
      # fallback (the runtime does not run the two streams concurrently, or IL_DEVICE_SYNC=0): one graph, fork after the gather, join before the critic loss
      self._ov_leave()
      self.sample_all()
      self.side.wait_stream(main)                                   # the discriminator needs the sampled batches
      with torch.cuda.stream(self.side):
        self._disc_step_and_relabel_on_gathered_rows(_lib.stream_ptr())
      st = _lib.stream_ptr()
      _lib.check(L.il_sac_update(C.byref(self.sac), C.byref(self.pb), None, None, _lib.ptr(self.logp), _lib.ptr(self.q), fwd, st))
      main.wait_stream(self.side)                                   # join: the critic loss reads the rewards
      _lib.check(L.il_sac_update(C.byref(self.sac), C.byref(self.pb), None, None, _lib.ptr(self.logp), _lib.ptr(self.q), _lib.IL_FLAG_SAC_SKIP_FORWARD, st))
      self._prepared = True
      return
    if self.algorithm != 'GAIL' and self.resident_sampler:
      # SAC / PWIL: the index draw is a resident launch on the second stream (il_replay_draw_resident: it waits on the device for the previous update's end), the update
      # itself one chain of four launches that reads its rows from the ring through the indices. Captured as two graphs with no edge between them, like GAIL's.
      main = torch.cuda.current_stream()
      if self._capturing != 'main':
        if self._capturing is None:
          self.side.wait_stream(main)   # eager: appends the caller enqueued before this update precede the draw
        with torch.cuda.stream(self.side):
          m, e = self.memory, (self.expert_memory if self.has_expert else None)
          mt = m.stream().device_state(m.device)
          _lib.check(L.il_replay_draw_resident(_lib.ptr(mt), self.B, _lib.ptr(m._ring_state), _lib.ptr(self.idx), _lib.ptr(e._ring_state) if e else None, _lib.ptr(self.eidx) if e else None,
                                               _lib.ptr(self.sync), _lib.stream_ptr()))
      if self._capturing != 'side':
        self._sac_gather(self._ring_batches()[0], None, None, None, self.prepared_flag() | _lib.IL_FLAG_SAC_WAIT_INDICES)
        self._prepared = True
      if self._capturing is None:
        main.wait_stream(self.side)
        if self._ov_active: main.wait_stream(self.ov_stream)
      return
    self._ov_leave()
    self.sample_all()
    st = _lib.stream_ptr()
    sync_kept = self.sac.sync
    if self.algorithm != 'GAIL':
      self.sac.sync = None   # one stream from here on: no kernel of this schedule hands anything over on the device (the critic loss would wait for [IL_SYNC_REWARDS])
    if self.algorithm == 'GAIL':
      self._disc_step_and_relabel_on_gathered_rows(st)
    else:
      self._enqueue_reward_model(st)
    flag = self.prepared_flag()
    if self.bc_aux:   # train.py:201: a behavioural-cloning step on the expert batch with the ACTOR's optimiser; it moves the actor, so the lane-ordered copies are re-derived
      a, ao = self._keep[0], self._keep[4]
      od = ao.desc()
      if self.general:
        from .models import ACTIVATION_IDS
        ws = a._general_workspace(self.B)
        _lib.check(L.il_bc_step_general(_lib.ptr(a.flat), _lib.ptr(ao.grad), C.byref(od), a.state_size, a.action_size, a.hidden, a.depth, ACTIVATION_IDS[a.activation], C.byref(self.eb), _lib.ptr(ws),
                                        ws.numel(), None, 0, st))
      else:
        _lib.check(L.il_bc_step(_lib.ptr(a.flat), _lib.ptr(ao.grad), C.byref(od), a.state_size, a.action_size, a.hidden, C.byref(self.eb), C.c_void_p(self.sac.workspace), self.sac.workspace_floats,
                                None, 0, st))
      flag = 0
    if self.general:
      from .models import ACTIVATION_IDS
      a, c = self._keep[0], self._keep[1]
      _lib.check(L.il_sac_update_general(C.byref(self.sac), C.byref(self.pb), a.depth, ACTIVATION_IDS[a.activation], c.hidden, c.depth, ACTIVATION_IDS[c.activation], None, None,
                                         _lib.ptr(self.logp), _lib.ptr(self.q), 0, st))
    else:
      _lib.check(L.il_sac_update(C.byref(self.sac), C.byref(self.pb), None, None, _lib.ptr(self.logp), _lib.ptr(self.q), flag, st))
    self.sac.sync = sync_kept
    self._prepared = True

  def relabel_args(self, step: int, num_trajectories: int):
    """AdRIL / SQIL: the per-update scalars of RewardRelabeller.resample_and_relabel (models.py:300-318) for the NEXT update: a stream-ordered copy into the device
    buffer the captured k_mix_relabel reads. Call once before every run() / replay()."""
    r = self.discriminator
    if r.balanced:
      n_expert = self.B if r.sample_expert else 0
      r.sample_expert = not r.sample_expert
    else:
      n_expert = self.B // 2
    rnd = -(-int(step) // r.update_freq) if r.update_freq > 0 else 0
    self.dyn.copy_(torch.tensor([n_expert, rnd, int(num_trajectories)], dtype=torch.int64))   # pageable source: staged synchronously, ordered on the current stream
    self._dyn_set = True

  def _enqueue_reward_model(self, st):
    """train.py:183-198 for the non-adversarial algorithms, on the sampled batches (`self.rows` / `self.erows`)."""
    L, alg, m = _lib.lib(), self.algorithm, self.memory
    if self.mix_expert:   # models.py:287-290: the first half of every field <- expert rows
      _lib.check(L.il_batch_mix_relabel(_lib.ptr(self.rows), _lib.ptr(self.erows), self.B, m.state_size, m.action_size, self.B // 2, 0, 0, 0, 0.0, 0, st))
    if alg == 'AdRIL':
      capturing = torch.cuda.is_current_stream_capturing() or self._recording   # a captured / recorded launch reads whatever relabel_args() stored before it is replayed / re-issued
      assert capturing or self._dyn_set, 'UpdatePlan(AdRIL): call relabel_args(step, memory.num_trajectories) before every update'
      r = self.discriminator
      import numpy as np
      reward_expert = float(np.float32(1 / self.expert_memory.num_trajectories)) if r.update_freq > 0 else 0.0
      _lib.check(L.il_batch_mix_relabel_dyn(_lib.ptr(self.rows), _lib.ptr(self.erows), self.B, m.state_size, m.action_size, 2 if r.update_freq > 0 else 1, r.update_freq, reward_expert,
                                            _lib.ptr(self.dyn), st))
      if not capturing: self._dyn_set = False
    elif alg == 'GMMIL':
      d, t, e = self.discriminator, self.transitions, self.expert_transitions
      if d.gamma_1 is None:   # the median heuristic of the FIRST batch (models.py:193-195): host-side, once - the first update must be an eager run()
        assert not torch.cuda.is_current_stream_capturing(), 'UpdatePlan(GMMIL): run() once before capture() (the first batch fixes the kernel bandwidths)'
        self.rewards.copy_(gmmil_predict_reward(d, t['states'], t['actions'], e['states'], e['actions'], t['weights'].contiguous(), e['weights'].contiguous()))
        return
      D = d.state_size + (0 if d.state_only else d.action_size)
      if getattr(self, '_gmmil_ws', None) is None:   # this learner's own counters and partial sums, kept with the plan (a captured graph holds the pointer)
        self._gmmil_ws = torch.zeros(int(L.il_gmmil_workspace_floats(self.B, self.B, D)), dtype=torch.float32, device=self.rows.device)
      ws = self._gmmil_ws
      _lib.check(L.il_gmmil_reward(C.byref(self.pb), C.byref(self.eb), d.state_size, d.action_size, int(d.state_only), float(d.gamma_1), float(d.gamma_2), _lib.ptr(self.rewards), None, None,
                                   _lib.ptr(ws), ws.numel(), st))
    elif alg == 'RED':
      assert not self.discriminator.training, 'UpdatePlan(RED): discriminator.eval() first (train.py:147)'
      _lib.check(L.il_red_forward(C.byref(self.red), C.byref(self.pb), 0, None, None, None, 0, _lib.ptr(self.rewards), None, None, st))
    elif alg == 'DRIL':
      _lib.check(L.il_dril_uncertainty(C.byref(self.dril), C.byref(self.pb), None, None, None, 0x40000000, None, _lib.ptr(self.rewards), st))   # offset range apart from the eager calls' small counters

  def capture(self, warmup: int = 3, updates: int = 1):
    """`updates` > 1: that many CONSECUTIVE updates per replay (offline training, several updates per environment step)."""
    assert self.device_index_draw, 'graph capture needs device-side index draws (no H2D inside the graph)'
    if self.device_sync and not self._probe_device_sync(graph=True):
      self._set_device_sync(False)   # the runtime serialises graph branches here (e.g. a counter-collecting profiler): keep stream dependencies
    self.memory.stream().device_state(self.rows.device)  # materialise the device copy of the MT19937 state before capture starts
    # Warm-up updates run on the CALLER's stream, i.e. on the (main, side) pair the probe above validated. A fresh warm-up stream may be mapped onto the
    # hardware queue the side stream uses (HIP multiplexes streams onto a few HSA queues, round-robin at creation): the two branches then serialise, the
    # device-side waits of those updates expire and they consume stale rewards (seen once in the full test suite: 48 expired waits per warm-up update).
    for _ in range(warmup):
      self.run()
    torch.cuda.synchronize()
    if warmup and self.sync_timeouts():
      raise RuntimeError(f'UpdatePlan.capture: {self.sync_timeouts()} device-side waits expired during the warm-up updates (the two branches did not run concurrently); '
                         'their results are invalid. Set IL_DEVICE_SYNC=0 to keep plain stream dependencies.')
    self._ov_leave()   # (the warm-up updates may have alternated over `ov_stream`: a captured update is the in-order schedule)
    if self.device_sync and (self.algorithm == 'GAIL' or self.resident_sampler):   # two graphs, one per branch, replayed on two streams; no edge between them (see _run_update)
      self._captured_resident = self.resident_sampler
      self.graph_side, self._capturing = torch.cuda.CUDAGraph(), 'side'
      with torch.cuda.graph(self.graph_side, stream=self.side):
        for _ in range(updates): self._run_update()
      self.graph, self._capturing = torch.cuda.CUDAGraph(), 'main'
      with torch.cuda.graph(self.graph):
        for _ in range(updates): self.run()
      self._capturing = None
      return self
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      for _ in range(updates): self.run()
    return self

  def direct_launch_ok(self) -> bool:
    """record_direct / launch_direct apply: the device hand-off schedule with two unjoined branches and nothing hooked into the update (an ActingWorker attached for
    +acting.schedule=overlap rides in the captured graph instead)."""
    recordable = all(getattr(h, '_il_recordable', False) for h in self.pre_hooks + self.post_hooks)   # (round 6) an ActingWorker's append / publish launches are library calls too
    return bool(recordable and (self._two_branch_schedule() or self._one_stream_recordable()))

  def _two_branch_schedule(self) -> bool:
    return bool(self.device_sync and (self.algorithm == 'GAIL' or self.resident_sampler))

  def _one_stream_recordable(self) -> bool:
    """(round 6) The one-stream plans - GMMIL / RED / DRIL / AdRIL, SAC / PWIL with mixed batches or the BC auxiliary step, general actor / critic shapes - are library calls
    on the caller's stream and nothing else (no torch operation, no stream edge: checked on the GPU by wrapping every tensor factory / fill / copy around run()), so they can be
    recorded and re-issued like the two-branch schedule. A hipGraph replay of such a plan costs the runtime's replay-to-replay gap (~6 us) AND one fill kernel of
    torch.cuda.CUDAGraph.replay(), which re-seeds the default generator's graph-safe Philox offset on every replay (5 us on the stream; profiles/r06_gmmil_plan_kernel_stats.md).
    Not recordable: GAIL on stream dependencies (host-side stream waits between its branches) and the discriminator variants that run torch operations inside the plan."""
    if self.device_sync or not self.device_index_draw or self.algorithm == 'GAIL': return False
    if self.algorithm == 'GMMIL' and self.discriminator.gamma_1 is None: return False   # (the first update fixes the bandwidths on the host)
    return True

  def record_direct(self):
    """The two-branch schedule `capture()` records, as DIRECT launches: walks the update's host code once with a recording stand-in for the library (nothing is launched),
    keeping each branch's entry point + ctypes arguments (stream included), so that `launch_direct()` re-issues exactly those calls - two calls into libil_hip.so per
    update, six kernel launches, no hipGraph and no stream edge. A hipGraph replay pays the runtime's graph-launch bookkeeping between two replays on a stream (~4.5 us per
    update here); a direct launch is one AQL packet per kernel (profiles/r05_launch_ab.txt). Run at least one update first (run(): code objects loaded, LDS
    attributes set, lane-ordered weight copies built). The main branch is recorded on the CALLER's current stream and must be launched from it; descriptors are passed by
    reference, so later changes of their fields (watch_timeouts, widen_handoff_bound) apply, unlike in a captured graph."""
    assert self.direct_launch_ok(), 'record_direct: the device hand-off schedule (two unjoined branches) or a one-stream plan of library calls; hooks that are library calls only'
    assert self._prepared, 'record_direct: run() at least one update first'
    self.launcher_wait()   # (passes of an earlier recording handed to the launcher thread go out before the descriptors are walked again)
    self.memory.stream().device_state(self.rows.device)

    class Recorder:
      def __init__(self, real): self.real, self.calls = real, []
      def __getattr__(self, name):
        fn = getattr(self.real, name)
        # a LAUNCH is an il_* entry point that returns a status and takes the stream as its last argument; everything else (size / layout / grid queries - which are
        # c_int32 = c_int on this platform and were swallowed by the first form of this recorder -, host-side reads) is answered by the library itself
        args_t = getattr(fn, 'argtypes', None) or ()
        if not (name.startswith('il_') and not name.startswith('il_launcher_') and fn.restype is C.c_int and len(args_t) > 0 and args_t[-1] is _lib._P):
          return fn
        def call(*args):
          self.calls.append((fn, args))
          return 0
        return call

    real = _lib.lib()
    out = []
    if self.main_overlap and not self._ov_active:
      self._ov_enter()   # (a real launch, ahead of the recording: the recorded update is the steady state)
    try:
      self._recording = True
      for branch in (('side', 'main') if self._two_branch_schedule() else ('one stream',)):
        if branch == 'one stream':   # everything on the caller's stream: one recorded list
          rec = Recorder(real)
          _lib._lib = rec
          self.run()
          out += [[], list(rec.calls)]
          break
        rec = Recorder(real)
        _lib._lib, self._capturing = rec, branch
        if branch == 'side':
          with torch.cuda.stream(self.side): self._run_update()
        else:
          self.run()   # (with the hooks: an attached ActingWorker's append before, its parameter snapshot after)
        out.append(list(rec.calls))
    finally:
      _lib._lib, self._capturing, self._recording = real, None, False
    self._direct_side, self._direct_main = out
    self._captured_resident = self.resident_sampler if self._two_branch_schedule() else False
    self._direct_overlap = bool(self._ov_active)
    return self

  def launch_direct(self, join: bool = True):
    """`join` (overlapped schedule only): order the caller's stream behind the update's last launch, which sits on `ov_stream` - whatever the caller enqueues next on its own
    stream (an acting forward, an evaluation, a checkpoint) then sees the stepped networks, as it does behind an in-order update. join=False: back-to-back updates with
    nothing in between (offline training, several updates per environment step, bench.py): the next update's first launch is dispatched while this one's last still runs;
    call `join()` before reading anything."""
    self._raise_if_poisoned()
    self.launcher_wait()   # (passes handed to the launcher thread go out first)
    if self.main_feeds_ring and self._captured_resident:
      self.side.wait_stream(torch.cuda.current_stream())   # (as replay(): appends enqueued since the last update precede the resident index draw)
    if self._direct_overlap and not self._ov_active:
      self._ov_enter()   # (something else ran on this learner since the last overlapped update: run(), replay())
    for fn, args in self._direct_side:
      if fn(*args) != 0: _lib.check(1)
    for fn, args in self._direct_main:
      if fn(*args) != 0: _lib.check(1)
    if join and self._direct_overlap:
      torch.cuda.current_stream().wait_stream(self.ov_stream)

  # --- (round 6) the recorded launches issued by a thread of the library (csrc/launcher.hip): the host returns at once and goes on with its environment step
  @staticmethod
  def _word(a) -> int:
    """One recorded ctypes argument as the 64-bit word the launcher passes on (pointers / integers only)."""
    if a is None: return 0
    if isinstance(a, int): return a & 0xFFFFFFFFFFFFFFFF
    if isinstance(a, (C.c_void_p, C.c_char_p)): return int(a.value or 0)
    if isinstance(a, (C.c_int, C.c_uint, C.c_int32, C.c_uint32, C.c_int64, C.c_uint64, C.c_long, C.c_ulong, C.c_longlong, C.c_ulonglong, C.c_size_t)): return int(a.value) & 0xFFFFFFFFFFFFFFFF
    if hasattr(a, '_obj'): return C.addressof(a._obj)          # C.byref(descriptor): the descriptor stays this plan's, later changes of its fields apply
    if isinstance(a, (C.Structure, C.Array)): raise TypeError('a structure passed by value cannot be re-issued by the launcher')
    if hasattr(a, 'contents'): return C.cast(a, C.c_void_p).value or 0   # a ctypes pointer
    raise TypeError(f'launch_async: argument of type {type(a).__name__} is not a pointer or an integer')

  def launch_async(self):
    """`launch_direct()` issued by the library's launcher thread (il_launcher_*): the two recorded branches - and an attached ActingWorker's append before / parameter
    snapshot after them - go out in the recorded order while the caller returns at once (~1 us on this thread instead of the ~20 us of launch work), e.g. to post the next
    observation and step the environment (profiles/r06_acting.json). Call `launcher_wait()` (or `join()`) before synchronising a stream, reading a result, or issuing
    anything else of this learner: a device synchronisation only waits for what has been ISSUED. Same launches in the same order as `launch_direct()`: bit-identical
    (test_launcher_thread_issues_the_recorded_update)."""
    self._raise_if_poisoned()
    L = _lib.lib()
    recorded = getattr(self, '_launcher_of', (None, None))
    if getattr(self, '_launcher', None) is None or recorded[0] is not self._direct_side or recorded[1] is not self._direct_main:
      assert getattr(self, '_direct_main', None) is not None, 'launch_async: record_direct() first'
      assert not self._direct_overlap, 'launch_async: the overlapped schedule joins streams on the host after every update; use launch_direct()'
      if getattr(self, '_launcher', None) is None:
        h = C.c_void_p()
        _lib.check(L.il_launcher_create(C.byref(h)))
        self._launcher = h
        import weakref
        weakref.finalize(self, lambda hv=h.value: L.il_launcher_destroy(C.c_void_p(hv)))
      _lib.check(L.il_launcher_clear(self._launcher))
      for fn, args in self._direct_side + self._direct_main:
        if any(t is C.c_float or t is C.c_double for t in (fn.argtypes or ())):
          raise NotImplementedError(f'launch_async: {fn.__name__} takes floating-point arguments; the launcher re-issues integer / pointer arguments only (use launch_direct)')
        words = (C.c_uint64 * 16)(*[self._word(a) for a in args])
        _lib.check(L.il_launcher_add(self._launcher, C.cast(fn, C.c_void_p), words, len(args)))
      self._launcher_of = (self._direct_side, self._direct_main)
    if self.main_feeds_ring and self._captured_resident:
      self.side.wait_stream(torch.cuda.current_stream())   # (as launch_direct(): appends enqueued since the last update precede the resident index draw)
    _lib.check(L.il_launcher_submit(self._launcher))

  def launcher_wait(self):
    """Every update submitted with `launch_async()` has been issued to its streams (raises if a recorded call failed)."""
    if getattr(self, '_launcher', None) is not None and not getattr(self, '_recording', False):   # (record_direct walks run() with a recording stand-in for the library: nothing of the launcher's own belongs in a recorded pass)
      _lib.check(_lib.lib().il_launcher_wait(self._launcher))

  def join(self):
    """Order the caller's stream after the plan's second stream: `replay()` leaves the two branches unjoined (no edge between the graphs), so anything the caller reads
    on its own stream that the other branch wrote - the discriminator's parameters / buffers for a checkpoint, the rewards on the fallback schedules - needs this first."""
    self.launcher_wait()
    if self.side is not None:
      torch.cuda.current_stream().wait_stream(self.side)
    if self._ov_active:
      torch.cuda.current_stream().wait_stream(self.ov_stream)

  def _raise_if_poisoned(self):
    """watch_timeouts(): a bounded device-side wait of this learner has given up (pinned host word, no synchronisation). From that launch on the optimiser launches have
    skipped their stores ([IL_SYNC_POISON]): the weights are those of the last complete update. Raised at the NEXT launch, not at the next logging interval."""
    w = self._watch_np
    if w is not None and w[0]:
      raise RuntimeError(f'UpdatePlan: {int(w[0])} device-side hand-off wait(s) expired (the branches of the update did not run concurrently: a profiler that serialises kernels, a CU '
                         'mask, a co-tenant process). The optimiser launches of that update and of every later one skipped their stores - the networks hold the last complete '
                         'update - and will keep doing so until clear_poison(). IL_DEVICE_SYNC=0 selects plain stream dependencies.')

  def replay(self):
    self._raise_if_poisoned()
    self.launcher_wait()
    self._ov_leave()
    if self.graph_side is not None:
      if self.main_feeds_ring and self._captured_resident:
        self.side.wait_stream(torch.cuda.current_stream())   # appends enqueued on the caller's stream since the last update must precede the resident index draw
      with torch.cuda.stream(self.side):
        self.graph_side.replay()
    self.graph.replay()


class PopulationPlan:
  """N independent learners (seeds / hyper-parameter trials: how the reference is actually used, README.md:96-99, train_all.py) advanced
  by ONE hipGraph replay: every learner's update block is a branch of the graph on its own stream, so their latency-bound kernels
  (16-64 workgroups each) fill the 256 CUs side by side.  Each learner owns its replay ring, index stream, scratch and noise state."""

  def __init__(self, plans):
    self.plans = list(plans)
    for p in self.plans:
      p.overlap = False  # one stream per learner: the learners themselves are the concurrency (nested fork/join breaks hipGraph capture on ROCm 7.2)
      p._set_device_sync(False)
    self.streams = [torch.cuda.Stream() for _ in self.plans]
    self.graph = None

  def run(self):
    main = torch.cuda.current_stream()
    for p, s in zip(self.plans, self.streams):
      s.wait_stream(main)
      with torch.cuda.stream(s):
        p.run()
    for s in self.streams:
      main.wait_stream(s)

  def capture(self, warmup: int = 0):
    for p in self.plans:
      p.memory.stream().device_state(p.rows.device)
    for _ in range(warmup):
      self.run()
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      self.run()
    return self

  def replay(self):
    self.graph.replay()


def _device_array(structs, device) -> Tensor:
  """ctypes descriptors -> one device byte tensor (the kernels index it with the learner id)."""
  raw = b''.join(bytes(x) for x in structs)
  return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


class BatchedPopulationPlan:
  """N independent learners with identical shapes advanced by the SAME kernel launches (`il_*_population`): the learner id is a grid
  dimension, so one update of the whole population costs 11 launches instead of 11 N.  This is the route from the latency-bound
  single-learner regime towards the HBM roofline.  Every learner keeps its own replay ring, MT19937 index stream,
  networks, optimiser state, scratch and Philox counter (build the `UpdatePlan`s with distinct `learner_id`s)."""

  def __init__(self, plans, groups: Optional[int] = None):
    self.plans = list(plans)
    # groups > 1: the population is cut into `groups` contiguous sub-populations, each advanced by its own launches on its own stream (parallel branches of the
    # captured graph). A population launch ends with a tail in which the last workgroups of every learner drain while most CUs idle, and the next kernel cannot
    # start before it: with two or more independent branches the other sub-population's workgroups fill those CUs. IL_POP_GROUPS overrides the default of 1.
    groups = int(os.environ.get('IL_POP_GROUPS', '1')) if groups is None else int(groups)
    self.subs = None
    if groups > 1 and len(self.plans) >= 2 * groups:
      per = (len(self.plans) + groups - 1) // groups
      self.subs = [BatchedPopulationPlan(self.plans[i:i + per], groups=1) for i in range(0, len(self.plans), per)]
      self.streams = [torch.cuda.Stream() for _ in self.subs]
      for sub in self.subs: sub.side = None   # nested fork / join breaks hipGraph capture on ROCm 7.2: inside a branch the discriminator kernels stay in stream order (the other branches run beside them)
      self.algorithm, self.B, self.L, self.graph = self.subs[0].algorithm, self.subs[0].B, len(self.plans), None
      return
    for p in self.plans:
      p._set_device_sync(False)   # one stream, one set of launches for all learners: plain stream order
    p0 = self.plans[0]
    assert all(p.algorithm == p0.algorithm and p.B == p0.B for p in self.plans) and p0.algorithm in ('SAC', 'GAIL'), 'the population launches exist for SAC and GAIL learners'
    if any(getattr(p, 'general', False) for p in self.plans):
      raise NotImplementedError('BatchedPopulationPlan: actor / critic shapes outside depth 2 / ReLU / hidden <= 256 / action_size <= 8 have no population launches (PopulationPlan runs them as independent graph branches)')
    self.algorithm, self.B, self.L, dev = p0.algorithm, p0.B, len(self.plans), p0.rows.device
    self.sac_descs = _device_array([p.sac for p in self.plans], dev)
    self.batches = _device_array([p.pb for p in self.plans], dev)
    args = []
    for p in self.plans:
      m, e = p.memory, (p.expert_memory if p.has_expert else None)
      st = m.stream().device_state(dev)
      args.append(_lib.SampleArgs(st.data_ptr(), m._ring_state.data_ptr(), m.ring.data_ptr(), m.size, m.row, p.idx.data_ptr(), p.rows.data_ptr(),
                                  e._ring_state.data_ptr() if e else None, e.ring.data_ptr() if e else None, e.size if e else 0, e.row if e else 0,
                                  p.eidx.data_ptr() if e else None, p.erows.data_ptr() if e else None))
    self.sample_args = _device_array(args, dev)
    self.max_row = max([p.memory.row for p in self.plans] + [p.expert_memory.row for p in self.plans if p.has_expert])
    if self.algorithm == 'GAIL':
      self.disc_descs = _device_array([p.disc for p in self.plans], dev)
      self.expert_batches = _device_array([p.eb for p in self.plans], dev)
      self.reward_ptrs = torch.tensor([p.rewards.data_ptr() for p in self.plans], dtype=torch.int64, device=dev)
    self.graph = None
    self._prepared = False
    self.side = torch.cuda.Stream() if self.algorithm == 'GAIL' and os.environ.get('IL_POP_OVERLAP', '1') != '0' else None

  def run(self):
    if self.subs is not None:
      main = torch.cuda.current_stream()
      for sub, s in zip(self.subs, self.streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
          sub.run()
      for s in self.streams:
        main.wait_stream(s)
      return
    L, st, p0 = _lib.lib(), _lib.stream_ptr(), self.plans[0]
    prepared = _lib.IL_FLAG_SAC_PREPARED if self._prepared else 0
    _lib.check(L.il_replay_sample_population(_lib.ptr(self.sample_args), self.L, self.B, self.max_row, st))
    if self.algorithm == 'GAIL' and self.side is not None:
      # The discriminator kernels (48 small workgroups per learner, latency-bound: ~60 us of a ~450 us replay at 32 learners) run on a second stream beside the
      # reward-independent forward kernels of every learner and join before the critic loss. One fork / join per replay (its ~10 us of queue signalling is paid once
      # for the whole population, unlike in the single-learner update where it is why the branches hand over on the device instead).
      main = torch.cuda.current_stream()
      self.side.wait_stream(main)
      with torch.cuda.stream(self.side):
        _lib.check(L.il_gail_step_population(_lib.ptr(self.disc_descs), _lib.ptr(self.batches), _lib.ptr(self.expert_batches), _lib.ptr(self.reward_ptrs), self.L, C.byref(p0.disc), _lib.stream_ptr()))
      _lib.check(L.il_sac_update_population(_lib.ptr(self.sac_descs), _lib.ptr(self.batches), self.L, C.byref(p0.sac), prepared | _lib.IL_FLAG_SAC_FORWARD_ONLY, st))
      main.wait_stream(self.side)
      _lib.check(L.il_sac_update_population(_lib.ptr(self.sac_descs), _lib.ptr(self.batches), self.L, C.byref(p0.sac), _lib.IL_FLAG_SAC_SKIP_FORWARD, st))
    else:
      if self.algorithm == 'GAIL':
        _lib.check(L.il_gail_step_population(_lib.ptr(self.disc_descs), _lib.ptr(self.batches), _lib.ptr(self.expert_batches), _lib.ptr(self.reward_ptrs), self.L, C.byref(p0.disc), st))
      _lib.check(L.il_sac_update_population(_lib.ptr(self.sac_descs), _lib.ptr(self.batches), self.L, C.byref(p0.sac), prepared, st))
    self._prepared = True

  def capture(self, warmup: int = 0):
    for _ in range(warmup):
      self.run()
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      self.run()
    return self

  def replay(self):
    self.graph.replay()
