"""SAC / BC update oracle (TEST ORACLE, numpy float32, manual backward).

Restates reference `training.py:14-54` (`sac_update`) and `training.py:57-64`
(`behavioural_cloning_update`) with the noise injected (`eps_next` for
`policy.sample()` on s' at :21, `eps_cur` for `rsample()` on s at :35).
"""
from __future__ import annotations

import numpy as np

from . import nets
from .nets import f32


class SacState:
  """Flat parameter / optimiser arenas for one learner (reference objects at train.py:64-67)."""

  def __init__(self, state_size, action_size, hidden=256, depth=2, activation='relu', critic_hidden=None, critic_depth=None, critic_activation=None):
    """The critics take their own (hidden, depth, activation) when given (reinforcement.critic of the reference's configuration); default: the actor's."""
    self.S, self.A, self.H, self.depth, self.activation = state_size, action_size, hidden, depth, activation
    self.Hc, self.depth_c, self.activation_c = critic_hidden or hidden, critic_depth or depth, critic_activation or activation
    self.actor_shapes = nets.mlp_shapes(state_size, hidden, depth, 2 * action_size)
    self.critic_shapes = nets.mlp_shapes(state_size + action_size, self.Hc, self.depth_c, 1)
    self.Pa = nets.mlp_numel(state_size, hidden, depth, 2 * action_size)
    self.Pc = nets.mlp_numel(state_size + action_size, self.Hc, self.depth_c, 1)
    z = lambda n: np.zeros(n, f32)
    self.actor, self.critic, self.target = z(self.Pa), z(2 * self.Pc), z(2 * self.Pc)
    self.log_alpha = z(1)
    self.actor_m, self.actor_v, self.critic_m, self.critic_v = z(self.Pa), z(self.Pa), z(2 * self.Pc), z(2 * self.Pc)
    self.alpha_m, self.alpha_v = z(1), z(1)
    self.t_actor = self.t_critic = self.t_alpha = 0

  def actor_layers(self):
    return nets.unpack(self.actor, self.actor_shapes)

  def critic_layers(self, flat, k):
    return nets.unpack(flat[k * self.Pc:(k + 1) * self.Pc], self.critic_shapes)


def critic_forward(st, flat, s, a, masks=None):
  x = np.concatenate([s, a], axis=1).astype(f32)
  outs = []
  for k in range(2):
    q, acts = nets.mlp_forward(st.critic_layers(flat, k), x, None if masks is None else masks[k], activation=st.activation_c)
    outs.append((q[:, 0], acts))
  return outs


def sac_update(st: SacState, batch, eps_next, eps_cur, *, discount, entropy_target, polyak_factor, lr=3e-4, weight_decay=0.0, lr_alpha=None, return_grads=False, masks=None):
  """masks (tests only): {'actor': [m1, m2], 'critic': [[m1, m2], [m1, m2]], 'pcritic': [[m1, m2], [m1, m2]]} - the ReLU decisions of the three back-propagated
  passes as another evaluation made them (nets.mlp_forward); the no-grad passes (actor and targets on s') have no backward to disagree in."""
  s, a, r, s2 = batch['states'], batch['actions'], batch['rewards'], batch['next_states']
  term, w, absb = batch['terminals'], batch['weights'], batch['absorbing']
  B, A = s.shape[0], st.A
  lr_alpha = lr if lr_alpha is None else lr_alpha
  alpha = np.exp(st.log_alpha[0]).astype(f32)  # training.py:16
  m = (f32(1) - absb).astype(f32)

  # --- target values (training.py:19-25, no grad)
  out, _ = nets.mlp_forward(st.actor_layers(), s2, activation=st.activation)
  mean2, _, _, std2 = nets.actor_head(out, A)
  x2 = eps_next * std2 + mean2                      # torch.normal: z*std + mean
  a2 = np.tanh(x2)
  logp2 = nets.tanh_gaussian_logp(x2, mean2, std2)
  a2 = m[:, None] * a2
  (t1, _), (t2, _) = critic_forward(st, st.target, s2, a2)
  tv = np.minimum(t1, t2) - m * alpha * logp2
  y = (r + (f32(1) - term) * f32(discount) * tv).astype(f32)

  # --- critic loss + step (training.py:26-31)
  mk = masks or {}
  (q1, acts1), (q2, acts2) = critic_forward(st, st.critic, s, a, mk.get('critic'))
  g_c = []
  for k, (q, acts) in enumerate(((q1, acts1), (q2, acts2))):
    dq = (w * (f32(2) * (q - y))) / f32(B)
    g, _ = nets.mlp_backward(st.critic_layers(st.critic, k), acts, dq[:, None], need_dx=False, masks=mk['critic'][k] if masks else None, activation=st.activation_c)
    g_c.append(g)
  g_c = np.concatenate(g_c)
  st.t_critic += 1
  nets.adam_step(st.critic, g_c, st.critic_m, st.critic_v, st.t_critic, lr, weight_decay)

  # --- policy loss + step (training.py:34-42), critic already updated
  a_layers = st.actor_layers()
  out, acts_a = nets.mlp_forward(a_layers, s, mk.get('actor'), activation=st.activation)
  mean, ls_raw, _, std = nets.actor_head(out, A)
  x = mean + eps_cur * std                          # rsample: loc + eps*scale
  an = np.tanh(x)
  logp = nets.tanh_gaussian_logp(x, mean, std)
  (qn1, actsn1), (qn2, actsn2) = critic_forward(st, st.critic, s, an, mk.get('pcritic'))
  sel1 = np.where(qn1 < qn2, f32(1), np.where(qn1 == qn2, f32(0.5), f32(0)))
  da = np.zeros((B, A), f32)
  for k, (acts, sel) in enumerate(((actsn1, sel1), (actsn2, f32(1) - sel1))):
    _, dx = nets.mlp_backward(st.critic_layers(st.critic, k), acts, (-(sel) / f32(B))[:, None], need_dx=True, masks=mk['pcritic'][k] if masks else None, activation=st.activation_c)
    da += dx[:, st.S:]
  c = (w * m * alpha) / f32(B)                       # dL/dlogp
  dx_pre = c[:, None] * (f32(2) * np.tanh(x)) + da * (f32(1) - an * an)
  dmean = dx_pre
  dstd = dx_pre * eps_cur - c[:, None] / std
  dls = dstd * std * ((ls_raw >= nets.LOG_STD_MIN) & (ls_raw <= nets.LOG_STD_MAX))
  g_a, _ = nets.mlp_backward(a_layers, acts_a, np.concatenate([dmean, dls], axis=1), need_dx=False, masks=mk.get('actor'), activation=st.activation)
  st.t_actor += 1
  nets.adam_step(st.actor, g_a, st.actor_m, st.actor_v, st.t_actor, lr, weight_decay)

  # --- temperature (training.py:45-49): Adam (not AdamW) on log_alpha
  g_alpha = np.array([-(alpha) * np.mean(w * m * (logp + f32(entropy_target)), dtype=f32)], f32)
  st.t_alpha += 1
  nets.adam_step(st.log_alpha, g_alpha, st.alpha_m, st.alpha_v, st.t_alpha, lr_alpha, 0.0)

  # --- polyak (training.py:52)
  nets.polyak(st.target, st.critic, polyak_factor)

  out_q = np.minimum(q1, q2)
  if return_grads:
    return logp, out_q, dict(critic=g_c, actor=g_a, alpha=g_alpha, y=y, logp_next=logp2, a_next=a2, a_new=an, qn1=qn1, qn2=qn2)
  return logp, out_q


def bc_update(actor_flat, m_, v_, t, shapes, action_size, batch, *, lr, weight_decay=0.0, return_grads=False, activation='relu'):
  """behavioural_cloning_update (training.py:57-64) + SoftActor.log_prob (models.py:97-99)."""
  s, a, w = batch['states'], batch['actions'], batch['weights']
  B, A = s.shape[0], action_size
  a = np.clip(a, f32(-1 + 1e-6), f32(1 - 1e-6))
  layers = nets.unpack(actor_flat, shapes)
  out, acts = nets.mlp_forward(layers, s, activation=activation)
  mean, ls_raw, _, std = nets.actor_head(out, A)
  x = np.arctanh(a).astype(f32)
  logp = nets.tanh_gaussian_logp(x, mean, std)
  up = (-w / f32(B))[:, None]                        # d(loss)/d(logp)
  d = x - mean
  dmean = up * d / (std * std)
  dstd = up * (d * d / (std * std * std) - f32(1) / std)
  dls = dstd * std * ((ls_raw >= nets.LOG_STD_MIN) & (ls_raw <= nets.LOG_STD_MAX))
  g, _ = nets.mlp_backward(layers, acts, np.concatenate([dmean, dls], axis=1), need_dx=False, activation=activation)
  nets.adam_step(actor_flat, g, m_, v_, t, lr, weight_decay)
  loss = np.mean(w * -logp, dtype=f32)
  return (loss, g, logp) if return_grads else loss
