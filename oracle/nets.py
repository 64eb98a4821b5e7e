"""MLP / tanh-Gaussian / AdamW building blocks (TEST ORACLE, numpy float32).

Restates the torch ops the reference composes:
  * `_create_fcnn` Linear-ReLU stacks, `y = x W^T + b` (models.py:48-69);
  * `SoftActor.forward` chunk + clamp(-20, 2) + Tanh(Normal) (models.py:90-94);
  * `TransformedDistribution.log_prob` with `TanhTransform.log_abs_det_jacobian`
    = 2(log 2 - x - softplus(-2x)) and `Normal.log_prob`
    = -(v-mu)^2/(2 var) - log(scale) - log(sqrt(2 pi))  (torch.distributions);
  * `optim.AdamW` / `optim.Adam` single-tensor step (train.py:66,84,95);
  * `update_target_network` polyak (models.py:79-81).

Parameters are flat float32 vectors in torch `parameters()` order
(W1[H,in], b1[H], W2[H,H], b2[H], ..., Wout[out,H], bout[out]).
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0  # models.py:87


def mlp_shapes(in_dim, hidden, depth, out_dim):
  dims = [in_dim] + [hidden] * depth + [out_dim]
  return [((dims[i + 1], dims[i]), (dims[i + 1],)) for i in range(len(dims) - 1)]


def mlp_numel(in_dim, hidden, depth, out_dim):
  return sum(w[0] * w[1] + b[0] for w, b in mlp_shapes(in_dim, hidden, depth, out_dim))


def unpack(flat, shapes):
  """Views (W, b) per layer into the flat vector."""
  out, o = [], 0
  for ws, bs in shapes:
    nW = ws[0] * ws[1]
    W = flat[o:o + nW].reshape(ws); o += nW
    b = flat[o:o + bs[0]]; o += bs[0]
    out.append((W, b))
  assert o == flat.size
  return out


def _act(z, activation):
  """models.py:16 ACTIVATION_FUNCTIONS: relu / tanh / sigmoid (torch op by op in float32)."""
  if activation == 'relu': return np.maximum(z, f32(0))
  if activation == 'tanh': return np.tanh(z).astype(f32)
  if activation == 'sigmoid': return (f32(1) / (f32(1) + np.exp(-z))).astype(f32)
  raise ValueError(activation)


def _act_grad(h, activation):
  """d act / d z from the POST-activation value h."""
  if activation == 'relu': return (h > 0)
  if activation == 'tanh': return (f32(1) - h * h).astype(f32)
  if activation == 'sigmoid': return (h * (f32(1) - h)).astype(f32)
  raise ValueError(activation)


def mlp_forward(layers, x, masks=None, activation='relu'):
  """Returns (out, acts) with acts[i] = input of layer i (post-ReLU hidden).

  masks (tests only): one boolean [B, H] array per hidden layer = "this pre-activation was > 0" as ANOTHER correct fp32 evaluation of the same network decided it (the
  HIP path's il_sac.debug_masks). A pre-activation within rounding of 0 takes either sign depending on the summation order; with `masks` the ReLU passes z where the mask
  says so (z itself is then ~1e-8: the forward value barely moves) and `mlp_backward(..., masks=...)` back-propagates through the same units - which isolates every other
  source of difference from that one."""
  acts, h = [], x.astype(f32)
  for i, (W, b) in enumerate(layers):
    acts.append(h)
    z = h @ W.T + b
    if i == len(layers) - 1: h = z
    elif masks is None: h = _act(z, activation)
    else: h = np.where(masks[i], z, f32(0)).astype(f32)   # (masks: ReLU networks only)
  return h, acts


def mlp_backward(layers, acts, dout, need_dx=True, masks=None, activation='relu'):
  """Gradient of sum(out * dout). Returns (flat grad in parameter order, dx). masks: see mlp_forward (masks[i - 1] gates the input of layer i)."""
  grads, dz = [None] * len(layers), dout.astype(f32)
  for i in range(len(layers) - 1, -1, -1):
    W, _ = layers[i]
    grads[i] = (dz.T @ acts[i], dz.sum(axis=0))
    if i > 0 or need_dx:
      dh = dz @ W
      if i > 0:
        dz = (dh * (_act_grad(acts[i], activation) if masks is None else masks[i - 1])).astype(f32)  # ReLU: the post-ReLU input of layer i is > 0 iff its pre-activation was
  flat = np.concatenate([np.concatenate([g[0].ravel(), g[1].ravel()]) for g in grads]).astype(f32)
  return flat, (dh if need_dx else None)


def softplus(z):
  z = z.astype(f32)
  return np.where(z > 20, z, np.log1p(np.exp(np.minimum(z, f32(20))))).astype(f32)


def actor_head(out, action_size):
  mean, ls_raw = out[:, :action_size], out[:, action_size:]
  ls = np.clip(ls_raw, f32(LOG_STD_MIN), f32(LOG_STD_MAX))
  return mean, ls_raw, ls, np.exp(ls)


LOG_SQRT_2PI = f32(math.log(math.sqrt(2 * math.pi)))
LOG2 = f32(math.log(2.0))


def tanh_gaussian_logp(x, mean, std):
  """log pi(tanh(x)) for pre-tanh x; per-row sum, same op order as torch."""
  var = std * std
  nlp = -((x - mean) ** 2) / (f32(2) * var) - np.log(std) - LOG_SQRT_2PI
  ladj = f32(2) * (LOG2 - x - softplus(f32(-2) * x))
  return (f32(0) - ladj.sum(axis=1)) + nlp.sum(axis=1)


def adam_step(p, g, m, v, t, lr, wd=0.0, b1=0.9, b2=0.999, eps=1e-8, decoupled=True):
  """In-place torch `_single_tensor_adam` step; `t` is the 1-based step count."""
  if wd != 0 and decoupled:
    p *= f32(1 - lr * wd)
  m += f32(1 - b1) * (g - m)  # exp_avg.lerp_(grad, 1 - beta1)
  v *= f32(b2)
  v += f32(1 - b2) * g * g
  bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
  step_size, bc2_sqrt = lr / bc1, bc2 ** 0.5
  denom = np.sqrt(v) / f32(bc2_sqrt) + f32(eps)
  p -= f32(step_size) * (m / denom)


def polyak(target, param, tau):
  target *= f32(tau)
  target += f32(1 - tau) * param
