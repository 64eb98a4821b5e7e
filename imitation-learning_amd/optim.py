"""Adam / AdamW state holders for the fused kernels (stand-ins for torch.optim.AdamW / Adam at reference train.py:66,84,95).

The arithmetic (torch `_single_tensor_adam` op order) runs inside the HIP kernels; these objects only own exp_avg / exp_avg_sq
arenas and the DEVICE step counter, and expose the small part of the torch.optim surface the reference loop touches.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


class AdamW:
  decoupled = True

  def __init__(self, target, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """`target`: a module with a `.flat` arena (SoftActor, TwinCritic, GAILDiscriminator) or a flat fp32 tensor (log_alpha)."""
    self.flat: Tensor = target.flat if hasattr(target, 'flat') else target
    assert self.flat.dtype == torch.float32 and self.flat.is_contiguous()
    self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
    self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
    self.grad = torch.zeros_like(self.flat)          # only used by the data-parallel path (all-reduce between backward and step)
    self.step_count = torch.zeros(16, dtype=torch.int32, device=self.flat.device)  # [0] step, [4..11] per-step fp32 constants (library-owned)

  def desc(self) -> _lib.Adam:
    return _lib.Adam(self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.step_count.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay)

  def zero_grad(self, set_to_none: bool = True):  # gradients never outlive a kernel on the fused path
    pass

  def step(self, grad: Tensor = None):
    """Stand-alone AdamW step from an explicit gradient arena (ticks the device step counter first)."""
    g = self.grad if grad is None else grad
    d = self.desc()
    _lib.check(_lib.lib().il_adam_step(_lib.ptr(self.flat), _lib.ptr(g), d, self.flat.numel(), _lib.IL_FLAG_TICK, _lib.stream_ptr()))

  def state_dict(self):
    return dict(step=int(self.step_count[0].item()), exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)

  def load_state_dict(self, sd):
    self.step_count[0] = int(sd['step']); self.exp_avg.copy_(sd['exp_avg']); self.exp_avg_sq.copy_(sd['exp_avg_sq'])


class Adam(AdamW):
  """torch.optim.Adam with weight_decay=0 (the temperature optimiser, reference train.py:66)."""
  decoupled = False

  def __init__(self, target, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
    super().__init__(target, lr=lr, betas=betas, eps=eps, weight_decay=0.0)
