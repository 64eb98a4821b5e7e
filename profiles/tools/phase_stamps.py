"""Developer tool: per-phase cycle counts (s_memtime) inside k_gail_grad and k_policy_critic.
Build the stamped library first (never shipped):
  cd imitation-learning_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DIL_PHASE_STAMPS -shared *.hip -o build/libil_hip_stamps.so
then run this script on the GPU box."""
import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
from imitation_learning_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get('IL_STAMP_LIB', 'imitation-learning_amd/csrc/build/libil_hip_stamps.so'))  # built with -DIL_PHASE_STAMPS, see the header of this file
import torch
import imitation_learning_amd as il
from imitation_learning_amd import training as il_training
from test_gpu_parity import _make_plan
plan, nets = _make_plan('GAIL', 13)
plan.overlap = False
for _ in range(20): plan.run()
torch.cuda.synchronize()
L = _lib.lib()
for fn, idxs in (('il_debug_stamps_gail', range(0, 9)), ('il_debug_stamps_sac', range(16, 30))):
  f = getattr(L, fn); f.restype = C.c_int
  buf = (C.c_ulonglong * 64)()
  f(buf)
  vals = [buf[i] for i in idxs]
  print(fn, 'cycles (100 MHz s_memtime ticks?):', [vals[i + 1] - vals[i] for i in range(len(vals) - 1)], 'total', vals[-1] - vals[0])
