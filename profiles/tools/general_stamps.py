"""Developer tool: phase stamps (s_memtime of one workgroup; -DIL_PHASE_STAMPS build: bash profiles/tools/build_variants.sh stamps:"-DIL_PHASE_STAMPS -w") of the tile
engine's k_gt_fwd / k_gt_bwd on depth 3 / tanh / 256:  IL_HIP_LIBRARY=variants/stamps/libil_hip.so python profiles/tools/general_stamps.py [hidden depth activation]"""
import ctypes as C, sys
sys.path[:0] = ['.', 'tests', 'tests/golden']
import numpy as np, torch
import imitation_learning_amd as il
import inputs as gi
from gpu_util import Cfg, tbatch
from imitation_learning_amd import _lib
dev = torch.device('cuda', 0)
S, A, B = 18, 6, 256
hidden, depth, act = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 3, sys.argv[3] if len(sys.argv) > 3 else 'tanh'
cfg = Cfg(hidden_size=hidden, depth=depth, activation=act)
actor, critic = il.SoftActor(S, A, cfg, device=dev), il.TwinCritic(S, A, cfg, device=dev)
if depth == 2 and act == 'relu': actor.general = critic.general = True
target, log_alpha = il.create_target_network(critic), torch.zeros(1, device=dev)
ao, co, to = il.AdamW(actor, lr=3e-4, weight_decay=0), il.AdamW(critic, lr=3e-4, weight_decay=0), il.Adam(log_alpha, lr=3e-4)
b = tbatch(gi.transitions(np.random.RandomState(0), B, S, A, weighted=True))
for _ in range(30): il.sac_update(actor, critic, log_alpha, target, b, ao, co, to, 0.97, -3.0, 0.99)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 64)()
assert raw.il_debug_stamps_general(buf) == 0
v = [int(x) for x in buf]
us = lambda a, b: (v[b] - v[a]) / 100.0
print(f'shape: hidden {hidden} depth {depth} {act}; s_memtime at 100 MHz; the LAST launch of each kernel (fwd: critic_2(s, a~) pass; bwd: actor head pass), workgroup 0')
print('k_gt_fwd: rows', us(0, 1), '| layer 0', us(1, 2), '| hidden layers', [us(2 + l - 1, 2 + l) for l in range(1, depth)], '| output layer', us(2 + depth - 1, 10), '| head', us(10, 11), '| total', us(0, 11))
print('k_gt_bwd: seed', us(16, 17), '| through the output layer', us(17, 18), '| hidden layers', [us(18 + i - 1, 18 + i) for i in range(1, depth)], '| rest', us(18 + depth - 1, 27), '| total', us(16, 27))
