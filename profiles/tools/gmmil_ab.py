"""Developer tool: A/B of k_gmmil_tile builds (same ABI, IL_HIP_LIBRARY selects the build). GMMIL.predict_reward at B = 1024, Ant dims, plus a ragged size;
prints per-kernel average durations (HIP events on the launch stream, il_trace_*) and a digest of the rewards, which must not depend on the build
(every variant keeps each pair's accumulation order over the features and the 64-column partial sums).
  IL_HIP_LIBRARY=imitation-learning_amd/csrc/build/ab/libil_hip_<v>.so python profiles/tools/gmmil_ab.py"""
import ctypes as C
import hashlib
import os
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
import imitation_learning_amd as il
from imitation_learning_amd import _lib

dev = torch.device('cuda', 0)
L = _lib.lib()


def case(Bp, Be, Sg, Ag, calls):
  rs = np.random.RandomState(5)
  mk = lambda n, shift: (torch.from_numpy((rs.standard_normal((n, Sg)) + shift).astype(np.float32)).to(dev), torch.from_numpy(rs.uniform(-1, 1, (n, Ag)).astype(np.float32)).to(dev))
  (xs, xa), (es, ea) = mk(Bp, 0.0), mk(Be, 0.5)
  wp, we = torch.ones(Bp, device=dev), torch.ones(Be, device=dev)
  gm = il.GMMILDiscriminator(Sg, Ag, bench.Cfg(state_only=False))
  for _ in range(20): r = gm.predict_reward(xs, xa, es, ea, wp, we)
  torch.cuda.synchronize()
  digest = hashlib.sha256(r.cpu().numpy().tobytes()).hexdigest()[:16]
  L.il_trace_enable(1)
  for _ in range(calls): gm.predict_reward(xs, xa, es, ea, wp, we)
  torch.cuda.synchronize()
  buf = C.create_string_buffer(1 << 16)
  _lib.check(L.il_trace_report(buf, len(buf)))
  L.il_trace_enable(0)
  kern = {}
  for line in buf.value.decode().strip().splitlines():
    name, cnt, tot = line.split()
    kern[name] = round(float(tot) / int(cnt) * 1e3, 2)
  t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0.record()
  for _ in range(calls): gm.predict_reward(xs, xa, es, ea, wp, we)
  t1.record(); torch.cuda.synchronize()
  return dict(us_per_call=round(t0.elapsed_time(t1) * 1e3 / calls, 2), kernels_us=kern, digest=digest)


if os.environ.get('IL_GMMIL_AB_SWEEP') == '1':   # round 6: where the centred Gram launch (IL_GMMIL_MFMA=1) wins over the direct-difference launch: batch x feature sweep
  for B in (128, 256, 512, 1024, 2048):
    for Sg, Ag in ((18, 6), (112, 8)):
      r = case(B, B, Sg, Ag, 200)
      print(f"IL_GMMIL_MFMA={os.environ.get('IL_GMMIL_MFMA', '1')} B={B} D={Sg + Ag}: kernel {r['kernels_us'].get('k_gmmil_tile')} us, {r['us_per_call']} us per call", flush=True)
else:
  print(os.environ.get('IL_HIP_LIBRARY', 'default'), dict(b1024_ant=case(1024, 1024, 112, 8, 300), ragged_200x333=case(200, 333, 18, 6, 100), b256_hc=case(256, 256, 18, 6, 100)), flush=True)
