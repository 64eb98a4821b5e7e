"""Developer tool: workgroup-level timeline of ONE steady-state SAC+GAIL update of the headline schedule (two graphs, device hand-off, resident sampler), from a
-DIL_TIMELINE build of the library (every workgroup of the instrumented kernels stores s_memrealtime, 100 MHz, device-wide, at its phase boundaries).

  cd imitation-learning_amd/csrc && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -ffp-contract=off -I. -I../../include -DIL_TIMELINE -c $f -o build/ab/tl/${f%.hip}.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC build/ab/tl/*.o -o build/ab/libil_hip_tl.so
  IL_HIP_LIBRARY=imitation-learning_amd/csrc/build/ab/libil_hip_tl.so python profiles/tools/update_timeline.py [replays]

Times are microseconds relative to the moment the sampler workgroup signals [IL_SYNC_INDICES] for the LAST replayed update (the stamps of earlier updates are overwritten).
min / median / max over the workgroups of a role."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
from imitation_learning_amd import _lib

K, W, S = 12, 512, 8   # IL_TL_K, IL_TL_WGS, IL_TL_SLOTS (csrc/il_common.hpp)
dev = torch.device('cuda', 0)
plan, nets, _ = bench.build(dev, 0)
plan.capture(warmup=3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(n): plan.replay()
torch.cuda.synchronize()
assert plan.sync_timeouts() == 0
raw = C.CDLL(_lib.LIB_PATH)


def read(fn):
  buf = (C.c_ulonglong * (K * W * S))()
  assert getattr(raw, fn)(buf) == 0
  return np.frombuffer(buf, dtype=np.uint64).reshape(K, W, S).astype(np.int64)


sac, gail = read('il_debug_timeline_sac'), read('il_debug_timeline_gail')
B, nt = plan.B, plan.B // 16
# sampler = the last column's pass-0 workgroup of k_gail_grad: linear id nt (grid (nt + 1, 3))
t0 = gail[0, nt, 3]
us = lambda a: (np.asarray(a, np.float64) - t0) / 100.0


def row(name, a):
  a = np.asarray(a, np.float64)
  a = a[np.isfinite(a)]
  print(f'  {name:54s} {a.min():8.2f} {np.median(a):8.2f} {a.max():8.2f}   (n = {a.size})')


def chain_roles():
  role, net, tile = [], [], []
  for bid in range(6 * nt):
    x, q, ra, rc = bid & 7, bid >> 3, nt >> 3, nt >> 2
    if q < ra: r, k, t = 0, 0, q * 8 + x
    elif q < ra + 2 * rc: r, k, t = (1 if q < ra + rc else 2), x >> 2, (x & 3) * rc + (q - ra) % rc
    else: r, k, t = 3, 0, (q - ra - 2 * rc) * 8 + x
    role.append(r); net.append(k); tile.append(t)
  return np.array(role), np.array(net), np.array(tile)


print(f'update period {1e6 / 1:.0f}' if False else f'B = {B}; all times in us after the sampler signalled [IL_SYNC_INDICES]; min / median / max over workgroups')
print('discriminator branch (side stream)')
sam = gail[0, nt]
row('sampler: previous update over ([IL_SYNC_MAIN_EPOCH] seen)', us([sam[2]]))
row('sampler: indices drawn (signal issued)', us([sam[3]]))
gw = np.array([x + (nt + 1) * y for y in range(3) for x in range(nt)])
row('k_gail_grad: launched', us(gail[0, gw, 0]))
row('k_gail_grad: preparation done, waiting for the indices', us(gail[0, gw, 1]))
row('k_gail_grad: indices seen', us(gail[0, gw, 2]))
row('k_gail_grad: done', us(gail[0, gw, 7]))
nr = int((gail[1, :, 0] > t0 - 10_000_000).sum()) if False else int((gail[1, :, 7] >= t0).sum())
row('k_gail_reduce: launched', us(gail[1, :nr, 0]))
row('k_gail_reduce: done ([IL_SYNC_PARAMS] signalled)', us(gail[1, :nr, 7]))
print('SAC branch (main stream)')
role, net, tile = chain_roles()
c = sac[0]
row('k_sac_chain: launched', us(c[:6 * nt + 3, 0]))
row('k_sac_chain: indices seen', us(c[:6 * nt + 3, 1]))
for r, name in ((0, "actor(s')"), (3, 'actor(s)')):
  row(f'  {name}: done', us(c[:6 * nt][role == r, 7]))
tg = c[:6 * nt][role == 1]
row("  targets: actor(s') of the tile seen", us(tg[:, 2]))
row('  targets: forward done', us(tg[:, 6]))
row('  targets: arrival signalled', us(tg[:, 7]))
cr = c[:6 * nt][role == 2]
row('  critics: forward done', us(cr[:, 2]))
row('  critics: backward GEMM done', us(cr[:, 3]))
row('  critics: discriminator step seen ([IL_SYNC_PARAMS])', us(sac[4, :6 * nt][role == 2, 0]))
row('  critics: relabel done', us(cr[:, 4]))
row('  critics: targets of the tile seen', us(cr[:, 5]))
row('  critics: done', us(cr[:, 7]))
row('  row-copy workgroups: done', us(c[6 * nt:6 * nt + 3, 7]))
def dw_rows(kid, name, n_blocks):
  m = sac[kid, :, 7] >= t0
  row(f'{name}: launched', us(sac[kid, m, 0]))
  row(f'{name}: done', us(sac[kid, m, 7]))
  idx = np.where(m)[0]
  for lo, hi, what in ((0, n_blocks, '32 x 32 LDS block jobs (every layer + biases)'), (n_blocks, 10 ** 6, 'tail blocks (polyak, alpha)')):
    sel = idx[(idx >= lo) & (idx < hi)]
    if sel.size:
      if lo == 0 and (sac[kid, sel, 1] >= t0).all():
        row(f'    {what}: products done', us(sac[kid, sel, 1]))
      row(f'    {what}: done', us(sac[kid, sel, 7]))
H = 256
jobs = lambda IN, OUT: (H // 32) ** 2 + (H // 32) * ((IN + 31) // 32) + ((OUT + 31) // 32) * (H // 32)
dw_rows(1, 'k_dw_adam (critic)', 2 * jobs(bench.S + bench.A, 1))
p = sac[3]
row('k_policy_critic: launched', us(p[:2 * nt + 4 * nt][p[:6 * nt, 0] >= t0, 0]))
row('  critic workgroups: arrival signalled', us(p[:2 * nt, 7]))
h = p[2 * nt:6 * nt]
h = h[h[:, 7] >= t0]
row('  helpers: own work done, waiting for the critics', us(h[:, 1]))
row('  helpers: both critics of the tile seen', us(h[:, 2]))
row('  helpers: done', us(h[:, 7]))
dw_rows(2, 'k_dw_adam (actor) + tail', jobs(bench.S, 2 * bench.A))
print('phases inside the forward tiles of k_sac_chain and the critic workgroups of k_policy_critic (us since the workgroup entered the phase list; median over workgroups)')
for kid, name, idx in ((5, "actor(s') tile", np.where(role == 0)[0]), (6, 'actor(s) tile', np.where(role == 3)[0])):
  a = sac[kid][idx]
  d = (a[:, 1:6] - a[:, 0:5]) / 100.0
  print(f'  {name:16s} rows gathered {np.median(d[:, 0]):.2f} | layer 1 {np.median(d[:, 1]):.2f} | layer 2 {np.median(d[:, 2]):.2f} | head GEMM {np.median(d[:, 3]):.2f} | sample, log-prob, stores {np.median(d[:, 4]):.2f}')
a = sac[7][:2 * nt]
d = (a[:, [1, 2, 3, 5, 6, 7]] - a[:, [0, 1, 2, 3, 5, 6]]) / 100.0
print(f'  policy critic    rows {np.median(d[:, 0]):.2f} | layer 1 {np.median(d[:, 1]):.2f} | layer 2 {np.median(d[:, 2]):.2f} | Q + mask {np.median(d[:, 3]):.2f} | layer 2 backward {np.median(d[:, 4]):.2f} | dQ/da columns {np.median(d[:, 5]):.2f}')
