"""Developer tool (round 5): the discriminator branch on CUs of its own (hipExtStreamCreateWithCUMask) instead of the whole-CU LDS requests that keep its
workgroups off the pair-mode workgroups' CUs. Both branches run on masked streams (side: the first `n` mask bits per XCD x 8; main: the rest); direct launches.
  python profiles/tools/cu_mask_experiment.py <side CUs per XCD, 0 = no masks>      [IL_PAIR_LDS_KB=124 drops the 160 KB request of the pair kernels]
Prints updates/s, the stamps' kernel durations, and where the side / main workgroups actually ran (distinct CUs per XCD)."""
import collections, ctypes as C, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch, bench
from imitation_learning_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device('cuda', 0)
L = _lib.lib()
cus = torch.cuda.get_device_properties(dev).multi_processor_count
words = (cus + 31) // 32


def masked(bits):
  m = (C.c_uint32 * words)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(words)])
  out = C.c_void_p()
  _lib.check(L.il_stream_create_cu_mask(m, words, C.byref(out)))
  return torch.cuda.ExternalStream(out.value, device=dev)

plan, nets, _ = bench.build(dev, 0)
main = torch.cuda.current_stream()
if n > 0:
  side_bits = set(range(8 * n))             # the runtime deals mask bits round-robin to the XCDs: bits 0 .. 8n-1 = n CUs on each of the 8 XCDs
  plan.side = masked(side_bits)
  main = masked(set(range(cus)) - side_bits)
with torch.cuda.stream(main):
  for _ in range(5): plan.run()
  torch.cuda.synchronize()
  assert plan.sync_timeouts() == 0, 'the two masked streams did not run concurrently'
  plan.record_direct()
  for _ in range(300): plan.launch_direct()
  torch.cuda.synchronize()
  rates = []
  for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(3000): plan.launch_direct()
    torch.cuda.synchronize()
    rates.append(3000 / (time.perf_counter() - t0))
st = _lib.kernel_stamps()
place = {}
for k in ('k_sac_chain_pair', 'k_policy_critic_pair', 'k_gail_grad', 'k_gail_reduce', 'k_dw_adam_critic'):
  per = collections.Counter(p >> 16 for _, _, p in _lib.kernel_stamp_rows(k))
  place[k] = (len({p for _, _, p in _lib.kernel_stamp_rows(k)}), dict(sorted(per.items())))
side_cus = {p for k in ('k_gail_grad', 'k_gail_reduce') for _, _, p in _lib.kernel_stamp_rows(k)}
main_cus = {p for k in ('k_sac_chain_pair', 'k_policy_critic_pair', 'k_dw_adam_critic') for _, _, p in _lib.kernel_stamp_rows(k)}
print(f'side CUs per XCD {n}, IL_PAIR_LDS_KB={os.environ.get("IL_PAIR_LDS_KB", "160")}: {np.median(rates):.0f} updates/s (min {min(rates):.0f}, max {max(rates):.0f}); timeouts {plan.sync_timeouts()}; '
      f'durations {({k: round(v["duration_us"], 2) for k, v in st.items()})}; distinct CUs and workgroups per XCD: {place}; CUs used by both branches: {len(side_cus & main_cus)}')
