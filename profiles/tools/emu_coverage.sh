#!/bin/bash
# Line coverage of the kernel SOURCES under the host-emulated CPU tests (tests/host_emu): which lines of csrc/*.hip / *.hpp does `pytest -m "not gpu"` execute?
#   bash profiles/tools/emu_coverage.sh [pytest -k expression]        (gcov build of the emulated library, ~1 min to compile; report on stdout)
cd "$(dirname "$0")/../.."
export IL_EMU_COVERAGE=1
rm -rf tests/host_emu/_build/cov_report; mkdir -p tests/host_emu/_build/cov_report
python -m pytest tests/test_kernels_host_emulation.py -q -x -p no:cacheprovider ${1:+-k "$1"} --deselect tests/test_kernels_host_emulation.py::test_emulated_kernels_do_not_depend_on_the_wave_schedule \
  --deselect tests/test_kernels_host_emulation.py::test_emulator_schedule_perturbation_exposes_a_missing_barrier --deselect tests/test_kernels_host_emulation.py::test_emulator_sanitised_build_sees_one_element_past_a_buffer 2>&1 | tail -1
d=$(ls -td tests/host_emu/_build/*/x/y 2>/dev/null | head -1)
cd "$d" || exit 1
for f in *.gcda; do gcov -l -r -s "$PWD" "${f%.gcda}.o" > /dev/null 2>&1; done
python3 - <<'PY'
import glob, re, collections
lines = collections.defaultdict(dict)   # file -> line -> executed in ANY translation unit (a header is compiled into several)
for g in sorted(glob.glob('*.gcov')):
  src = None
  for line in open(g, errors='replace'):
    m = re.match(r'\s*([^:]+):\s*(\d+):(.*)', line)
    if not m: continue
    cnt, ln, text = m.group(1).strip(), int(m.group(2)), m.group(3)
    if ln == 0:
      if text.startswith('Source:'): src = text[7:].split('/')[-1].replace('.cpp', '.hip')
      continue
    if cnt == '-' or not src or not src.endswith(('.hip', '.hpp')): continue
    hit = cnt not in ('#####', '=====')
    lines[src][ln] = lines[src].get(ln, False) or hit
print(f"{'file':28s} executable lines   executed")
R = N = 0
for k in sorted(lines):
  n, r = len(lines[k]), sum(lines[k].values()); R += r; N += n
  print(f'{k:28s} {n:8d}        {r:6d}  {100.0 * r / n:5.1f} %')
print(f"{'all kernel sources':28s} {N:8d}        {R:6d}  {100.0 * R / N:5.1f} %")
import json; json.dump({k: sorted(l for l, h in v.items() if not h) for k, v in lines.items()}, open('../../../cov_report/not_executed.json', 'w'))
PY
