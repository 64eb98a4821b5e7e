#!/usr/bin/env python3
"""Static check for the pattern that cost k_dw_adam 2x in rounds 1-2: an inner loop whose body issues ONE or two global loads and then waits for them
(`s_waitcnt vmcnt(0)`) before the next trip - every trip is a full L2 / fabric round trip, and hipcc neither unrolls a runtime-bounded loop nor software-pipelines it.
Compiles each .hip to ISA (no GPU needed) and lists the self-looping basic blocks with <= `--max-loads` global loads and a full vmcnt wait.

  python profiles/tools/serial_load_loops.py [file.hip ...] [--max-loads 2]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'imitation-learning_amd', 'csrc')


def scan(path, max_loads, extra):
  with tempfile.NamedTemporaryFile(suffix='.s') as f:
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-S', '--cuda-device-only', '-o', f.name, path] + extra,
                   check=True, stderr=subprocess.DEVNULL, cwd=CSRC)
    lines = open(f.name).read().splitlines()
  kernel, block, body, found = None, None, [], []
  for ln in lines:
    m = re.match(r'^(_Z\w+|k_\w+):', ln)
    if m: kernel = m.group(1)
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
      block, body = m.group(1), []
      continue
    if block is None: continue
    body.append(ln)
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', ln)
    if m and m.group(1) == block:   # self loop
      loads = sum(1 for b in body if re.search(r'\b(global|buffer|flat)_load', b))
      waits0 = any(re.search(r's_waitcnt.*vmcnt\(0\)', b) for b in body)
      mfma = sum(1 for b in body if 'v_mfma' in b)
      if 0 < loads <= max_loads and waits0:
        found.append((kernel, block, loads, mfma, len(body)))
  return found


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('files', nargs='*')
  ap.add_argument('--max-loads', type=int, default=2)
  ap.add_argument('--extra', default='')
  a = ap.parse_args()
  files = a.files or sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
  for f in files:
    for kernel, block, loads, mfma, n in scan(os.path.join(CSRC, f), a.max_loads, a.extra.split()):
      m = re.match(r'_Z(\d+)', kernel)
      name = kernel[len(m.group(0)):len(m.group(0)) + int(m.group(1))] if m else kernel
      print(f'{f:18s} {name:28s} {block:12s} loads/trip {loads}  mfma/trip {mfma:2d}  instructions {n}')


if __name__ == '__main__':
  main()
