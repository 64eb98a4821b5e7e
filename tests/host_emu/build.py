"""TEST INFRASTRUCTURE: builds ONE host-emulation library from the HIP kernel sources of imitation-learning_amd/csrc (see emu_hip.hpp for the execution model).

`load()` translates every source file of csrc/ and every device header into
tests/host_emu/_build/, compiles them with g++ (-ffp-contract=off like the device build) against emu_hip.hpp standing in for <hip/hip_runtime.h>, and returns a ctypes
handle exporting the same `extern "C"` entry points as libil_hip.so - to be called with HOST pointers. Only the source TEXT is transformed, never its logic:
  * `kernel<<<grid, block, lds, stream>>>(args);`          -> `EMU_LAUNCH(kernel, grid, block, lds, args);`
  * `extern __shared__ ... float smem[];`                   -> a pointer to the emulated workgroup's LDS
  * `__attribute__((ext_vector_type(n)))`                   -> GCC's `vector_size(4 n)` (same indexing and arithmetic)
  * `__attribute__((address_space(n)))`                     -> dropped (one flat host address space)
  * `asm volatile("" ...)` register pins, `s_waitcnt` waits -> dropped (program order on one OS thread)
MFMA, DPP, readlane, ballot, shuffles, wave barriers, buffer loads / stores and the atomics are functions of emu_hip.hpp. The library is cached by a hash of every input.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'imitation-learning_amd', 'csrc')
BUILD = os.path.join(HERE, '_build')
# IL_EMU_ASAN=1: AddressSanitizer build (run python with LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0):
# every load / store of every emulated kernel - global buffers (numpy / torch allocations go through the intercepted malloc) and the workgroup's LDS - bounds-checked
SANITIZE = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer'] if os.environ.get('IL_EMU_ASAN', '0') == '1' else []
# IL_EMU_COVERAGE=1: gcov instrumentation (the translation keeps the sources' line numbers: profiles/tools/emu_coverage.sh reports executed lines per kernel file)
if os.environ.get('IL_EMU_COVERAGE', '0') == '1':
  SANITIZE = SANITIZE + ['--coverage']
NOT_EMULATED = ()

def _split_top_level(s: str):
  parts, depth, cur = [], 0, ''
  for ch in s:
    if ch in '([{': depth += 1
    elif ch in ')]}': depth -= 1
    if ch == ',' and depth == 0:
      parts.append(cur.strip()); cur = ''
    else:
      cur += ch
  parts.append(cur.strip())
  return parts


# `__shared__ [aligned] T a[..][..];` / `__shared__ T a, b;`: a workgroup's static LDS. Several workgroups can be resident at once (emu_hip.hpp), so a C++ `static` would be
# shared between them: each declaration becomes a reference to per-workgroup storage, allocated to the byte (the sanitised build then bounds-checks it too).
_SHARED = re.compile(r'__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?((?:unsigned\s+)?(?:long\s+long|short|int|float|\w+))\s+(\w+(?:\s*,\s*\w+)*)((?:\[[^\]]*\])*)\s*;')
_shared_count = [0]


def _shared_decl(m):
  typ, names, dims = m.group(1), [n.strip() for n in m.group(2).split(',')], m.group(3)
  out = []
  for n in names:
    _shared_count[0] += 1
    key = f'(const void*)"static-lds-{_shared_count[0]}-{n}"'
    if dims:
      out.append(f'{typ} (&{n}){dims} = *({typ} (*){dims})emu::block_static({key}, sizeof({typ}{dims}));')
    else:
      out.append(f'{typ}& {n} = *({typ}*)emu::block_static({key}, sizeof({typ}));')
  return ' '.join(out)


_LAUNCH = re.compile(r'([A-Za-z_][\w:]*(?:<[\w, ]+>)?)\s*<<<(.*?)>>>\s*\(')


def translate(src: str) -> str:
  src = re.sub(r'extern\s+__shared__\s+__attribute__\(\(aligned\(16\)\)\)\s+float\s+(\w+)\[\];', r'float* \1 = (float*)emu::dyn_smem;', src)
  src = re.sub(r'__attribute__\(\(ext_vector_type\((\d+)\)\)\)', lambda m: f'__attribute__((vector_size({4 * int(m.group(1))})))', src)
  src = re.sub(r'__attribute__\(\(address_space\(\d+\)\)\)', '', src)
  src = re.sub(r'asm volatile\(""[^;]*\);', '', src)
  src = re.sub(r'asm volatile\("s_waitcnt [^"]*"[^;]*\);', '', src)
  src = re.sub(r'asm volatile\("s_getreg_b32 %0, hwreg\(HW_REG_XCC_ID\)" : "=s"\((\w+)\)\);', r'\1 = emu::xcc_id();', src)   # which XCD: the block's linear id mod 8, as observed on the device
  src = re.sub(r'asm volatile\("s_getreg_b32 %0, hwreg\(HW_REG_HW_ID\)" : "=s"\((\w+)\)\);', r'\1 = ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 3 & 31u) << 8;', src)   # CU within the XCD: workgroups are dealt round-robin to the XCDs, then to their CUs
  assert 'asm volatile' not in src, 'an inline-assembly statement the emulator does not know'
  src = _SHARED.sub(_shared_decl, src)
  assert '__shared__' not in re.sub(r'//[^\n]*', '', src), 'a static __shared__ declaration the translation did not recognise'
  out, pos = '', 0
  for m in _LAUNCH.finditer(src):
    cfg = _split_top_level(m.group(2))
    assert 2 <= len(cfg) <= 4, m.group(0)
    grid, block, lds, stream = cfg[0], cfg[1], (cfg[2] if len(cfg) > 2 else '0'), (cfg[3] if len(cfg) > 3 else '0')
    out += src[pos:m.start()] + f'EMU_LAUNCH({m.group(1)}, {grid}, {block}, {lds}, {stream}, '
    pos = m.end()
  out += src[pos:]
  if os.environ.get('IL_EMU_DEBUG_WAITS', '0') == '1':   # name the device-side wait that expired
    out = out.replace('if (++spins > limit) { sync_timed_out(sync); break; }',
                      'if (++spins > limit) { fprintf(stderr, "emu: wait expired in block (%u,%u) of (%u,%u): counter %d is %lld, target %lld\\n", blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, which, (long long)sync[which], (long long)target); sync_timed_out(sync); break; }')
  assert '<<<' not in re.sub(r'//[^\n]*', '', out), 'a kernel launch the translation did not recognise'
  return out


_EXPORTS = '''// GENERATED by tests/host_emu/build.py: what a test drives the emulated streams with
#include <hip/hip_runtime.h>
extern "C" void emu_drain() { emu::drain(); }                                                          // hipDeviceSynchronize
extern "C" void emu_stream_wait(uintptr_t waiter, uintptr_t on) { emu::stream_wait(waiter, on); }      // waiter.wait_stream(on)
extern "C" void emu_capture_begin(uint64_t graph, uintptr_t stream) { emu::capture_begin(graph, stream); }   // hipStreamBeginCapture
extern "C" void emu_capture_end(uint64_t graph) { emu::capture_end(graph); }
extern "C" void emu_graph_launch(uint64_t graph, uintptr_t stream) { emu::graph_launch(graph, stream); }
extern "C" int emu_is_capturing() { return emu::capturing != 0; }
'''


def sources():
  return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip') and f not in NOT_EMULATED)


_HANDLE = None


def load() -> C.CDLL:
  global _HANDLE
  if _HANDLE is not None:
    return _HANDLE
  os.makedirs(BUILD, exist_ok=True)
  texts = {h: translate(open(os.path.join(CSRC, h)).read()) for h in sorted(os.listdir(CSRC)) if h.endswith('.hpp')}
  for f in sources():
    texts[f[:-4] + '.cpp'] = translate(open(os.path.join(CSRC, f)).read())
  texts['emu_exports.cpp'] = _EXPORTS
  hashed = dict(texts, **{'flags': ' '.join(SANITIZE) + os.environ.get('IL_EMU_DEBUG_WAITS', ''), 'emu_hip.hpp': open(os.path.join(HERE, 'emu_hip.hpp')).read(), 'il_hip.h': open(os.path.join(ROOT, 'include', 'il_hip.h')).read()})
  tag = hashlib.sha256('\0'.join(k + '\0' + v for k, v in sorted(hashed.items())).encode()).hexdigest()[:16]
  so = os.path.join(BUILD, f'libil_emu_{tag}.so')
  if not os.path.exists(so):
    d = os.path.join(BUILD, tag)
    src_dir = os.path.join(d, 'x', 'y')                     # the sources include "../../include/il_hip.h", as they do from csrc/
    for sub in (src_dir, os.path.join(d, 'hip'), os.path.join(d, 'include')):
      os.makedirs(sub, exist_ok=True)
    open(os.path.join(d, 'include', 'il_hip.h'), 'w').write(hashed['il_hip.h'])
    open(os.path.join(d, 'hip', 'hip_runtime.h'), 'w').write(f'#pragma once\n#include "{os.path.join(HERE, "emu_hip.hpp")}"\n')
    for k, v in texts.items():
      open(os.path.join(src_dir, k), 'w').write(v)
    flags = [*SANITIZE, '-O1', '-g', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-function', '-Wno-unused-variable', '-Wno-attributes', '-Wno-unknown-pragmas', '-Wno-ignored-attributes', '-I', d]
    procs = []
    for k in texts:
      if k.endswith('.cpp'):
        o = os.path.join(src_dir, k[:-4] + '.o')
        procs.append((k, o, subprocess.Popen(['g++', *flags, '-c', os.path.join(src_dir, k), '-o', o], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for k, o, p in procs:
      _, err = p.communicate()
      if p.returncode != 0:
        raise RuntimeError(f'host-emulation build of {k} failed:\n' + err[-6000:])
    r = subprocess.run(['g++', *SANITIZE, '-shared', '-o', so + '.tmp', *[o for _, o, _ in procs]], capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('host-emulation link failed:\n' + r.stderr[-4000:])
    os.replace(so + '.tmp', so)
  _HANDLE = C.CDLL(so)
  return _HANDLE


def load_selftest() -> C.CDLL:
  """The emulator's own check kernels (selftest_kernels.hip), through the same translation and flags."""
  os.makedirs(BUILD, exist_ok=True)
  text = translate(open(os.path.join(HERE, 'selftest_kernels.hip')).read())
  tag = hashlib.sha256((text + open(os.path.join(HERE, 'emu_hip.hpp')).read() + ' '.join(SANITIZE)).encode()).hexdigest()[:16]
  so = os.path.join(BUILD, f'libemu_selftest_{tag}.so')
  if not os.path.exists(so):
    d = os.path.join(BUILD, 'selftest_' + tag)
    os.makedirs(os.path.join(d, 'hip'), exist_ok=True)
    open(os.path.join(d, 'hip', 'hip_runtime.h'), 'w').write(f'#pragma once\n#include "{os.path.join(HERE, "emu_hip.hpp")}"\n')
    open(os.path.join(d, 'selftest.cpp'), 'w').write(text)
    r = subprocess.run(['g++', *SANITIZE, '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-Wno-attributes', '-I', d, os.path.join(d, 'selftest.cpp'), '-o', so + '.tmp'],
                       capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('host-emulation selftest build failed:\n' + r.stderr[-4000:])
    os.replace(so + '.tmp', so)
  return C.CDLL(so)
