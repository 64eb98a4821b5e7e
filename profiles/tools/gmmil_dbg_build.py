"""Developer tool: builds imitation-learning_amd/csrc/build/ab/libil_hip_dbg.so = the product library with k_gmmil_tile<0> instrumented to store s_memtime at its
phase boundaries for EVERY workgroup (start, first barrier, end of chunk 0, end of the feature loop, partial sums written, ticket taken, last arriver's sums written)
plus (HW_REG_XCC_ID, HW_REG_HW_ID). The instrumented source is generated from gmmil.hip (its IL_STAMP lines are the insertion points); nothing of it is part of the
product. Read with profiles/tools/gmmil_timeline.py.   python profiles/tools/gmmil_dbg_build.py [-DGMMIL_RB=4 ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, 'imitation-learning_amd', 'csrc')
AB = os.path.join(CS, 'build', 'ab')
os.makedirs(AB, exist_ok=True)
s = open(os.path.join(CS, 'gmmil.hip')).read()


def sub(old, new, count=1):
  global s
  assert s.count(old) >= 1, old
  s = s.replace(old, new, count)


sub('#include "il_common.hpp"\n', '''#include "il_common.hpp"
static __device__ unsigned long long gm_tl[8 * 4096];
extern "C" int il_debug_gmmil_timeline(unsigned long long* out_host) { return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(gm_tl), sizeof(unsigned long long) * 8 * 4096) == hipSuccess ? 0 : 3; }
#define TL(i) do { if (MODE == 0 && threadIdx.x == 0) gm_tl[8 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
''')
sub('  IL_STAMP(stamp, 0);\n', '''  TL(0);
  if (MODE == 0 && threadIdx.x == 0) { unsigned x, h; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    gm_tl[8 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + 7] = ((unsigned long long)(x & 0xf) << 32) | h; }
''')
sub('    __syncthreads();\n    if (k0 + NPF * GKC < D) fetch(', '    __syncthreads();\n    if (k0 == 0) TL(1);\n    if (k0 + NPF * GKC < D) fetch(')
sub('    }\n    __syncthreads();\n   }\n  }\n  IL_STAMP(stamp, 1);', '    }\n    if (k0 == 0) TL(6);\n    __syncthreads();\n   }\n  }\n  TL(2);')
sub('  IL_STAMP(stamp, 2);', '  TL(3);')
sub('  IL_STAMP(stamp, 3);', '  TL(4);')
sub('  out_r[i] = sim - self;', '  out_r[i] = sim - self;\n  TL(5);')
open(os.path.join(AB, 'gmmil_dbg.hip'), 'w').write(s)
flags = '--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -ffp-contract=off'.split() + ['-I' + CS, '-I' + os.path.join(ROOT, 'include')] + sys.argv[1:]
subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-c', os.path.join(AB, 'gmmil_dbg.hip'), '-o', os.path.join(AB, 'gmmil_dbg.o')])
others = [os.path.join(CS, 'build', f) for f in sorted(os.listdir(os.path.join(CS, 'build'))) if f.endswith('.o') and f != 'gmmil.o']
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + others + [os.path.join(AB, 'gmmil_dbg.o'), '-o', os.path.join(AB, 'libil_hip_dbg.so')])
print('built', os.path.join(AB, 'libil_hip_dbg.so'))
