"""Developer tool (round 6): profiles/tools/direct_soak_matrix.py run as the child of a process that itself holds an (idle) GPU context with live allocations and a few
streams - the situation of tests/test_soak_gpu.py inside a pytest run, where the old hand-offs failed far more often than in a bare shell.
  python profiles/tools/soak_with_idle_parent.py <updates> <repeats> [switches]"""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
keep = [torch.zeros(64 << 20, device='cuda') for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
for s in streams:
  with torch.cuda.stream(s): keep[0].add_(1)
torch.cuda.synchronize()
sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, 'profiles', 'tools', 'direct_soak_matrix.py')] + sys.argv[1:], cwd=ROOT).returncode)
