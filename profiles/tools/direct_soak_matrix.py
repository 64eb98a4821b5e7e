"""Developer tool (round 6): the soak of profiles/tools/pair_soak.py, copies load only, repeated for a matrix of schedule switches and launch modes: how often does the
digest of N back-to-back updates under an uneven load differ from the quiet run's?  python profiles/tools/direct_soak_matrix.py [updates] [repeats]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
src = open(os.path.join(ROOT, 'profiles', 'tools', 'pair_soak.py')).read()
LEARNER = src.split('LEARNER = f"""')[1].split('"""')[0].replace('{N}', str(N))
COPIES = "import torch, time\na = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)\nt = time.time()\nwhile time.time() - t < 120: b.copy_(a); torch.cuda.synchronize()\n"
def learner(env):
  r = subprocess.run([sys.executable, '-c', LEARNER], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=900)
  if r.returncode != 0: return 'FAILED: ' + r.stderr.strip().splitlines()[-1][:120]
  return [l for l in r.stdout.splitlines() if l.startswith('DIGEST')][-1].split()[1]
cases = [dict(IL_SOAK_LAUNCH='direct'), dict(IL_SOAK_LAUNCH='direct', IL_PAIR='0'), dict(IL_SOAK_LAUNCH='direct', IL_EARLY_DRAW='0'), dict(IL_SOAK_LAUNCH='graph'),
         dict(IL_SOAK_LAUNCH='direct', IL_INLINE_RELABEL='0'), dict(IL_SOAK_LAUNCH='direct', IL_RING_GATHER='0'), dict(IL_SOAK_LAUNCH='thread')]
if len(sys.argv) > 3: cases = [dict(kv.split('=') for kv in c.split(',')) for c in sys.argv[3:]]
quiet = {}
for env in cases:
  key = tuple(sorted(env.items()))
  want = learner(env)
  got = []
  for _ in range(R):
    bg = subprocess.Popen([sys.executable, '-c', COPIES], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(3)
    got.append(learner(env))
    bg.kill(); bg.wait()
  print(f'{env}: quiet {want}; under copies: {got}; mismatches {sum(g != want for g in got)} / {R}', flush=True)
