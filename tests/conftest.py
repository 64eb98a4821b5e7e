import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  # libil_hip.so is a build artefact (git-ignored): bring it up to date with the sources before anything imports it. `make` is a no-op when it is
  # current; without hipcc the tests that need the library fail loudly (there is no fallback to test instead).
  import shutil
  import subprocess
  csrc = os.path.join(ROOT, 'imitation-learning_amd', 'csrc')
  if shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc'):
    env = dict(os.environ, PATH=os.environ.get('PATH', '') + ':/opt/rocm/bin')
    subprocess.run(['make', '-C', csrc, '-j8', '-s'], env=env, check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')


LEDGER = os.path.join(ROOT, 'tests', 'tolerance_ledger.json')   # the sites DESIGN.md 4 documents as running close to their allowance: {site prefix: why}


def pytest_sessionfinish(session, exitstatus):
  """Round 6: a measured outlier fraction above 60 % of its allowance fails the run unless its site has an entry in tests/tolerance_ledger.json (and with it in DESIGN.md 4):
  a drift towards an allowance is a finding, not noise to be absorbed by the next widening."""
  import json
  gu = sys.modules.get('gpu_util')
  if gu is None or not getattr(gu, 'FRACTIONS', None) or os.environ.get('IL_TOLERANCE_GATE', '1') == '0':
    return
  try:
    ledger = json.load(open(LEDGER))
  except Exception:
    ledger = {}
  over = sorted({(n, f, a) for n, f, a in gu.FRACTIONS if a > 0 and f > 0.6 * a and not any(n.startswith(k) for k in ledger)})
  if over:
    tr = session.config.pluginmanager.get_plugin('terminalreporter')
    msg = 'outlier fractions above 60 % of their allowance without an entry in tests/tolerance_ledger.json / DESIGN.md 4: ' + '; '.join(f'{n}: {f:.2e} / {a:.0e}' for n, f, a in over)
    if tr is not None: tr.write_line('TOLERANCE GATE: ' + msg, red=True)
    session.exitstatus = 1


def pytest_terminal_summary(terminalreporter):
  """The largest measured outlier fractions of the Adam-state comparisons (tests/gpu_util.py close_params / close_sparse) against their allowance."""
  import sys
  gu = sys.modules.get('gpu_util')
  if gu is None or not getattr(gu, 'FRACTIONS', None):
    return
  worst = sorted(gu.FRACTIONS, key=lambda t: -t[1])[:5]
  terminalreporter.write_line('largest outlier fractions (measured / allowed): ' + '; '.join(f'{n}: {f:.1e} / {a:.0e}' for n, f, a in worst))
  out = os.environ.get('IL_FRACTIONS_OUT')   # every measured fraction as JSON (the tolerance ledger of DESIGN.md 4 is built from a GPU run's file)
  if out:
    import json
    with open(out, 'w') as f:
      json.dump(dict(outlier_fractions=[dict(site=n, measured=fr, allowed=a) for n, fr, a in gu.FRACTIONS],
                     f64_brackets=[dict(site=n, hip_err_over_scale=h, reference_f32_err_over_scale=r) for n, h, r in getattr(gu, 'BRACKETS', [])]), f, indent=1)
