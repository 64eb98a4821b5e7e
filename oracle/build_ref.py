#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE, not product code: builds `oracle/_ref/` from the reference's own sources where they lie.

The reference's hot path (`training.py`, `models.py`, `memory.py`) is Python: "building" it means byte-compiling those three files, from /root/reference, into
sourceless `oracle/_ref/<module>.pyc` (what `hipcc` does to a .hip file, `py_compile` does to these) - no source text is copied, nothing under oracle/_ref/ is
tracked by git (.gitignore), and like the in-tree .so files it travels to the GPU box with the snapshot, so that `bench.py`'s `cpu_baseline` leg can time the reference's
OWN code on the bench host's cores instead of quoting a number measured elsewhere. The 12-line `omegaconf` stand-in (ours: the real package is not
installed; the reference only uses `DictConfig` for a type annotation and attribute / `.get` access) is written next to them.

  python oracle/build_ref.py            # no-op with a message when /root/reference is absent (the GPU box: it uses what the container built)

Only `bench.py`'s cpu_baseline leg (through oracle/ref_cpu_baseline.py, in a subprocess) and tests may load oracle/_ref; nothing under imitation-learning_amd/ may
(tests/test_abi_and_layout.py::test_product_package_never_imports_the_oracle).
"""
import hashlib
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
MODULES = ('memory', 'models', 'training')   # reference memory.py:1-68, models.py:1-318, training.py:1-134

STUB = '''"""Stand-in for the `omegaconf` package (not installed here): the reference's hot-path modules use DictConfig for a type annotation and attribute / .get access only."""


class DictConfig(dict):
  def __getattr__(self, k):
    try: v = self[k]
    except KeyError: raise AttributeError(k)
    return DictConfig(v) if isinstance(v, dict) and not isinstance(v, DictConfig) else v

  def __setattr__(self, k, v): self[k] = v


class OmegaConf: pass
'''


def build(reference: str = None) -> bool:
  reference = reference or os.environ.get('IL_REFERENCE', '/root/reference')
  if not all(os.path.isfile(os.path.join(reference, m + '.py')) for m in MODULES):
    print(f'[oracle/build_ref] {reference} is not here: keeping whatever oracle/_ref/ holds ({"present" if os.path.isdir(OUT) else "absent"})')
    return False
  os.makedirs(os.path.join(OUT, 'omegaconf'), exist_ok=True)
  manifest = dict(python=sys.version.split()[0], magic=__import__('importlib.util').util.MAGIC_NUMBER.hex(), reference=reference, modules={})
  for m in MODULES:
    src = os.path.join(reference, m + '.py')
    py_compile.compile(src, cfile=os.path.join(OUT, m + '.pyc'), dfile=f'<reference>/{m}.py', doraise=True)
    manifest['modules'][m] = dict(source_sha256=hashlib.sha256(open(src, 'rb').read()).hexdigest())
  with open(os.path.join(OUT, 'omegaconf', '__init__.py'), 'w') as f:
    f.write(STUB)
  json.dump(manifest, open(os.path.join(OUT, 'MANIFEST.json'), 'w'), indent=1)
  print(f'[oracle/build_ref] byte-compiled {", ".join(MODULES)} from {reference} into {OUT}')
  return True


if __name__ == '__main__':
  build(sys.argv[1] if len(sys.argv) > 1 else None)
