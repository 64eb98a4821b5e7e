"""Importable alias of the `imitation-learning_amd/` package (a hyphen cannot appear in a Python module name).

All code lives in `imitation-learning_amd/`; this module only points its package search path there and runs that
package's __init__ so `import imitation_learning_amd.training` etc. resolve to the real files.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'imitation-learning_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
  exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
