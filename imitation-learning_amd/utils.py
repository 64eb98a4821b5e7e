"""Small helpers of the reference's utils.py (plotting is optional: seaborn / matplotlib are not part of this environment)."""


def cycle(iterable):
  """Cycles over an iterable without caching the order (reference utils.py:10-13)."""
  while True:
    for x in iterable:
      yield x


def lineplot(x, y, y2=None, filename='', xaxis='Steps', yaxis='Return', title=''):
  try:
    import numpy as np
    from matplotlib import pyplot as plt
  except Exception:
    return  # plotting is a convenience of the reference loop, never a dependency of the update path
  y = np.array(y)
  mean, std = y.mean(axis=1), y.std(axis=1)
  plt.plot(x, mean, color='coral'); plt.fill_between(x, mean - std, mean + std, color='coral', alpha=0.3)
  plt.xlim(left=0, right=x[-1]); plt.xlabel(xaxis); plt.ylabel(yaxis); plt.title(title)
  plt.savefig(f'{filename}.png'); plt.close()
