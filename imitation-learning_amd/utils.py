"""Host-side conveniences of the training loop (reference utils.py). Plotting is optional: matplotlib is not part of this image."""
import itertools


def cycle(iterable):
  """Endless pass over `iterable`, re-iterating it each epoch so a shuffling loader reshuffles (reference utils.py:10-13)."""
  return itertools.chain.from_iterable(iter(lambda: iterable, None))


def lineplot(x, y, y2=None, filename='', xaxis='Steps', yaxis='Return', title=''):
  """Mean +- std band of per-evaluation returns over steps, written to `<filename>.png`; silently skipped without matplotlib."""
  try:
    import numpy as np
    from matplotlib import pyplot as plt
  except ImportError:
    return
  series = np.asarray(y, dtype=np.float64)
  centre, spread = series.mean(axis=1), series.std(axis=1)
  fig, ax = plt.subplots()
  ax.plot(x, centre, color='coral')
  ax.fill_between(x, centre - spread, centre + spread, color='coral', alpha=0.3)
  ax.set(xlim=(0, x[-1]), xlabel=xaxis, ylabel=yaxis, title=title)
  fig.savefig(f'{filename}.png')
  plt.close(fig)
