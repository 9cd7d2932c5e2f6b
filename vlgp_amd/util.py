"""Host-side data-layout helpers of the hot path (vlgp/util.py:446-499)."""
import math

import numpy as np


def clip(a, lbound, ubound=None):
    """In-place symmetric (or two-sided) clamp; vlgp/util.py:446-454."""
    if ubound is None:
        if not lbound > 0:
            raise AssertionError("bound must be positive")
        lbound, ubound = -lbound, lbound
    elif not ubound > lbound:
        raise AssertionError("ubound must exceed lbound")
    np.clip(a, lbound, ubound, out=a)


def segment_starts(length, window):
    """Start rows of the ceil(length/window) segments of one trial.

    When ``length`` is not a multiple of ``window`` the surplus is spread as
    random overlaps drawn from the global NumPy RNG, with the same call the
    reference makes (vlgp/util.py:482-493)."""
    k = math.ceil(length / window)
    surplus = k * window - length
    starts = np.arange(k) * window
    shift = np.cumsum(np.append([0], np.random.multinomial(surplus, np.ones(k - 1) / (k - 1))))
    return starts - shift


def cut_trials(trials, params, config):
    """util.cut_trials (vlgp/util.py:457-499): window-sized segments that are
    NumPy *views* of the parent trial's y, x, mu, w, v."""
    window = config["window"]
    if not window:
        return trials
    segs = []
    for tr in trials:
        for s in segment_starts(tr["y"].shape[0], window):
            sl = slice(int(s), int(s) + window)
            segs.append({k: tr[k][sl] for k in ("y", "x", "mu", "w", "v")})
    return np.array(segs, dtype=object)


def save(result, path, ext="npy"):
    """Store a ``fit`` result (vlgp/util.py:181-190): one pickled object in ``.npy``, or the
    top-level keys as arrays in ``.npz``."""
    import pathlib

    path = pathlib.Path(path)
    if ext == "npy":
        np.save(path.with_suffix(".npy"), result, allow_pickle=True)
    elif ext == "npz":
        np.savez(path.with_suffix(".npz"), **result)
    else:
        raise NotImplementedError("unknown file type {}".format(ext))


def load(path):
    """Read what ``save`` wrote (vlgp/util.py:193-208).  The reference calls ``np.load`` without
    ``allow_pickle``, which NumPy >= 1.16.3 refuses for the pickled ``.npy`` it writes itself;
    pickles are allowed here because the file format is one."""
    import pathlib

    path = pathlib.Path(path)
    if not path.exists():
        raise FileNotFoundError(path.as_posix())
    if path.suffix == ".npy":
        return np.load(path, allow_pickle=True)[()]
    if path.suffix == ".npz":
        with np.load(path, allow_pickle=True) as f:
            return {k: (f[k][()] if f[k].dtype == object and f[k].shape == () else f[k]) for k in f.files}
    raise NotImplementedError("unknown file type {}".format(path.suffix))
