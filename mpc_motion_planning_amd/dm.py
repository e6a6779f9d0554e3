"""A minimal stand-in for the casadi.DM values the reference's drivers touch: `.full()` and array conversion
(main_cbf_kin_c_sim.py:17-18,102,118-119)."""
import numpy as np


class DM:
    __slots__ = ("_a",)

    def __init__(self, a):
        a = np.array(a, dtype=np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        self._a = a

    def full(self):
        return self._a.copy()

    @property
    def shape(self):
        return self._a.shape

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __getitem__(self, idx):
        return DM(self._a.reshape(-1)[idx]) if not isinstance(idx, tuple) else DM(self._a[idx])

    def __float__(self):
        return float(self._a.reshape(-1)[0])

    def __len__(self):
        return self._a.shape[0]

    def __repr__(self):
        return "DM(%r)" % (self._a.tolist(),)
