"""Helpers for the -m gpu tests: move numpy arrays through the C ABI's raw device entry points."""
import ctypes as C

import numpy as np


class DevArray:
    def __init__(self, ctx, arr=None, nbytes=None):
        self.ctx = ctx
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            self.nbytes = arr.nbytes
            self.p = ctx.malloc(max(16, arr.nbytes))
            ctx.h2d(self.p, arr)
        else:
            self.nbytes = nbytes
            self.p = ctx.malloc(max(16, nbytes))
            ctx.memset(self.p, 0, nbytes)

    def get(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.d2h(out, self.p)
        return out

    def free(self):
        if self.p:
            self.ctx.free(self.p)
            self.p = None


def pad_rows(a, ld):
    """row-major 2-D array -> same rows with leading dimension ld (zero padded)."""
    a = np.asarray(a)
    out = np.zeros((a.shape[0], ld), dtype=a.dtype)
    out[:, : a.shape[1]] = a
    return out


def rup(x, m):
    return (x + m - 1) // m * m
