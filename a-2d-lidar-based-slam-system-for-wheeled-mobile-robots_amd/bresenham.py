"""``bresenham(start, end).path`` on the GPU.

Mirrors the class of the reference (W12m/bresenham.py:2-58): integer endpoints in, a
list of ``(x, y)`` tuples from start to end inclusive out, empty for identical endpoints
(:10-11).  The rasteriser keeps the reference's float64 error accumulation (:34-55), so it
is NOT integer Bresenham.  The walk runs in libslamhip's ``k_bresenham`` kernel through
``slam_bresenham_batch``; ``rasterize`` exposes the batched form.
"""
from __future__ import annotations

import numpy as np

from . import _abi


def rasterize(starts, ends, context=None):
    """Batch form: starts, ends [B,2] int -> list of B int32 arrays [len_b, 2]."""
    ctx = context or _abi.default_context()
    s = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1, 2)
    e = np.ascontiguousarray(ends, dtype=np.int32).reshape(-1, 2)
    B = s.shape[0]
    if B == 0:
        return []
    d = np.abs(e.astype(np.int64) - s.astype(np.int64))
    cap = d.max(axis=1) + 1
    offsets = np.zeros(B, dtype=np.int64)
    np.cumsum(cap[:-1], out=offsets[1:])
    total = int(cap.sum())
    lens = np.zeros(B, dtype=np.int32)
    cells = np.zeros((total, 2), dtype=np.int32)
    _abi.check(_abi.lib().slam_bresenham_batch(ctx.handle, _abi.ptr(s), _abi.ptr(e), B, _abi.ptr(offsets),
                                               _abi.ptr(lens), _abi.ptr(cells), total))
    return [cells[offsets[b]:offsets[b] + lens[b]] for b in range(B)]


class bresenham:  # noqa: N801 (the reference's class name)
    def __init__(self, start, end):
        self.start = start
        self.end = end
        self.path = []
        self.flag = 0
        if (start[0] == end[0]) and (start[1] == end[1]):
            return
        # bookkeeping attributes the reference leaves behind (bresenham.py:14-21)
        self.steep = abs(end[1] - start[1]) > abs(end[0] - start[0])
        a, b = (start[1], end[1]) if self.steep else (start[0], end[0])
        self.flag = 1 if a > b else 0
        cells = rasterize([[int(start[0]), int(start[1])]], [[int(end[0]), int(end[1])]])[0]
        self.path = [(int(x), int(y)) for x, y in cells]

    def swap(self, n1, n2):
        return [n2, n1]
