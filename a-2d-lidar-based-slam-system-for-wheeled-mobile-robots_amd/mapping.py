"""``Mapping`` occupancy-grid accumulator on the GPU.

Mirrors W12m/mapping.py:8-51: ``Mapping(xw, yw, xyreso)`` holds ``pmap`` (50 = unknown,
0 = free, 100 = occupied) and ``datamap``; ``update(ox, oy, center_x, center_y)`` casts one
ray per beam from the centre to each world-frame endpoint and returns the live ``pmap``.
The rays are walked by libslamhip's ``k_grid_update`` kernel (float-error Bresenham,
integer pass / hit counters) through ``slam_grid_update``; ``pmap`` comes from
``slam_grid_read``.

Fidelity notes:
* the reference converts world coordinates with ``int(10 * (x + 10))`` whatever ``xyreso``
  is (mapping.py:33-36).  That is the default here; pass ``index_scale`` / ``index_offset``
  (or use :meth:`Mapping.metric`) for grids of another resolution such as 400x400 @ 0.05 m.
* ``datamap`` is rebuilt as ``0.01*pass + 20*hit`` from the counters; it equals the
  reference's running float sum to ~1e-12, ``pmap`` is exact.
* ``hit_inc=4`` gives the w12-mapping-online variant (W12o/mapping.py:46).  There the
  reference's own ``pmap`` depends on the order in which a cell's +4 and +0.01 arrived when
  the pass count sits exactly on a threshold (e.g. 2 hits + 200 passes: 8 + 200 x 0.01 is
  occupied if the passes came first, free if the hits did); the device applies the
  hits-first order (include/slam_hip.h, slam_grid_create).
* NaN coordinates raise ``ValueError`` and infinite ``oy`` / centre raise ``OverflowError``
  as ``int()`` does in the reference.  The reference stops at the offending beam (earlier
  beams are already in the map, later ones are not); so does ``update``.  A cell index
  beyond 2^20 also raises ``OverflowError`` (the reference would walk that ray for hours).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


class Mapping:
    def __init__(self, xw, yw, xyreso, index_scale=10.0, index_offset=10.0, index_offset_y=None,
                 free_inc=0.01, hit_inc=20.0, thresh=10.0, context=None):
        self.width_x = xw * xyreso
        self.width_y = yw * xyreso
        self.xyreso = xyreso
        self.xw = int(xw)
        self.yw = int(yw)
        self.pmap = 50 * np.ones((self.xw, self.yw))   # default 50: unknown (mapping.py:14)
        self._datamap = np.zeros((self.xw, self.yw))   # mapping.py:15, fetched on demand (see ``datamap``)
        self._datamap_stale = False
        self.minx = -self.width_x / 2.0
        self.maxx = self.width_x / 2.0
        self.miny = -self.width_y / 2.0
        self.maxy = self.width_y / 2.0
        self.index_scale = float(index_scale)
        self.index_offset_x = float(index_offset)
        self.index_offset_y = float(index_offset if index_offset_y is None else index_offset_y)
        self._ctx = context or _abi.default_context()
        h = C.c_void_p()
        _abi.check(_abi.lib().slam_grid_create(self._ctx.handle, 1, self.xw, self.yw, self.index_scale,
                                               self.index_offset_x, self.index_offset_y, float(free_inc),
                                               float(hit_inc), float(thresh), C.byref(h)))
        self._grid = h
        self._p8 = np.empty((self.xw, self.yw), dtype=np.int8)
        # one scan per call owns the map: the ray cast keeps pmap current on the device itself
        live = C.c_void_p()
        _abi.check(_abi.lib().slam_grid_live_pmap(self._ctx.handle, self._grid, C.byref(live)))

    @property
    def datamap(self):
        """The evidence map of mapping.py:15 (float64 [xw, yw]).  ``update`` only brings ``pmap``
        back from the device; this view (8 bytes per cell) is read when it is asked for."""
        if self._datamap_stale:
            _abi.check(_abi.lib().slam_grid_read(self._ctx.handle, self._grid, 0, None, _abi.ptr(self._datamap), None, None))
            self._datamap_stale = False
        return self._datamap

    def _fetch_pmap(self):
        _abi.check(_abi.lib().slam_grid_read(self._ctx.handle, self._grid, 0, _abi.ptr(self._p8), None, None, None))
        self.pmap[...] = self._p8
        self._datamap_stale = True
        return self.pmap

    @classmethod
    def metric(cls, xw, yw, xyreso, **kw):
        """Grid whose index rule follows its resolution: scale = 1/xyreso, offsets = half
        the map width (the reference's rule is the xyreso = 0.1, 20 m special case)."""
        scale = round(1.0 / xyreso) if abs(round(1.0 / xyreso) - 1.0 / xyreso) < 1e-9 else 1.0 / xyreso
        return cls(xw, yw, xyreso, index_scale=scale, index_offset=xw / (2.0 * scale),
                   index_offset_y=yw / (2.0 * scale), **kw)

    def __del__(self):
        try:
            if getattr(self, "_grid", None) is not None and self._ctx._h is not None:
                _abi.lib().slam_grid_destroy(self._ctx.handle, self._grid)
            self._grid = None
        except Exception:
            pass

    def update(self, ox, oy, center_x, center_y):
        ox = np.ascontiguousarray(np.asarray(ox, dtype=np.float64).reshape(-1))
        oy = np.ascontiguousarray(np.asarray(oy, dtype=np.float64).reshape(-1))
        n = len(ox)
        if len(oy) < n:
            raise IndexError("oy is shorter than ox")
        if n == 0:
            return self.pmap
        cx = np.array([float(np.asarray(center_x).reshape(-1)[0])])
        cy = np.array([float(np.asarray(center_y).reshape(-1)[0])])
        L = _abi.lib()
        try:
            _abi.check(L.slam_grid_update(self._ctx.handle, self._grid, _abi.ptr(ox), _abi.ptr(oy), _abi.ptr(cx),
                                          _abi.ptr(cy), 1, n, None))
        finally:
            self._datamap_stale = True            # on a NaN / inf beam the beams before it were still applied
        return self._fetch_pmap()

    def update_scans(self, ranges, angle_min, angle_max, poses, centres=None):
        """S raw scans at once (``slam_grid_update_scans``): ranges [S, n] (inf -> 30 m), poses
        [S, 3] the xEst each scan is transformed with (W12m/slam_ekf.py:89), centres [S, 2] the
        ray origins (default: the pose; w12-mapping-online passes the /tf position,
        W12o/slam_ekf.py:104).  Returns the live ``pmap``."""
        r = np.ascontiguousarray(np.asarray(ranges, dtype=np.float32))
        if r.ndim == 1:
            r = r[None]
        S, n = r.shape
        p = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(S, 3))
        c = None if centres is None else np.ascontiguousarray(np.asarray(centres, dtype=np.float64).reshape(S, 2))
        ct, st = _abi.trig_tables(angle_min, angle_max, n)
        L = _abi.lib()
        try:
            _abi.check(L.slam_grid_update_scans(self._ctx.handle, self._grid, _abi.ptr(r), _abi.ptr(ct), _abi.ptr(st),
                                                _abi.ptr(p), _abi.ptr(c), S, n))
        finally:
            self._datamap_stale = True
        return self._fetch_pmap()

    def counters(self):
        """(pass, hit) uint32 [xw, yw]: the integer evidence behind ``datamap``."""
        p = np.empty((self.xw, self.yw), dtype=np.uint32)
        h = np.empty((self.xw, self.yw), dtype=np.uint32)
        _abi.check(_abi.lib().slam_grid_read(self._ctx.handle, self._grid, 0, None, None, _abi.ptr(p), _abi.ptr(h)))
        return p, h

    def occupancy_grid_data(self):
        """int8 [yw*xw] in the OccupancyGrid layout of publishMap
        (W12m/slam_ekf.py:270-271): data[y*width + x] = pmap[x][y]."""
        out = np.empty(self.xw * self.yw, dtype=np.int8)
        _abi.check(_abi.lib().slam_grid_occupancy_data(self._ctx.handle, self._grid, 0, _abi.ptr(out)))
        return out

    def visits(self):
        v = C.c_uint64(0)
        _abi.check(_abi.lib().slam_grid_visits(self._ctx.handle, self._grid, C.byref(v)))
        return int(v.value)
