"""Landmark extraction from one scan (host side of the W12 node).

Mirrors ``Extraction`` / ``LandMarkSet`` of W12m/extraction.py:9-89: consecutive beams closer
than ``range_threshold`` form a cluster; a cluster of at least three beams whose largest
point-to-point distance stays below ``radius_max_th`` is a landmark, reported as the mean of
its points.  This is O(N) sequential segmentation of one 360-beam scan (SURVEY.md section 2
row 9: no data parallelism), so it stays on the host; the arithmetic that matters for parity
(gap norms, pairwise distances, means) is NumPy float64 as in the reference.

Behaviour kept from the reference, quirks included:
* only the first N-1 points are ever labelled (the loop runs over gaps, :36);
* the point that CLOSES a cluster (gap to its successor >= threshold) still belongs to it
  (:44-45), and a cluster needs two earlier members to be closed that way (:41);
* a lone point in front of a gap is labelled -1, but a single earlier member it leaves behind
  keeps the cluster number it was given (:56-61) - that number is never a landmark;
* a cluster still open at the end of the scan is never tested (no wrap-around);
* ``process`` returns ``None`` and sets ``flag`` when there is no landmark (:68-71).
"""
from __future__ import annotations

import numpy as np

from .param import get_param


class LandMarkSet:
    def __init__(self):
        self.position_x = []
        self.position_y = []
        self.id = []


class Extraction:
    def __init__(self):
        self.flag = 0
        self.range_threshold = get_param('/extraction/range_threshold', 1.0)   # same cluster (:19)
        self.radius_max_th = get_param('/extraction/radius_max_th', 0.3)       # landmark size (:21)
        self.landMark_min_pt = get_param('"/extraction/landMark_min_pt', 2)    # read, unused (:22, key typo kept)

    def labels(self, pc):
        """Per-point cluster labels [N-1] and the list of landmark cluster numbers."""
        xy = np.asarray(pc, dtype=np.float64)[:2, :]
        m = xy.shape[1]
        labels = np.empty(max(m - 1, 0), dtype=np.int64)
        landmarks = []
        if m < 2:
            return labels, landmarks
        step = xy[:, :-1] - xy[:, 1:]
        gap = np.sqrt(step[0] * step[0] + step[1] * step[1])      # np.linalg.norm of the 2-vector (:37)
        cluster, first = 0, 0                                     # members of the open cluster: first .. i-1
        for i in range(m - 1):
            members = i - first
            if gap[i] < self.range_threshold:
                labels[i] = cluster
                continue
            if members >= 2:
                labels[i] = cluster
                pts = xy[:, first:i + 1].T
                d = pts[:, None, :] - pts[None, :, :]
                extent = np.sqrt((d * d).sum(axis=2)).max()       # nanmax(squareform(pdist(.))) (:47-49)
                if extent < self.radius_max_th:
                    landmarks.append(cluster)
            else:
                labels[i] = -1
            cluster += 1
            first = i + 1
        return labels, landmarks

    def process(self, msg, trust=False):
        """msg: 3xN points of one scan (``laserToNumpy`` output) -> LandMarkSet or None."""
        pc = np.asarray(msg, dtype=np.float64)[:2, :]
        labels, found = self.labels(pc)
        if not found:
            self.flag = 1
            return None
        out = LandMarkSet()
        for c in found:                                           # ascending = order of first occurrence (:74-87)
            idx = np.nonzero(labels == c)[0]
            sx = sy = 0
            for k in idx:                                         # running sums in index order, as :81-83
                sx += pc[0][k]
                sy += pc[1][k]
            out.id.append(int(c))
            out.position_x.append(sx / len(idx))
            out.position_y.append(sy / len(idx))
        return out
