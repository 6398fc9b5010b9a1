#!/usr/bin/env python3
"""CPU model of the beam-window search's lane efficiency in k_icp (csrc/icp_kernels.hip, nn_polar::scan): every lane of
a wave runs as many 4-candidate trips as the wave's WIDEST window needs.  The benchmark replay (configs[1]) is solved
with exhaustive nearest neighbours in NumPy, the window of every query in every iteration is computed with the kernel's
own float32 formula (tests/test_polar_window_bound.py::window), and the trips are counted for the shipped layout
(query i -> lane i % threads, slot i // threads) and for queries dealt to lanes in order of window width.

usage: icp_lane_model.py [pairs=60] [threads=192]        (no GPU needed)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

from test_polar_window_bound import window, f32

PKG = "a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd"
AMIN, AMAX = -3.14159, 3.14159


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    syn = importlib.import_module(PKG + ".synthetic")
    n = 360
    rep = syn.make_replay(1000, n, seed=1, stride=5)
    ang = np.linspace(AMIN, AMAX, n)
    ct, st = np.cos(ang), np.sin(ang)
    cr = ct[:-1] * st[1:] - st[:-1] * ct[1:]
    inv_db = f32(f32(1.000002) / f32(f32(cr.min()) * f32(0.999999)))
    rng = np.random.default_rng(0)
    ks = np.sort(rng.choice(np.arange(1, rep.ranges.shape[0]), size=pairs, replace=False))
    tot = {"own": 0, "wave": 0, "wave_sorted": 0, "wave_bins": 0, "cand": 0, "big": 0, "q": 0, "iters": 0}
    widths_all = []
    for k in ks:
        rt = rep.ranges[k - 1].astype(np.float64)
        rs = rep.ranges[k].astype(np.float64)
        tx, ty = ct * rt, st * rt
        sx, sy = ct * rs, st * rs
        seed = np.arange(n)
        pre = 0.0
        for it in range(30):
            d2 = (sx[:, None] - tx[None, :]) ** 2 + (sy[:, None] - ty[None, :]) ** 2
            j = np.argmin(d2, axis=1)
            if it >= 1:
                small, lo, hi = window(sx, sy, tx, ty, seed, inv_db, 2e-7, 0.0)
                w = (hi - lo + 1)
                fits = small & (hi - lo < 96)
                tot["big"] += int((~fits).sum())
                tot["q"] += n
                tot["iters"] += 1
                trips = np.where(fits, (np.minimum(hi, n - 1) - np.maximum(lo, 0)) // 4 + 1, 0)
                tot["cand"] += int(np.where(fits, np.minimum(hi, n - 1) - np.maximum(lo, 0) + 1, 0).sum())
                widths_all.append(w[fits])
                tot["own"] += int(trips.sum())
                # shipped layout
                lane = np.arange(n) % threads
                slot = np.arange(n) // threads
                wave = lane // 64
                for s_ in range(slot.max() + 1):
                    for wv in range(threads // 64):
                        m = (slot == s_) & (wave == wv)
                        if m.any():
                            tot["wave"] += int(trips[m].max())
                # dealt in order of trips (a full sort)
                order = np.argsort(-trips, kind="stable")
                ts = trips[order]
                for a in range(0, n, 64):
                    tot["wave_sorted"] += int(ts[a:a + 64].max())
                # partition only: queries with more than `thr` trips go to the end of the position range
                for thr in (1, 2):
                    wide = trips > thr
                    order_p = np.concatenate([np.flatnonzero(~wide), np.flatnonzero(wide)])
                    tp = trips[order_p]
                    tot_p = 0
                    lane_p = np.arange(n) % threads; slot_p = np.arange(n) // threads; wave_p = lane_p // 64
                    for s_ in range(slot_p.max() + 1):
                        for wv in range(threads // 64):
                            m_ = (slot_p == s_) & (wave_p == wv)
                            if m_.any(): tot_p += int(tp[m_].max())
                    tot["part%d" % thr] = tot.get("part%d" % thr, 0) + tot_p
                # counting sort into 4 bins by trips (1, 2, 3-4, 5+), dealt consecutively
                b = np.digitize(trips, [2, 3, 5])
                order = np.argsort(-b, kind="stable")
                tb = trips[order]
                for a in range(0, n, 64):
                    tot["wave_bins"] += int(tb[a:a + 64].max())
            # Kabsch update with the matches
            mx, my = tx[j], ty[j]
            ca, cb = np.array([sx.mean(), sy.mean()]), np.array([mx.mean(), my.mean()])
            A_ = np.stack([sx - ca[0], sy - ca[1]])
            B_ = np.stack([mx - cb[0], my - cb[1]])
            W = B_ @ A_.T
            th = np.arctan2(W[1, 0] - W[0, 1], W[0, 0] + W[1, 1])
            c, s = np.cos(th), np.sin(th)
            t = cb - np.array([c * ca[0] - s * ca[1], s * ca[0] + c * ca[1]])
            sx, sy = c * sx - s * sy + t[0], s * sx + c * sy + t[1]
            seed = j
            err = float(np.sqrt(d2[np.arange(n), j]).mean())
            if abs(pre - err) < 1e-3:
                break
            pre = err
    w = np.concatenate(widths_all)
    print("pairs %d, later iterations %d, queries %d, without a window %d (%.2f %%)" % (pairs, tot["iters"], tot["q"], tot["big"], 100.0 * tot["big"] / tot["q"]))
    print("window width: median %d, mean %.1f, 90 %% %d, 99 %% %d, max %d; candidates per query %.2f" % (np.median(w), w.mean(), np.percentile(w, 90), np.percentile(w, 99), w.max(), tot["cand"] / tot["q"]))
    print("trips per query (own windows) %.2f" % (tot["own"] / tot["q"]))
    for name in ("wave", "wave_sorted", "wave_bins", "part1", "part2"):
        print("%-12s wave-trips per iteration %.1f, lane efficiency %.3f" % (name, tot[name] / tot["iters"], tot["own"] / (64.0 * tot[name])))


if __name__ == "__main__":
    main()
