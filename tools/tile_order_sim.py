#!/usr/bin/env python3
"""Model behind k_tile_order (dmesh2_renderer_amd/csrc/dm2_binning.hip): list scheduling of a frame's tiles on the resident
workgroup slots of one XCD (128 = 32 CUs x 4 blocks), tile cost = list length (+ a fixed overhead), in index order against
longest-list-first.  CPU only: the bench's synthetic scene, tile lists counted with numpy.
usage: python tools/tile_order_sim.py [cfg4]"""
import heapq
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def makespan(costs, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    args, _, _, _ = bench.build_inputs(cfg, torch.device("cpu"), 0, 1)
    W, H = int(args[2]), int(args[3])
    faces, vi = args[5].numpy(), args[9][0].numpy()
    p = vi[faces]
    mn, mx = p.min(1), p.max(1)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x0 = np.clip(np.floor(mn[:, 0] / 16), 0, gx).astype(int); x1 = np.clip(np.ceil(mx[:, 0] / 16), 0, gx).astype(int)
    y0 = np.clip(np.floor(mn[:, 1] / 16), 0, gy).astype(int); y1 = np.clip(np.ceil(mx[:, 1] / 16), 0, gy).astype(int)
    cnt = np.zeros((gy, gx), np.int64)
    span = int(max((x1 - x0).max(), (y1 - y0).max()))
    for dx in range(span):
        for dy in range(span):
            m = (x0 + dx < x1) & (y0 + dy < y1)
            np.add.at(cnt, (y0[m] + dy, x0[m] + dx), 1)
    c = cnt.ravel()
    per = (c.size + 7) // 8
    print(f"{cfg}: {c.size} tiles, {c.sum()} entries, list length {c.mean():.1f} +- {c.std():.1f} (min {c.min()}, max {c.max()})")
    for ovh in (0, 30):
        cur = lpt = ideal = 0.0
        for x in range(8):
            seg = c[x * per:(x + 1) * per] + ovh
            cur = max(cur, makespan(list(seg), 128)); lpt = max(lpt, makespan(sorted(seg, reverse=True), 128))
            ideal = max(ideal, seg.sum() / 128)
        print(f"  fixed cost {ovh}: index order {cur / ideal:.3f} x the mean load, longest first {lpt / ideal:.3f} x")


if __name__ == "__main__":
    main()
