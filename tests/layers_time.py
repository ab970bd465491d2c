#!/usr/bin/env python3
"""Time LayeredRenderer.generate on BASELINE config 3 (1024x1024, 26^3 Kuhn lattice: T = 93 750 tets,
F = 191 250 faces, 4 layers) next to the CPU oracle on the same inputs.  `python tests/layers_time.py [--cpu]`."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dmesh2_renderer_amd as dm2  # noqa: E402
from dmesh2_renderer_amd import _C, scenes  # noqa: E402


def main():
    W = H = 1024
    L = 4
    sc = scenes.tet_lattice(W, H, 25, seed=scenes.SEED_BASE + 3)
    scd = sc.to("cuda")
    for fused in (False, True):
        lr = dm2.LayeredRenderer(scd.mv, scd.proj, W, H, "cuda", fused_prep=fused)

        def run():
            return lr.generate([0], scd.verts, scd.faces, scd.tets, scd.face_tets, scd.tet_faces, scd.faces_existence, L)

        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            out = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"LayeredRenderer.generate cfg3 (fused_prep={fused}): {ms:.3f} ms  ({W * H / ms / 1e3:.1f} Mpixels/s), "
              f"mean layers/pixel {out[1].float().mean().item():.2f}")
    _C.profile_enable(True)
    run()
    print("stages (ms):", {k: round(v, 4) for k, v in _C.profile_read().items()})
    _C.profile_enable(False)
    if "--cpu" in sys.argv:
        from oracle import cpu as orc
        lr = dm2.LayeredRenderer(scd.mv, scd.proj, W, H, "cuda")
        ndc, img = lr.compute_verts_ndc_image(scd.verts, scd.mv[[0]], scd.proj[[0]])
        a = [sc.verts.numpy(), sc.faces.numpy(), sc.tets.numpy(), sc.face_tets.numpy(), sc.tet_faces.numpy(),
             sc.faces_existence.numpy(), ndc.cpu().numpy(), img.cpu().numpy(), lr.ray_o[[0]].cpu().numpy(), lr.ray_d[[0]].cpu().numpy()]
        t0 = time.perf_counter()
        orc.generate_render_layers_cuda(W, H, *a, L, nthreads=orc.max_threads())
        print(f"CPU oracle ({orc.max_threads()} threads): {(time.perf_counter() - t0) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
