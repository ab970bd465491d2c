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
    stages = {k: round(v, 4) for k, v in _C.profile_read().items()}
    print("stages (ms):", stages)
    _C.profile_enable(False)
    if "--json" in sys.argv:
        # One line in the shape of bench.py's: `ms` is the last loop's (fused prep, the default).  Algorithmic HBM bytes of one
        # call ("LayeredRenderer.generate at cfg 3"): the inputs read once, the per-call intermediates written and
        # read once (face state, tile lists, 256-B tet records), the outputs written once.  The walk's ~10 record fetches per
        # pixel are L2 hits by design and not HBM bytes; what bounds the call is their latency, which `frac` makes visible.
        import json
        P, F, T, N = sc.verts.shape[0], sc.faces.shape[0], sc.tets.shape[0], W * H
        R = int(_C.generate_render_layers_cuda.last_debug[0])                # (tile, face) list entries of the call
        inputs = 12 * P + 12 * F + 16 * T + 8 * F + 16 * T + F + 20 * P + 24 * N
        interm = 2 * (28 * F + 12 * R + 256 * T + 8 * N)
        outputs = 4 * L * N + 4 * N
        alg = inputs + interm + outputs
        print(json.dumps({"metric": "Mpixels/s LayeredRenderer.generate @1024x1024, 93 750 tets, L=4 (BASELINE configs[2])",
                          "value": round(W * H / ms / 1e3, 1), "unit": "Mpixels/s", "ms_per_call": round(ms, 4), "n_gpus": 1,
                          "dtype": "f32", "data": "synthetic", "stage_ms": stages,
                          "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(alg / (ms * 1e-3) / 8e12, 4), "alg_bytes_per_call": alg, "traffic": None,
                                       "note": "latency bound: a dependent walk of ~10 steps per pixel over L2-resident records"}}))
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
