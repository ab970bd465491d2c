#!/usr/bin/env python3
"""Randomised sweep of LayeredRenderer.generate on the HIP path against the CPU oracle: random lattice sizes, jitter,
face-existence density, image sizes, cameras and layer counts; both tet walks (packed per-tet records -- the default -- and
the reference's access pattern under DM2_FLAG_LEGACY_KERNELS).  Face ids and counts must match exactly.
`python tests/fuzz_layers.py [seconds] [seed]`.  A development tool (needs a GPU); tests/test_gpu_parity.py::test_layers_exact
and tests/test_gpu_scale.py::test_cfg3_layered_renderer_full_size are the gate."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dmesh2_renderer_amd as dm2  # noqa: E402
from dmesh2_renderer_amd import _C, scenes  # noqa: E402
from oracle import cpu as orc  # noqa: E402


def one_case(seed, idx):
    rng = np.random.default_rng([seed, idx])
    W, H = int(rng.integers(8, 200)), int(rng.integers(8, 140))
    n = int(rng.integers(1, 8))
    L = int(rng.integers(1, 7))
    cams = int(rng.integers(1, 3))
    jitter = float(rng.choice([0.0, 0.1, 0.2, 0.35]))
    ex = float(rng.choice([0.05, 0.3, 0.7, 1.0]))
    sc = scenes.tet_lattice(W, H, n, seed=scenes.SEED_BASE + 5000 + idx + 7919 * seed, jitter=jitter, existence_p=ex, num_cams=cams)
    bidx = [int(b) for b in rng.integers(0, cams, size=int(rng.integers(1, 3)))]
    scd = sc.to("cuda")
    lr = dm2.LayeredRenderer(scd.mv, scd.proj, W, H, "cuda", fused_prep=False)
    ndc, img = lr.compute_verts_ndc_image(scd.verts, scd.mv[bidx], scd.proj[bidx])
    ro, rd = lr.ray_o[bidx], lr.ray_d[bidx]
    rl, rc = orc.generate_render_layers_cuda(W, H, sc.verts.numpy(), sc.faces.numpy(), sc.tets.numpy(), sc.face_tets.numpy(),
                                             sc.tet_faces.numpy(), sc.faces_existence.numpy(), ndc.cpu().numpy(), img.cpu().numpy(),
                                             ro.cpu().numpy(), rd.cpu().numpy(), L)[:2]
    ok = True
    for legacy in (0, _C.DM2_FLAG_LEGACY_KERNELS):
        old = _C.set_flags(legacy)
        try:
            layers, cnt = lr.generate(bidx, scd.verts, scd.faces, scd.tets, scd.face_tets, scd.tet_faces, scd.faces_existence, L)
        finally:
            _C.set_flags(old)
        ok = ok and np.array_equal(layers.cpu().numpy(), rl) and np.array_equal(cnt.cpu().numpy(), rc)
    return ok, dict(W=W, H=H, n=n, L=L, cams=cams, jitter=jitter, existence=ex, bidx=bidx, hit=float((rc > 0).mean()))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        ok, desc = one_case(seed, n)
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", dict(desc, seed=seed, idx=n - 1), flush=True)
        if n % 100 == 0:
            print(f"... {n} cases, {time.time() - t0:.0f} s", flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s, mismatches: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
