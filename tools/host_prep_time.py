#!/usr/bin/env python3
"""Time the Python host prep of Renderer.forward (projection + Triangles AA tables, reference
__init__.py:239-262,334-344 / pyrenderer.py:6-30) and its autograd backward on the bench workload,
next to the fused HIP prep (dmesh2_renderer_amd.prep) when the library exports it.
SURVEY.md §8(d): this time is reported separately from the op's metric."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dmesh2_renderer_amd as dm2  # noqa: E402
from dmesh2_renderer_amd import scenes  # noqa: E402
from dmesh2_renderer_amd.pyrenderer import Triangles  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    dev = torch.device("cuda", 0)
    W, H, F, ci = bench.CONFIGS[cfg]
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + ci).to(dev)
    r = dm2.Renderer(sc.mv, sc.proj, W, H, dev)
    mv, proj = r.mv[[0]], r.proj[[0]]
    verts = sc.verts.clone().requires_grad_(True)
    B = 1

    def torch_prep():
        ndc, image = r.compute_verts_ndc_image(verts, mv, proj)
        corners = image[:, sc.faces.flatten()].view(-1, 3, 2)
        tri = Triangles(corners[:, 0], corners[:, 1], corners[:, 2])
        return ndc, tri.verts.reshape(B, F, 3, 2), tri

    ndc, aav, _ = torch_prep()
    g_ndc = torch.randn_like(ndc)
    g_aa = torch.randn_like(aav)

    def torch_fwd():
        with torch.no_grad():
            torch_prep()

    def torch_fwd_bwd():
        verts.grad = None
        n, a, _ = torch_prep()
        torch.autograd.backward([n, a], [g_ndc, g_aa])

    print(f"{cfg}: P={verts.shape[0]} F={F}")
    print(f"torch host prep forward          {timed(torch_fwd):8.3f} ms")
    print(f"torch host prep forward+backward {timed(torch_fwd_bwd):8.3f} ms")
    try:
        from dmesh2_renderer_amd import prep
    except ImportError:
        return
    faces = sc.faces.to(torch.int32)

    def fused_fwd():
        with torch.no_grad():
            prep.prepare(verts, faces, mv, proj, W, H)

    def fused_fwd_bwd():
        verts.grad = None
        out = prep.prepare(verts, faces, mv, proj, W, H)
        torch.autograd.backward([out[0], out[2]], [g_ndc, g_aa])

    print(f"fused HIP prep forward           {timed(fused_fwd):8.3f} ms")
    print(f"fused HIP prep forward+backward  {timed(fused_fwd_bwd):8.3f} ms")


if __name__ == "__main__":
    main()
