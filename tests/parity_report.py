#!/usr/bin/env python3
"""Parity report of a library build against the CPU oracle on cfg2 (512x512, 50k triangles):
max relative L_inf per output / gradient and the flipped-pixel count (pixels whose last contributor or
AA record count differ).  Used to characterise non-default builds (e.g. -ffp-contract=fast) via
DM2_HIP_LIB=<path> python tests/parity_report.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from dmesh2_renderer_amd import _C  # noqa: E402
from oracle import cpu as orc  # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def main():
    dev = torch.device("cuda", 0)
    args, dLc, dLd, _ = bench.build_inputs("cfg2", dev, 0, 1)
    out = _C.render_forward_cuda(*args)
    g = _C.render_backward_cuda(out[0], *args, dLc, dLd, out[7], out[8], out[9], out[3], out[4], out[5], out[6])
    na = [a.detach().cpu().numpy() if torch.is_tensor(a) else a for a in args]
    ref = orc.render_forward_cuda(*na, nthreads=orc.max_threads())
    gref = orc.render_backward_cuda(ref, dLc.cpu().numpy(), dLd.cpu().numpy())
    B, H, W = ref.depth.shape
    N, Tn = B * H * W, B * ((W + 15) // 16) * ((H + 15) // 16)
    nc = _C.debug_fetch(4, N, Tn, out[0], out[9], torch.int32, N).cpu().numpy().view(np.uint32)
    print("lib:", _C.LIB_PATH)
    print("flipped pixels (n_contrib):", int((nc != ref.n_contrib).sum()), "of", N,
          "| (record count):", int((out[5].cpu().numpy() != ref.buf_tri_cnt).sum()))
    print("color rel L_inf %.3e   depth rel L_inf %.3e   bit-exact color: %s" % (
        rel(out[1].cpu().numpy(), ref.color), rel(out[2].cpu().numpy(), ref.depth),
        np.array_equal(out[1].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))))
    for name, x in zip(["verts", "verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"], g):
        print("grad %-14s rel L_inf %.3e" % (name, rel(x.cpu().numpy(), gref[name])))


if __name__ == "__main__":
    main()
