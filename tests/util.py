"""Shared helpers for the tests (CPU side)."""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import dmesh2_renderer_amd as dm2  # noqa: E402
from dmesh2_renderer_amd import _C, scenes  # noqa: E402

ARG_NAMES = [
    "background", "patch_min", "patch_width", "patch_height", "verts", "faces", "verts_color",
    "faces_opacity", "verts_ndc", "verts_image", "faces_intense", "aa_temperature", "aa_face_verts",
    "aa_face_edges", "aa_face_edges_iszero", "aa_face_edges_recip", "aa_face_edges_normal",
    "aa_face_edges_normal_c", "len_oarea_buffer", "image_ray_o", "image_ray_d"]


@contextlib.contextmanager
def patched_C(**fns):
    """Temporarily replace functions of dmesh2_renderer_amd._C (capture / fake backends)."""
    old = {k: getattr(_C, k) for k in fns}
    try:
        for k, v in fns.items():
            setattr(_C, k, v)
        yield
    finally:
        for k, v in old.items():
            setattr(_C, k, v)


def capture_forward_args(sc, batch_idx, patch_min, pw, ph, temp=1.0, K=20, device="cpu"):
    """Run the package's host prep (Renderer.forward) and return the 21 boundary args."""
    got = {}

    def fake(*args):
        got["args"] = args
        B = args[8].shape[0]
        z = torch.zeros
        return (0, z((B, ph, pw, 3), device=device), z((B, ph, pw), device=device), z(0), z(0), z(0), z(0), z(0), z(0), z(0))

    scd = sc.to(device)
    r = dm2.Renderer(scd.mv, scd.proj, sc.width, sc.height, device, aa_grad_buffer_size=K, tables_from_image=False)   # (the arguments are replayed as they are)
    with patched_C(render_forward_cuda=fake):
        r(batch_idx, torch.tensor(patch_min, dtype=torch.int64, device=device), pw, ph, scd.verts, scd.faces,
          scd.verts_color, scd.faces_opacity, scd.faces_intense[batch_idx], scd.background, aa_temperature=temp)
    return got["args"], r


def soup_args(W, H, F, seed, temp=1.0, K=20, cams=1, batch_idx=(0,), patch_min=None, pw=None, ph=None, shared=True,
              depth_complexity=4.0):
    sc = scenes.triangle_soup(W, H, F, seed, num_cams=cams, shared_verts=shared, depth_complexity=depth_complexity)
    batch_idx = list(batch_idx)
    if patch_min is None:
        patch_min = [[0, 0]] * len(batch_idx)
    return capture_forward_args(sc, batch_idx, patch_min, pw or W, ph or H, temp, K)[0], sc


def to_numpy_args(args):
    return [a.detach().cpu().numpy() if torch.is_tensor(a) else a for a in args]


def rel_linf(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-12)) if a.size else 0.0
