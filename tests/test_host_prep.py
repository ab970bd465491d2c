"""The package's Python host prep (rays, projection, AA tables, dtype casts, depth post-map)
reproduces, bit for bit on CPU, the 21 boundary arguments the reference's Python hands to
_C.render_forward_cuda (tests/golden/boundary_*.npz, captured from /root/reference with a stub _C)."""
import os

import numpy as np
import pytest
import torch

from util import ARG_NAMES, dm2, patched_C, scenes

LAYER_ARGS = ["width", "height", "verts", "faces", "tets", "face_tets", "tet_faces", "face_existence",
              "verts_ndc", "verts_image", "image_ray_o", "image_ray_d", "num_layers"]


def _same(a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b)
    return np.array_equal(a, b)


@pytest.mark.parametrize("name", ["boundary_full.npz", "boundary_patch.npz"])
def test_forward_boundary_args_match_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    W, H = int(g["width"]), int(g["height"])
    mv, proj = torch.from_numpy(g["in_mv"]), torch.from_numpy(g["in_proj"])
    batch_idx = g["batch_idx"].tolist()
    pw, ph = int(g["arg_patch_width"]), int(g["arg_patch_height"])
    got = {}

    def fake(*args):
        got["args"] = args
        e = torch.zeros(0)
        return 0, torch.from_numpy(g["stub_color"]), torch.from_numpy(g["stub_depth"]), e, e, e, e, e, e, e

    r = dm2.Renderer(mv, proj, W, H, "cpu", aa_grad_buffer_size=int(g["in_K"]))
    assert _same(r.ray_o.numpy(), g["full_ray_o"]) and _same(r.ray_d.numpy(), g["full_ray_d"])
    with patched_C(render_forward_cuda=fake):
        color, depth = r(batch_idx, torch.from_numpy(g["in_patch_min"]).long(), pw, ph,
                         torch.from_numpy(g["in_verts"]), torch.from_numpy(g["in_faces"]),
                         torch.from_numpy(g["in_verts_color"]), torch.from_numpy(g["in_faces_opacity"]),
                         torch.from_numpy(g["in_faces_intense"])[batch_idx], torch.from_numpy(g["in_background"]),
                         aa_temperature=float(g["arg_aa_temperature"]))
    assert len(got["args"]) == 21
    for k, a in zip(ARG_NAMES, got["args"]):
        ref = g["arg_" + k]
        if torch.is_tensor(a):
            assert _same(a.numpy(), ref), k
        else:
            assert type(a)(ref) == a, k
    # depth post-map 1 - (d+1)/2 (reference __init__.py:377-378); colour passes through
    assert _same(color.numpy(), g["out_color"]) and _same(depth.numpy(), g["out_depth"])


def test_layers_boundary_args_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "boundary_layers.npz"))
    W, H = int(g["arg_width"]), int(g["arg_height"])
    sc = scenes.tet_lattice(W, H, int(g["n"]), seed=int(g["seed"]), num_cams=2)
    got = {}

    def fake(*args):
        got["args"] = args
        return torch.zeros(1), torch.zeros(1)

    lr = dm2.LayeredRenderer(sc.mv, sc.proj, W, H, "cpu")
    with patched_C(generate_render_layers_cuda=fake):
        lr.generate(g["batch_idx"].tolist(), sc.verts, sc.faces, sc.tets, sc.face_tets, sc.tet_faces, sc.faces_existence, 3)
    assert len(got["args"]) == 13
    for k, a in zip(LAYER_ARGS, got["args"]):
        ref = g["arg_" + k]
        if torch.is_tensor(a):
            assert _same(a.numpy(), ref), k
        else:
            assert int(ref) == a, k


def test_autograd_routes_through_host_prep():
    """grad flows from the op's aa_face_verts / verts_ndc / verts inputs back to `verts` (incl. CCW un-permute)."""
    sc = scenes.triangle_soup(32, 32, 12, 5)
    verts = sc.verts.clone().requires_grad_(True)

    class FakeOp:
        @staticmethod
        def fwd(*args):
            B, H, W = args[8].shape[0], args[3], args[2]
            z = torch.zeros
            return 0, z((B, H, W, 3)), z((B, H, W)), z(0), z(0), z(0), z(0), z(0), z(0), z(0)

        @staticmethod
        def bwd(*args):
            verts_, vc, fo, ndc, fi, aa = args[5], args[7], args[8], args[9], args[11], args[13]
            return (torch.ones_like(verts_), torch.zeros_like(vc), torch.zeros_like(fo), torch.ones_like(ndc),
                    torch.zeros_like(fi), torch.ones_like(aa))

    r = dm2.Renderer(sc.mv, sc.proj, 32, 32, "cpu")
    with patched_C(render_forward_cuda=FakeOp.fwd, render_backward_cuda=FakeOp.bwd):
        color, depth = r([0], torch.zeros((1, 2), dtype=torch.int64), 32, 32, verts, sc.faces, sc.verts_color,
                         sc.faces_opacity, sc.faces_intense, sc.background)
        (color.sum() + depth.sum()).backward()
    assert verts.grad is not None and torch.isfinite(verts.grad).all() and verts.grad.abs().sum() > 0


@pytest.mark.parametrize("name", ["boundary_full.npz", "boundary_patch.npz"])
def test_analytic_rays_match_reference_rays(golden_dir, name):
    """The closed form the kernels evaluate per pixel under DM2_FLAG_ANALYTIC_RAYS (oracle.cpu.analytic_rays states it in
    numpy) against the rays the reference's Renderer._init_rays produced (committed fixtures): 1e-6, the rounding of a
    4x4 product whose summation order the reference leaves to its BLAS."""
    from oracle import cpu as orc
    g = np.load(os.path.join(golden_dir, name))
    ro, rd = orc.analytic_rays(g["in_mv"], g["in_proj"], int(g["width"]), int(g["height"]))
    assert ro.shape == g["full_ray_o"].shape and rd.shape == g["full_ray_d"].shape
    assert np.abs(ro - g["full_ray_o"]).max() <= 1e-6
    assert np.abs(rd - g["full_ray_d"]).max() <= 1e-6
