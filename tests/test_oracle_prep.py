"""Oracle restatement of the Python host prep (projection + AA tables) pinned
against vectors the reference's own Python produced (tests/golden/boundary_*.npz, make_golden.py), and its
backward against torch autograd of the host twin and fp64 finite differences."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu as orc
import dmesh2_renderer_amd as dm2
from dmesh2_renderer_amd import scenes
from dmesh2_renderer_amd.pyrenderer import Triangles

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


@pytest.mark.parametrize("name", ["boundary_full.npz", "boundary_patch.npz"])
def test_prepare_matches_reference_vectors(name):
    g = np.load(os.path.join(GOLD, name))
    bi = g["batch_idx"].tolist()
    W, H = int(g["width"]), int(g["height"])
    out = orc.prepare_faces(g["in_verts"], g["in_faces"], g["in_mv"][bi], g["in_proj"][bi], W, H)
    # projection: a 4x4 matmul, the BLAS summation order of the reference run is not defined -> 1e-6 relative
    assert rel(out["verts_ndc"], g["arg_verts_ndc"]) <= 1e-6
    assert rel(out["verts_image"], g["arg_verts_image"]) <= 1e-6
    # tables: pure element-wise functions of verts_image -> bit-exact when fed the reference's own verts_image
    tv = g["arg_verts_image"][:, g["in_faces"].reshape(-1)].reshape(len(bi), -1, 3, 2)
    t = orc.aa_tables(tv)
    for key, gk in (("verts", "arg_aa_face_verts"), ("edges", "arg_aa_face_edges"), ("recip", "arg_aa_face_edges_recip"),
                    ("normal", "arg_aa_face_edges_normal"), ("normal_c", "arg_aa_face_edges_normal_c")):
        assert np.array_equal(t[key].view(np.uint32), g[gk].view(np.uint32)), key
    assert np.array_equal(t["iszero"], g["arg_aa_face_edges_iszero"])
    # and the fused restatement agrees with the reference's tables to projection accuracy (same CCW decisions)
    assert np.array_equal(out["iszero"], g["arg_aa_face_edges_iszero"])
    assert rel(out["verts"], g["arg_aa_face_verts"]) <= 1e-6
    assert rel(out["edges"], g["arg_aa_face_edges"]) <= 1e-4          # differences of nearby pixel coordinates
    assert rel(out["normal_c"], g["arg_aa_face_edges_normal_c"]) <= 1e-5


def test_prepare_tables_are_functions_of_its_own_image():
    """The fused restatement's tables are exactly aa_tables(its verts_image): one definition, two code paths."""
    sc = scenes.triangle_soup(64, 48, 300, scenes.SEED_BASE + 31, num_cams=2)
    sc.faces[1::2] = sc.faces[1::2][:, [0, 2, 1]]
    out = orc.prepare_faces(sc.verts, sc.faces, sc.mv, sc.proj, 64, 48)
    tv = out["verts_image"][:, sc.faces.numpy().reshape(-1)].reshape(2, -1, 3, 2)
    t = orc.aa_tables(tv)
    for key in ("verts", "edges", "recip", "normal", "normal_c"):
        assert np.array_equal(t[key].view(np.uint32), out[key].view(np.uint32)), key
    assert np.array_equal(t["iszero"], out["iszero"])


def _torch_host_prep(verts, faces, mv, proj, W, H):
    r = dm2.Renderer.__new__(dm2.Renderer)
    r.width, r.height = W, H
    ndc, image = dm2.Renderer.compute_verts_ndc_image(r, verts, mv, proj)
    corners = image[:, faces.flatten().long()].view(-1, 3, 2)
    tri = Triangles(corners[:, 0], corners[:, 1], corners[:, 2])
    return ndc, image, tri.verts.reshape(mv.shape[0], -1, 3, 2)


def test_prepare_backward_matches_torch_autograd():
    W, H = 80, 60
    sc = scenes.triangle_soup(W, H, 400, scenes.SEED_BASE + 32, num_cams=3, shared_verts=True)
    sc.faces[1::2] = sc.faces[1::2][:, [0, 2, 1]]          # both orientations: the CCW reorder must be un-permuted
    verts = sc.verts.clone().requires_grad_(True)
    ndc, image, aav = _torch_host_prep(verts, sc.faces, sc.mv, sc.proj, W, H)
    g = torch.Generator().manual_seed(5)
    g_ndc, g_img, g_aa = (torch.randn(x.shape, generator=g) for x in (ndc, image, aav))
    torch.autograd.backward([ndc, image, aav], [g_ndc, g_img, g_aa])
    got = orc.prepare_faces_backward(sc.verts, sc.faces, sc.mv, sc.proj, W, H, g_ndc, g_img, g_aa)
    assert rel(got, verts.grad.numpy()) <= 1e-5
    # each upstream alone (null pointers for the others)
    for kw, outs, gs in ((dict(g_ndc=g_ndc), [ndc], [g_ndc]), (dict(g_aa=g_aa), [aav], [g_aa])):
        v2 = sc.verts.clone().requires_grad_(True)
        o = _torch_host_prep(v2, sc.faces, sc.mv, sc.proj, W, H)
        sel = [o[0]] if "g_ndc" in kw else [o[2]]
        torch.autograd.backward(sel, gs)
        got = orc.prepare_faces_backward(sc.verts, sc.faces, sc.mv, sc.proj, W, H, **kw)
        assert rel(got, v2.grad.numpy()) <= 1e-5


def test_prepare_backward_finite_differences_fp64():
    W, H = 40, 30
    sc = scenes.triangle_soup(W, H, 30, scenes.SEED_BASE + 33, num_cams=2)
    v = sc.verts.numpy().astype(np.float64)
    f = sc.faces.numpy().copy()
    f[1::2] = f[1::2][:, [0, 2, 1]]
    mv, pr = sc.mv.numpy().astype(np.float64), sc.proj.numpy().astype(np.float64)
    rng = np.random.default_rng(3)
    o = orc.prepare_faces(v, f, mv, pr, W, H, dtype=np.float64)
    g_ndc, g_aa = rng.standard_normal(o["verts_ndc"].shape), rng.standard_normal(o["verts"].shape)
    ana = orc.prepare_faces_backward(v, f, mv, pr, W, H, g_ndc=g_ndc, g_aa=g_aa, dtype=np.float64)

    def loss(vv):
        oo = orc.prepare_faces(vv, f, mv, pr, W, H, dtype=np.float64)
        return float((oo["verts_ndc"] * g_ndc).sum() + (oo["verts"] * g_aa).sum())

    h = 1e-6
    idx = rng.choice(v.size, size=24, replace=False)
    for i in idx:
        vp, vm = v.copy().reshape(-1), v.copy().reshape(-1)
        vp[i] += h; vm[i] -= h
        fd = (loss(vp.reshape(v.shape)) - loss(vm.reshape(v.shape))) / (2 * h)
        assert abs(fd - ana.reshape(-1)[i]) <= 1e-6 * max(1.0, abs(fd)), (i, fd, ana.reshape(-1)[i])


def test_w_clamp_blocks_gradient():
    """|w| < 1e-4 is clamped with the sign kept (__init__.py:254-255) and passes no gradient into w."""
    verts = np.array([[0.0, 0.0, 0.0], [0.3, 0.2, 0.00005], [0.1, -0.2, -0.00005]], np.float32)
    faces = np.array([[0, 1, 2]], np.int32)
    mv = np.eye(4, dtype=np.float32)[None]
    pr = np.array([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 0]]], np.float32)     # w = z
    o = orc.prepare_faces(verts, faces, mv, pr, 8, 8)
    assert np.allclose(o["verts_ndc"][0, 1, 0], 0.3 / 1e-4, rtol=1e-6) and np.allclose(o["verts_ndc"][0, 2, 0], 0.1 / -1e-4, rtol=1e-6)
    vt = torch.from_numpy(verts).requires_grad_(True)
    ndc, image, aav = _torch_host_prep(vt, torch.from_numpy(faces), torch.from_numpy(mv), torch.from_numpy(pr), 8, 8)
    g_ndc = torch.ones_like(ndc)
    ndc.backward(g_ndc)
    got = orc.prepare_faces_backward(verts, faces, mv, pr, 8, 8, g_ndc=g_ndc)
    assert rel(got, vt.grad.numpy()) <= 1e-6
