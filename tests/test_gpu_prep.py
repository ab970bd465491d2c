"""GPU parity of the fused host prep (dm2_prepare_faces / dm2_prepare_faces_backward, SURVEY.md §8(f)
rank 1) against the CPU oracle (oracle/dm2_oracle_prep.cpp, itself pinned by the reference's own vectors in
tests/test_oracle_prep.py), the reference-produced golden tensors, and torch autograd of the host twin.

Bars: forward bit-exact vs the oracle (same fp32 operation order, no FMA contraction); backward 1e-6
relative vs the oracle (only the atomic scatter order differs) and 1e-5 vs torch autograd."""
import os

import numpy as np
import pytest
import torch

from util import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ("verts_ndc", "verts_image", "verts", "edges", "iszero", "recip", "normal", "normal_c")


def _C():
    from dmesh2_renderer_amd import _C as c
    return c


def _orc():
    from oracle import cpu as orc
    return orc


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def mixed_orientation(sc):
    """The soup generator emits consistently oriented triangles; reverse every other face so that the CCW
    reorder (pyrenderer.py:521-529) and its un-permutation in the backward are exercised."""
    f = sc.faces.clone()
    f[1::2] = f[1::2][:, [0, 2, 1]]
    sc.faces = f
    return sc


def hip_prepare(verts, faces, mv, proj, W, H):
    outs = _C().prepare_faces(torch.as_tensor(verts).cuda(), torch.as_tensor(faces).cuda(), torch.as_tensor(mv).cuda(),
                              torch.as_tensor(proj).cuda(), W, H)
    torch.cuda.synchronize()
    return dict(zip(KEYS, [o.cpu().numpy() for o in outs]))


CASES = [
    dict(W=64, H=48, F=300, seed=41, cams=1),
    dict(W=100, H=70, F=1000, seed=42, cams=3),
    dict(W=128, H=128, F=5000, seed=43, cams=2, shared=True),
    dict(W=1920, H=1080, F=20000, seed=44, cams=1),
]


@pytest.mark.parametrize("c", CASES)
def test_prepare_bit_exact_vs_oracle(c):
    sc = mixed_orientation(scenes.triangle_soup(c["W"], c["H"], c["F"], scenes.SEED_BASE + c["seed"], num_cams=c["cams"],
                                                shared_verts=c.get("shared", False)))
    ref = _orc().prepare_faces(sc.verts, sc.faces, sc.mv, sc.proj, c["W"], c["H"])
    got = hip_prepare(sc.verts, sc.faces, sc.mv, sc.proj, c["W"], c["H"])
    for k in KEYS:
        assert np.array_equal(bits(got[k]), bits(ref[k])), k
    # both orientations occur, so the CCW reorder is exercised
    flipped = np.any(got["verts"][:, :, 1] != got["verts_image"][:, sc.faces.numpy()[:, 1]], axis=-1)
    assert flipped.any() and (~flipped).any()


def test_prepare_degenerate_and_clamped_inputs():
    """Zero-area faces, repeated vertices, axis-parallel edges (+-inf reciprocals) and |w| < 1e-4 vertices."""
    verts = np.array([[0.0, 0.0, 0.5], [0.3, 0.0, 0.5], [0.3, 0.2, 0.5], [0.0, 0.2, 0.5],      # axis-parallel quad at w = 0.5
                      [0.1, 0.1, 0.00005], [0.2, -0.1, -0.00005], [0.5, 0.5, 0.0],              # clamped w (both signs, zero)
                      [0.4, 0.4, 2.0]], np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3], [0, 2, 1], [0, 0, 1], [4, 5, 6], [1, 1, 1], [7, 4, 2]], np.int32)
    mv = np.eye(4, dtype=np.float32)[None]
    pr = np.array([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 0]]], np.float32)     # w = z
    ref = _orc().prepare_faces(verts, faces, mv, pr, 16, 12)
    got = hip_prepare(verts, faces, mv, pr, 16, 12)
    for k in KEYS:
        assert np.array_equal(bits(got[k]), bits(ref[k])), k
    assert np.isinf(got["recip"]).any() and got["iszero"].any()


def test_prepare_empty_inputs():
    C = _C()
    z3 = torch.zeros((0, 3), device="cuda")
    mv = torch.eye(4, device="cuda")[None].repeat(2, 1, 1)
    outs = C.prepare_faces(z3, torch.zeros((0, 3), dtype=torch.int32, device="cuda"), mv, mv, 8, 8)
    assert [tuple(o.shape) for o in outs[:3]] == [(2, 0, 3), (2, 0, 2), (2, 0, 3, 2)]
    g = C.prepare_faces_backward(z3, torch.zeros((0, 3), dtype=torch.int32, device="cuda"), mv, mv, 8, 8)
    assert tuple(g.shape) == (0, 3)
    with pytest.raises(RuntimeError):
        C.prepare_faces(torch.zeros((4, 3)), torch.zeros((1, 3), dtype=torch.int32), mv.cpu(), mv.cpu(), 8, 8)   # CPU tensors: no CPU path


@pytest.mark.parametrize("name", ["boundary_full.npz", "boundary_patch.npz"])
def test_prepare_vs_reference_vectors(name):
    """Against the tensors the reference's own Python produced (torch CPU): projection to 1e-6 (BLAS summation
    order), tables to the accuracy that implies, identical orientation / axis-parallel decisions."""
    g = np.load(os.path.join(GOLD, name))
    bi = g["batch_idx"].tolist()
    got = hip_prepare(g["in_verts"], g["in_faces"], g["in_mv"][bi], g["in_proj"][bi], int(g["width"]), int(g["height"]))
    assert rel(got["verts_ndc"], g["arg_verts_ndc"]) <= 1e-6
    assert rel(got["verts_image"], g["arg_verts_image"]) <= 1e-6
    assert rel(got["verts"], g["arg_aa_face_verts"]) <= 1e-6
    assert np.array_equal(got["iszero"], g["arg_aa_face_edges_iszero"])
    assert rel(got["edges"], g["arg_aa_face_edges"]) <= 1e-4
    assert rel(got["normal_c"], g["arg_aa_face_edges_normal_c"]) <= 1e-5


@pytest.mark.parametrize("c", CASES[:3])
def test_prepare_backward_vs_oracle_and_autograd(c):
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd.pyrenderer import Triangles
    W, H = c["W"], c["H"]
    sc = mixed_orientation(scenes.triangle_soup(W, H, c["F"], scenes.SEED_BASE + c["seed"], num_cams=c["cams"],
                                                shared_verts=c.get("shared", False)))
    B, P, F = c["cams"], sc.verts.shape[0], sc.faces.shape[0]
    gen = torch.Generator().manual_seed(c["seed"])
    g_ndc, g_img, g_aa = torch.randn((B, P, 3), generator=gen), torch.randn((B, P, 2), generator=gen), torch.randn((B, F, 3, 2), generator=gen)
    ref = _orc().prepare_faces_backward(sc.verts, sc.faces, sc.mv, sc.proj, W, H, g_ndc, g_img, g_aa)
    scd = sc.to("cuda")
    got = _C().prepare_faces_backward(scd.verts, scd.faces.to(torch.int32), scd.mv, scd.proj, W, H,
                                      g_ndc.cuda(), g_img.cuda(), g_aa.cuda()).cpu().numpy()
    assert rel(got, ref) <= 1e-6
    # single upstreams (NULL pointers for the others)
    for kw in (dict(g_verts_ndc=g_ndc.cuda()), dict(g_aa_face_verts=g_aa.cuda()), dict(g_verts_image=g_img.cuda())):
        okw = {"g_verts_ndc": "g_ndc", "g_aa_face_verts": "g_aa", "g_verts_image": "g_image"}
        r1 = _orc().prepare_faces_backward(sc.verts, sc.faces, sc.mv, sc.proj, W, H, **{okw[k]: v.cpu() for k, v in kw.items()})
        g1 = _C().prepare_faces_backward(scd.verts, scd.faces.to(torch.int32), scd.mv, scd.proj, W, H, **kw).cpu().numpy()
        assert rel(g1, r1) <= 1e-6
    # torch autograd through the reference-shaped host twin on the same device
    v = scd.verts.clone().requires_grad_(True)
    r = dm2.Renderer.__new__(dm2.Renderer)
    r.width, r.height = W, H
    ndc, image = dm2.Renderer.compute_verts_ndc_image(r, v, scd.mv, scd.proj)
    corners = image[:, scd.faces.flatten().long()].view(-1, 3, 2)
    tri = Triangles(corners[:, 0], corners[:, 1], corners[:, 2])
    torch.autograd.backward([ndc, image, tri.verts.reshape(B, F, 3, 2)], [g_ndc.cuda(), g_img.cuda(), g_aa.cuda()])
    assert rel(got, v.grad.cpu().numpy()) <= 1e-5


def test_renderer_fused_prep_end_to_end():
    """Renderer(fused_prep=True) against the reference-shaped torch prep: same image, same gradients."""
    import dmesh2_renderer_amd as dm2
    W, H, F = 96, 80, 800
    sc = mixed_orientation(scenes.triangle_soup(W, H, F, scenes.SEED_BASE + 45, num_cams=2, shared_verts=True)).to("cuda")
    pm = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
    gen = torch.Generator().manual_seed(9)
    gc, gd = torch.randn((2, H, W, 3), generator=gen).cuda(), torch.randn((2, H, W), generator=gen).cuda()
    res = []
    for fused in (False, True):
        r = dm2.Renderer(sc.mv, sc.proj, W, H, "cuda", fused_prep=fused)
        leaves = [t.clone().requires_grad_(True) for t in (sc.verts, sc.verts_color, sc.faces_opacity, sc.faces_intense)]
        color, depth = r([0, 1], pm, W, H, leaves[0], sc.faces, leaves[1], leaves[2], leaves[3], sc.background, aa_temperature=1.0)
        torch.autograd.backward([color, depth], [gc, gd])
        res.append((color.detach().cpu().numpy(), depth.detach().cpu().numpy(), [t.grad.cpu().numpy() for t in leaves]))
    (c0, d0, g0), (c1, d1, g1) = res
    assert np.abs(c0 - c1).max() <= 1e-4 and np.abs(d0 - d1).max() <= 1e-4
    for a, b in zip(g0, g1):
        assert rel(b, a) <= 1e-3      # a 1-ulp difference of verts_image moves AA areas by ~1e-6; gradients amplify it


@pytest.mark.parametrize("legacy", [False, True])
def test_fused_aa_gradient_routing(legacy):
    """With the fused prep the op returns its AA-corner gradients per VERTEX (DM2_FLAG_AA_GRAD_TO_VERTS: the gradient of
    verts_image, the CCW reorder undone from the packed record) instead of dL/d(aa_face_verts): same leaf gradients as the
    reference's route through the (B,F,3,2) tensor, on a scene with both orientations and shared vertices, on the dense
    and on the per-pixel-walk kernels."""
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import _C
    W, H, F = 112, 72, 900
    sc = mixed_orientation(scenes.triangle_soup(W, H, F, scenes.SEED_BASE + 47, num_cams=2, shared_verts=True)).to("cuda")
    pm = torch.tensor([[8, 4], [0, 0]], dtype=torch.int64, device="cuda")
    pw, ph = 96, 64
    gen = torch.Generator().manual_seed(11)
    gc, gd = torch.randn((2, ph, pw, 3), generator=gen).cuda(), torch.randn((2, ph, pw), generator=gen).cuda()
    res = []
    old_flags = _C.set_flags(_C.DM2_FLAG_LEGACY_KERNELS if legacy else 0)
    old_mode = dm2._FUSED_AA_GRAD
    try:
        for routed in (False, True):
            dm2._FUSED_AA_GRAD = routed
            r = dm2.Renderer(sc.mv, sc.proj, W, H, "cuda", fused_prep=True)
            leaves = [t.clone().requires_grad_(True) for t in (sc.verts, sc.verts_color, sc.faces_opacity, sc.faces_intense)]
            color, depth = r([1, 0], pm, pw, ph, leaves[0], sc.faces, leaves[1], leaves[2], leaves[3], sc.background, aa_temperature=1.0)
            torch.autograd.backward([color, depth], [gc, gd])
            res.append((color.detach().cpu().numpy(), [t.grad.cpu().numpy() for t in leaves]))
    finally:
        dm2._FUSED_AA_GRAD = old_mode
        _C.set_flags(old_flags)
    (c0, g0), (c1, g1) = res
    assert np.array_equal(c0, c1)
    assert np.abs(g0[0]).max() > 0
    for a, b in zip(g0, g1):
        assert rel(b, a) <= 1e-5


@pytest.mark.parametrize("legacy,temp", [(False, 1.0), (True, 1.0), (False, 0.0)])
def test_tables_from_image_matches_materialised_tables(legacy, temp):
    """The fused prep's default never materialises the six AA tables: the op's plan builds them per (view, face) from
    verts_image straight into its packed records (DM2_FLAG_TABLES_FROM_IMAGE).  Same bits in the image, same leaf gradients
    as with the tables written by dm2_prepare_faces and read back -- both orientations, shared vertices, two views, patches."""
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import _C
    W, H, F = 112, 72, 900
    sc = mixed_orientation(scenes.triangle_soup(W, H, F, scenes.SEED_BASE + 48, num_cams=2, shared_verts=True)).to("cuda")
    pm = torch.tensor([[8, 4], [0, 0]], dtype=torch.int64, device="cuda")
    pw, ph = 96, 64
    gen = torch.Generator().manual_seed(12)
    gc, gd = torch.randn((2, ph, pw, 3), generator=gen).cuda(), torch.randn((2, ph, pw), generator=gen).cuda()
    res, seen = [], []
    old_flags = _C.set_flags(_C.DM2_FLAG_LEGACY_KERNELS if legacy else 0)
    old_mode = dm2._TABLES_FROM_IMAGE
    real = _C.render_forward_cuda

    def spy(*args):
        seen.append(tuple(args[12].shape))
        return real(*args)

    _C.render_forward_cuda = spy
    try:
        for from_image in (False, True):
            dm2._TABLES_FROM_IMAGE = from_image
            r = dm2.Renderer(sc.mv, sc.proj, W, H, "cuda", fused_prep=True)
            leaves = [t.clone().requires_grad_(True) for t in (sc.verts, sc.verts_color, sc.faces_opacity, sc.faces_intense)]
            color, depth = r([1, 0], pm, pw, ph, leaves[0], sc.faces, leaves[1], leaves[2], leaves[3], sc.background, aa_temperature=temp)
            torch.autograd.backward([color, depth], [gc, gd])
            res.append((color.detach().cpu().numpy(), depth.detach().cpu().numpy(), [t.grad.cpu().numpy() for t in leaves]))
    finally:
        dm2._TABLES_FROM_IMAGE = old_mode
        _C.set_flags(old_flags)
        _C.render_forward_cuda = real
    assert seen == [(2, F, 3, 2), (2, 0, 3, 2)]                 # tables handed over, then placeholders
    (c0, d0, g0), (c1, d1, g1) = res
    assert np.array_equal(c0, c1) and np.array_equal(d0, d1)
    assert np.abs(g0[0]).max() > 0
    for a, b in zip(g0, g1):
        assert rel(b, a) <= 1e-5


def test_layered_renderer_fused_projection():
    import dmesh2_renderer_amd as dm2
    sc = scenes.tet_lattice(64, 64, 4, scenes.SEED_BASE + 46).to("cuda")
    outs = []
    for fused in (False, True):
        lr = dm2.LayeredRenderer(sc.mv, sc.proj, 64, 64, "cuda", fused_prep=fused)
        outs.append(lr.generate([0], sc.verts, sc.faces, sc.tets, sc.face_tets, sc.tet_faces, sc.faces_existence, 3))
    # identical layers wherever the projection's last bit does not decide a tile/depth tie (everywhere, on this lattice)
    assert (outs[0][1] == outs[1][1]).float().mean().item() >= 0.999
    assert (outs[0][0] == outs[1][0]).float().mean().item() >= 0.999
