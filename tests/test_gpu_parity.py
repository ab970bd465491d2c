"""GPU parity: the HIP path (through the C ABI, via dmesh2_renderer_amd._C) against the CPU oracle
on the same seeded inputs.

Bars: forward outputs and integer state bit-exact (both sides evaluate the same fp32 operation
sequence without FMA contraction); gradients within 1e-5 relative L-inf (north_star tolerance; fp32
atomic summation order differs, as it does run-to-run in the reference itself)."""
import os

import numpy as np
import pytest
import torch

from util import ARG_NAMES, rel_linf, scenes, soup_args, to_numpy_args

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-5


def _C():
    from dmesh2_renderer_amd import _C as c
    return c


@pytest.fixture(autouse=True, params=["dense", "legacy"])
def kernels(request):
    """Every test runs against both work distributions of the composite kernels (same results required)."""
    c = _C()
    old = c.set_flags(c.DM2_FLAG_LEGACY_KERNELS if request.param == "legacy" else 0)
    yield request.param
    c.set_flags(old)


def _orc():
    from oracle import cpu as orc
    return orc


def to_dev(args, dev="cuda"):
    return [a.to(dev) if torch.is_tensor(a) else a for a in args]


def run_both(args, seed=0, backward=True):
    C, orc = _C(), _orc()
    dargs = to_dev(args)
    out = C.render_forward_cuda(*dargs)
    ref = orc.render_forward_cuda(*to_numpy_args(args))
    res = dict(out=out, ref=ref)
    if backward:
        rng = np.random.RandomState(seed)
        gc = rng.randn(*ref.color.shape).astype(np.float32)
        gd = rng.randn(*ref.depth.shape).astype(np.float32)
        grads = C.render_backward_cuda(out[0], *dargs, torch.from_numpy(gc).cuda(), torch.from_numpy(gd).cuda(),
                                       out[7], out[8], out[9], out[3], out[4], out[5], out[6])
        res["grads"] = [g.cpu().numpy() for g in grads]
        res["ref_grads"] = orc.render_backward_cuda(ref, gc, gd)
    torch.cuda.synchronize()
    return res


def check_forward(res, args):
    C = _C()
    out, ref = res["out"], res["ref"]
    R, color, depth, oarea, tri_id, tri_cnt, doarea, face_buf, bin_buf, img_buf = out
    B, H, W = ref.depth.shape
    assert R == ref.num_rendered
    assert color.shape == (B, H, W, 3) and depth.shape == (B, H, W) and tri_cnt.shape == (B, H, W)
    assert oarea.dim() == 4 and tri_id.dim() == 4 and doarea.dim() == 6
    N, Tn = B * H * W, B * ((W + 15) // 16) * ((H + 15) // 16)
    if R > 0:
        ranges = C.debug_fetch(0, N, Tn, R, img_buf, torch.int32, Tn * 2).cpu().numpy().view(np.uint32).reshape(Tn, 2)
        flist = C.debug_fetch(1, N, Tn, R, bin_buf, torch.int32, R).cpu().numpy().view(np.uint32)
        assert np.array_equal(ranges, ref.binning.ranges)
        assert np.array_equal(flist, ref.binning.face_list)
        fT = C.debug_fetch(2, N, Tn, R, img_buf, torch.float32, N).cpu().numpy()
        fpT = C.debug_fetch(3, N, Tn, R, img_buf, torch.float32, N).cpu().numpy()
        nc = C.debug_fetch(4, N, Tn, R, img_buf, torch.int32, N).cpu().numpy().view(np.uint32)
        flipped = int((nc != ref.n_contrib).sum())
        assert flipped == 0, f"{flipped} pixels with a different last contributor"
        assert np.array_equal(fT.view(np.uint32), ref.final_T.view(np.uint32))
        assert np.array_equal(fpT.view(np.uint32), ref.final_prev_T.view(np.uint32))
    c, d = color.cpu().numpy(), depth.cpu().numpy()
    assert np.array_equal(c.view(np.uint32), ref.color.view(np.uint32)), f"color max abs diff {np.abs(c - ref.color).max()}"
    assert np.array_equal(d.view(np.uint32), ref.depth.view(np.uint32)), f"depth max abs diff {np.abs(d - ref.depth).max()}"
    assert np.array_equal(tri_cnt.cpu().numpy(), ref.buf_tri_cnt)


GRAD_NAMES = ["verts", "verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"]


def check_backward(res, tol=GRAD_TOL):
    worst = {}
    for name, g in zip(GRAD_NAMES, res["grads"]):
        r = res["ref_grads"][name]
        assert g.shape == r.shape, name
        assert np.isfinite(g).all() == np.isfinite(r).all(), name
        m = np.isfinite(r)
        worst[name] = rel_linf(g[m], r[m])
    assert all(v <= tol for v in worst.values()), worst
    ndc = res["grads"][3]
    assert not ndc[..., :2].any()                       # only the z channel receives gradient (backward.cu:516-518)
    return worst


CASES = [
    # W, H, F, seed, temp, K, cams, batch_idx, patch_min, pw, ph
    dict(W=64, H=64, F=300, seed=1, temp=1.0, K=20),
    dict(W=80, H=48, F=500, seed=2, temp=1.0, K=2),                       # K overflow path of the reference
    dict(W=50, H=37, F=200, seed=3, temp=0.5, K=20),                      # ragged: W,H not multiples of 16
    dict(W=64, H=64, F=300, seed=4, temp=0.0, K=20),                      # point-sampled coverage, K forced to 0
    dict(W=96, H=64, F=400, seed=5, temp=1.0, K=0),                       # no record buffer
    dict(W=96, H=80, F=600, seed=6, temp=1.0, K=20, cams=3, batch_idx=(2, 0),
         patch_min=[[16, 8], [5, 3]], pw=40, ph=33),                      # batch of patches at different offsets
    dict(W=128, H=128, F=3000, seed=7, temp=1.0, K=20, dc=12.0),          # deep: early termination T < 1e-4
    dict(W=64, H=64, F=20000, seed=8, temp=1.0, K=20, dc=1.5),            # sub-pixel faces: chunks full of 1-4 pixel rectangles
    dict(W=48, H=48, F=40, seed=9, temp=1.0, K=20, dc=60.0),              # huge faces: every face covers whole tiles (pair / survivor cuts)
    dict(W=48, H=48, F=60, seed=10, temp=0.0, K=20, dc=60.0),             # point sampling, huge faces: several rounds of hits per chunk
    dict(W=70, H=50, F=2500, seed=11, temp=0.0, K=0, dc=10.0),            # point sampling, many chunks per tile
]


def make_args(c):
    args, sc = soup_args(c["W"], c["H"], c["F"], scenes.SEED_BASE + c["seed"], temp=c["temp"], K=c["K"],
                         cams=c.get("cams", 1), batch_idx=c.get("batch_idx", (0,)), patch_min=c.get("patch_min"),
                         pw=c.get("pw"), ph=c.get("ph"), depth_complexity=c.get("dc", 4.0))
    return args


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_forward_backward_parity(case):
    args = make_args(case)
    res = run_both(args, seed=case["seed"])
    check_forward(res, args)
    check_backward(res)


def test_opaque_faces_alpha_one_and_early_exit():
    args = make_args(dict(W=64, H=64, F=400, seed=11, temp=0.0, K=0, dc=8.0))
    args = list(args)
    args[7] = torch.ones_like(args[7])                 # opacity 1, temp 0 -> alpha == 1 (backward.cu:396-401)
    res = run_both(args, seed=3)
    check_forward(res, args)
    check_backward(res)


def test_background_and_intensity():
    args = list(make_args(dict(W=48, H=48, F=150, seed=12, temp=1.0, K=20)))
    args[0] = torch.tensor([0.2, 0.7, 0.4])
    args[10] = torch.rand_like(args[10]) + 0.5
    res = run_both(args, seed=4)
    check_forward(res, args)
    check_backward(res)


@pytest.mark.parametrize("name", ["boundary_full.npz", "boundary_patch.npz"])
def test_golden_boundary_inputs(golden_dir, name):
    """Inputs exactly as the reference's Python produced them (committed fixtures)."""
    g = np.load(os.path.join(golden_dir, name))
    args = []
    for k in ARG_NAMES:
        a = g["arg_" + k]
        args.append(torch.from_numpy(a) if a.ndim > 0 else (float(a) if k == "aa_temperature" else int(a)))
    res = run_both(args, seed=9)
    check_forward(res, args)
    check_backward(res)


def test_degenerate_inputs():
    C = _C()
    args = list(make_args(dict(W=40, H=24, F=50, seed=13, temp=1.0, K=20)))
    # every face behind the camera / out of the depth range -> culled (forward.cu:71)
    a2 = list(args); a2[8] = args[8].clone(); a2[8][..., 2] = 5.0
    res = run_both(a2, seed=1)
    assert res["out"][0] == 0
    check_forward(res, a2); check_backward(res, tol=0)
    # F == 0
    a3 = list(args)
    a3[5] = args[5][:0]; a3[7] = args[7][:0]; a3[10] = args[10][:, :0]
    for i in (12, 13, 14, 15, 16, 17):
        a3[i] = args[i][:, :0]
    out = C.render_forward_cuda(*to_dev(a3))
    assert out[0] == 0 and not out[1].any() and not out[2].any()        # zero images (render.cu:128-129,149)
    # P == 0 and F == 0
    a4 = list(a3); a4[4] = args[4][:0]; a4[6] = args[6][:0]; a4[8] = args[8][:, :0]; a4[9] = args[9][:, :0]
    out = C.render_forward_cuda(*to_dev(a4))
    assert out[0] == 0 and not out[1].any()
    g = C.render_backward_cuda(0, *to_dev(a4), torch.zeros_like(out[1]), torch.zeros_like(out[2]), out[7], out[8], out[9],
                               out[3], out[4], out[5], out[6])
    assert [tuple(x.shape) for x in g] == [(0, 3), (0, 3), (0,), (1, 0, 3), (1, 0), (1, 0, 3, 2)]


def test_argument_errors():
    C = _C()
    args = to_dev(make_args(dict(W=32, H=32, F=20, seed=14, temp=1.0, K=20)))
    bad = list(args); bad[11] = 1.5
    with pytest.raises(RuntimeError, match="aa_temperature must be in the range"):
        C.render_forward_cuda(*bad)
    bad = list(args); bad[18] = -1
    with pytest.raises(RuntimeError, match="len_oarea_buffer must be non-negative"):
        C.render_forward_cuda(*bad)
    bad = list(args); bad[0] = torch.zeros(4, device="cuda")
    with pytest.raises(RuntimeError, match=r"background must have dimensions \(3,\)"):
        C.render_forward_cuda(*bad)
    bad = list(args); bad[7] = bad[7][:-1]
    with pytest.raises(RuntimeError, match="face opacity must have dimensions"):
        C.render_forward_cuda(*bad)


def test_one_big_triangle_covers_everything():
    """Opaque single triangle, temp 0 -> every covered pixel shows the interpolated vertex colours."""
    W = H = 64
    sc = scenes.triangle_soup(W, H, 1, 5)
    sc.verts = torch.tensor([[-10.0, -10.0, 0.0], [10.0, -10.0, 0.0], [0.0, 12.0, 0.0]])
    sc.verts_color = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]])
    sc.faces_opacity = torch.ones(1)
    from util import capture_forward_args
    args, _ = capture_forward_args(sc, [0], [[0, 0]], W, H, temp=0.0, K=0)
    res = run_both(args)
    check_forward(res, args)
    c = res["out"][1].cpu().numpy()
    assert np.allclose(c.sum(-1), 1.0, atol=1e-5)       # barycentric weights sum to one, nothing else blends
    assert res["out"][0] == 16                          # 4 x 4 tiles


def test_layers_exact():
    C, orc = _C(), _orc()
    W, H = 200, 120
    sc = scenes.tet_lattice(W, H, 6, seed=scenes.SEED_BASE + 3, num_cams=2)
    import dmesh2_renderer_amd as dm2
    scd = sc.to("cuda")
    lr = dm2.LayeredRenderer(scd.mv, scd.proj, W, H, "cuda")
    # both walks: over packed per-tet records (default) and with the reference's access pattern (DM2_FLAG_LEGACY_KERNELS)
    for L, legacy in ((4, False), (1, False), (4, True), (2, True)):
        old_flags = C.set_flags(C.DM2_FLAG_LEGACY_KERNELS if legacy else 0)
        try:
            layers, cnt = lr.generate([1, 0], scd.verts, scd.faces, scd.tets, scd.face_tets, scd.tet_faces, scd.faces_existence, L)
        finally:
            C.set_flags(old_flags)
        ndc, img = lr.compute_verts_ndc_image(scd.verts, scd.mv[[1, 0]], scd.proj[[1, 0]])
        ro, rd = lr.ray_o[[1, 0]], lr.ray_d[[1, 0]]
        rl, rc, rff, rft, bn = orc.generate_render_layers_cuda(
            W, H, sc.verts.numpy(), sc.faces.numpy(), sc.tets.numpy(), sc.face_tets.numpy(), sc.tet_faces.numpy(),
            sc.faces_existence.numpy(), ndc.cpu().numpy(), img.cpu().numpy(), ro.cpu().numpy(), rd.cpu().numpy(), L,
            return_first=True)
        R, face_buf, bin_buf, img_buf = C.generate_render_layers_cuda.last_debug
        assert R == bn.num_rendered
        N, Tn = 2 * H * W, 2 * ((W + 15) // 16) * ((H + 15) // 16)
        ff = C.debug_fetch(5, N, Tn, R, img_buf, torch.int32, N).cpu().numpy().reshape(2, H, W)
        ft = C.debug_fetch(6, N, Tn, R, img_buf, torch.int32, N).cpu().numpy().reshape(2, H, W)
        assert np.array_equal(ff, rff) and np.array_equal(ft, rft)
        assert np.array_equal(cnt.cpu().numpy(), rc)
        assert np.array_equal(layers.cpu().numpy(), rl)
        assert (rc > 0).mean() > 0.3                     # the lattice is actually hit
        assert layers.dtype == torch.int32 and layers.shape == (2, H, W, L)


def test_layers_with_an_unreachable_padded_tet():
    """A tet no face points to, whose face ids are padding (-1 and beyond F): the reference's walk never reads it.  The packed
    tet records are built for ALL tets, so their builder has to leave such entries alone: same layers, no out-of-bounds read."""
    C = _C()
    W, H = 96, 64
    sc = scenes.tet_lattice(W, H, 4, seed=scenes.SEED_BASE + 5)
    import dmesh2_renderer_amd as dm2
    scd = sc.to("cuda")
    lr = dm2.LayeredRenderer(scd.mv, scd.proj, W, H, "cuda")
    base = lr.generate([0], scd.verts, scd.faces, scd.tets, scd.face_tets, scd.tet_faces, scd.faces_existence, 3)
    F = scd.faces.shape[0]
    tets2 = torch.cat([scd.tets, scd.tets[:2]])
    tf2 = torch.cat([scd.tet_faces, torch.tensor([[-1, -1, -1, -1], [F + 1000, 2 ** 30, -7, F]], dtype=scd.tet_faces.dtype, device="cuda")])
    got = lr.generate([0], scd.verts, scd.faces, tets2, scd.face_tets, tf2, scd.faces_existence, 3)
    torch.cuda.synchronize()
    assert torch.equal(base[0], got[0]) and torch.equal(base[1], got[1])
