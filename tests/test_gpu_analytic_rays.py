"""Renderer(analytic_rays=True) / LayeredRenderer(analytic_rays=True) -- no (B,H,W,3) ray tensors, the
kernels compute each pixel's ray from inv(mv), inv(proj) (DM2_FLAG_ANALYTIC_RAYS).

* against the tensor path fed with the SAME closed form evaluated by the oracle (oracle.cpu.analytic_rays_from_inverse):
  bit-equal images (the per-pixel arithmetic is identical), gradients to summation order;
* against the default path (rays built by the reference-shaped torch ops of Renderer._init_rays): within the port's
  1e-5 (the rays themselves agree to 1.2e-7, tests/test_host_prep.py pins that against the reference's fixtures)."""
import numpy as np
import pytest
import torch

from util import rel_linf, scenes

pytestmark = pytest.mark.gpu


def _render(r, sc, bidx, pm, pw, ph, temp, wc, wd):
    leaves = [sc.verts.clone().requires_grad_(True), sc.verts_color.clone().requires_grad_(True),
              sc.faces_opacity.clone().requires_grad_(True), sc.faces_intense.clone().requires_grad_(True)]
    color, depth = r(bidx, pm, pw, ph, leaves[0], sc.faces, leaves[1], leaves[2], leaves[3][bidx], sc.background, aa_temperature=temp)
    ((color * wc).sum() + (depth * wd).sum()).backward()
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), depth.detach().cpu().numpy(), [t.grad.cpu().numpy() for t in leaves]


@pytest.mark.parametrize("temp", [1.0, 0.0])
@pytest.mark.parametrize("kernels", ["dense", "legacy"])
def test_renderer_analytic_rays(temp, kernels):
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import _C
    from oracle import cpu as orc
    W, H = 96, 72
    sc = scenes.triangle_soup(W, H, 500, scenes.SEED_BASE + 41, num_cams=3).to("cuda")
    bidx = [2, 0]
    pm = torch.tensor([[16, 8], [5, 3]], dtype=torch.int64, device="cuda")
    pw, ph = 60, 41
    g = torch.Generator().manual_seed(5)
    wc = torch.randn((2, ph, pw, 3), generator=g).cuda(); wd = torch.randn((2, ph, pw), generator=g).cuda()
    old = _C.set_flags(_C.DM2_FLAG_LEGACY_KERNELS if kernels == "legacy" else 0)
    try:
        ra = dm2.Renderer(sc.mv, sc.proj, W, H, "cuda", analytic_rays=True)
        assert ra.ray_o is None and ra.ray_d is None                       # no ray tensors are held
        ca, da, ga = _render(ra, sc, bidx, pm, pw, ph, temp, wc, wd)
        # the tensor path on the oracle's evaluation of the same closed form, from the SAME fp32 inverses
        cam = ra.ray_cam.cpu().numpy()
        ro, rd = orc.analytic_rays_from_inverse(cam[:, :16].reshape(-1, 4, 4), cam[:, 16:].reshape(-1, 4, 4), W, H)
        rt = dm2.Renderer(sc.mv, sc.proj, W, H, "cuda")
        rt.ray_o, rt.ray_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
        ct, dt, gt = _render(rt, sc, bidx, pm, pw, ph, temp, wc, wd)
        assert np.array_equal(ca.view(np.uint32), ct.view(np.uint32)) and np.array_equal(da.view(np.uint32), dt.view(np.uint32))
        for a, b in zip(ga, gt):
            assert rel_linf(a, b) <= 2e-6
        # and the default path (rays from the reference-shaped torch ops)
        rd_ = dm2.Renderer(sc.mv, sc.proj, W, H, "cuda")
        cd, dd, gd = _render(rd_, sc, bidx, pm, pw, ph, temp, wc, wd)
        assert np.abs(ca - cd).max() <= 1e-5 * max(np.abs(cd).max(), 1.0) and np.abs(da - dd).max() <= 1e-5
        for a, b in zip(ga, gd):
            assert rel_linf(a, b) <= 1e-5
    finally:
        _C.set_flags(old)


def test_layered_renderer_analytic_rays():
    import dmesh2_renderer_amd as dm2
    from oracle import cpu as orc
    W, H = 120, 88
    sc = scenes.tet_lattice(W, H, 6, seed=scenes.SEED_BASE + 5, num_cams=2).to("cuda")
    la = dm2.LayeredRenderer(sc.mv, sc.proj, W, H, "cuda", analytic_rays=True)
    layers_a, cnt_a = la.generate([1, 0], sc.verts, sc.faces, sc.tets, sc.face_tets, sc.tet_faces, sc.faces_existence, 4)
    cam = la.ray_cam.cpu().numpy()
    ro, rd = orc.analytic_rays_from_inverse(cam[:, :16].reshape(-1, 4, 4), cam[:, 16:].reshape(-1, 4, 4), W, H)
    lt = dm2.LayeredRenderer(sc.mv, sc.proj, W, H, "cuda")
    lt.ray_o, lt.ray_d = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
    layers_t, cnt_t = lt.generate([1, 0], sc.verts, sc.faces, sc.tets, sc.face_tets, sc.tet_faces, sc.faces_existence, 4)
    torch.cuda.synchronize()
    assert torch.equal(layers_a, layers_t) and torch.equal(cnt_a, cnt_t)
    assert (cnt_a > 0).float().mean() > 0.3


def test_analytic_rays_argument_errors():
    from dmesh2_renderer_amd import _C
    from util import soup_args
    args, _ = soup_args(32, 32, 20, scenes.SEED_BASE + 42)
    dargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
    dargs[19] = dargs[20] = torch.empty((1, 0, 0, 3), device="cuda")
    with pytest.raises(RuntimeError, match="image_ray_o"):
        _C.render_forward_cuda(*dargs)                                     # placeholders without the side channel
    with _C.analytic_rays(torch.zeros((1, 31), device="cuda"), 32, 32):
        with pytest.raises(RuntimeError, match="ray_cam must be float32"):
            _C.render_forward_cuda(*dargs)
