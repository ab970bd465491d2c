"""CPU-only checks of the oracle (test infrastructure) beyond the AA vectors:

* its backward is the derivative of its forward (fp64 central differences) for every input that
  receives a gradient -- with `verts` checked in the opt-in corrected mode only, because the
  reference's as-written d(bary v)/d(verts) is really d(t)/d(verts) (auxiliary.h:272-280);
* domain invariants that hold at any size: tiling independence (any patch decomposition reproduces
  the full frame bit for bit), aa_temperature = 0 is point sampling, K does not change results,
  binning order / key packing, config-1 (256x256 / 2k triangles, CPU tensors) end to end.
"""
import numpy as np
import pytest
import torch

from oracle import cpu as orc
from util import scenes, soup_args, to_numpy_args


def _f64(args):
    out = []
    for a in to_numpy_args(args):
        if isinstance(a, np.ndarray) and a.dtype == np.float32:
            a = a.astype(np.float64)
        out.append(a)
    return out


def _rebuild_tables(a):
    """AA edge tables as functions of aa_face_verts (pyrenderer.py:10-30), fp64, no reordering."""
    t = orc.aa_tables(a[12], np.float64, reorder=False)
    a[12], a[13], a[14], a[15], a[16], a[17] = t["verts"], t["edges"], t["iszero"], t["recip"], t["normal"], t["normal_c"]
    return a


def _loss(a, gc, gd):
    f = orc.render_forward_cuda(*a, dtype=np.float64)
    return float((f.color * gc).sum() + (f.depth * gd).sum()), f


@pytest.fixture(scope="module")
def small_scene():
    args, sc = soup_args(40, 32, 30, scenes.SEED_BASE + 21, temp=0.7, K=20, depth_complexity=3.0)
    a = _rebuild_tables(_f64(args))
    rng = np.random.RandomState(3)
    gc = rng.randn(1, 32, 40, 3); gd = rng.randn(1, 32, 40)
    L0, f = _loss(a, gc, gd)
    g = orc.render_backward_cuda(f, gc, gd)
    gcorr = orc.render_backward_cuda(f, gc, gd, corrected_dv=True)
    return a, gc, gd, g, gcorr, f


def _fd(a, gc, gd, idx, pos, h, rebuild=False):
    ap = [x.copy() if isinstance(x, np.ndarray) else x for x in a]
    am = [x.copy() if isinstance(x, np.ndarray) else x for x in a]
    ap[idx][pos] += h; am[idx][pos] -= h
    if rebuild:
        ap = _rebuild_tables(ap); am = _rebuild_tables(am)
    return (_loss(ap, gc, gd)[0] - _loss(am, gc, gd)[0]) / (2 * h)


def _check(a, gc, gd, grad, idx, rng, n, h, rebuild=False, only_last=None, rtol=2e-4, atol=1e-6):
    nz = np.argwhere(np.abs(grad) > 1e-9)
    assert len(nz) > 0
    ok = 0
    for pos in nz[rng.permutation(len(nz))[:n]]:
        pos = tuple(pos)
        if only_last is not None and pos[-1] != only_last:
            continue
        fd = _fd(a, gc, gd, idx, pos, h, rebuild)
        an = grad[pos]
        # piecewise-smooth function: a perturbation that crosses a kink (pixel-corner tie, clamp region)
        # is retried with a smaller step before it counts as a failure
        if not np.isclose(an, fd, rtol=rtol, atol=atol):
            fd = _fd(a, gc, gd, idx, pos, h * 0.1, rebuild)
        assert np.isclose(an, fd, rtol=5 * rtol, atol=10 * atol), (idx, pos, an, fd)
        ok += 1
    return ok


def test_backward_is_derivative_of_forward(small_scene):
    a, gc, gd, g, gcorr, f = small_scene
    rng = np.random.RandomState(0)
    assert _check(a, gc, gd, g["verts_color"], 6, rng, 12, 1e-6) >= 8
    assert _check(a, gc, gd, g["faces_opacity"], 7, rng, 10, 1e-6) >= 8
    assert _check(a, gc, gd, g["faces_intense"], 10, rng, 8, 1e-6) >= 6
    assert _check(a, gc, gd, g["verts_ndc"], 8, rng, 40, 1e-6, only_last=2) >= 6
    assert _check(a, gc, gd, g["aa_face_verts"], 12, rng, 12, 1e-7, rebuild=True, rtol=1e-3, atol=1e-5) >= 8
    assert not g["verts_ndc"][..., :2].any()


def test_dverts_matches_fd_only_in_corrected_mode(small_scene):
    a, gc, gd, g, gcorr, f = small_scene
    rng = np.random.RandomState(1)
    assert _check(a, gc, gd, gcorr["verts"], 4, rng, 12, 1e-7, rtol=1e-3, atol=1e-5) >= 8
    # the as-written mode mixes grad(u) with grad(t): it must differ from the true derivative ...
    d = np.abs(g["verts"] - gcorr["verts"]).max() / np.abs(gcorr["verts"]).max()
    assert d > 1e-3
    # ... while every other gradient is untouched by the flag
    for k in ("verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"):
        assert np.array_equal(g[k], gcorr[k])


def test_f32_oracle_close_to_f64(small_scene):
    a, gc, gd, g, gcorr, f = small_scene
    a32 = [x.astype(np.float32) if isinstance(x, np.ndarray) and x.dtype == np.float64 else x for x in a]
    f32 = orc.render_forward_cuda(*a32)
    same = f32.n_contrib == f.n_contrib           # pixels where fp32 took the same branches
    assert same.mean() > 0.98
    m = same.reshape(f.depth.shape)
    assert np.abs(f32.color - f.color)[m].max() < 5e-5


def test_tiling_independence_and_patches():
    """Per-pixel output depends only on the depth-ordered faces overlapping that pixel:
    band / patch renders equal the same rows / window of the full frame bit for bit."""
    W, H = 96, 80
    args, sc = soup_args(W, H, 400, scenes.SEED_BASE + 22)
    full = orc.render_forward_cuda(*to_numpy_args(args))
    na = to_numpy_args(args)
    for (x0, y0, pw, ph) in [(0, 16, W, 32), (32, 0, 48, H), (16, 48, 64, 32), (5, 3, 37, 29)]:
        a = list(na)
        a[1] = np.array([[x0, y0]], np.int32); a[2] = pw; a[3] = ph
        a[19] = np.ascontiguousarray(na[19][:, y0:y0 + ph, x0:x0 + pw]); a[20] = np.ascontiguousarray(na[20][:, y0:y0 + ph, x0:x0 + pw])
        part = orc.render_forward_cuda(*a)
        assert np.array_equal(part.color.view(np.uint32), full.color[:, y0:y0 + ph, x0:x0 + pw].view(np.uint32))
        assert np.array_equal(part.depth.view(np.uint32), full.depth[:, y0:y0 + ph, x0:x0 + pw].view(np.uint32))


def test_record_buffer_depth_does_not_change_gradients():
    """K only changes how backward obtains the AA Jacobian (pop vs recompute, backward.cu:241-284)."""
    res = []
    rng = np.random.RandomState(5)
    for K in (20, 3, 0):
        args, sc = soup_args(64, 48, 300, scenes.SEED_BASE + 23, K=K)
        f = orc.render_forward_cuda(*to_numpy_args(args))
        if not res:
            gc = rng.randn(*f.color.shape).astype(np.float32); gd = rng.randn(*f.depth.shape).astype(np.float32)
        res.append((f, orc.render_backward_cuda(f, gc, gd)))
    assert res[0][0].buf_tri_cnt.max() > 3                      # K=3 really overflows
    for f, g in res[1:]:
        assert np.array_equal(f.color, res[0][0].color)
        for k in g:
            assert np.array_equal(g[k], res[0][1][k]), k


def test_temperature_zero_is_point_sampling():
    args, sc = soup_args(48, 48, 120, scenes.SEED_BASE + 24, temp=0.0)
    f = orc.render_forward_cuda(*to_numpy_args(args))
    assert f.buf_oarea.shape[-1] == 0 and not f.buf_tri_cnt.any()           # K forced to 0 (render.cu:141-142)
    na = to_numpy_args(args)
    # a pixel is touched iff its centre ray hits a face inside (code 0): compare with a brute-force test
    ro, rd = na[19][0], na[20][0]
    verts, faces = na[4], na[5]
    hit = np.zeros((48, 48), bool)
    for fidx in range(faces.shape[0]):
        p = verts[faces[fidx]].reshape(-1)
        for y in range(48):
            for x in range(0, 48, 7):
                ok, tuv, _ = orc.ray_tri(ro[y, x], rd[y, x], p)
                if ok and tuv[1] >= 0 and tuv[2] >= 0 and tuv[1] + tuv[2] <= 1:
                    hit[y, x] = True
    touched = (f.n_contrib.reshape(48, 48) > 0)
    assert np.array_equal(touched[:, ::7], hit[:, ::7])


def test_binning_order_and_keys():
    args, sc = soup_args(100, 70, 500, scenes.SEED_BASE + 25, cams=2, batch_idx=(0, 1))
    na = to_numpy_args(args)
    bn = orc.Binning(2, na[4].shape[0], na[5].shape[0], 100, 70, na[1], na[5], na[8], na[9])
    gx, gy = 7, 5
    assert bn.num_rendered == int(bn.tiles_touched.sum()) == len(bn.keys)
    tiles = (bn.keys >> np.uint64(32)).astype(np.int64)
    assert (np.diff(tiles) >= 0).all()                                        # sorted by tile
    depth_bits = (bn.keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    same = np.diff(tiles) == 0
    assert (np.diff(depth_bits)[same] >= 0).all()                             # then by depth bits
    tie = same & (np.diff(depth_bits) == 0)
    assert (np.diff(bn.face_list.astype(np.int64))[tie] > 0).all()            # ties: ascending face id (stable sort)
    for t in range(2 * gx * gy):
        s, e = bn.ranges[t]
        assert (tiles[s:e] == t).all()
    assert bn.ranges[:, 1].max() == bn.num_rendered
    assert orc.higher_msb(8160) == 13 and orc.higher_msb(256) == 9 and orc.higher_msb(1024) == 11   # sort bits
    # half-open tile rect, clamped (auxiliary.h:72-92)
    assert list(orc.patch_rect((0, 0), [[1.0, 1.0], [17.0, 2.0], [3.0, 33.0]], (4, 4))) == [0, 0, 2, 3]
    assert list(orc.patch_rect((16, 16), [[1.0, 1.0], [17.0, 2.0], [3.0, 33.0]], (4, 4))) == [0, 0, 1, 2]
    assert list(orc.patch_rect((0, 0), [[-50.0, -50.0], [-40.0, -45.0], [-45.0, -40.0]], (4, 4))) == [0, 0, 0, 0]


def test_clamp_bary_regions():
    pts = {0: (0.2, 0.3), 1: (-0.1, -0.2), 2: (1.4, -0.1), 3: (-0.2, 1.5), 4: (-0.3, 0.4), 5: (0.5, -0.2), 6: (0.8, 0.7)}
    for code, (u, v) in pts.items():
        c, out = orc.clamp_bary(u, v)
        assert c == code
        uc, vc = out[0], out[1]
        assert uc >= 0 and vc >= 0 and uc + vc <= 1 + 1e-6
        # Jacobian consistent with the clamp (finite differences inside the region)
        h = 1e-3
        for k, (du, dv) in enumerate([(h, 0), (0, h)]):
            c2, o2 = orc.clamp_bary(u + du, v + dv, np.float64)
            c1, o1 = orc.clamp_bary(u - du, v - dv, np.float64)
            if c1 == c2 == code:
                assert abs((o2[0] - o1[0]) / (2 * h) - out[2 + k]) < 1e-6
                assert abs((o2[1] - o1[1]) / (2 * h) - out[4 + k]) < 1e-6


def test_config1_cpu_end_to_end():
    """BASELINE config 1: forward 256x256, 2k random triangles, CPU tensors (oracle + Python host layer)."""
    args, sc = soup_args(256, 256, 2000, scenes.SEED_BASE + 1, shared=False)
    f = orc.render_forward_cuda(*to_numpy_args(args), nthreads=orc.max_threads())
    assert 6000 < f.num_rendered < 10000                       # R ~ 7.8k
    assert np.isfinite(f.color).all() and np.isfinite(f.depth).all()
    assert 0.0 <= f.color.min() and f.color.max() <= 1.0 + 1e-5
    covered = f.n_contrib > 0
    assert 0.9 < covered.mean() <= 1.0
    # depth post-map of the module (reference __init__.py:377-378): background +1 -> 0
    d = 1.0 - (f.depth + 1.0) / 2.0
    assert d[~covered.reshape(d.shape)].max(initial=0.0) <= 1e-7
    # single-thread and OpenMP runs agree exactly (per-pixel independence)
    f1 = orc.render_forward_cuda(*to_numpy_args(args), nthreads=1)
    assert np.array_equal(f1.color, f.color) and np.array_equal(f1.n_contrib, f.n_contrib)


def test_layers_oracle_on_lattice():
    """Faces come out in strictly increasing ray parameter, each exists, and the first one is the nearest hit."""
    W, H = 48, 40
    sc = scenes.tet_lattice(W, H, 3, seed=scenes.SEED_BASE + 3)
    import dmesh2_renderer_amd as dm2
    lr = dm2.LayeredRenderer(sc.mv, sc.proj, W, H, "cpu")
    ndc, img = lr.compute_verts_ndc_image(sc.verts, sc.mv, sc.proj)
    layers, cnt, ff, ft, bn = orc.generate_render_layers_cuda(
        W, H, sc.verts.numpy(), sc.faces.numpy(), sc.tets.numpy(), sc.face_tets.numpy(), sc.tet_faces.numpy(),
        sc.faces_existence.numpy(), ndc.numpy(), img.numpy(), lr.ray_o.numpy(), lr.ray_d.numpy(), 4, return_first=True)
    assert (cnt > 0).mean() > 0.3 and cnt.max() <= 4
    verts, faces = sc.verts.numpy(), sc.faces.numpy()
    exist = sc.faces_existence.numpy()
    ro, rd = lr.ray_o.numpy()[0], lr.ray_d.numpy()[0]
    checked = 0
    for y in range(0, H, 5):
        for x in range(0, W, 5):
            n = cnt[0, y, x]
            ids = layers[0, y, x]
            assert (ids[n:] == -1).all() and (ids[:n] >= 0).all()
            ts = []
            for fid in ids[:n]:
                assert exist[fid]
                ok, tuv, _ = orc.ray_tri(ro[y, x], rd[y, x], verts[faces[fid]].reshape(-1))
                assert ok and tuv[0] >= 0 and tuv[1] >= -1e-5 and tuv[2] >= -1e-5 and tuv[1] + tuv[2] <= 1 + 1e-5
                ts.append(tuv[0])
            assert all(b > a for a, b in zip(ts, ts[1:]))
            if ff[0, y, x] >= 0:
                # brute force nearest hit over ALL faces
                best, bt = -1, np.inf
                for fid in range(faces.shape[0]):
                    ok, tuv, _ = orc.ray_tri(ro[y, x], rd[y, x], verts[faces[fid]].reshape(-1))
                    if ok and tuv[0] >= 0 and tuv[1] >= 0 and tuv[2] >= 0 and tuv[1] + tuv[2] <= 1 and tuv[0] < bt:
                        best, bt = fid, tuv[0]
                assert best == ff[0, y, x]
                checked += 1
    assert checked > 10
