"""Every device clipper, directly, on the vectors the reference's own Python AA code produced
(tests/golden/aa_pairs.npz and aa_error_pairs.npz <- pyrenderer.py:207-425; the second file holds >= 8 inputs for
each of its exceptions "Error code 00".."05" plus near-tie inputs it accepts), through the C ABI
(dm2_debug_aa_overlap).

Bars: the generic clipper of the per-pixel-walk kernels (variant 0) and the forward's straight-line area clipper
(variant 1) are bit-equal to the CPU oracle -- area, and for variant 0 the Jacobian in the reference's fan order; the
backward's segment formulation (variant 2; the same polynomial regrouped, see dm2_clip_seg.h; variant 3: with the
reference's fan sum over its corners, the area bit-equal as well) agrees to 2 ulp of the
pixel area and 1e-6 absolute in the Jacobian.  All three report an error exactly where the reference raises.
Variant 4 is the default backward's Jacobian without a polygon (dm2_clip_fast.h): every pair it does NOT hand to the
segment formulation as a tie (code -1) is held to the oracle's Jacobian; the ties are variant 2's."""
import os

import numpy as np
import pytest
import torch

from util import ROOT

pytestmark = pytest.mark.gpu


# variant 2 regroups the reference's fan sum (same polynomial, same corner coordinates): per pair its Jacobian agrees with
# the oracle's to this fraction of max(1, largest Jacobian entry) -- 1e-6 typically; the bound is reached where an edge
# component is just above the 1e-3 "iszero" threshold and 1/e^2 ~ 10^5 amplifies the rounding of either order of summation
SEG_GRAD_TOL = 3e-5
SEG_AREA_TOL = 2.4e-7          # 2 ulp of the pixel area


def _grad_err(g, og):
    scale = np.maximum(1.0, np.abs(og).reshape(len(og), -1).max(axis=1))
    return np.abs(g - og).reshape(len(og), -1).max(axis=1) / scale


def _load(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    return {k: g[k] for k in g.files}


def _device_run(g, variant):
    from dmesh2_renderer_amd import _C
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).cuda()
    area, grad, code = _C.debug_aa_overlap(variant, t("t_verts"), t("t_edges"), t("t_edges_iszero"), t("t_edges_recip"),
                                           t("t_edges_normal"), t("t_edges_normal_c"), t("pixmin"))
    torch.cuda.synchronize()
    return area.cpu().numpy(), grad.cpu().numpy(), code.cpu().numpy()


def _oracle_run(g):
    from oracle import cpu as orc
    t = dict(verts=g["t_verts"], edges=g["t_edges"], iszero=g["t_edges_iszero"], recip=g["t_edges_recip"],
             normal=g["t_edges_normal"], normal_c=g["t_edges_normal_c"])
    n = len(g["pixmin"])
    area = np.zeros(n, np.float32); grad = np.zeros((n, 3, 2), np.float32); code = np.zeros(n, np.int32)
    for i in range(n):
        a, gr, c = orc.aa_overlap(t, i, g["pixmin"][i], np.float32)
        code[i] = c
        if c == 0:
            area[i] = a; grad[i] = gr
    return area, grad, code


@pytest.mark.parametrize("name", ["aa_pairs.npz", "aa_error_pairs.npz"])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_device_clipper_on_reference_vectors(name, variant):
    g = _load(name)
    ref_err = np.array([bool(m) for m in g["msg_analytic"]])
    assert ref_err.sum() >= 10
    area, grad, code = _device_run(g, variant)
    o_area, o_grad, o_code = _oracle_run(g)
    # error <=> the reference raised (and the oracle agrees)
    assert np.array_equal(code != 0, ref_err), np.where((code != 0) != ref_err)[0]
    assert np.array_equal(o_code != 0, ref_err)
    ok = ~ref_err
    if variant in (0, 1, 3):      # 3: the segment formulation with the reference's fan sum over its corners: the area to the bit
        assert np.array_equal(area.view(np.uint32), o_area.view(np.uint32)), np.abs(area - o_area).max()
    else:
        assert np.abs(area - o_area).max() <= SEG_AREA_TOL, np.abs(area - o_area).max()
    if variant == 0:
        assert np.array_equal(grad.view(np.uint32), o_grad.view(np.uint32)), np.abs(grad - o_grad).max()
    elif variant in (2, 3):
        # a pair whose area is exactly 0 never blends (forward.cu:337-338): the backward never sees it, the hook returns zeros
        ok = ok & (o_area != 0)
        assert _grad_err(grad, o_grad)[ok].max() <= SEG_GRAD_TOL, _grad_err(grad, o_grad)[ok].max()
    if variant != 1:
        # and against the reference's own numbers, with the tolerance tests/test_oracle_aa.py uses for the oracle
        assert np.allclose(grad[ok], g["grad_analytic"][ok], rtol=1e-5, atol=2e-6)
    assert np.abs(area[ok] - g["area_analytic"][ok]).max() <= 2e-6


def test_full_cover_and_reject_are_exact():
    g = _load("aa_pairs.npz")
    full = (g["area_analytic"] == 1.0) & (g["err_analytic"] == 0)
    zero = (g["area_analytic"] == 0.0) & (g["err_analytic"] == 0)
    assert full.sum() >= 5 and zero.sum() >= 20
    for variant in (0, 1, 2):
        area, grad, code = _device_run(g, variant)
        assert (area[full] == 1.0).all() and not grad[full].any()    # aa.h:493-496: zero Jacobian
        assert (area[zero] == 0.0).all()


def _random_pairs(seed, n, base, radius):
    rng = np.random.RandomState(seed)
    tris = np.zeros((n, 3, 2), np.float32); pms = np.zeros((n, 2), np.float32)
    for it in range(n):
        c = rng.uniform(0, 3, 2) + base
        r = radius * rng.uniform(0.6, 1.4, 3)
        th = rng.uniform(0, 2 * np.pi) + np.arange(3) * 2 * np.pi / 3
        tri = (c[None, :] + np.stack([r * np.cos(th), r * np.sin(th)], -1)).astype(np.float32)
        if it % 7 == 0:
            tri[1, 0] = tri[0, 0] + np.float32(rng.uniform(-2e-3, 2e-3))      # near-vertical edge (|e.x| < 1e-3 on some)
        if it % 11 == 0:
            tri[2, 1] = tri[1, 1] + np.float32(rng.uniform(-2e-3, 2e-3))      # near-horizontal edge
        tris[it] = tri
        pms[it] = np.floor(c + rng.uniform(-radius - 1, radius + 1, 2))
    return tris, pms


@pytest.mark.parametrize("base,radius", [(10.0, 2.5), (1000.0, 2.5), (1900.0, 0.7), (500.0, 12.0), (300.0, 60.0)])
def test_device_clippers_random_pairs(base, radius):
    """Random (triangle, pixel) pairs at image coordinates up to 1080p / 4K magnitudes and triangle sizes from
    sub-pixel to 100 px: variants 0 / 1 bit-equal to the oracle, variant 2 within 2 ulp (area) / 1e-6 (Jacobian)."""
    from oracle import cpu as orc
    tris, pms = _random_pairs(int(base) + int(radius * 10), 3000, base, radius)
    with np.errstate(divide="ignore"):
        t = orc.aa_tables(tris, np.float32, reorder=True)
    g = dict(t_verts=t["verts"], t_edges=t["edges"], t_edges_iszero=t["iszero"], t_edges_recip=t["recip"],
             t_edges_normal=t["normal"], t_edges_normal_c=t["normal_c"], pixmin=pms)
    o_area, o_grad, o_code = _oracle_run(g)
    partial = (o_code == 0) & (o_area > 0) & (o_area < 1)
    assert partial.sum() >= 40, partial.sum()
    for variant in (0, 1, 2, 3):
        area, grad, code = _device_run(g, variant)
        assert np.array_equal(code != 0, o_code != 0), variant
        if variant >= 2:
            live = (o_code == 0) & (o_area != 0)
            if variant == 3:      # exact_area: the reference's fan sum over the rebuilt corners, bit for bit
                assert np.array_equal(area[live].view(np.uint32), o_area[live].view(np.uint32)), np.abs(area - o_area)[live].max()
            assert np.abs(area - o_area).max() <= SEG_AREA_TOL, (variant, np.abs(area - o_area).max())
            assert _grad_err(grad, o_grad)[live].max() <= SEG_GRAD_TOL, (variant, _grad_err(grad, o_grad)[live].max())
        else:
            assert np.array_equal(area.view(np.uint32), o_area.view(np.uint32)), variant
            if variant == 0:
                assert np.array_equal(grad.view(np.uint32), o_grad.view(np.uint32))


def _tie_pairs(seed, n):
    """Triangles with a corner EXACTLY on a boundary line of the tested pixel (or 1 ulp off it), two such corners, or all
    coordinates on a quarter-pixel grid: the ties under which the reference's polygon stops being the geometric
    intersection (it still returns an area and a Jacobian, and the forward blends with them)."""
    rng = np.random.RandomState(seed)

    def jit(x, k):
        x = np.float32(x)
        for _ in range(abs(k)):
            x = np.nextafter(x, np.float32(np.inf) if k > 0 else np.float32(-np.inf))
        return x
    tris = np.zeros((n, 3, 2), np.float32); pms = np.zeros((n, 2), np.float32)
    for it in range(n):
        base = rng.choice([8, 300, 1500])
        pm = (rng.randint(2, 20, size=2) + base).astype(np.float32)
        c = pm + rng.uniform(0, 1, 2)
        r = float(rng.choice([0.8, 2.5, 8.0])) * rng.uniform(0.6, 1.4, 3)
        th = rng.uniform(0, 2 * np.pi) + np.arange(3) * 2 * np.pi / 3
        tri = (c[None, :] + np.stack([r * np.cos(th), r * np.sin(th)], -1)).astype(np.float32)
        k = int(rng.choice([0, 0, 0, -1, 1]))
        vi = rng.randint(0, 3)
        kind = it % 4
        if kind == 0:
            tri[vi, 0] = jit(pm[0] + rng.randint(0, 2), k); tri[vi, 1] = pm[1] + rng.uniform(0.05, 0.95)
        elif kind == 1:
            tri[vi, 1] = jit(pm[1] + rng.randint(0, 2), k); tri[vi, 0] = pm[0] + rng.uniform(0.05, 0.95)
        elif kind == 2:
            tri[vi, 0] = jit(pm[0] + rng.randint(0, 2), k); tri[vi, 1] = pm[1] + rng.uniform(0.05, 0.95)
            vj = (vi + 1) % 3
            tri[vj, 1] = jit(pm[1] + rng.randint(0, 2), 0); tri[vj, 0] = pm[0] + rng.uniform(-1.5, 2.5)
        else:
            tri = (np.round(tri * 4) / 4).astype(np.float32)
        tris[it] = tri; pms[it] = pm
    return tris, pms


def test_device_clippers_exact_ties():
    """Exact ties: about a third of these inputs make the reference raise, a few per cent make it return an area that
    is NOT the geometric overlap (e.g. 0.92 where the overlap is 0.53) -- every variant has to follow it there."""
    from oracle import cpu as orc
    tris, pms = _tie_pairs(5, 12000)
    with np.errstate(all="ignore"):
        t = orc.aa_tables(tris, np.float32, reorder=True)
    g = dict(t_verts=t["verts"], t_edges=t["edges"], t_edges_iszero=t["iszero"], t_edges_recip=t["recip"],
             t_edges_normal=t["normal"], t_edges_normal_c=t["normal_c"], pixmin=pms)
    with np.errstate(all="ignore"):
        o_area, o_grad, o_code = _oracle_run(g)
    assert (o_code != 0).sum() >= 1500 and ((o_code == 0) & (o_area > 0) & (o_area < 1)).sum() >= 4000
    for variant in (0, 1, 2, 3):
        area, grad, code = _device_run(g, variant)
        assert np.array_equal(code != 0, o_code != 0), variant
        if variant >= 2:
            live = (o_code == 0) & (o_area != 0)
            if variant == 3:
                assert np.array_equal(area[live].view(np.uint32), o_area[live].view(np.uint32)), np.abs(area - o_area)[live].max()
            assert np.abs(area - o_area).max() <= SEG_AREA_TOL, np.abs(area - o_area).max()
            assert _grad_err(grad, o_grad)[live].max() <= SEG_GRAD_TOL, _grad_err(grad, o_grad)[live].max()
        else:
            assert np.array_equal(area.view(np.uint32), o_area.view(np.uint32)), variant
            if variant == 0:
                assert np.array_equal(grad.view(np.uint32), o_grad.view(np.uint32))


# ---- variant 4: the Jacobian without a polygon (dm2_clip_fast.h) + its tie flag ------------------------------------
FAST_GRAD_TOL = 1e-5           # of max(1, largest Jacobian entry); 2e-7 typical, 8e-6 the largest seen


def _check_fast(g, o_area, o_grad, o_code, max_tie_fraction=None, min_ok=10):
    area, grad, code = _device_run(g, 4)
    assert np.array_equal(code > 0, o_code != 0)                      # errors exactly where the reference raises
    live = (o_code == 0) & (o_area != 0)
    assert np.array_equal(area[live].view(np.uint32), o_area[live].view(np.uint32))     # (the forward's area)
    tie = code == -1
    assert not (tie & ~live).any()
    ok = live & ~tie
    assert ok.sum() >= min_ok, ok.sum()
    err = _grad_err(grad, o_grad)
    assert err[ok].max() <= FAST_GRAD_TOL, (err[ok].max(), np.where(ok & (err > FAST_GRAD_TOL))[0][:10])
    if max_tie_fraction is not None:
        assert tie[live].mean() <= max_tie_fraction, tie[live].mean()
    # and the ties go to the segment formulation, held to its own bar above
    return tie[live].mean(), err[ok].max()


@pytest.mark.parametrize("name", ["aa_pairs.npz", "aa_error_pairs.npz"])
def test_fast_jacobian_on_reference_vectors(name):
    g = _load(name)
    o_area, o_grad, o_code = _oracle_run(g)
    _check_fast(g, o_area, o_grad, o_code, min_ok=100 if name == "aa_pairs.npz" else 1)     # (the second file is all near-ties)
    area, grad, code = _device_run(g, 4)
    ok = (o_code == 0) & (o_area != 0) & (code == 0)
    assert np.allclose(grad[ok], g["grad_analytic"][ok], rtol=1e-5, atol=2e-6)    # the reference's own numbers


@pytest.mark.parametrize("base,radius,n", [(10.0, 2.5, 20000), (1000.0, 2.5, 40000), (1900.0, 0.7, 20000), (500.0, 12.0, 20000),
                                            (300.0, 60.0, 10000), (3800.0, 3.0, 40000)])
def test_fast_jacobian_random_pairs(base, radius, n):
    """Pairs in general position: the polygon-free Jacobian IS the oracle's (same corner coordinates, same polynomial);
    a few per cent are flagged as ties (more at 4K coordinates and for sub-pixel triangles, whose every pair has a corner
    near the pixel)."""
    from oracle import cpu as orc
    tris, pms = _random_pairs(int(base) * 7 + int(radius * 10) + 1, n, base, radius)
    with np.errstate(divide="ignore"):
        t = orc.aa_tables(tris, np.float32, reorder=True)
    g = dict(t_verts=t["verts"], t_edges=t["edges"], t_edges_iszero=t["iszero"], t_edges_recip=t["recip"],
             t_edges_normal=t["normal"], t_edges_normal_c=t["normal_c"], pixmin=pms)
    o_area, o_grad, o_code = _oracle_run(g)
    frac, worst = _check_fast(g, o_area, o_grad, o_code, max_tie_fraction=0.08)
    print(f"fast Jacobian base {base} radius {radius}: ties {frac:.3%}, worst error {worst:.2e}")


def _axis_pairs(seed, n):
    """Faces with exactly axis-parallel edges (a regular grid mesh seen head-on): recip = inf, crossing parameters inf."""
    rng = np.random.RandomState(seed)
    tris = np.zeros((n, 3, 2), np.float32); pms = np.zeros((n, 2), np.float32)
    for it in range(n):
        base = rng.choice([8, 300, 1500])
        o = (rng.uniform(0, 6, 2) + base).astype(np.float32)
        s = np.float32(rng.uniform(0.8, 6.0)); u = np.float32(rng.uniform(0.8, 6.0))
        kind = it % 3
        if kind == 0:
            tri = np.array([o, o + [s, 0], o + [0, u]], np.float32)                  # right angle, two axis-parallel edges
        elif kind == 1:
            tri = np.array([o + [s, u], o + [0, u], o + [s, 0]], np.float32)
        else:
            tri = np.array([o, o + [s, 0], o + rng.uniform(0.5, 5, 2)], np.float32)  # one axis-parallel edge
        tris[it] = tri
        pms[it] = np.floor(o + rng.uniform(-1, 6, 2))
    return tris, pms


def test_fast_jacobian_axis_parallel_and_ties():
    from oracle import cpu as orc
    for (tris, pms), cap in ((_axis_pairs(3, 12000), 0.08), (_tie_pairs(5, 12000), None), (_tie_pairs(11, 12000), None)):
        with np.errstate(all="ignore"):
            t = orc.aa_tables(tris, np.float32, reorder=True)
            g = dict(t_verts=t["verts"], t_edges=t["edges"], t_edges_iszero=t["iszero"], t_edges_recip=t["recip"],
                     t_edges_normal=t["normal"], t_edges_normal_c=t["normal_c"], pixmin=pms)
            o_area, o_grad, o_code = _oracle_run(g)
        frac, worst = _check_fast(g, o_area, o_grad, o_code, max_tie_fraction=cap)
        print(f"fast Jacobian: ties {frac:.3%}, worst error {worst:.2e}")
