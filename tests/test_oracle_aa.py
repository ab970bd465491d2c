"""Oracle AA clipper vs vectors produced by the reference's own Python AA oracle
(tests/golden/aa_pairs.npz <- pyrenderer.py:207-425 analytic, :66-205 autograd)."""
import os

import numpy as np
import pytest

from oracle import cpu as orc


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "aa_pairs.npz"))


def test_tables_match_reference_triangles(g):
    t = orc.aa_tables(g["tri_in"], np.float32, reorder=True)
    assert np.array_equal(t["verts"], g["t_verts"])
    assert np.array_equal(t["edges"], g["t_edges"])
    assert np.array_equal(t["iszero"], g["t_edges_iszero"])
    assert np.array_equal(t["recip"].view(np.uint32), g["t_edges_recip"].view(np.uint32))   # incl. +-inf
    assert np.array_equal(t["normal"], g["t_edges_normal"])
    assert np.array_equal(t["normal_c"], g["t_edges_normal_c"])


def _tables(g):
    return dict(verts=g["t_verts"], edges=g["t_edges"], iszero=g["t_edges_iszero"], recip=g["t_edges_recip"],
                normal=g["t_edges_normal"], normal_c=g["t_edges_normal_c"])


def test_area_grad_and_errors_f32(g):
    t = _tables(g)
    n = len(g["pixmin"])
    n_err = n_partial = 0
    for i in range(n):
        area, grad, code = orc.aa_overlap(t, i, g["pixmin"][i], np.float32)
        if g["err_analytic"][i]:
            # the Python raises ValueError (caught -> area 0); the native code returns a code 1..6
            assert code != 0, i
            n_err += 1
            continue
        assert code == 0, (i, code)
        # the Python accumulates in fp32 tensors as well; allow a few ulp for op-order differences
        assert abs(area - g["area_analytic"][i]) <= 2e-6, (i, area, g["area_analytic"][i])
        assert np.allclose(grad, g["grad_analytic"][i], rtol=1e-5, atol=2e-6), (i, grad, g["grad_analytic"][i])
        if 0 < area < 1:
            n_partial += 1
    assert n_err >= 10 and n_partial >= 100


def test_area_grad_f64_agrees_with_autograd_flavour(g):
    t = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in _tables(g).items()}
    with np.errstate(divide="ignore"):
        t["recip"] = 1.0 / t["edges"]
    ok = 0
    for i in range(len(g["pixmin"])):
        if g["err_analytic"][i] or np.isnan(g["grad_autograd"][i]).any():
            continue
        area, grad, code = orc.aa_overlap(t, i, g["pixmin"][i].astype(np.float64), np.float64)
        if code != 0:
            continue      # fp64 tables can resolve an fp32 tie differently; not counted
        assert abs(area - g["area_autograd"][i]) <= 5e-6
        assert np.allclose(grad, g["grad_autograd"][i], rtol=2e-4, atol=2e-5), i
        ok += 1
    assert ok >= 180


def test_full_cover_and_disjoint(g):
    t = _tables(g)
    full = np.where((g["area_analytic"] == 1.0) & (g["err_analytic"] == 0))[0]
    zero = np.where((g["area_analytic"] == 0.0) & (g["err_analytic"] == 0))[0]
    assert len(full) >= 5 and len(zero) >= 20
    for i in full:
        area, grad, code = orc.aa_overlap(t, i, g["pixmin"][i])
        assert code == 0 and area == 1.0 and not grad.any()        # aa.h:493-496: zero gradient
    for i in zero:
        area, grad, code = orc.aa_overlap(t, i, g["pixmin"][i])
        assert code == 0 and area == 0.0


def test_aa_gradient_finite_difference_f64():
    """d(area)/d(verts) of the restatement vs central differences (tables rebuilt per perturbation)."""
    rng = np.random.RandomState(7)
    checked = 0
    for _ in range(60):
        pm = rng.randint(0, 10, size=2).astype(np.float64)
        c = pm + rng.uniform(-0.5, 1.5, size=2)
        ang = rng.uniform(0, 2 * np.pi) + np.arange(3) * 2 * np.pi / 3
        tri = np.stack([c[0] + rng.uniform(0.5, 2.5, 3) * np.cos(ang), c[1] + rng.uniform(0.5, 2.5, 3) * np.sin(ang)], -1)
        t = orc.aa_tables(tri[None], np.float64)
        area, grad, code = orc.aa_overlap(t, 0, pm, np.float64)
        if code != 0 or area in (0.0, 1.0):
            continue
        v = t["verts"][0]
        h = 1e-6
        fd = np.zeros((3, 2))
        for i in range(3):
            for k in range(2):
                vp = v.copy(); vp[i, k] += h
                vm = v.copy(); vm[i, k] -= h
                ap = orc.aa_overlap(orc.aa_tables(vp[None], np.float64, reorder=False), 0, pm, np.float64)[0]
                am = orc.aa_overlap(orc.aa_tables(vm[None], np.float64, reorder=False), 0, pm, np.float64)[0]
                fd[i, k] = (ap - am) / (2 * h)
        assert np.allclose(grad, fd, atol=1e-5), (grad, fd)
        checked += 1
    assert checked >= 25


# ---- the reference's exceptions E00..E05 (tests/golden/aa_error_pairs.npz) -------------------------------------
@pytest.fixture(scope="module")
def ge(golden_dir):
    return np.load(os.path.join(golden_dir, "aa_error_pairs.npz"))


def test_error_vectors_cover_every_code(ge):
    msgs = list(ge["msg_analytic"])
    for k in range(6):
        assert msgs.count("[pyrasterizer] Error code %02d" % k) >= 5, k
    assert msgs.count("") >= 30                      # near-tie inputs the reference accepts


def test_error_codes_match_reference(ge):
    """Reference exception "Error code 0k" (pyrenderer.py:274,294,364,380,391,423) <-> native code k+1
    (aa.h:265,301,325,355,398/411,437); accepted near-tie inputs give the reference's area and gradient."""
    t = _tables(ge)
    assert np.array_equal(orc.aa_tables(ge["tri_in"], np.float32)["verts"], ge["t_verts"])
    for i in range(len(ge["pixmin"])):
        area, grad, code = orc.aa_overlap(t, i, ge["pixmin"][i], np.float32)
        msg = str(ge["msg_analytic"][i])
        if msg:
            assert code == int(msg[-2:]) + 1, (i, msg, code)
            continue
        assert code == 0, (i, code)
        assert abs(area - ge["area_analytic"][i]) <= 2e-6, (i, area, ge["area_analytic"][i])
        assert np.allclose(grad, ge["grad_analytic"][i], rtol=1e-5, atol=2e-6), (i, grad, ge["grad_analytic"][i])
