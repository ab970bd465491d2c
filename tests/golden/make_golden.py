#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It imports the reference's *Python* host layer with a stub
in place of its (unbuildable here: nvcc + glm missing) native ``_C`` module,
and records

  aa_pairs.npz        tri/pixel overlap area + d(area)/d(tri verts) from the
                      reference's scalar Python AA oracle, both flavours
                      (pyrenderer.py:66-205 autograd, :207-425 analytic), plus
                      the per-triangle tables ``Triangles`` builds (:6-30).
  aa_error_pairs.npz  the same fields for inputs found by a seeded search near the clipper's tie cases: at
                      least 8 inputs for each of the reference's exceptions "Error code 00" .. "05"
                      (pyrenderer.py:274,294,364,380,391,423), labelled by the message the reference printed,
                      plus near-tie inputs it accepts.
  boundary_*.npz      the exact 21 positional arguments the reference hands to
                      ``_C.render_forward_cuda`` (__init__.py:48-78) for small
                      seeded scenes, the final ``(color, depth)`` post-map
                      (__init__.py:377-378) for a stub-returned image, and the
                      11 arguments of ``_C.generate_render_layers_cuda``
                      (__init__.py:432-449).

Only arrays are written; no reference source or bytecode is copied.  The GPU
box never runs this script (no /root/reference there) -- tests read the .npz.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def load_scenes():
    spec = importlib.util.spec_from_file_location(
        "dm2_scenes", os.path.join(REPO, "dmesh2_renderer_amd", "scenes.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dm2_scenes"] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    sys.path.insert(0, REF)
    stub = types.ModuleType("dmesh2_renderer._C")
    stub.calls = []
    sys.modules["dmesh2_renderer._C"] = stub
    import dmesh2_renderer  # noqa: E402
    return dmesh2_renderer, stub


def t2n(x):
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# --------------------------------------------------------------------------
# AA pairs
# --------------------------------------------------------------------------
def aa_cases(rng):
    """(tri (3,2), pixmin (2,)) pairs covering every branch of the clipper."""
    cases = []
    # 1. random partial overlaps: small triangles around a pixel
    for _ in range(140):
        pm = rng.randint(0, 20, size=2).astype(np.float32)
        c = pm + rng.uniform(-0.8, 1.8, size=2)
        ang = rng.uniform(0, 2 * np.pi) + np.arange(3) * 2 * np.pi / 3 + rng.uniform(-0.4, 0.4, 3)
        rad = rng.uniform(0.3, 3.0, size=3)
        tri = np.stack([c[0] + rad * np.cos(ang), c[1] + rad * np.sin(ang)], -1)
        if rng.uniform() < 0.5:
            tri = tri[::-1].copy()          # clockwise input -> exercises order_ccw
        cases.append((tri.astype(np.float32), pm))
    # 2. big triangles fully covering the pixel
    for _ in range(10):
        pm = rng.randint(0, 20, size=2).astype(np.float32)
        tri = np.array([[pm[0] - 30, pm[1] - 20], [pm[0] + 40, pm[1] - 25], [pm[0] + 2, pm[1] + 50]], np.float32)
        tri += rng.uniform(-1, 1, size=(3, 2)).astype(np.float32)
        cases.append((tri, pm))
    # 3. far away (no overlap), incl. bbox-overlapping-but-outside
    for _ in range(20):
        pm = rng.randint(0, 20, size=2).astype(np.float32)
        off = rng.uniform(1.5, 6.0, size=2) * rng.choice([-1, 1], size=2)
        tri = (pm + off + rng.uniform(-1.0, 1.0, size=(3, 2))).astype(np.float32)
        cases.append((tri, pm))
    for _ in range(10):   # thin diagonal sliver whose bbox contains the pixel
        pm = rng.randint(2, 18, size=2).astype(np.float32)
        d = rng.uniform(1.2, 2.0)
        tri = np.array([[pm[0] - 3, pm[1] - 3 + d + 1], [pm[0] + 4, pm[1] + 4 + d + 1],
                        [pm[0] - 3, pm[1] + 4 + d + 2]], np.float32)
        cases.append((tri, pm))
    # 4. triangle entirely inside the pixel
    for _ in range(10):
        pm = rng.randint(0, 20, size=2).astype(np.float32)
        tri = (pm + rng.uniform(0.1, 0.9, size=(3, 2))).astype(np.float32)
        cases.append((tri, pm))
    # 5. axis-aligned edges (edges_iszero true) crossing the pixel
    for _ in range(20):
        pm = rng.randint(2, 18, size=2).astype(np.float32)
        x0 = pm[0] + rng.uniform(0.1, 0.9)
        y0 = pm[1] + rng.uniform(0.1, 0.9)
        tri = np.array([[x0, pm[1] - 2.3], [x0, pm[1] + 3.1], [x0 + rng.uniform(1.5, 4), y0]], np.float32)
        if rng.uniform() < 0.5:
            tri = tri[:, ::-1].copy()
        cases.append((tri, pm))
    # 6. error codes: edge through a pixel corner exactly (E00), vertex on a
    #    pixel edge / corner, degenerate triangles
    for _ in range(8):
        pm = rng.randint(2, 18, size=2).astype(np.float32)
        tri = np.array([[pm[0] - 2, pm[1] - 2], [pm[0] + 3, pm[1] + 3], [pm[0] - 2, pm[1] + 5]], np.float32)
        cases.append((tri, pm))
    for _ in range(6):
        pm = rng.randint(2, 18, size=2).astype(np.float32)
        tri = np.array([[pm[0], pm[1] + 0.5], [pm[0] + 3, pm[1] - 1.25], [pm[0] + 2.5, pm[1] + 4]], np.float32)
        cases.append((tri, pm))
    for _ in range(6):
        pm = rng.randint(2, 18, size=2).astype(np.float32)
        tri = np.array([[pm[0] + 0.5, pm[1] + 0.5], [pm[0] + 0.5, pm[1] + 0.5], [pm[0] + 2.5, pm[1] + 4]], np.float32)
        cases.append((tri, pm))
    for _ in range(6):   # vertex exactly on a pixel corner
        pm = rng.randint(2, 18, size=2).astype(np.float32)
        tri = np.array([[pm[0] + 1, pm[1] + 1], [pm[0] - 3, pm[1] + 0.25], [pm[0] + 0.25, pm[1] - 3]], np.float32)
        cases.append((tri, pm))
    return cases


def aa_error_candidates(rng):
    """Endless stream of (tri, pixmin, family) near the clipper's tie cases: families aimed at the reference's
    exceptions E00-E05 (pyrenderer.py:274,294,364,380,391,423 <-> aa.h:265,301,325,355,398/411,437)."""
    def jit(x, k=3):             # move a float32 by up to k ulps
        x = np.float32(x)
        for _ in range(rng.randint(0, k + 1)):
            x = np.nextafter(x, np.float32(np.inf) if rng.rand() < 0.5 else np.float32(-np.inf))
        return x
    it = 0
    while True:
        fam = it % 7
        it += 1
        if fam == 6:
            # E01: a long edge past a pixel corner at small coordinates: the rounding of t * e (|e| ~ 100) exceeds the
            # ulp of the crossing, so the crossings with BOTH pixel edges at that corner can test valid
            pm = rng.randint(1, 8, size=2).astype(np.float32)
            cx, cy = pm[0] + rng.randint(0, 2), pm[1] + rng.randint(0, 2)
            d = rng.uniform(-1, 1, 2); d /= np.linalg.norm(d)
            L = rng.uniform(10, 300)
            nrm = np.array([-d[1], d[0]])
            off = nrm * rng.uniform(-2e-6, 2e-6)
            a = np.array([cx, cy]) + off - d * L * rng.uniform(0.2, 1); b = np.array([cx, cy]) + off + d * L * rng.uniform(0.2, 1)
            c = np.array([cx, cy]) + nrm * rng.uniform(1, 40) * rng.choice([-1, 1])
            yield np.array([a, b, c], np.float32), pm, fam
            continue
        pm = rng.randint(2, 60, size=2).astype(np.float32)
        cx, cy = pm[0] + rng.randint(0, 2), pm[1] + rng.randint(0, 2)       # a pixel corner
        if fam == 0:      # edge through (almost) a pixel corner, generic direction            -> E00, E04, E05
            d = rng.uniform(-1, 1, 2); d /= np.linalg.norm(d)
            a = np.array([cx, cy]) - d * rng.uniform(0.5, 4); b = np.array([cx, cy]) + d * rng.uniform(0.5, 4)
            a = [jit(a[0]), jit(a[1])]; b = [jit(b[0]), jit(b[1])]
            c = np.array([cx, cy]) + np.array([-d[1], d[0]]) * rng.uniform(1, 4) * rng.choice([-1, 1])
            tri = [a, b, c]
        elif fam == 1:    # near-vertical edge (|e.x| < 1e-3: "iszero") across an x = const pixel edge   -> E02, E03
            x = cx + rng.uniform(-5e-4, 5e-4); dx = rng.uniform(-9e-4, 9e-4)
            tri = [[x, pm[1] + rng.uniform(-2, 0.9)], [x + dx, pm[1] + rng.uniform(0.1, 3)],
                   [x + rng.uniform(1, 4) * rng.choice([-1, 1]), pm[1] + rng.uniform(-1, 2)]]
        elif fam == 2:    # near-horizontal edge across a y = const pixel edge                  -> E02, E03
            y = cy + rng.uniform(-5e-4, 5e-4); dy = rng.uniform(-9e-4, 9e-4)
            tri = [[pm[0] + rng.uniform(-2, 0.9), y], [pm[0] + rng.uniform(0.1, 3), y + dy],
                   [pm[0] + rng.uniform(-1, 2), y + rng.uniform(1, 4) * rng.choice([-1, 1])]]
        elif fam == 3:    # sliver hugging a pixel edge                                           -> E00, E03, E04
            y = cy + rng.uniform(-2e-6, 2e-6)
            tri = [[pm[0] - rng.uniform(0.2, 2), jit(y)], [pm[0] + 1 + rng.uniform(0.2, 2), jit(y)],
                   [pm[0] + rng.uniform(0, 1), jit(y + rng.uniform(-3e-6, 3e-6))]]
        elif fam == 4:    # triangle corner (almost) on a pixel corner                            -> E00, E04, E05
            tri = [[jit(cx), jit(cy)], [cx + rng.uniform(-3, 3), cy + rng.uniform(-3, 3)], [cx + rng.uniform(-3, 3), cy + rng.uniform(-3, 3)]]
        else:             # edge along the pixel diagonal through two corners
            tri = [[jit(pm[0] - 1), jit(pm[1] - 1)], [jit(pm[0] + 2), jit(pm[1] + 2)], [pm[0] + rng.uniform(-3, 3), pm[1] + rng.uniform(2, 5)]]
        yield np.array(tri, np.float32), pm, fam


def aa_error_cases(ref, per_code=8, seed=4321):
    """Search the candidate stream for inputs on which the reference's analytic clipper raises each of
    "[pyrasterizer] Error code 00" .. "05" (the message it prints is the label); ``per_code`` inputs per code,
    plus, per family, the same number of near-tie inputs it accepts.  The search calls the reference only."""
    import contextlib
    import io
    pr = ref.pyrenderer
    rng = np.random.RandomState(seed)
    want = {"[pyrasterizer] Error code %02d" % k: per_code for k in range(6)}
    ok_left = {fam: per_code for fam in range(7)}
    cases = []
    tried = 0
    for tri, pm, fam in aa_error_candidates(rng):
        tried += 1
        if tried > 400000 or (not any(want.values()) and not any(ok_left.values())):
            break
        if fam != 6 and not any(want[m] for m in want if not m.endswith("01")) and not ok_left[fam]:
            continue
        tv = torch.from_numpy(tri.copy()).unsqueeze(0)
        tris = pr.Triangles(tv[:, 0], tv[:, 1], tv[:, 2])
        pixs = pr.Pixels(torch.from_numpy(pm).unsqueeze(0), torch.from_numpy(pm + 1.0).unsqueeze(0))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            pr.tri_pixel_overlap_area(tris, pixs, 0, 0, use_autograd=False)
        msg = buf.getvalue().strip()
        if msg:
            if want.get(msg, 0) > 0:
                want[msg] -= 1
                cases.append((tri, pm))
        elif ok_left[fam] > 0:
            ok_left[fam] -= 1
            cases.append((tri, pm))
    missing = {m: k for m, k in want.items() if k}
    assert not missing, f"no input found for {missing} in {tried} candidates"
    return cases


def make_aa(ref, cases=None, name="aa_pairs.npz"):
    pr = ref.pyrenderer
    if cases is None:
        cases = aa_cases(np.random.RandomState(1234))
    n = len(cases)
    tri_in = np.stack([c[0] for c in cases]).astype(np.float32)     # (n,3,2) as given
    pixmin = np.stack([c[1] for c in cases]).astype(np.float32)     # (n,2)
    tv = torch.from_numpy(tri_in.copy())
    # Triangles() reorders its inputs in place (pyrenderer.py:521-529)
    tris = pr.Triangles(tv[:, 0], tv[:, 1], tv[:, 2])
    pixs = pr.Pixels(torch.from_numpy(pixmin), torch.from_numpy(pixmin + 1.0))
    out = dict(
        tri_in=tri_in, pixmin=pixmin,
        t_verts=t2n(tris.verts), t_edges=t2n(tris.edges),
        t_edges_iszero=t2n(tris.edges_iszero), t_edges_recip=t2n(tris.edges_recip),
        t_edges_normal=t2n(tris.edges_normal), t_edges_normal_c=t2n(tris.edges_normal_c),
    )
    area_an = np.zeros(n, np.float32)
    grad_an = np.zeros((n, 3, 2), np.float32)
    err_an = np.zeros(n, np.int32)          # 1 = the reference raised ValueError
    msg_an = []
    area_ag = np.zeros(n, np.float32)
    grad_ag = np.zeros((n, 3, 2), np.float32)
    err_ag = np.zeros(n, np.int32)
    npoly = np.zeros(n, np.int32)
    import contextlib
    import io
    for i in range(n):
        # analytic flavour (gradient hand-derived, same algebra as aa.h:151-441)
        tris.verts.requires_grad_(False)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            a, poly = pr.tri_pixel_overlap_area(tris, pixs, i, i, use_autograd=False)
        msg = buf.getvalue().strip()
        msg_an.append(msg)
        if msg:
            err_an[i] = 1
        area_an[i] = float(a)
        npoly[i] = len(poly)
        # autograd flavour
        v = tris.verts.detach().clone().requires_grad_(True)
        t2 = types.SimpleNamespace(
            verts=v, edges=torch.stack([v[:, 1] - v[:, 0], v[:, 2] - v[:, 1], v[:, 0] - v[:, 2]], dim=1),
            edges_iszero=tris.edges_iszero, edges_recip=None,
            edges_normal=tris.edges_normal, edges_normal_c=tris.edges_normal_c)
        t2.edges_recip = 1.0 / t2.edges
        try:
            a2, poly2 = pr.tri_pixel_overlap_area(t2, pixs, i, i, use_autograd=True)
            if torch.is_tensor(a2) and a2.requires_grad:
                a2.backward()
                grad_ag[i] = t2n(v.grad[i])
            area_ag[i] = float(a2)
        except ValueError:
            err_ag[i] = 1
    # analytic gradient: run the inner analytic routine once more to read it out
    for i in range(n):
        if err_an[i]:
            continue
        tvs = tris.verts[i]
        pix_verts = pixs.verts[i]
        inside = [True] * 4
        outside_any = False
        all_inside = True
        for ti in range(3):
            all_out = True
            for pvi in range(4):
                ins = bool(pr.is_vert_inside_triangle_edge(pix_verts[pvi], tris.edges_normal[i][ti], tris.edges_normal_c[i][ti]))
                all_out = all_out and (not ins)
                all_inside = all_inside and ins
                inside[pvi] = inside[pvi] and ins
            if all_out:
                outside_any = True
                break
        if outside_any or all_inside:
            continue
        try:
            a, poly, g = pr._tri_pixel_overlap_area_analytic(
                tvs, tris.edges[i], tris.edges_iszero[i], tris.edges_recip[i], pix_verts, inside, 1.0)
            grad_an[i] = t2n(g)
            assert abs(float(a) - area_an[i]) < 1e-6
        except ValueError:
            raise AssertionError("analytic flavour raised although the first pass did not")
    out.update(area_analytic=area_an, grad_analytic=grad_an, err_analytic=err_an,
               area_autograd=area_ag, grad_autograd=grad_ag, err_autograd=err_ag,
               npoly=npoly, msg_analytic=np.array(msg_an))
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, ":", n, "cases; errors", int(err_an.sum()), "partial", int(((area_an > 0) & (area_an < 1)).sum()),
          "messages", dict(zip(*np.unique(np.array(msg_an), return_counts=True))))


# --------------------------------------------------------------------------
# boundary captures
# --------------------------------------------------------------------------
ARG_NAMES = [
    "background", "patch_min", "patch_width", "patch_height", "verts", "faces", "verts_color",
    "faces_opacity", "verts_ndc", "verts_image", "faces_intense", "aa_temperature", "aa_face_verts",
    "aa_face_edges", "aa_face_edges_iszero", "aa_face_edges_recip", "aa_face_edges_normal",
    "aa_face_edges_normal_c", "len_oarea_buffer", "image_ray_o", "image_ray_d"]


def make_boundary(ref, stub, scenes, name, W, H, F, seed, cams, batch_idx, patch_min, pw, ph, temp, K):
    sc = scenes.triangle_soup(W, H, F, seed, num_cams=cams, shared_verts=True)
    captured = {}

    def fake_forward(*args):
        assert len(args) == 21
        for k, a in zip(ARG_NAMES, args):
            captured["arg_" + k] = t2n(a)
        B = args[8].shape[0]
        g = torch.Generator().manual_seed(seed + 7)
        color = torch.rand((B, ph, pw, 3), generator=g)
        depth = torch.rand((B, ph, pw), generator=g) * 2 - 1
        captured["stub_color"], captured["stub_depth"] = t2n(color), t2n(depth)
        e = torch.zeros(0)
        return 0, color, depth, e, e, e, e, e, e, e

    stub.render_forward_cuda = fake_forward
    r = ref.Renderer(sc.mv, sc.proj, W, H, "cpu", aa_grad_buffer_size=K)
    pmin = torch.tensor(patch_min, dtype=torch.int64)
    color, depth = r(batch_idx, pmin, pw, ph, sc.verts.clone(), sc.faces, sc.verts_color,
                     sc.faces_opacity, sc.faces_intense[batch_idx], sc.background, aa_temperature=temp)
    out = dict(width=W, height=H, num_faces=F, seed=seed, num_cams=cams,
               batch_idx=np.array(batch_idx), in_patch_min=np.array(patch_min), in_K=K,
               in_mv=t2n(sc.mv), in_proj=t2n(sc.proj), in_verts=t2n(sc.verts), in_faces=t2n(sc.faces),
               in_verts_color=t2n(sc.verts_color), in_faces_opacity=t2n(sc.faces_opacity),
               in_faces_intense=t2n(sc.faces_intense), in_background=t2n(sc.background),
               full_ray_o=t2n(r.ray_o), full_ray_d=t2n(r.ray_d),
               out_color=t2n(color), out_depth=t2n(depth))
    out.update(captured)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 0 and k.startswith("arg_")})


LAYER_ARGS = ["width", "height", "verts", "faces", "tets", "face_tets", "tet_faces", "face_existence",
              "verts_ndc", "verts_image", "image_ray_o", "image_ray_d", "num_layers"]


def make_layers_boundary(ref, stub, scenes):
    W, H = 40, 24
    sc = scenes.tet_lattice(W, H, 2, seed=scenes.SEED_BASE + 3, num_cams=2)
    captured = {}

    def fake_layers(*args):
        assert len(args) == 13
        for k, a in zip(LAYER_ARGS, args):
            captured["arg_" + k] = t2n(a)
        B = args[8].shape[0]
        return torch.zeros((B, H, W, args[-1]), dtype=torch.int32), torch.zeros((B, H, W), dtype=torch.int32)

    stub.generate_render_layers_cuda = fake_layers
    lr = ref.LayeredRenderer(sc.mv, sc.proj, W, H, "cpu")
    lr.generate([1, 0], sc.verts, sc.faces, sc.tets, sc.face_tets, sc.tet_faces, sc.faces_existence, 3)
    out = dict(in_mv=t2n(sc.mv), in_proj=t2n(sc.proj), batch_idx=np.array([1, 0]), n=2, seed=scenes.SEED_BASE + 3)
    out.update(captured)
    np.savez_compressed(os.path.join(HERE, "boundary_layers.npz"), **out)
    print("boundary_layers", {k: getattr(v, "shape", None) for k, v in captured.items()})


def main():
    scenes = load_scenes()
    ref, stub = import_reference()
    if "--only-aa-errors" in sys.argv:
        make_aa(ref, aa_error_cases(ref), "aa_error_pairs.npz")
        return
    make_aa(ref)
    make_aa(ref, aa_error_cases(ref), "aa_error_pairs.npz")
    make_boundary(ref, stub, scenes, "boundary_full.npz", 48, 32, 60, scenes.SEED_BASE + 11,
                  cams=2, batch_idx=[0, 1], patch_min=[[0, 0], [0, 0]], pw=48, ph=32, temp=1.0, K=20)
    make_boundary(ref, stub, scenes, "boundary_patch.npz", 48, 32, 60, scenes.SEED_BASE + 12,
                  cams=3, batch_idx=[2, 0], patch_min=[[16, 8], [5, 3]], pw=24, ph=20, temp=0.5, K=4)
    make_layers_boundary(ref, stub, scenes)


if __name__ == "__main__":
    main()
