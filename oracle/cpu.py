"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

ctypes front-end of the CPU oracle (oracle/dm2_oracle.cpp).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` import
this module; the product package ``dmesh2_renderer_amd`` never does.

The three entry points mirror the reference's ``_C`` functions (ext.cpp:6-9):
``render_forward_cuda`` (21 positional args, render.h:13-45),
``render_backward_cuda`` (render.h:48-94) and ``generate_render_layers_cuda``
(render.h:102-119), on numpy arrays.  Parity status: see dm2_oracle_math.hpp.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_c = ctypes


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "build", "libdm2_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("dm2_oracle.cpp", "dm2_oracle_prep.cpp", "dm2_oracle_math.hpp", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # DM2_ORACLE_LIB: an alternative build of the same sources, e.g. build/libdm2_oracle_asan.so from `make asan`
        # (run as  LD_PRELOAD=$(g++ -print-file-name=libasan.so) DM2_ORACLE_LIB=... python -m pytest tests -m "not gpu")
        _LIB = ctypes.CDLL(os.environ.get("DM2_ORACLE_LIB") or build())
        _LIB.orc_binning_create.restype = _c.c_void_p
        _LIB.orc_binning_num_rendered.restype = _c.c_int64
        _LIB.orc_binning_num_rendered.argtypes = [_c.c_void_p]
        _LIB.orc_binning_free.argtypes = [_c.c_void_p]
        _LIB.orc_higher_msb.restype = _c.c_uint32
    return _LIB


def max_threads() -> int:
    return int(lib().orc_max_threads())


def _np(x, dtype):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(_c.c_void_p)


# --------------------------------------------------------------------------
# host-side AA tables (numpy restatement of pyrenderer.py:6-30, 521-535)
# --------------------------------------------------------------------------
def aa_tables(tri_verts, dtype=np.float32, reorder=True):
    """tri_verts (...,3,2) -> dict of the six per-triangle tables."""
    v = np.array(tri_verts, dtype=dtype, copy=True)
    p0, p1, p2 = v[..., 0, :], v[..., 1, :], v[..., 2, :]
    if reorder:
        half = dtype(0.5)
        area = half * ((p1[..., 0] - p0[..., 0]) * (p2[..., 1] - p0[..., 1]) - (p2[..., 0] - p0[..., 0]) * (p1[..., 1] - p0[..., 1]))
        swap = area < 0
        p1s = np.where(swap[..., None], p2, p1)
        p2s = np.where(swap[..., None], p1, p2)
        p1, p2 = p1s, p2s
    verts = np.stack([p0, p1, p2], axis=-2)
    edges = np.stack([p1 - p0, p2 - p1, p0 - p2], axis=-2)
    with np.errstate(divide="ignore"):
        recip = (dtype(1.0) / edges).astype(dtype)
    iszero = np.abs(edges) < dtype(1e-3)
    normal = np.stack([-edges[..., 1], edges[..., 0]], axis=-1)
    starts = np.stack([p0, p1, p2], axis=-2)
    normal_c = (normal * starts).sum(axis=-1).astype(dtype)
    return dict(verts=verts.astype(dtype), edges=edges.astype(dtype), iszero=iszero,
                recip=recip, normal=normal.astype(dtype), normal_c=normal_c)


def prepare_faces(verts, faces, mv, proj, width, height, dtype=np.float32):
    """Host prep of Renderer.forward (reference __init__.py:239-262, pyrenderer.py:6-30) for the cameras
    mv/proj (B,4,4): -> dict(verts_ndc (B,P,3), verts_image (B,P,2), verts/edges/recip/normal (B,F,3,2),
    iszero (B,F,3,2) bool, normal_c (B,F,3))."""
    suf = "f32" if dtype == np.float32 else "f64"
    v = _np(verts, dtype); fc = _np(faces, np.int32); m = _np(mv, dtype); pr = _np(proj, dtype)
    B, P, F = m.shape[0], v.shape[0], fc.shape[0]
    out = dict(verts_ndc=np.zeros((B, P, 3), dtype), verts_image=np.zeros((B, P, 2), dtype),
               verts=np.zeros((B, F, 3, 2), dtype), edges=np.zeros((B, F, 3, 2), dtype),
               iszero=np.zeros((B, F, 3, 2), np.uint8), recip=np.zeros((B, F, 3, 2), dtype),
               normal=np.zeros((B, F, 3, 2), dtype), normal_c=np.zeros((B, F, 3), dtype))
    with np.errstate(divide="ignore"):
        getattr(lib(), "orc_prepare_" + suf)(B, P, F, int(width), int(height), _p(v), _p(fc), _p(m), _p(pr),
                                             _p(out["verts_ndc"]), _p(out["verts_image"]), _p(out["verts"]), _p(out["edges"]),
                                             _p(out["iszero"]), _p(out["recip"]), _p(out["normal"]), _p(out["normal_c"]))
    out["iszero"] = out["iszero"].astype(bool)
    return out


def prepare_faces_backward(verts, faces, mv, proj, width, height, g_ndc=None, g_image=None, g_aa=None, dtype=np.float32):
    """d(verts) (P,3) that torch autograd sends back through the host prep for upstream gradients of
    verts_ndc (B,P,3), verts_image (B,P,2) and aa_face_verts (B,F,3,2)."""
    suf = "f32" if dtype == np.float32 else "f64"
    v = _np(verts, dtype); fc = _np(faces, np.int32); m = _np(mv, dtype); pr = _np(proj, dtype)
    B, P, F = m.shape[0], v.shape[0], fc.shape[0]
    gn = None if g_ndc is None else _np(g_ndc, dtype)
    gi = None if g_image is None else _np(g_image, dtype)
    ga = None if g_aa is None else _np(g_aa, dtype)
    out = np.zeros((P, 3), dtype)
    getattr(lib(), "orc_prepare_backward_" + suf)(B, P, F, int(width), int(height), _p(v), _p(fc), _p(m), _p(pr),
                                                  _p(gn), _p(gi), _p(ga), _p(out))
    return out


def analytic_rays(mv, proj, width, height, dtype=np.float32):
    """Per-pixel primary rays of cameras mv / proj (B,4,4) in closed form: the reference's Renderer._init_rays
    (__init__.py:198-237) -- pixel centre -> NDC (x, y, -1, 1) @ inv(proj)^T @ inv(mv)^T without perspective divide,
    direction normalised with + 1e-6 on the length -- with the 4-term sums taken in index order (the reference leaves
    that order to its BLAS).  -> ray_o, ray_d (B,H,W,3).  The HIP kernels evaluate exactly this per pixel
    (DM2_FLAG_ANALYTIC_RAYS, csrc/dm2_device_math.h analytic_ray)."""
    f = dtype
    imv = np.linalg.inv(np.asarray(mv, np.float64)).astype(f) if dtype == np.float32 else np.linalg.inv(np.asarray(mv, np.float64))
    ipr = np.linalg.inv(np.asarray(proj, np.float64)).astype(f) if dtype == np.float32 else np.linalg.inv(np.asarray(proj, np.float64))
    return analytic_rays_from_inverse(imv, ipr, width, height, dtype)


def analytic_rays_from_inverse(imv, ipr, width, height, dtype=np.float32):
    """Same, from inv(mv), inv(proj) as given (B,4,4) -- what `Renderer(analytic_rays=True)` hands to the kernels."""
    f = dtype
    imv = np.asarray(imv, f); ipr = np.asarray(ipr, f)
    B = imv.shape[0]
    xs = np.arange(width, dtype=f); ys = np.arange(height, dtype=f)
    hx = ((xs + f(0.5)) / f(width) * f(2.0)) - f(1.0)
    hy = ((ys + f(0.5)) / f(height) * f(2.0)) - f(1.0)
    h = [np.broadcast_to(hx[None, None, :], (B, height, width)), np.broadcast_to(hy[None, :, None], (B, height, width)), f(-1.0), f(1.0)]
    e = lambda m, j, k: m[:, j, k][:, None, None]
    v = [((h[0] * e(ipr, j, 0) + h[1] * e(ipr, j, 1)).astype(f) + (h[2] * e(ipr, j, 2)).astype(f)).astype(f) + (h[3] * e(ipr, j, 3)).astype(f)
         for j in range(4)]
    v = [x.astype(f) for x in v]
    w = [((v[0] * e(imv, j, 0) + v[1] * e(imv, j, 1)).astype(f) + (v[2] * e(imv, j, 2)).astype(f)).astype(f) + (v[3] * e(imv, j, 3)).astype(f)
         for j in range(3)]
    ro = np.stack([np.broadcast_to(e(imv, j, 3), (B, height, width)) for j in range(3)], axis=-1).astype(f)
    d = np.stack([w[j].astype(f) - ro[..., j] for j in range(3)], axis=-1).astype(f)
    ln = (np.sqrt(((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(f) + (d[..., 2] * d[..., 2]).astype(f)).astype(f)) + f(1e-6)).astype(f)
    rd = (d / ln[..., None]).astype(f)
    return np.ascontiguousarray(ro), np.ascontiguousarray(rd)


def aa_overlap(tables, idx, pixmin, dtype=np.float32):
    """Overlap of triangle ``idx`` of ``tables`` with the unit pixel at pixmin."""
    suf, ct = ("f32", _c.c_float) if dtype == np.float32 else ("f64", _c.c_double)
    fn = getattr(lib(), "orc_aa_overlap_" + suf)
    tv = _np(tables["verts"][idx], dtype); te = _np(tables["edges"][idx], dtype)
    tz = _np(tables["iszero"][idx], np.uint8); tr = _np(tables["recip"][idx], dtype)
    tn = _np(tables["normal"][idx], dtype); tc = _np(tables["normal_c"][idx], dtype)
    area = np.zeros(1, dtype); grad = np.zeros((3, 2), dtype)
    fn.restype = _c.c_int
    code = fn(_p(tv), _p(te), _p(tz), _p(tr), _p(tn), _p(tc), ct(float(pixmin[0])), ct(float(pixmin[1])), _p(area), _p(grad))
    return float(area[0]), grad, int(code)


def ray_tri(ro, rd, p, dtype=np.float32, corrected=False):
    suf = "f32" if dtype == np.float32 else "f64"
    fn = getattr(lib(), "orc_ray_tri_" + suf)
    tuv = np.zeros(3, dtype); grads = np.zeros((6, 3), dtype)
    ok = fn(_p(_np(ro, dtype)), _p(_np(rd, dtype)), _p(_np(p, dtype)), _p(tuv), _p(grads), int(corrected))
    return bool(ok), tuv, grads


def clamp_bary(u, v, dtype=np.float32):
    suf, ct = ("f32", _c.c_float) if dtype == np.float32 else ("f64", _c.c_double)
    out = np.zeros(6, dtype)
    code = getattr(lib(), "orc_clamp_bary_" + suf)(ct(u), ct(v), _p(out))
    return int(code), out


def patch_rect(pm, tri, grid):
    out = np.zeros(4, np.uint32)
    lib().orc_patch_rect(_c.c_uint32(pm[0]), _c.c_uint32(pm[1]), _p(_np(tri, np.float32)),
                         _c.c_uint32(grid[0]), _c.c_uint32(grid[1]), _p(out))
    return out


def higher_msb(n):
    return int(lib().orc_higher_msb(_c.c_uint32(n)))


# --------------------------------------------------------------------------
# binning
# --------------------------------------------------------------------------
class Binning:
    def __init__(self, B, P, F, W, H, patch_min, faces, verts_ndc, verts_image, key_min_depth=False):
        self.B, self.P, self.F, self.W, self.H = B, P, F, W, H
        self.gx, self.gy = (W + 15) // 16, (H + 15) // 16
        self._pm = _np(patch_min, np.int32); self._faces = _np(faces, np.int32)
        self._ndc = _np(verts_ndc, np.float32); self._img = _np(verts_image, np.float32)
        self._h = lib().orc_binning_create(B, P, F, W, H, _p(self._pm), _p(self._faces), _p(self._ndc), _p(self._img), int(key_min_depth))
        self.num_rendered = int(lib().orc_binning_num_rendered(self._h))
        BF, R, Tn = B * F, self.num_rendered, B * self.gx * self.gy
        self.depths = np.zeros(BF, np.float32); self.min_depths = np.zeros(BF, np.float32)
        self.max_depths = np.zeros(BF, np.float32); self.tiles_touched = np.zeros(BF, np.uint32)
        self.keys = np.zeros(R, np.uint64); self.face_list = np.zeros(R, np.uint32)
        self.ranges = np.zeros((Tn, 2), np.uint32)
        lib().orc_binning_copy(_c.c_void_p(self._h), _p(self.depths), _p(self.min_depths), _p(self.max_depths),
                               _p(self.tiles_touched), _p(self.keys), _p(self.face_list), _p(self.ranges))

    @property
    def handle(self):
        return _c.c_void_p(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_binning_free(_c.c_void_p(self._h)); self._h = None
        except Exception:
            pass


# --------------------------------------------------------------------------
# render forward / backward, `_C`-shaped
# --------------------------------------------------------------------------
class ForwardResult:
    """Outputs of render_forward_cuda plus the internal state backward needs."""


def _common(args, dtype):
    (background, patch_min, pw, ph, verts, faces, verts_color, faces_opacity, verts_ndc, verts_image,
     faces_intense, temp, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, K, ray_o, ray_d) = args
    d = dict(
        background=_np(background, dtype), patch_min=_np(patch_min, np.int32), W=int(pw), H=int(ph),
        verts=_np(verts, dtype), faces=_np(faces, np.int32), verts_color=_np(verts_color, dtype),
        faces_opacity=_np(faces_opacity, dtype), verts_ndc=_np(verts_ndc, dtype), verts_image=_np(verts_image, dtype),
        faces_intense=_np(faces_intense, dtype), temp=float(temp), aa_v=_np(aa_v, dtype), aa_e=_np(aa_e, dtype),
        aa_z=_np(aa_z, np.uint8), aa_r=_np(aa_r, dtype), aa_n=_np(aa_n, dtype), aa_c=_np(aa_c, dtype),
        K=int(K), ray_o=_np(ray_o, dtype), ray_d=_np(ray_d, dtype))
    d["B"] = d["verts_ndc"].shape[0]; d["P"] = d["verts"].shape[0]; d["F"] = d["faces"].shape[0]
    if not (0.0 <= d["temp"] <= 1.0):
        raise RuntimeError("aa_temperature must be in the range [0, 1]")
    if d["K"] < 0:
        raise RuntimeError("len_oarea_buffer must be non-negative")
    if d["temp"] == 0.0:
        d["K"] = 0                                               # render.cu:141-142
    return d


def _call_args(d):
    return [d["B"], d["P"], d["F"], d["W"], d["H"], d["K"], _c.c_double(d["temp"]), _p(d["patch_min"]),
            _p(d["background"]), _p(d["verts"]), _p(d["faces"]), _p(d["verts_color"]), _p(d["faces_opacity"]),
            _p(d["verts_ndc"]), _p(d["faces_intense"]), _p(d["aa_v"]), _p(d["aa_e"]), _p(d["aa_z"]), _p(d["aa_r"]),
            _p(d["aa_n"]), _p(d["aa_c"]), _p(d["ray_o"]), _p(d["ray_d"])]


def render_forward_cuda(*args, dtype=np.float32, nthreads=1):
    """Oracle twin of _C.render_forward_cuda (render.cu:28-195); 21 positional args."""
    assert len(args) == 21
    d = _common(args, dtype)
    B, P, F, W, H, K = d["B"], d["P"], d["F"], d["W"], d["H"], d["K"]
    res = ForwardResult()
    res.d = d
    N = B * H * W
    res.color = np.zeros((B, H, W, 3), dtype); res.depth = np.zeros((B, H, W), dtype)
    res.final_T = np.zeros(N, dtype); res.final_prev_T = np.zeros(N, dtype); res.n_contrib = np.zeros(N, np.uint32)
    res.buf_oarea = np.zeros((B, H, W, K), dtype); res.buf_tri_id = np.zeros((B, H, W, K), np.int32)
    res.buf_tri_cnt = np.zeros((B, H, W), np.int32); res.buf_doarea = np.zeros((B, H, W, K, 3, 2), dtype)
    res.num_rendered = 0
    res.binning = None
    if P != 0:                                                   # render.cu:149
        bn = Binning(B, P, F, W, H, d["patch_min"], d["faces"], d["verts_ndc"], d["verts_image"])
        res.binning = bn
        res.num_rendered = bn.num_rendered
        suf = "f32" if dtype == np.float32 else "f64"
        getattr(lib(), "orc_render_forward_" + suf)(
            *_call_args(d), _p(bn.ranges), _p(bn.face_list), _p(res.color), _p(res.depth), _p(res.final_T),
            _p(res.final_prev_T), _p(res.n_contrib), _p(res.buf_oarea), _p(res.buf_tri_id), _p(res.buf_tri_cnt),
            _p(res.buf_doarea), int(nthreads))
    return res


def render_backward_cuda(fwd: ForwardResult, dL_dcolor, dL_ddepth, corrected_dv=False, nthreads=1):
    """Oracle twin of _C.render_backward_cuda (render.cu:198-373) given the forward's state."""
    d = fwd.d
    dtype = d["verts"].dtype.type
    B, P, F = d["B"], d["P"], d["F"]
    g = dict(verts=np.zeros((P, 3), dtype), verts_color=np.zeros((P, 3), dtype), faces_opacity=np.zeros(F, dtype),
             verts_ndc=np.zeros((B, P, 3), dtype), faces_intense=np.zeros((B, F), dtype),
             aa_face_verts=np.zeros((B, F, 3, 2), dtype))
    if F != 0 and fwd.binning is not None:                       # render.cu:320
        bn = fwd.binning
        dc = _np(dL_dcolor, dtype); dd = _np(dL_ddepth, dtype)
        suf = "f32" if dtype == np.float32 else "f64"
        getattr(lib(), "orc_render_backward_" + suf)(
            *_call_args(d), _p(bn.ranges), _p(bn.face_list), _p(dc), _p(dd), _p(fwd.final_T), _p(fwd.final_prev_T),
            _p(fwd.n_contrib), _p(fwd.buf_oarea), _p(fwd.buf_tri_id), _p(fwd.buf_tri_cnt), _p(fwd.buf_doarea),
            _p(g["verts"]), _p(g["verts_color"]), _p(g["faces_opacity"]), _p(g["verts_ndc"]), _p(g["faces_intense"]),
            _p(g["aa_face_verts"]), int(corrected_dv), int(nthreads))
    return g


def generate_render_layers_cuda(width, height, verts, faces, tets, face_tets, tet_faces, face_existence,
                                verts_ndc, verts_image, ray_o, ray_d, num_layers, nthreads=1, return_first=False):
    """Oracle twin of _C.generate_render_layers_cuda (render.cu:378-476)."""
    if num_layers < 0:
        raise RuntimeError("num_layers must be non-negative")
    verts = _np(verts, np.float32); faces = _np(faces, np.int32); tets = _np(tets, np.int32)
    face_tets = _np(face_tets, np.int32); tet_faces = _np(tet_faces, np.int32); fe = _np(face_existence, np.int32)
    ndc = _np(verts_ndc, np.float32); img = _np(verts_image, np.float32)
    ro = _np(ray_o, np.float32); rd = _np(ray_d, np.float32)
    B, P, F, T = ndc.shape[0], verts.shape[0], faces.shape[0], tets.shape[0]
    W, H, L = int(width), int(height), int(num_layers)
    pm = np.zeros((B, 2), np.int32)                              # renderer.cu:557-558
    bn = Binning(B, P, F, W, H, pm, faces, ndc, img, key_min_depth=True)
    layers = np.full((B, H, W, L), -1, np.int32); cnt = np.zeros((B, H, W), np.int32)
    ff = np.full((B, H, W), -1, np.int32); ft = np.full((B, H, W), -1, np.int32)
    lib().orc_render_layers(B, P, F, T, W, H, _p(verts), _p(faces), _p(tets), _p(face_tets), _p(tet_faces), _p(fe),
                            _p(ro), _p(rd), bn.handle, L, _p(ff), _p(ft), _p(layers), _p(cnt), int(nthreads))
    if return_first:
        return layers, cnt, ff, ft, bn
    return layers, cnt
