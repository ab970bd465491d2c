"""``_C``-compatible shim over libdm2_hip.so (include/dm2_hip.h).

Exposes the three functions of the reference's pybind module
``dmesh2_renderer._C`` (ext.cpp:5-9) with identical positional arguments and
tuple returns:

    render_forward_cuda(...21 args...)  -> 10-tuple   (render.cu:28-195)
    render_backward_cuda(...31 args...) -> 6-tuple    (render.cu:198-373)
    generate_render_layers_cuda(...13 args...) -> 2-tuple (render.cu:378-476)

PyTorch is used only as the owner of device memory and of the current stream;
every computation happens in the HIP library.  There is NO fallback: if the
library is missing, or tensors are not on a ROCm device, a RuntimeError is
raised.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DM2_HIP_LIB") or os.path.join(_HERE, "csrc", "libdm2_hip.so")   # env override: A/B builds
_lib = None
_lock = threading.Lock()

DM2_FLAG_CORRECTED_DV = 1
DM2_FLAG_LEGACY_KERNELS = 2
DM2_FLAG_NO_BACKWARD = 4
DM2_FLAG_ANALYTIC_RAYS = 8
DM2_FLAG_AA_GRAD_TO_VERTS = 16
DM2_FLAG_TABLES_FROM_IMAGE = 32
DM2_FLAG_NO_PAIR_POOL = 64
SCRATCH_FACE, SCRATCH_IMAGE, SCRATCH_BINNING, SCRATCH_LAYER_IMAGE, SCRATCH_LAYER_TETS, SCRATCH_PAIR_POOL, SCRATCH_TIE_QUEUE = range(7)
# what a forward left for its backward (include/dm2_hip.h DM2_FWD_*)
FWD_UNKNOWN, FWD_NONE, FWD_MASKS, FWD_POOL, FWD_POINT = 0, 1, 2, 3, 4
ABI_VERSION = 6

# opt-in flags applied to every call (tests use this for the corrected-gradient mode)
_flags = 0
_bin_hint: dict = {}      # (device, B, W, H, F) -> bytes of binning scratch (+ pair pool) that held the last forward of that shape; under _lock

_vp, _i32, _i64, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t


class RenderDesc(ctypes.Structure):
    _fields_ = [
        ("B", _i32), ("P", _i32), ("F", _i32), ("W", _i32), ("H", _i32), ("K", _i32),
        ("aa_temperature", ctypes.c_float), ("flags", _i32), ("full_W", _i32), ("full_H", _i32),
        ("background", _vp), ("patch_min", _vp), ("verts", _vp), ("faces", _vp), ("verts_color", _vp),
        ("faces_opacity", _vp), ("verts_ndc", _vp), ("verts_image", _vp), ("faces_intense", _vp),
        ("aa_face_verts", _vp), ("aa_face_edges", _vp), ("aa_face_edges_iszero", _vp),
        ("aa_face_edges_recip", _vp), ("aa_face_edges_normal", _vp), ("aa_face_edges_normal_c", _vp),
        ("image_ray_o", _vp), ("image_ray_d", _vp), ("ray_cam", _vp),
    ]


class PrepDesc(ctypes.Structure):
    _fields_ = [
        ("B", _i32), ("P", _i32), ("F", _i32), ("W", _i32), ("H", _i32),
        ("verts", _vp), ("faces", _vp), ("mv", _vp), ("proj", _vp),
        ("verts_ndc", _vp), ("verts_image", _vp), ("aa_face_verts", _vp), ("aa_face_edges", _vp),
        ("aa_face_edges_iszero", _vp), ("aa_face_edges_recip", _vp), ("aa_face_edges_normal", _vp),
        ("aa_face_edges_normal_c", _vp),
    ]


class LayersDesc(ctypes.Structure):
    _fields_ = [
        ("B", _i32), ("P", _i32), ("F", _i32), ("T", _i32), ("W", _i32), ("H", _i32), ("L", _i32), ("flags", _i32),
        ("verts", _vp), ("faces", _vp), ("tets", _vp), ("face_tets", _vp), ("tet_faces", _vp),
        ("face_existence", _vp), ("verts_ndc", _vp), ("verts_image", _vp), ("image_ray_o", _vp), ("image_ray_d", _vp),
        ("ray_cam", _vp),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "dm2_abi_version": (ctypes.c_int, []),
    "dm2_last_error": (ctypes.c_char_p, []),
    "dm2_scratch_bytes": (_sz, [ctypes.c_int, _i64, _i64]),
    "dm2_forward_plan": (ctypes.c_int, [ctypes.POINTER(RenderDesc), _vp, _sz, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dm2_forward_run": (ctypes.c_int, [ctypes.POINTER(RenderDesc), _i64, _i64, _i64, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp,
                                       ctypes.POINTER(_i32)]),
    "dm2_forward": (ctypes.c_int, [ctypes.POINTER(RenderDesc), _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp,
                                   ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i32)]),
    "dm2_backward": (ctypes.c_int, [ctypes.POINTER(RenderDesc), _i64, _i32, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dm2_layers_plan": (ctypes.c_int, [ctypes.POINTER(LayersDesc), _vp, _sz, _vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dm2_layers_run": (ctypes.c_int, [ctypes.POINTER(LayersDesc), _i64, _i64, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "dm2_prepare_faces": (ctypes.c_int, [ctypes.POINTER(PrepDesc), _vp]),
    "dm2_prepare_faces_backward": (ctypes.c_int, [ctypes.POINTER(PrepDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "dm2_exchange_mark": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp, _vp, _vp]),
    "dm2_exchange_pack": (ctypes.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dm2_exchange_unpack": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dm2_debug_aa_overlap": (ctypes.c_int, [ctypes.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dm2_debug_fetch": (ctypes.c_int, [ctypes.c_int, _i64, _i64, _i64, _vp, _sz, _vp, _vp]),
    "dm2_profile_enable": (None, [ctypes.c_int]),
    "dm2_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_float), ctypes.c_int]),
    "dm2_debug_stamps": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int]),
}

STAGE_NAMES = ["preprocess_scan", "bin_scatter", "tile_sort", "tile_ranges", "forward_composite", "backward_composite", "backward_ties"]


def profile_enable(on: bool):
    load_library().dm2_profile_enable(1 if on else 0)


def profile_read():
    """Per-stage milliseconds of the calling thread's most recent forward/backward (see dm2_profile_read)."""
    buf = (ctypes.c_float * len(STAGE_NAMES))()
    n = load_library().dm2_profile_read(buf, len(STAGE_NAMES))
    return {STAGE_NAMES[i]: float(buf[i]) for i in range(n)}


def load_library(path: str | None = None):
    """dlopen libdm2_hip.so and bind every symbol include/dm2_hip.h declares."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise RuntimeError(
                f"dmesh2_renderer_amd: native library not found at {p}; build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C dmesh2_renderer_amd/csrc` "
                f"(there is no CPU fallback)")
        lib = ctypes.CDLL(p)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.dm2_abi_version() != ABI_VERSION:
            raise RuntimeError("dmesh2_renderer_amd: ABI version mismatch")
        if path is None:
            _lib = lib
        return lib


def set_flags(flags: int):
    """Set opt-in behaviour flags (DM2_FLAG_*) for subsequent calls; returns the old value."""
    global _flags
    old, _flags = _flags, int(flags)
    return old


def _err(lib, what):
    msg = lib.dm2_last_error()
    return RuntimeError(f"{what}: {msg.decode() if msg else 'unknown error'}")


def _require_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "dmesh2_renderer_amd: all tensors must live on a ROCm GPU (got a CPU tensor); "
                "this build has no CPU path -- the CPU restatement under oracle/ is test infrastructure only")
    dev = tensors[0].device
    for t in tensors:
        if t.device != dev:
            raise RuntimeError("dmesh2_renderer_amd: tensors are on different devices")
    return dev


def _c(t, dtype):
    if t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")      # packed_accessor64<T> throws likewise
    return t.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else 0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _bytes(dev, n):
    return torch.empty((max(int(n), 0),), dtype=torch.uint8, device=dev)


def _check_render_shapes(background, patch_min, verts, faces, verts_color, faces_opacity, verts_ndc, verts_image,
                         faces_intense, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, ray_o, ray_d, temp, K):
    # messages of render.cu:62-118
    def bad(cond, msg):
        if cond:
            raise RuntimeError(msg)
    bad(background.dim() != 1 or background.size(0) != 3, "background must have dimensions (3,)")
    bad(patch_min.dim() != 2 or patch_min.size(1) != 2, "patch_min must have dimensions (B, 2)")
    bad(verts.dim() != 2 or verts.size(1) != 3, "verts must have dimensions (P, 3)")
    bad(faces.dim() != 2 or faces.size(1) != 3, "faces must have dimensions (F, 3)")
    bad(verts_color.dim() != 2 or verts_color.size(1) != 3, "vert color must have dimensions (P, 3)")
    bad(faces_opacity.dim() != 1 or faces_opacity.size(0) != faces.size(0), "face opacity must have dimensions (F,)")
    bad(verts_ndc.dim() != 3 or verts_ndc.size(2) != 3, "verts_ndc must have dimensions (B, P, 3)")
    bad(verts_image.dim() != 3 or verts_image.size(2) != 2, "verts_image must have dimensions (B, P, 2)")
    bad(faces_intense.dim() != 2 or faces_intense.size(1) != faces.size(0), "faces_intense must have dimensions (B, F,)")
    for t, nm in ((aa_v, "aa_face_verts"), (aa_e, "aa_face_edges"), (aa_z, "aa_face_edges_iszero"),
                  (aa_r, "aa_face_edges_recip"), (aa_n, "aa_face_edges_normal")):
        bad(t.dim() != 4 or t.size(2) != 3 or t.size(3) != 2, f"{nm} must have dimensions (B, F, 3, 2)")
    bad(aa_c.dim() != 3 or aa_c.size(2) != 3, "aa_face_edges_normal_c must have dimensions (B, F, 3)")
    bad(ray_o.dim() != 4 or ray_o.size(3) != 3, "image_ray_o must have dimensions (B, H, W, 3)")
    bad(ray_d.dim() != 4 or ray_d.size(3) != 3, "image_ray_d must have dimensions (B, H, W, 3)")
    bad(temp < 0 or temp > 1, "aa_temperature must be in the range [0, 1]")
    bad(K < 0, "len_oarea_buffer must be non-negative")


def _make_desc(args, keep):
    (background, patch_min, pw, ph, verts, faces, verts_color, faces_opacity, verts_ndc, verts_image, faces_intense,
     temp, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, K, ray_o, ray_d) = args
    temp = float(temp); K = int(K); pw = int(pw); ph = int(ph)
    _check_render_shapes(background, patch_min, verts, faces, verts_color, faces_opacity, verts_ndc, verts_image,
                         faces_intense, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, ray_o, ray_d, temp, K)
    dev = _require_gpu(background, patch_min, verts, faces, verts_color, faces_opacity, verts_ndc, verts_image,
                       faces_intense, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, ray_o, ray_d)
    f32, i32 = torch.float32, torch.int32
    B, P, F = verts_ndc.size(0), verts.size(0), faces.size(0)
    # sizes the kernels index with (the reference's accessors would fault on a mismatch)
    def need(t, shape, nm):
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"{nm} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    need(patch_min, (B, 2), "patch_min"); need(verts_color, (P, 3), "verts_color"); need(verts_ndc, (B, P, 3), "verts_ndc")
    need(verts_image, (B, P, 2), "verts_image"); need(faces_intense, (B, F), "faces_intense")
    from_image = bool(getattr(_tls, "tables_from_image", False))
    if not from_image:
        for t, nm in ((aa_v, "aa_face_verts"), (aa_e, "aa_face_edges"), (aa_z, "aa_face_edges_iszero"),
                      (aa_r, "aa_face_edges_recip"), (aa_n, "aa_face_edges_normal")):
            need(t, (B, F, 3, 2), nm)
        need(aa_c, (B, F, 3), "aa_face_edges_normal_c")
    ana = _analytic(B, dev)
    if ana is None:
        need(ray_o, (B, ph, pw, 3), "image_ray_o"); need(ray_d, (B, ph, pw, 3), "image_ray_d")
    if temp == 0.0:
        K = 0                                                     # render.cu:141-142
    ts = dict(
        background=_c(background, f32), patch_min=_c(patch_min, i32), verts=_c(verts, f32), faces=_c(faces, i32),
        verts_color=_c(verts_color, f32), faces_opacity=_c(faces_opacity, f32), verts_ndc=_c(verts_ndc, f32),
        verts_image=_c(verts_image, f32), faces_intense=_c(faces_intense, f32), aa_face_verts=_c(aa_v, f32),
        aa_face_edges=_c(aa_e, f32), aa_face_edges_iszero=_c(aa_z, torch.bool), aa_face_edges_recip=_c(aa_r, f32),
        aa_face_edges_normal=_c(aa_n, f32), aa_face_edges_normal_c=_c(aa_c, f32), image_ray_o=_c(ray_o, f32),
        image_ray_d=_c(ray_d, f32))
    keep.append(ts)
    d = RenderDesc()
    d.B, d.P, d.F, d.W, d.H, d.K = B, P, F, pw, ph, K
    d.aa_temperature = temp
    d.flags = _flags
    for k, t in ts.items():
        setattr(d, k, t.data_ptr() if t.numel() > 0 else None)
    if from_image:
        d.flags |= DM2_FLAG_TABLES_FROM_IMAGE
        for k in ("aa_face_verts", "aa_face_edges", "aa_face_edges_iszero", "aa_face_edges_recip", "aa_face_edges_normal", "aa_face_edges_normal_c"):
            setattr(d, k, None)
    if ana is not None:
        cam, fw, fh = ana
        keep.append(cam)
        d.flags |= DM2_FLAG_ANALYTIC_RAYS
        d.ray_cam, d.full_W, d.full_H = cam.data_ptr(), fw, fh
        d.image_ray_o = d.image_ray_d = None
    return d, dev, (B, P, F, pw, ph, K)


def _tiles(B, W, H):
    return B * ((W + 15) // 16) * ((H + 15) // 16)


_tls = threading.local()


class forward_only:
    """``with _C.forward_only(True): _C.render_forward_cuda(...)`` -- no backward will follow this forward (RenderFunction
    sets it when no input requires a gradient, e.g. under ``torch.no_grad()``): the blend masks a backward would use are
    not written.  A side channel on purpose: the function keeps the reference's exact 21-argument signature."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.old = getattr(_tls, "forward_only", False)
        _tls.forward_only = self.on

    def __exit__(self, *exc):
        _tls.forward_only = self.old


class analytic_rays:
    """``with _C.analytic_rays(ray_cam, width, height): _C.render_forward_cuda(...)`` (and the matching backward /
    generate_render_layers call): the primary rays are computed per pixel from ``ray_cam`` (B,32) float32 = inv(mv) then
    inv(proj) of each rendered view, row-major, for an image of (width, height) -- the operation order of the reference's
    ``Renderer._init_rays`` -- and the ``image_ray_o`` / ``image_ray_d`` arguments are placeholders of shape (B,0,0,3)
    that are never read.  A side channel like ``forward_only``: the 21 / 31 / 13-argument
    signatures stay the reference's."""

    def __init__(self, ray_cam, width, height):
        self.val = None if ray_cam is None else (ray_cam, int(width), int(height))

    def __enter__(self):
        self.old = getattr(_tls, "analytic", None)
        _tls.analytic = self.val

    def __exit__(self, *exc):
        _tls.analytic = self.old


class aa_grad_to_verts:
    """``with _C.aa_grad_to_verts(True): _C.render_backward_cuda(...)`` -- the sixth gradient is not dL/d(aa_face_verts)
    (B,F,3,2) but that gradient already scattered to the vertices the corners belong to, (B,P,2): the gradient of
    ``verts_image`` through the reordered copy (DM2_FLAG_AA_GRAD_TO_VERTS).  For a caller that owns the host prep as well
    (Renderer with the fused prep).  A side channel like ``forward_only``: the 31-argument signature stays the reference's."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.old = getattr(_tls, "aa_to_verts", False)
        _tls.aa_to_verts = self.on

    def __exit__(self, *exc):
        _tls.aa_to_verts = self.old


class tables_from_image:
    """``with _C.tables_from_image(True): _C.render_forward_cuda(...)`` (and the matching backward): the six ``aa_*`` arguments
    are placeholders (any tensors of the right rank, e.g. ``(B, 0, 3, 2)``) that are never read -- the plan builds the tables
    per (view, face) from ``verts_image[faces]`` in registers, exactly as ``Triangles`` would (DM2_FLAG_TABLES_FROM_IMAGE).  For a
    caller that owns the host prep (Renderer with the fused prep); the AA-corner gradients then come back per vertex
    (``aa_grad_to_verts``).  A side channel like ``forward_only``: the 21 / 31-argument signatures stay the reference's."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.old = getattr(_tls, "tables_from_image", False)
        _tls.tables_from_image = self.on

    def __exit__(self, *exc):
        _tls.tables_from_image = self.old


class forward_mode:
    """``with _C.forward_mode(mode): _C.render_backward_cuda(...)`` -- tells the backward what the forward of this frame left
    for it (FWD_NONE / FWD_MASKS / FWD_POOL, as ``_C.last_forward_mode()`` reported right after that forward), so that it
    launches exactly one composite kernel.  Without it the binning buffer's own note is used when the very tensor object the
    forward returned comes back; otherwise (FWD_UNKNOWN) every candidate kernel is launched and all but one return at once.
    A side channel like ``forward_only``: the 31-argument signature stays the reference's."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = getattr(_tls, "fwd_mode", None)
        _tls.fwd_mode = self.mode

    def __exit__(self, *exc):
        _tls.fwd_mode = self.old


def last_forward_mode():
    """FWD_* of the calling thread's most recent render_forward_cuda."""
    return getattr(_tls, "last_fwd_mode", FWD_UNKNOWN)


def _pool_budget(N, R):
    """Pairs the shim is willing to give pool room to (4 B each in the binning buffer + 16 B each of backward scratch, 8 GB at
    the cap): a frame whose plan counts more candidate pairs than this -- hundreds of faces over every pixel of a large image,
    thousands of screen-filling faces -- keeps blend masks only and its backward re-clips (DM2_FWD_MASKS)."""
    return min(max(256 * N, 16 * R, 1 << 24), 400_000_000)


def _analytic(B, dev):
    a = getattr(_tls, "analytic", None)
    if a is None:
        return None
    cam, w, h = a
    if cam.dtype != torch.float32 or tuple(cam.shape) != (B, 32) or cam.device != dev:
        raise RuntimeError(f"analytic_rays: ray_cam must be float32 (B, 32) on {dev}, got {cam.dtype} {tuple(cam.shape)} on {cam.device}")
    return cam.contiguous(), w, h


def render_forward_cuda(*args):
    """(num_rendered, color, depth, oarea, tri_id, tri_cnt, doarea, face_buffer, binning_buffer, img_buffer).

    The four AA-record tensors and three byte buffers are opaque to callers
    (reference __init__.py:103-109,160-166).  This implementation recomputes AA
    overlaps in the backward pass, so ``oarea``/``tri_id``/``doarea`` are empty
    (B,H,W,0[,3,2]) placeholders; ``tri_cnt`` (B,H,W) holds the number of
    records the reference would have taken (min(#overlaps visited, K)).
    """
    if len(args) != 21:
        raise TypeError(f"render_forward_cuda() takes 21 positional arguments ({len(args)} given)")
    lib = load_library()
    keep: list = []
    d, dev, (B, P, F, W, H, K) = _make_desc(args, keep)
    if getattr(_tls, "forward_only", False):
        d.flags |= DM2_FLAG_NO_BACKWARD
    if getattr(_tls, "aa_to_verts", False):
        d.flags |= DM2_FLAG_AA_GRAD_TO_VERTS          # the packed records note the CCW reorder for the backward
    with torch.cuda.device(dev):
        st = _stream(dev)
        f32, i32 = torch.float32, torch.int32
        color = torch.empty((B, H, W, 3), dtype=f32, device=dev)
        depth = torch.empty((B, H, W), dtype=f32, device=dev)
        oarea = torch.empty((B, H, W, 0), dtype=f32, device=dev)
        tri_id = torch.empty((B, H, W, 0), dtype=i32, device=dev)
        doarea = torch.empty((B, H, W, 0, 3, 2), dtype=f32, device=dev)
        N, Tn, BF = B * H * W, _tiles(B, W, H), B * F
        if P == 0 or BF == 0 or N == 0:
            # render.cu:149: nothing is rendered; outputs are the zero-initialised images
            color.zero_(); depth.zero_()
            tri_cnt = torch.zeros((B, H, W), dtype=i32, device=dev)
            e = _bytes(dev, 0)
            return 0, color, depth, oarea, tri_id, tri_cnt, doarea, e, _bytes(dev, 0), _bytes(dev, 0)
        tri_cnt = torch.empty((B, H, W), dtype=i32, device=dev)
        face_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_FACE, BF, 2 * Tn + 1))
        img_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_IMAGE, N, Tn))
        nr, longest, pairs, mode = _i64(0), _i64(0), _i64(0), _i32(0)
        # the binning scratch (+ pair pool) is sized from the last call on this device (+ 25 %): when it fits -- every step of
        # a training loop but the first -- plan and run are one C call and the GPU does not wait for Python in between
        key = (dev.index, B, W, H, F)
        with _lock:
            hint = _bin_hint.get(key, 0)
        if _pool_budget(N, 0) <= 0:
            d.flags |= DM2_FLAG_NO_PAIR_POOL
        bin_buf = _bytes(dev, hint)
        rc = lib.dm2_forward(ctypes.byref(d), _ptr(face_buf), face_buf.numel(), _ptr(bin_buf), bin_buf.numel(), _ptr(img_buf), img_buf.numel(),
                             _ptr(color), _ptr(depth), _ptr(tri_cnt), st, ctypes.byref(nr), ctypes.byref(longest), ctypes.byref(pairs),
                             ctypes.byref(mode))
        if rc not in (0, 2):
            raise _err(lib, "render_forward_cuda")
        R = int(nr.value)
        wants_pool = d.aa_temperature > 0.0 and not (d.flags & (DM2_FLAG_NO_BACKWARD | DM2_FLAG_LEGACY_KERNELS | DM2_FLAG_NO_PAIR_POOL))
        over = wants_pool and int(pairs.value) > _pool_budget(N, R)
        pool = lib.dm2_scratch_bytes(SCRATCH_PAIR_POOL, int(pairs.value), 0) if wants_pool and not over else 0
        need = lib.dm2_scratch_bytes(SCRATCH_BINNING, R, Tn) + pool
        if rc == 2 or (over and int(mode.value) == FWD_POOL):
            # (over budget although last frame's buffer happened to have room: rendered again with masks only, so that the
            # backward's tie scratch stays within the budget too)
            if over:
                d.flags |= DM2_FLAG_NO_PAIR_POOL
            bin_buf = _bytes(dev, need + need // 4)
            if lib.dm2_forward_run(ctypes.byref(d), R, int(longest.value), int(pairs.value), _ptr(face_buf), face_buf.numel(), _ptr(bin_buf),
                                   bin_buf.numel(), _ptr(img_buf), img_buf.numel(), _ptr(color), _ptr(depth), _ptr(tri_cnt), st,
                                   ctypes.byref(mode)):
                raise _err(lib, "render_forward_cuda (run)")
        with _lock:
            hint = _bin_hint.get(key, 0)
            if need > hint or 2 * (need + need // 4) < hint:
                if len(_bin_hint) > 64:
                    _bin_hint.clear()
                _bin_hint[key] = need + need // 4
    _tls.last_fwd_mode = int(mode.value)
    bin_buf._dm2_fwd_mode = int(mode.value)       # (survives only as long as this very tensor object is passed around)
    return R, color, depth, oarea, tri_id, tri_cnt, doarea, face_buf, bin_buf, img_buf


def render_backward_cuda(*args):
    """-> (dL_dverts, dL_dverts_color, dL_dfaces_opacity, dL_dverts_ndc, dL_dfaces_intense, dL_daa_face_verts).

    The six gradients are views of ONE packed fp32 buffer (attribute
    ``_dm2_packed`` on the first tensor) so that a multi-GPU caller can sum
    them with a single RCCL all-reduce (dmesh2_renderer_amd.sharding).
    """
    if len(args) != 31:
        raise TypeError(f"render_backward_cuda() takes 31 positional arguments ({len(args)} given)")
    lib = load_library()
    num_rendered = int(args[0])
    fwd_args = args[1:22]
    dL_dcolor, dL_ddepth = args[22], args[23]
    face_buf, bin_buf, img_buf = args[24], args[25], args[26]
    keep: list = []
    d, dev, (B, P, F, W, H, K) = _make_desc(fwd_args, keep)
    _require_gpu(dL_dcolor, dL_ddepth)
    f32 = torch.float32
    # physical order: the gradients of the LEAVES first ([dverts | dverts_color | dfaces_opacity | dfaces_intense], one
    # contiguous span for a multi-GPU caller), then the two intermediates of the host prep (dverts_ndc, daa_face_verts)
    to_verts = bool(getattr(_tls, "aa_to_verts", False))
    if to_verts:
        d.flags |= DM2_FLAG_AA_GRAD_TO_VERTS
    sizes = [P * 3, P * 3, F, B * F, B * P * 3, B * P * 2 if to_verts else B * F * 6]
    packed = torch.zeros((sum(sizes),), dtype=f32, device=dev)          # render.cu:313-318 zeros_like x6
    parts = torch.split(packed, sizes)
    g_verts = parts[0].view(P, 3); g_color = parts[1].view(P, 3); g_opac = parts[2].view(F)
    g_int = parts[3].view(B, F); g_ndc = parts[4].view(B, P, 3)
    g_aa = parts[5].view(B, P, 2) if to_verts else parts[5].view(B, F, 3, 2)
    if F != 0 and P != 0 and num_rendered > 0 and B * H * W > 0:
        if tuple(dL_dcolor.shape) != (B, H, W, 3) or tuple(dL_ddepth.shape) != (B, H, W):
            raise RuntimeError("dL_dout_color / dL_dout_depth must have dimensions (B, H, W, 3) / (B, H, W)")
        dc = _c(dL_dcolor, f32); dd = _c(dL_ddepth, f32)
        mode = getattr(_tls, "fwd_mode", None)
        if mode is None:
            mode = getattr(bin_buf, "_dm2_fwd_mode", FWD_UNKNOWN)
        with torch.cuda.device(dev):
            # scratch of this call: the queue of the pairs whose AA Jacobian the exact clipper has to supply -- room for one entry
            # per pair of the binning buffer's pool part (written only for the 1-5 % that are ties)
            tie_buf = None
            if mode in (FWD_POOL, FWD_UNKNOWN) and d.aa_temperature > 0.0 and not (d.flags & DM2_FLAG_LEGACY_KERNELS):
                pool_pairs = max(0, bin_buf.numel() - lib.dm2_scratch_bytes(SCRATCH_BINNING, num_rendered, _tiles(B, W, H))) // 4
                if pool_pairs > 0:
                    tie_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_TIE_QUEUE, pool_pairs, 0))
            if lib.dm2_backward(ctypes.byref(d), num_rendered, int(mode), _ptr(dc), _ptr(dd), _ptr(face_buf), face_buf.numel(),
                                _ptr(bin_buf), bin_buf.numel(), _ptr(img_buf), img_buf.numel(),
                                _ptr(tie_buf), tie_buf.numel() if tie_buf is not None else 0,
                                _ptr(g_verts), _ptr(g_color), _ptr(g_opac),
                                _ptr(g_ndc), _ptr(g_int), _ptr(g_aa), _stream(dev)):
                raise _err(lib, "render_backward_cuda")
    g_verts._dm2_packed = packed
    return g_verts, g_color, g_opac, g_ndc, g_int, g_aa


def generate_render_layers_cuda(width, height, verts, faces, tets, face_tets, tet_faces, face_existence,
                                verts_ndc, verts_image, image_ray_o, image_ray_d, num_layers):
    """-> (render_layers (B,H,W,L) int32, -1 = empty; render_layers_cnt (B,H,W) int32)."""
    lib = load_library()

    def bad(cond, msg):
        if cond:
            raise RuntimeError(msg)
    # messages of render.cu:397-429
    bad(verts.dim() != 2 or verts.size(1) != 3, "verts must have dimensions (P, 3)")
    bad(faces.dim() != 2 or faces.size(1) != 3, "faces must have dimensions (F, 3)")
    bad(tets.dim() != 2 or tets.size(1) != 4, "tets must have dimensions (T, 4)")
    bad(face_tets.dim() != 2 or face_tets.size(1) != 2, "face_tets must have dimensions (F, 2)")
    bad(tet_faces.dim() != 2 or tet_faces.size(1) != 4, "tet_faces must have dimensions (T, 4)")
    bad(face_existence.dim() != 1 or face_existence.size(0) != faces.size(0), "face_existence must have dimensions (F,)")
    bad(verts_ndc.dim() != 3 or verts_ndc.size(2) != 3, "verts_ndc must have dimensions (B, P, 3)")
    bad(verts_image.dim() != 3 or verts_image.size(2) != 2, "verts_image must have dimensions (B, P, 2)")
    bad(image_ray_o.dim() != 4 or image_ray_o.size(3) != 3, "image_ray_o must have dimensions (B, H, W, 3)")
    bad(image_ray_d.dim() != 4 or image_ray_d.size(3) != 3, "image_ray_d must have dimensions (B, H, W, 3)")
    num_layers = int(num_layers); width = int(width); height = int(height)
    bad(num_layers < 0, "num_layers must be non-negative")
    dev = _require_gpu(verts, faces, tets, face_tets, tet_faces, face_existence, verts_ndc, verts_image, image_ray_o, image_ray_d)
    f32, i32 = torch.float32, torch.int32
    B, P, F, T = verts_ndc.size(0), verts.size(0), faces.size(0), tets.size(0)
    bad(tuple(face_tets.shape) != (F, 2), "face_tets must have dimensions (F, 2)")
    bad(tuple(tet_faces.shape) != (T, 4), "tet_faces must have dimensions (T, 4)")
    bad(tuple(verts_ndc.shape) != (B, P, 3) or tuple(verts_image.shape) != (B, P, 2), "verts_ndc/verts_image shape mismatch")
    ana = _analytic(B, dev)
    bad(ana is None and (tuple(image_ray_o.shape) != (B, height, width, 3) or tuple(image_ray_d.shape) != (B, height, width, 3)),
        "image_ray_o/image_ray_d must have dimensions (B, H, W, 3)")
    ts = dict(verts=_c(verts, f32), faces=_c(faces, i32), tets=_c(tets, i32), face_tets=_c(face_tets, i32),
              tet_faces=_c(tet_faces, i32), face_existence=_c(face_existence, i32), verts_ndc=_c(verts_ndc, f32),
              verts_image=_c(verts_image, f32), image_ray_o=_c(image_ray_o, f32), image_ray_d=_c(image_ray_d, f32))
    d = LayersDesc()
    d.B, d.P, d.F, d.T, d.W, d.H, d.L, d.flags = B, P, F, T, width, height, num_layers, _flags
    for k, t in ts.items():
        setattr(d, k, t.data_ptr() if t.numel() > 0 else None)
    if ana is not None:
        if (ana[1], ana[2]) != (width, height):
            raise RuntimeError("analytic_rays: the image size differs from generate_render_layers_cuda's width / height")
        d.flags |= DM2_FLAG_ANALYTIC_RAYS
        d.ray_cam = ana[0].data_ptr()
        d.image_ray_o = d.image_ray_d = None
    with torch.cuda.device(dev):
        st = _stream(dev)
        cnt = torch.zeros((B, height, width), dtype=i32, device=dev)              # render.cu:437
        layers = torch.full((B, height, width, num_layers), -1, dtype=i32, device=dev)   # render.cu:438
        N, Tn, BF = B * height * width, _tiles(B, width, height), B * F
        if N == 0:
            return layers, cnt
        face_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_FACE, BF, 2 * Tn))
        img_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_LAYER_IMAGE, N, Tn))
        nr, longest = _i64(0), _i64(0)
        if lib.dm2_layers_plan(ctypes.byref(d), _ptr(face_buf), face_buf.numel(), st, ctypes.byref(nr), ctypes.byref(longest)):
            raise _err(lib, "generate_render_layers_cuda (plan)")
        R = int(nr.value)
        bin_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_BINNING, R, Tn))
        tet_buf = _bytes(dev, lib.dm2_scratch_bytes(SCRATCH_LAYER_TETS, T, 0))
        if lib.dm2_layers_run(ctypes.byref(d), R, int(longest.value), _ptr(face_buf), face_buf.numel(), _ptr(bin_buf), bin_buf.numel(),
                              _ptr(img_buf), img_buf.numel(), _ptr(tet_buf), tet_buf.numel(), _ptr(layers), _ptr(cnt), st):
            raise _err(lib, "generate_render_layers_cuda (run)")
    generate_render_layers_cuda.last_debug = (R, face_buf, bin_buf, img_buf)      # kept for tests
    return layers, cnt


def _prep_desc(verts, faces, mv, proj, width, height, keep):
    dev = _require_gpu(verts, faces, mv, proj)
    if verts.dim() != 2 or verts.size(1) != 3:
        raise RuntimeError("verts must have dimensions (P, 3)")
    if faces.dim() != 2 or faces.size(1) != 3:
        raise RuntimeError("faces must have dimensions (F, 3)")
    if mv.dim() != 3 or tuple(mv.shape[1:]) != (4, 4) or tuple(proj.shape) != tuple(mv.shape):
        raise RuntimeError("mv and proj must have dimensions (B, 4, 4)")
    f32, i32 = torch.float32, torch.int32
    v, fc, m, pr = _c(verts, f32), _c(faces, i32), _c(mv, f32), _c(proj, f32)
    keep += [v, fc, m, pr]
    d = PrepDesc()
    d.B, d.P, d.F, d.W, d.H = m.shape[0], v.shape[0], fc.shape[0], int(width), int(height)
    d.verts, d.faces, d.mv, d.proj = _ptr(v), _ptr(fc), _ptr(m), _ptr(pr)
    return d, dev


def prepare_faces(verts, faces, mv, proj, width, height, tables=True):
    """Fused host prep (include/dm2_hip.h: dm2_prepare_faces): -> (verts_ndc (B,P,3), verts_image (B,P,2),
    aa_face_verts, aa_face_edges, aa_face_edges_iszero (bool), aa_face_edges_recip, aa_face_edges_normal
    (all (B,F,3,2)), aa_face_edges_normal_c (B,F,3)); with tables=False only the first two."""
    lib = load_library()
    keep = []
    d, dev = _prep_desc(verts, faces, mv, proj, width, height, keep)
    B, P, F = d.B, d.P, d.F
    f32 = torch.float32
    ndc = torch.empty((B, P, 3), dtype=f32, device=dev)
    image = torch.empty((B, P, 2), dtype=f32, device=dev)
    d.verts_ndc, d.verts_image = _ptr(ndc), _ptr(image)
    outs = [ndc, image]
    if tables:
        aav, aae, aar, aan = (torch.empty((B, F, 3, 2), dtype=f32, device=dev) for _ in range(4))
        aaz = torch.empty((B, F, 3, 2), dtype=torch.bool, device=dev)
        aac = torch.empty((B, F, 3), dtype=f32, device=dev)
        d.aa_face_verts, d.aa_face_edges, d.aa_face_edges_iszero = _ptr(aav), _ptr(aae), _ptr(aaz)
        d.aa_face_edges_recip, d.aa_face_edges_normal, d.aa_face_edges_normal_c = _ptr(aar), _ptr(aan), _ptr(aac)
        outs += [aav, aae, aaz, aar, aan, aac]
    with torch.cuda.device(dev):
        if lib.dm2_prepare_faces(ctypes.byref(d), _stream(dev)):
            raise _err(lib, "dm2_prepare_faces")
    return tuple(outs)


def prepare_faces_backward(verts, faces, mv, proj, width, height, g_verts_ndc=None, g_verts_image=None, g_aa_face_verts=None):
    """d(verts) (P,3) through the fused host prep (dm2_prepare_faces_backward)."""
    lib = load_library()
    keep = []
    d, dev = _prep_desc(verts, faces, mv, proj, width, height, keep)
    f32 = torch.float32
    gs = []
    for g, shape in ((g_verts_ndc, (d.B, d.P, 3)), (g_verts_image, (d.B, d.P, 2)), (g_aa_face_verts, (d.B, d.F, 3, 2))):
        if g is not None:
            if tuple(g.shape) != shape:
                raise RuntimeError(f"upstream gradient has shape {tuple(g.shape)}, expected {shape}")
            g = _c(g, f32)
            _require_gpu(verts, g)
        gs.append(g)
    scratch = torch.empty((d.B * d.P * 2,), dtype=f32, device=dev) if gs[2] is not None else None
    out = torch.empty((d.P, 3), dtype=f32, device=dev)
    with torch.cuda.device(dev):
        if lib.dm2_prepare_faces_backward(ctypes.byref(d), _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(scratch), _ptr(out), _stream(dev)):
            raise _err(lib, "dm2_prepare_faces_backward")
    return out


def debug_fetch(what, count, aux, num_rendered, scratch, dtype, n):
    """Copy an internal array out of a scratch buffer (tests only; see dm2_debug_fetch)."""
    lib = load_library()
    out = torch.empty((n,), dtype=dtype, device=scratch.device)
    if n == 0:
        return out
    with torch.cuda.device(scratch.device):
        if lib.dm2_debug_fetch(what, count, aux, num_rendered, _ptr(scratch), scratch.numel(), _ptr(out), _stream(scratch.device)):
            raise _err(lib, "debug_fetch")
    return out


def debug_aa_overlap(variant, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, pixmin):
    """Run device clipper `variant` (see dm2_debug_aa_overlap) on n (triangle, pixel) pairs; tables (n,3,2) / (n,3),
    pixmin (n,2).  -> area (n), grad (n,3,2), code (n) int32."""
    lib = load_library()
    dev = _require_gpu(aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, pixmin)
    f32 = torch.float32
    n = aa_v.shape[0]
    ts = [_c(aa_v, f32), _c(aa_e, f32), _c(aa_z, torch.bool), _c(aa_r, f32), _c(aa_n, f32), _c(aa_c, f32), _c(pixmin, f32)]
    area = torch.zeros((n,), dtype=f32, device=dev)
    grad = torch.zeros((n, 3, 2), dtype=f32, device=dev)
    code = torch.zeros((n,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        if lib.dm2_debug_aa_overlap(int(variant), n, *[_ptr(t) for t in ts], _ptr(area), _ptr(grad), _ptr(code), _stream(dev)):
            raise _err(lib, "debug_aa_overlap")
    return area, grad, code


def touched_faces(face_buf, B, F):
    """(F) bool: faces binned into at least one tile of at least one view by the forward that produced ``face_buf``
    (tiles_touched of the face scratch; dm2_debug_fetch item 8).  The sharded exchange sends only those rows."""
    t = debug_fetch(8, B * F, 1, 0, face_buf, torch.int32, B * F)
    return (t.view(B, F) != 0).any(dim=0)


# ---- device side of the sharded step's sparse exchange (include/dm2_hip.h: dm2_exchange_*) --------------------------------
def exchange_mark(face_buf, faces, B, P, N):
    """-> (flags (F + P) uint8: face flags then vertex flags, counts (N, 2) int32: rows this rank will send to every owner)."""
    lib = load_library()
    dev = _require_gpu(face_buf, faces)
    F = faces.shape[0]
    fc = _c(faces, torch.int32)
    flags = torch.empty((F + P,), dtype=torch.uint8, device=dev)
    counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        if lib.dm2_exchange_mark(B, P, F, N, _ptr(fc), _ptr(face_buf), face_buf.numel(), _ptr(flags), _ptr(counts), _stream(dev)):
            raise _err(lib, "dm2_exchange_mark")
    return flags, counts


def exchange_pack(flags, counts, total_floats, dverts, dcolor, dopacity, dintense):
    """-> the send buffer (total_floats float32): per owner [face rows | vertex rows] (see dm2_exchange_pack)."""
    lib = load_library()
    dev = _require_gpu(flags, counts, dverts, dcolor, dopacity, dintense)
    f32 = torch.float32
    P, F, B, N = dverts.shape[0], dopacity.shape[0], dintense.shape[0], counts.shape[0]
    ts = [_c(dverts, f32), _c(dcolor, f32), _c(dopacity, f32), _c(dintense, f32)]
    send = torch.empty((int(total_floats),), dtype=f32, device=dev)
    cursors = torch.empty((N, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        if lib.dm2_exchange_pack(B, P, F, N, _ptr(flags), _ptr(counts), _ptr(cursors), *[_ptr(t) for t in ts], _ptr(send), _stream(dev)):
            raise _err(lib, "dm2_exchange_pack")
    return send


def exchange_unpack(recv, recv_counts, rows, rank, B, P, F):
    """-> (slice_v (ceil(P/N), 6), slice_f (ceil(F/N), 1 + B)): the owner's sums of the received rows.  recv_counts: (N, 2)
    int32 on the HOST (a list of pairs will do)."""
    lib = load_library()
    dev = _require_gpu(recv)
    recv_counts = torch.as_tensor(recv_counts, dtype=torch.int32, device="cpu").reshape(-1, 2).contiguous()
    N = recv_counts.shape[0]
    Ps, Fs = -(-P // N), -(-F // N)
    slice_v = torch.empty((Ps, 6), dtype=torch.float32, device=dev)
    slice_f = torch.empty((Fs, 1 + B), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if lib.dm2_exchange_unpack(B, P, F, N, int(rank), _ptr(recv), ctypes.c_void_p(recv_counts.data_ptr()), int(rows), _ptr(slice_v), _ptr(slice_f),
                                   _stream(dev)):
            raise _err(lib, "dm2_exchange_unpack")
    return slice_v, slice_f
