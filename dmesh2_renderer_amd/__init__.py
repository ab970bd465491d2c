"""dmesh2_renderer_amd -- MI355X-native drop-in for ``dmesh2_renderer``.

Same module surface as the reference's ``dmesh2_renderer/__init__.py``:

* ``RenderFunction``  (reference :11-177)  torch.autograd.Function, 21 inputs -> (color, depth)
* ``Renderer``        (reference :179-380) nn.Module: host prep (rays, projection, AA tables) + op
* ``LayeredRenderer`` (reference :388-451) ``generate()`` -> (render_layers, render_layers_cnt)

The native work goes through ``dmesh2_renderer_amd._C`` -- a ctypes shim over
the C-ABI library ``libdm2_hip.so`` (include/dm2_hip.h) whose three functions
have the reference extension's names, argument order and tuple returns
(ext.cpp:6-9).  There is no CPU fallback: CPU tensors or a missing library
raise ``RuntimeError``.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch

from . import _C
from .pyrenderer import Triangles

__all__ = ["RenderFunction", "Renderer", "LayeredRenderer", "Triangles"]

# Host prep of Renderer.forward (projection + the six AA tables): the fused HIP kernels of dmesh2_renderer_amd/prep.py by
# default on GPU tensors (two kernels each way instead of ~20 torch kernels each way; verts_image differs from the torch
# GEMM's in the last bit, well inside the 1e-5 of the port's tolerance); DM2_FUSED_PREP=0 or Renderer(fused_prep=False)
# selects the reference-shaped torch ops.
_FUSED_PREP_DEFAULT = os.environ.get("DM2_FUSED_PREP", "1") != "0"
_FUSED_AA_GRAD = os.environ.get("DM2_FUSED_AA_GRAD", "1") != "0"
_TABLES_FROM_IMAGE = os.environ.get("DM2_TABLES_FROM_IMAGE", "1") != "0"     # fused prep: AA tables built inside the op's plan, never materialised
_W_EPS = 1e-4   # |w| clamp of the projection, sign kept (reference __init__.py:254-255)


class RenderFunction(torch.autograd.Function):
    """Differentiable rasterize-and-composite op (reference __init__.py:11-177).

    Inputs (positional, as in the reference): background(3), patch_min(B,2) i32,
    patch_width, patch_height, verts(P,3)*, faces(F,3) i32, verts_color(P,3)*,
    faces_opacity(F)*, verts_ndc(B,P,3)* [grad only in z], verts_image(B,P,2),
    faces_intense(B,F)*, aa_temperature, aa_face_verts(B,F,3,2)*, aa_face_edges,
    aa_face_edges_iszero (bool), aa_face_edges_recip, aa_face_edges_normal
    (all (B,F,3,2)), aa_face_edges_normal_c(B,F,3), len_oarea_buffer,
    image_ray_o(B,H,W,3), image_ray_d(B,H,W,3).  (* = receives a gradient.)
    """

    N_INPUTS = 21
    # positions (in the 21-tuple) of the inputs that get a gradient, in the
    # order render_backward_cuda returns them (render.cu:372)
    _GRAD_SLOTS = (4, 6, 7, 8, 10, 12)

    @staticmethod
    def forward(ctx, *inputs):
        if len(inputs) != RenderFunction.N_INPUTS:
            raise TypeError(f"RenderFunction takes {RenderFunction.N_INPUTS} inputs, got {len(inputs)}")
        # analytic rays (Renderer(analytic_rays=True)): the camera block rides along in the thread-local side channel and is
        # kept for the backward
        ctx.analytic = getattr(_C._tls, "analytic", None)
        # fused host prep: the AA-corner gradients come back already scattered to the vertices' image coordinates
        # (input 9, verts_image) instead of as dL/d(aa_face_verts) (input 12) -- see _C.aa_grad_to_verts
        ctx.aa_to_verts = bool(getattr(_C._tls, "aa_to_verts", False))
        ctx.tables_from_image = bool(getattr(_C._tls, "tables_from_image", False))
        try:
            with _C.forward_only(not any(ctx.needs_input_grad)):
                out = _C.render_forward_cuda(*inputs)
        except Exception as ex:
            print("\nAn error occured in renderer forward.")
            print(ex)
            raise
        ctx.fwd_mode = _C.last_forward_mode()       # what this forward left for its backward (masks, pair pool): picks the backward's kernel
        num_rendered, color, depth = out[0], out[1], out[2]
        opaque = out[3:]            # 4 AA-record tensors + face/binning/image byte buffers
        tensors_in = [x for x in inputs if torch.is_tensor(x)]
        ctx.save_for_backward(*tensors_in, *opaque)
        ctx.n_tensor_in = len(tensors_in)
        ctx.tensor_slots = [i for i, x in enumerate(inputs) if torch.is_tensor(x)]
        ctx.scalars = {i: x for i, x in enumerate(inputs) if not torch.is_tensor(x)}
        ctx.num_rendered = num_rendered
        return color, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth):
        saved = ctx.saved_tensors
        inputs: list = [None] * RenderFunction.N_INPUTS
        for slot, t in zip(ctx.tensor_slots, saved[:ctx.n_tensor_in]):
            inputs[slot] = t
        for slot, v in ctx.scalars.items():
            inputs[slot] = v
        oarea, tri_id, tri_cnt, doarea, face_buf, binning_buf, image_buf = saved[ctx.n_tensor_in:]
        try:
            ana = ctx.analytic
            with _C.analytic_rays(*(ana if ana is not None else (None, 0, 0))), _C.aa_grad_to_verts(ctx.aa_to_verts), \
                    _C.forward_mode(ctx.fwd_mode), _C.tables_from_image(ctx.tables_from_image):
                grads = _C.render_backward_cuda(
                    ctx.num_rendered, *inputs, grad_out_color, grad_out_depth,
                    face_buf, binning_buf, image_buf, oarea, tri_id, tri_cnt, doarea)
        except Exception as ex:
            print("\nAn error occured in renderer backward.")
            print(ex)
            raise
        result: list = [None] * RenderFunction.N_INPUTS
        for slot, g in zip(RenderFunction._GRAD_SLOTS, grads):
            result[slot] = g
        if ctx.aa_to_verts:
            result[9], result[12] = grads[5], None          # (B,P,2): the gradient of verts_image, not of aa_face_verts
        return tuple(result)


class Renderer(torch.nn.Module):
    """Reference ``Renderer`` (__init__.py:179-380).

    ``mv``/``proj`` are (Bcam,4,4) column-vector matrices applied as row
    vectors (``v @ mv^T @ proj^T``).  One primary ray per pixel centre is
    precomputed for every camera at construction.
    """

    def __init__(self, mv, proj, width, height, device, aa_grad_buffer_size=20, fused_prep=None, analytic_rays=False,
                 tables_from_image=None):
        super().__init__()
        # not part of the reference's signature: with the fused prep, False hands the op the six materialised AA tables (the
        # reference's 21 arguments as they are); the default lets the op build them from verts_image in its plan
        self.tables_from_image = tables_from_image
        # not part of the reference's signature: analytic_rays=True keeps no (Bcam,H,W,3) ray tensors (49.8 MB per camera at
        # 1080p); the kernels compute each pixel's ray from inv(mv), inv(proj) in the operation order of _init_rays
        self.analytic_rays = bool(analytic_rays)
        self._setup(mv, proj, width, height, device)
        self.aa_grad_buffer_size = aa_grad_buffer_size
        # not part of the reference's signature: projection + AA tables by the fused HIP prep (dmesh2_renderer_amd/prep.py;
        # the default) or by the reference-shaped torch ops below (False)
        self.fused_prep = _FUSED_PREP_DEFAULT if fused_prep is None else bool(fused_prep)

    def _setup(self, mv, proj, width, height, device):
        self.mv = mv
        self.proj = proj
        self.width = width
        self.height = height
        self.device = device
        self.num_batch = mv.shape[0]
        self.ray_o = None
        self.ray_d = None
        if getattr(self, "analytic_rays", False):
            self.ray_cam = torch.cat((torch.inverse(mv).reshape(-1, 16), torch.inverse(proj).reshape(-1, 16)), dim=1) \
                .to(device=device, dtype=torch.float32).contiguous()
        else:
            self._init_rays()

    # -- rays ------------------------------------------------------------------
    def _init_rays(self):
        """Per-pixel world-space rays for every camera (reference :198-237).

        The ray target is the NDC point (x, y, -1, 1) taken through
        inv(proj), inv(mv) WITHOUT a perspective divide, and the direction is
        normalised with ``+1e-6`` on the length -- both as in the reference.
        """
        Bc, H, W, dev = self.num_batch, self.height, self.width, self.device
        inv_mv = torch.inverse(self.mv)
        inv_proj = torch.inverse(self.proj)
        cam_pos = inv_mv[:, :3, 3]                                           # (Bc,3)
        self.ray_o = cam_pos.reshape(Bc, 1, 1, 3).expand(Bc, H, W, 3).to(dev).contiguous()

        xs = torch.arange(W, device=dev).float() + 0.5                       # pixel centres
        ys = torch.arange(H, device=dev).float() + 0.5
        ndc_x = (xs / W * 2) - 1                                             # (W,)
        ndc_y = (ys / H * 2) - 1                                             # (H,)
        pix_h = torch.empty((Bc, H, W, 1, 4), device=dev, dtype=torch.float32)
        pix_h[..., 0, 0] = ndc_x.view(1, 1, W)
        pix_h[..., 0, 1] = ndc_y.view(1, H, 1)
        pix_h[..., 0, 2] = -1.0
        pix_h[..., 0, 3] = 1.0
        to_view = inv_proj.transpose(1, 2).unsqueeze(1).unsqueeze(1)         # (Bc,1,1,4,4)
        to_world = inv_mv.transpose(1, 2).unsqueeze(1).unsqueeze(1)
        target = torch.matmul(torch.matmul(pix_h, to_view), to_world)[..., 0, :3]   # (Bc,H,W,3), no /w
        d = target - self.ray_o
        self.ray_d = d / (torch.norm(d, dim=-1, keepdim=True) + 1e-6)

    def _camera_rows(self, t, batch_mvp_idx):
        """t[batch_mvp_idx] -- as a view (no copy) when the cameras are one or a run of consecutive ones."""
        idx = [int(i) for i in batch_mvp_idx]
        if idx and all(idx[k + 1] == idx[k] + 1 for k in range(len(idx) - 1)) and 0 <= idx[0] and idx[-1] < t.shape[0]:
            return t[idx[0]:idx[0] + len(idx)]
        return t[idx]

    def select_rays(self, batch_mvp_idx, batch_patch_min, patch_width, patch_height):
        """Rays of the (patch_height, patch_width) window at patch_min of each batch item (reference :264-302)."""
        # one read-back of the (B,2) patch origins serves the reference's two bound checks (same messages) and tells whether
        # every window is the whole frame -- then the ray tensors are handed over as views, not gathered into a copy
        # (2 x 24.9 MB per camera at 1080p)
        pm = [[int(v) for v in row] for row in batch_patch_min.tolist()]
        assert all(p[0] + patch_width <= self.width for p in pm), "Some b_patch_max_x exceed self.width"
        assert all(p[1] + patch_height <= self.height for p in pm), "Some b_patch_max_y exceed self.height"
        if patch_width == self.width and patch_height == self.height and all(p[0] == 0 and p[1] == 0 for p in pm):
            return self._camera_rows(self.ray_o, batch_mvp_idx), self._camera_rows(self.ray_d, batch_mvp_idx)
        px0 = batch_patch_min[:, 0].long()
        py0 = batch_patch_min[:, 1].long()
        dev = self.ray_o.device
        cams = torch.as_tensor(list(batch_mvp_idx), device=dev, dtype=torch.long)
        rows = py0.to(dev).view(-1, 1, 1) + torch.arange(patch_height, device=dev).view(1, -1, 1)
        cols = px0.to(dev).view(-1, 1, 1) + torch.arange(patch_width, device=dev).view(1, 1, -1)
        cam = cams.view(-1, 1, 1)
        return self.ray_o[cam, rows, cols], self.ray_d[cam, rows, cols]

    # -- projection --------------------------------------------------------------
    def compute_verts_ndc_image(self, verts, mv, proj):
        """verts (P,3) -> verts_ndc (B,P,3), verts_image (B,P,2) in full-image pixel units (reference :239-262)."""
        hom = torch.cat((verts, torch.ones_like(verts[:, :1])), dim=-1)      # (P,4)
        clip = torch.matmul(torch.matmul(hom, mv.transpose(1, 2)), proj.transpose(1, 2))   # (B,P,4)
        w = clip[..., 3:4]
        w = torch.where((w >= 0.0) & (w < _W_EPS), torch.full_like(w, _W_EPS), w)
        w = torch.where((w < 0.0) & (w > -_W_EPS), torch.full_like(w, -_W_EPS), w)
        ndc = clip[..., :3] / w
        half = (ndc[..., :2] + 1) * 0.5
        image = torch.stack((half[..., 0] * self.width, half[..., 1] * self.height), dim=-1)
        return ndc, image

    # -- forward -----------------------------------------------------------------
    def forward(self, batch_mvp_idx: List[int], batch_patch_min: torch.Tensor, patch_width: int,
                patch_height: int, verts: torch.Tensor, faces: torch.Tensor, verts_color: torch.Tensor,
                faces_opacity: torch.Tensor, faces_intense: torch.Tensor, background: torch.Tensor,
                aa_temperature: float = 1.0):
        """Render ``len(batch_mvp_idx)`` patches; returns color (B,H,W,3), depth (B,H,W) in [0,1] (0 = background)."""
        B = len(batch_mvp_idx)
        F = faces.shape[0]
        mv = self.mv[batch_mvp_idx]
        proj = self.proj[batch_mvp_idx]
        f32 = torch.float32
        if getattr(self, "analytic_rays", False):
            cams = torch.as_tensor(list(batch_mvp_idx), device=self.ray_cam.device, dtype=torch.long)
            ray_o = ray_d = torch.empty((B, 0, 0, 3), dtype=f32, device=self.ray_cam.device)      # placeholders, never read
            with _C.analytic_rays(self.ray_cam[cams].contiguous(), self.width, self.height):
                return self._forward_with_rays(B, F, mv, proj, ray_o, ray_d, batch_patch_min, patch_width, patch_height, verts, faces,
                                               verts_color, faces_opacity, faces_intense, background, aa_temperature)
        ray_o, ray_d = self.select_rays(batch_mvp_idx, batch_patch_min, patch_width, patch_height)
        return self._forward_with_rays(B, F, mv, proj, ray_o, ray_d, batch_patch_min, patch_width, patch_height, verts, faces,
                                       verts_color, faces_opacity, faces_intense, background, aa_temperature)

    def _forward_with_rays(self, B, F, mv, proj, ray_o, ray_d, batch_patch_min, patch_width, patch_height, verts, faces,
                           verts_color, faces_opacity, faces_intense, background, aa_temperature):
        f32 = torch.float32
        if getattr(self, "fused_prep", False) and verts.is_cuda:
            from . import prep
            tfi = getattr(self, "tables_from_image", None)
            if _FUSED_AA_GRAD and (_TABLES_FROM_IMAGE if tfi is None else tfi):
                # the fused prep owns the AA tables end to end: they are never materialised -- the op's plan builds them per
                # face from verts_image straight into its packed records (DM2_FLAG_TABLES_FROM_IMAGE; 114 B per face less to
                # write and to read back), and the corner gradients come back per vertex, as the gradient of verts_image
                verts_ndc, verts_image = prep.project(verts.to(f32), faces.to(torch.int32), mv.to(f32), proj.to(f32), self.width, self.height)
                ph4 = torch.empty((B, 0, 3, 2), dtype=f32, device=verts.device)
                with _C.tables_from_image(True), _C.aa_grad_to_verts(True):
                    color, depth = RenderFunction.apply(
                        background.to(f32), batch_patch_min.to(torch.int32), patch_width, patch_height,
                        verts.to(f32), faces.to(torch.int32), verts_color.to(f32), faces_opacity.to(f32),
                        verts_ndc, verts_image, faces_intense.to(f32), aa_temperature,
                        ph4, ph4, ph4.to(torch.bool), ph4, ph4, torch.empty((B, 0, 3), dtype=f32, device=verts.device),
                        self.aa_grad_buffer_size, ray_o.to(f32), ray_d.to(f32))
                return color, 1.0 - (depth + 1.0) / 2.0
            (verts_ndc, verts_image, aa_v, aa_e, aa_z, aa_r, aa_n, aa_c) = prep.prepare(
                verts.to(f32), faces.to(torch.int32), mv.to(f32), proj.to(f32), self.width, self.height)
            # the fused prep owns both ends of aa_face_verts: the op hands its corner gradients back per VERTEX (as the
            # gradient of verts_image) and prepare_faces_backward needs no (B,F,3,2) scatter pass (DM2_FUSED_AA_GRAD=0: the
            # reference's route through dL/d(aa_face_verts))
            with _C.aa_grad_to_verts(_FUSED_AA_GRAD and verts_image.requires_grad):
                color, depth = RenderFunction.apply(
                    background.to(f32), batch_patch_min.to(torch.int32), patch_width, patch_height,
                    verts.to(f32), faces.to(torch.int32), verts_color.to(f32), faces_opacity.to(f32),
                    verts_ndc, verts_image, faces_intense.to(f32), aa_temperature,
                    aa_v, aa_e, aa_z, aa_r, aa_n, aa_c, self.aa_grad_buffer_size, ray_o.to(f32), ray_d.to(f32))
            return color, 1.0 - (depth + 1.0) / 2.0
        verts_ndc, verts_image = self.compute_verts_ndc_image(verts, mv, proj)

        corners = verts_image[:, faces.flatten()].view(-1, 3, 2)             # (B*F,3,2)
        tri = Triangles(corners[:, 0], corners[:, 1], corners[:, 2])
        color, depth = RenderFunction.apply(
            background.to(f32),
            batch_patch_min.to(torch.int32), patch_width, patch_height,
            verts.to(f32), faces.to(torch.int32), verts_color.to(f32), faces_opacity.to(f32),
            verts_ndc.to(f32), verts_image.to(f32), faces_intense.to(f32),
            aa_temperature,
            tri.verts.reshape(B, F, 3, 2).to(f32),
            tri.edges.reshape(B, F, 3, 2).to(f32),
            tri.edges_iszero.reshape(B, F, 3, 2).to(torch.bool),
            tri.edges_recip.reshape(B, F, 3, 2).to(f32),
            tri.edges_normal.reshape(B, F, 3, 2).to(f32),
            tri.edges_normal_c.reshape(B, F, 3).to(f32),
            self.aa_grad_buffer_size,
            ray_o.to(f32), ray_d.to(f32),
        )
        # NDC z in [-1,1] (background +1) -> [0,1] with background 0 (reference :377-378)
        depth = 1.0 - (depth + 1.0) / 2.0
        return color, depth


class LayeredRenderer(Renderer):
    """Reference ``LayeredRenderer`` (__init__.py:388-451): non-differentiable per-pixel face layers.

    Like the reference it is used through ``generate`` only.  (The reference
    skips ``nn.Module.__init__``; here the module is initialised properly,
    which changes nothing observable.)
    """

    def __init__(self, mv, proj, width, height, device, fused_prep=None, analytic_rays=False):
        torch.nn.Module.__init__(self)
        self.analytic_rays = bool(analytic_rays)
        self._setup(mv, proj, width, height, device)
        self.fused_prep = _FUSED_PREP_DEFAULT if fused_prep is None else bool(fused_prep)

    def generate(self, batch_mvp_idx: Sequence[int], verts: torch.Tensor, faces: torch.Tensor,
                 tets: torch.Tensor, face_tets: torch.Tensor, tet_faces: torch.Tensor,
                 faces_existence: torch.Tensor, num_layers: int):
        """-> render_layers (B,H,W,L) int32 face ids (-1 = empty), render_layers_cnt (B,H,W) int32."""
        mv = self.mv[batch_mvp_idx]
        proj = self.proj[batch_mvp_idx]
        i32, f32 = torch.int32, torch.float32
        if getattr(self, "fused_prep", False) and verts.is_cuda:
            from . import prep
            with torch.no_grad():
                verts_ndc, verts_image = prep.project(verts.to(f32), faces.to(i32), mv.to(f32), proj.to(f32), self.width, self.height)
        else:
            verts_ndc, verts_image = self.compute_verts_ndc_image(verts, mv, proj)
        if getattr(self, "analytic_rays", False):
            cams = torch.as_tensor(list(batch_mvp_idx), device=self.ray_cam.device, dtype=torch.long)
            ph = torch.empty((len(cams), 0, 0, 3), dtype=f32, device=self.ray_cam.device)
            with _C.analytic_rays(self.ray_cam[cams].contiguous(), self.width, self.height):
                return _C.generate_render_layers_cuda(
                    self.width, self.height, verts.to(f32), faces.to(i32), tets.to(i32), face_tets.to(i32), tet_faces.to(i32),
                    faces_existence.to(i32), verts_ndc.to(f32), verts_image.to(f32), ph, ph, num_layers)
        ray_o, ray_d = self._camera_rows(self.ray_o, batch_mvp_idx), self._camera_rows(self.ray_d, batch_mvp_idx)
        return _C.generate_render_layers_cuda(
            self.width, self.height,
            verts.to(f32), faces.to(i32), tets.to(i32), face_tets.to(i32), tet_faces.to(i32),
            faces_existence.to(i32), verts_ndc.to(f32), verts_image.to(f32),
            ray_o.to(f32), ray_d.to(f32), num_layers)
