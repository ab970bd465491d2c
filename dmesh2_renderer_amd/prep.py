"""Fused host prep of ``Renderer.forward``.

The reference prepares the op's inputs in Python with ~20 torch kernels -- projection
(``compute_verts_ndc_image``, dmesh2_renderer/__init__.py:239-262) and the per-face AA tables
(``Triangles``, pyrenderer.py:6-30) -- and autograd replays as many on the way back to ``verts``.
``prepare`` does the same work with two HIP kernels forward and two backward
(csrc/dm2_prep.hip behind ``dm2_prepare_faces`` / ``dm2_prepare_faces_backward`` of the C ABI).

Results: the six tables are bit-identical to the torch twin's for the same ``verts_image``;
``verts_ndc`` / ``verts_image`` agree to fp32 rounding of a 4x4 product (the reference's own value
depends on the BLAS it runs on).  Gradients reach ``verts`` through ``verts_ndc``, ``verts_image``
and ``aa_face_verts`` exactly as in the reference, where the other five tables are constants too.

Opt-in: ``Renderer(..., fused_prep=True)`` (default: the reference-shaped torch prep).
"""
from __future__ import annotations

import torch

from . import _C

__all__ = ["prepare", "project"]


class _Prepare(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, mv, proj, width, height, tables):
        outs = _C.prepare_faces(verts.detach(), faces, mv, proj, width, height, tables=tables)
        ctx.save_for_backward(verts.detach(), faces, mv, proj)
        ctx.size = (int(width), int(height))
        ctx.tables = tables
        # the reference's autograd graph reaches verts through ndc, image and the reordered corners only
        ctx.mark_non_differentiable(*outs[3:])
        # an output nobody differentiates (verts_image, the five constant tables) arrives as None in backward instead of
        # a freshly zero-filled tensor of its size: 140 MB of fills per 1 M-triangle step otherwise
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, g_ndc=None, g_image=None, g_aav=None, *unused):
        verts, faces, mv, proj = ctx.saved_tensors
        if g_ndc is None and g_image is None and g_aav is None:
            return None, None, None, None, None, None, None
        g = _C.prepare_faces_backward(verts, faces, mv, proj, ctx.size[0], ctx.size[1],
                                      g_verts_ndc=g_ndc, g_verts_image=g_image, g_aa_face_verts=g_aav)
        return g, None, None, None, None, None, None


def prepare(verts, faces, mv, proj, width, height):
    """verts (P,3) f32, faces (F,3) i32, mv/proj (B,4,4) of the selected cameras, full image size ->
    (verts_ndc, verts_image, aa_face_verts, aa_face_edges, aa_face_edges_iszero, aa_face_edges_recip,
    aa_face_edges_normal, aa_face_edges_normal_c), differentiable w.r.t. ``verts``."""
    return _Prepare.apply(verts, faces, mv, proj, width, height, True)


def project(verts, faces, mv, proj, width, height):
    """Projection only (LayeredRenderer.generate): -> (verts_ndc, verts_image)."""
    return _Prepare.apply(verts, faces, mv, proj, width, height, False)
