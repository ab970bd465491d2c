"""Host-side anti-aliasing tables (``Triangles``).

Behavioural twin of the reference's ``Triangles`` (dmesh2_renderer/pyrenderer.py:6-30,
``order_ccw`` :521-529, ``tri_area`` :531-535): per image-space triangle it
produces the six tables the renderer's AA clipper consumes.  Values are
bit-identical to the reference's on the same device (same fp32 operation per
element); pinned by tests/golden/boundary_*.npz and aa_pairs.npz.

Differences in form only: the counter-clockwise reordering is a ``torch.where``
select instead of an in-place masked swap, so the inputs are not modified and
autograd routes ``d/d(verts)`` back through the select.
"""
from __future__ import annotations

import torch

EPS = 1e-3  # an edge component with |e| < EPS is treated as axis-parallel (pyrenderer.py:4,11)


def signed_area(p0: torch.Tensor, p1: torch.Tensor, p2: torch.Tensor) -> torch.Tensor:
    """Signed area of (p0,p1,p2), positive when counter-clockwise; (T,2) -> (T,)."""
    ax = p1[:, 0] - p0[:, 0]
    ay = p1[:, 1] - p0[:, 1]
    bx = p2[:, 0] - p0[:, 0]
    by = p2[:, 1] - p0[:, 1]
    return 0.5 * (ax * by - bx * ay)


class Triangles:
    """Per-triangle AA tables for T image-space triangles.

    verts          (T,3,2)  CCW-reordered corners q0,q1,q2 (q1<->q2 swapped where area<0)
    edges          (T,3,2)  q1-q0, q2-q1, q0-q2
    edges_iszero   (T,3,2)  |edges| < 1e-3   (bool)
    edges_recip    (T,3,2)  1/edges (may be +-inf)
    edges_normal   (T,3,2)  inward normals (-e.y, e.x)
    edges_normal_c (T,3)    normal . edge start point
    """

    def __init__(self, p0: torch.Tensor, p1: torch.Tensor, p2: torch.Tensor):
        flip = (signed_area(p0, p1, p2) < 0).unsqueeze(-1)
        q0 = p0
        q1 = torch.where(flip, p2, p1)
        q2 = torch.where(flip, p1, p2)
        starts = torch.stack((q0, q1, q2), dim=1)
        ends = torch.stack((q1, q2, q0), dim=1)
        self.verts = starts
        self.edges = ends - starts
        self.edges_iszero = self.edges.abs() < EPS
        self.edges_recip = 1.0 / self.edges
        self.edges_normal = torch.stack((-self.edges[..., 1], self.edges[..., 0]), dim=-1)
        self.edges_normal_c = (self.edges_normal * starts).sum(dim=-1)
