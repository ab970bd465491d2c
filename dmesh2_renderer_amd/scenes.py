"""Seeded synthetic scenes for tests and bench.py.

Nothing here is on the product hot path: these generators only produce the
*inputs* the reference's callers would hand to ``Renderer`` /
``LayeredRenderer`` (camera matrices, triangle soups, tet lattices).

All generators are deterministic functions of their seed and run on CPU
(torch.Generator); callers move the tensors to the device they need.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

SEED_BASE = 20250224  # manual_seed(20250224 + cfg)
TAN_HALF_FOV = 0.5
Z_NEAR = 1.0   # the reference builds ray targets from NDC (x,y,-1,1) WITHOUT a perspective divide (__init__.py:225-231): that is a true pixel ray only when the near plane sits at distance 1 (inv(proj) then returns w=1). SURVEY 8d suggested 0.1, which makes every ray point away from the scene.
Z_FAR = 10.0
CAM_DIST = 3.0


def camera(width: int, height: int, shift=(0.0, 0.0, 0.0)):
    """mv = translate(-shift) * translate(0,0,-3); proj = OpenGL perspective.

    Returned as (4,4) float32 tensors in the column-vector convention; the
    renderer applies them as row vectors ``v @ mv^T @ proj^T`` like the
    reference (__init__.py:249-250).
    """
    mv = torch.eye(4, dtype=torch.float32)
    mv[0, 3] = -float(shift[0])
    mv[1, 3] = -float(shift[1])
    mv[2, 3] = -CAM_DIST - float(shift[2])
    aspect = width / height
    t = TAN_HALF_FOV
    n, f = Z_NEAR, Z_FAR
    proj = torch.zeros(4, 4, dtype=torch.float32)
    proj[0, 0] = 1.0 / (aspect * t)
    proj[1, 1] = 1.0 / t
    proj[2, 2] = -(f + n) / (f - n)
    proj[2, 3] = -2.0 * f * n / (f - n)
    proj[3, 2] = -1.0
    return mv, proj


@dataclass
class SoupScene:
    width: int
    height: int
    mv: torch.Tensor            # (Bcam,4,4)
    proj: torch.Tensor          # (Bcam,4,4)
    verts: torch.Tensor         # (P,3) f32
    faces: torch.Tensor         # (F,3) i32
    verts_color: torch.Tensor   # (P,3) f32
    faces_opacity: torch.Tensor  # (F,) f32
    faces_intense: torch.Tensor  # (Bcam,F) f32
    background: torch.Tensor    # (3,) f32

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if torch.is_tensor(v) else v
        return SoupScene(**kw)


def triangle_soup(width: int, height: int, num_faces: int, seed: int,
                  num_cams: int = 1, depth_complexity: float = 4.0,
                  shared_verts: bool = False) -> SoupScene:
    """Triangle soup of SURVEY.md §8(d).

    Centres uniform in image space, view depth U(2.5,3.5), three vertices at
    120 degree spacing with radius r*U(0.6,1.4) px where r gives a mean
    triangle area of ``depth_complexity * N / F`` px^2, vertex z jitter
    +-0.02.  ``shared_verts`` welds a fraction of vertices so that gradient
    scatter hits shared rows (exercises atomics on meshes, not only soups).
    """
    g = torch.Generator().manual_seed(seed)
    W, H, F = width, height, num_faces
    N = W * H
    mean_area = depth_complexity * N / max(F, 1)
    r_px = math.sqrt(4.0 * mean_area / (3.0 * math.sqrt(3.0)))

    def U(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float64) * (hi - lo) + lo

    cx = U((F,), 0.0, W)
    cy = U((F,), 0.0, H)
    zv = U((F,), 2.5, 3.5)
    th0 = U((F,), 0.0, 2.0 * math.pi)
    rad = r_px * U((F, 3), 0.6, 1.4)
    zj = U((F, 3), -0.02, 0.02)
    ang = th0[:, None] + torch.arange(3, dtype=torch.float64)[None, :] * (2.0 * math.pi / 3.0)
    px = cx[:, None] + rad * torch.cos(ang)
    py = cy[:, None] + rad * torch.sin(ang)
    # un-project image point (px,py) at view depth zv (camera looks down -z)
    t = TAN_HALF_FOV
    aspect = W / H
    xn = px / W * 2.0 - 1.0
    yn = py / H * 2.0 - 1.0
    depth = zv[:, None] + zj
    xv = xn * depth * aspect * t
    yv = yn * depth * t
    zw = -depth + CAM_DIST
    verts = torch.stack([xv, yv, zw], dim=-1).reshape(F * 3, 3).to(torch.float32)
    faces = torch.arange(F * 3, dtype=torch.int32).reshape(F, 3)
    if shared_verts and F >= 2:
        # weld pairs into quads: odd face f shares the edge (v1,v0) of face f-1 and
        # gets its third corner mirrored across that edge, so vertex rows are
        # shared between faces (gradient scatter collides) without giant triangles
        faces = faces.clone()
        odd = torch.arange(1, F, 2)
        v3 = verts.reshape(F, 3, 3)
        v3[odd, 2] = v3[odd - 1, 0] + v3[odd - 1, 1] - v3[odd - 1, 2]
        verts = v3.reshape(F * 3, 3).contiguous()
        faces[odd, 0] = faces[odd - 1, 1]
        faces[odd, 1] = faces[odd - 1, 0]
    P = verts.shape[0]
    verts_color = torch.rand((P, 3), generator=g, dtype=torch.float32)
    faces_opacity = (torch.rand((F,), generator=g, dtype=torch.float32) * 0.7 + 0.2)
    mvs, projs = [], []
    for c in range(num_cams):
        mv, proj = camera(W, H, shift=(0.05 * c, -0.03 * c, 0.1 * c))
        mvs.append(mv)
        projs.append(proj)
    faces_intense = torch.ones((num_cams, F), dtype=torch.float32)
    if num_cams > 1:
        faces_intense = torch.rand((num_cams, F), generator=g, dtype=torch.float32) * 0.5 + 0.75
    return SoupScene(W, H, torch.stack(mvs), torch.stack(projs), verts, faces,
                     verts_color, faces_opacity, faces_intense,
                     torch.zeros(3, dtype=torch.float32))


@dataclass
class TetScene:
    width: int
    height: int
    mv: torch.Tensor
    proj: torch.Tensor
    verts: torch.Tensor        # (P,3) f32
    faces: torch.Tensor        # (F,3) i32
    tets: torch.Tensor         # (T,4) i32
    face_tets: torch.Tensor    # (F,2) i32, -1 = none
    tet_faces: torch.Tensor    # (T,4) i32
    faces_existence: torch.Tensor  # (F,) i32

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if torch.is_tensor(v) else v
        return TetScene(**kw)


# Kuhn subdivision of the unit cube into 6 tets: one per permutation of axes.
_KUHN_PERMS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


def tet_lattice(width: int, height: int, n: int, seed: int, jitter: float = 0.2,
                existence_p: float = 0.3, num_cams: int = 1) -> TetScene:
    """Jittered (n+1)^3 vertex lattice in [-1,1]^3, 6 Kuhn tets per cube.

    SURVEY.md §8(d) cfg 3: n=25 -> T=93 750, F=191 250.
    """
    rng = np.random.RandomState(seed)
    m = n + 1
    ax = np.linspace(-1.0, 1.0, m)
    gx, gy, gz = np.meshgrid(ax, ax, ax, indexing="ij")
    verts = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3)
    cell = 2.0 / n
    verts = verts + rng.uniform(-jitter * cell, jitter * cell, size=verts.shape)

    def vid(i, j, k):
        return (i * m + j) * m + k

    ii, jj, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = np.stack([ii, jj, kk], axis=-1).reshape(-1, 3)
    tets = []
    for perm in _KUHN_PERMS:
        corner = base.copy()
        ids = [vid(corner[:, 0], corner[:, 1], corner[:, 2])]
        for a in perm:
            corner = corner.copy()
            corner[:, a] += 1
            ids.append(vid(corner[:, 0], corner[:, 1], corner[:, 2]))
        tets.append(np.stack(ids, axis=-1))
    tets = np.concatenate(tets, axis=0).astype(np.int64)          # (T,4)
    T = tets.shape[0]
    # faces: the 4 vertex triples of each tet, deduplicated by sorted key
    combos = [(1, 2, 3), (0, 2, 3), (0, 1, 3), (0, 1, 2)]
    tri = np.stack([tets[:, c] for c in combos], axis=1)            # (T,4,3)
    tri_sorted = np.sort(tri, axis=-1).reshape(-1, 3)
    P = verts.shape[0]
    key = (tri_sorted[:, 0] * P + tri_sorted[:, 1]) * P + tri_sorted[:, 2]
    uniq, inv = np.unique(key, return_inverse=True)
    F = uniq.shape[0]
    faces = np.zeros((F, 3), dtype=np.int64)
    faces[inv] = tri_sorted
    tet_faces = inv.reshape(T, 4)
    face_tets = -np.ones((F, 2), dtype=np.int64)
    tet_of = np.repeat(np.arange(T), 4)
    order = np.argsort(inv, kind="stable")
    inv_s, tet_s = inv[order], tet_of[order]
    first = np.ones_like(inv_s, dtype=bool)
    first[1:] = inv_s[1:] != inv_s[:-1]
    face_tets[inv_s[first], 0] = tet_s[first]
    face_tets[inv_s[~first], 1] = tet_s[~first]
    existence = (rng.uniform(size=F) < existence_p).astype(np.int32)
    mvs, projs = [], []
    for c in range(num_cams):
        mv, proj = camera(width, height, shift=(0.07 * c, 0.04 * c, 0.0))
        mvs.append(mv)
        projs.append(proj)
    return TetScene(width, height, torch.stack(mvs), torch.stack(projs),
                    torch.from_numpy(verts.astype(np.float32)),
                    torch.from_numpy(faces.astype(np.int32)),
                    torch.from_numpy(tets.astype(np.int32)),
                    torch.from_numpy(face_tets.astype(np.int32)),
                    torch.from_numpy(tet_faces.astype(np.int32)),
                    torch.from_numpy(existence))
