"""Tile-row band sharding of one frame across the GPUs of a node.

The reference is single-GPU, but its op already renders arbitrary
sub-rectangles through ``patch_min / patch_width / patch_height``
(render.h:17-19, auxiliary.h:72-92).  Per-pixel results depend only on the
faces overlapping that pixel, so GPU ``g`` of ``G`` renders tile rows
``[floor(g*Ty/G), floor((g+1)*Ty/G))`` as one band-shaped patch; band edges sit
on multiples of 16 so every band's tile grid coincides with the single-GPU one
and the band images are bit-identical to the corresponding rows of the full
frame.  Geometry is replicated.  The forward needs no exchange; the backward
produces a full-size partial gradient on every rank, packed in ONE fp32 buffer
(see ``_C.render_backward_cuda``), summed with ONE all-reduce (RCCL over xGMI
when the process group's backend is ``nccl``; ``gloo`` in the CPU tests).

``BandShardedOp.backward_leaves`` is the cheaper, data-parallel-idiomatic variant: what a training step needs on
every rank are the gradients of the LEAVES (verts, verts_color, faces_opacity, faces_intense).  ``verts_ndc`` and
``aa_face_verts`` are intermediates of the host prep; their partial gradients are pushed through the prep's
backward locally (it is linear in them, so summing over ranks commutes with it) and only
``[dverts | dverts_color | dfaces_opacity | dfaces_intense]`` crosses the links: 24P + 4F + 4BF bytes instead of
24P + 12BP + 4F + 28BF (80 MB instead of 140 MB at 1080p / 1 M triangles).

``reduce_leaves_sparse`` is the exchange that scales: a rank's partial gradient is non-zero only in the rows of the faces
its band touched (about 1/N of them, plus the faces that straddle a band edge) and of their vertices.  Rows are owned by
contiguous id ranges (face f by rank f // ceil(F/N), vertex v by rank v // ceil(P/N)); every rank sends each owner only the
touched rows of that owner's range (one all-to-all of [id | row] records, about (N-1)/N^2 of the leaf bytes per rank), the
owner sums them into its dense slice, and one all-gather of the slices leaves every rank with the full gradients -- the
(N-1)/N of the leaf bytes that any scheme must deliver to a rank that keeps all parameters.  Against a ring all-reduce of the
dense buffer (2 (N-1)/N of the bytes) that is about half the traffic at N = 8, and both collectives drive all seven xGMI
links of a GPU at once instead of being bound by one link of a ring.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

TILE = 16


def band_rows(height: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first pixel row, number of pixel rows) of rank's band; may be (y0, 0) for surplus ranks."""
    ty = (height + TILE - 1) // TILE
    t0 = (rank * ty) // world_size
    t1 = ((rank + 1) * ty) // world_size
    y0 = min(t0 * TILE, height)
    y1 = min(t1 * TILE, height)
    return y0, y1 - y0


def all_bands(height: int, world_size: int) -> List[Tuple[int, int]]:
    return [band_rows(height, world_size, r) for r in range(world_size)]


def allreduce_packed_grads(grads, group=None):
    """Sum the six gradient tensors over ranks with a single collective.

    ``grads`` is the 6-tuple of ``_C.render_backward_cuda``; its tensors are
    views of one packed buffer (``grads[0]._dm2_packed``), which is reduced in
    place.  Falls back to flatten/unflatten for foreign tensors.
    """
    import torch.distributed as dist
    packed = getattr(grads[0], "_dm2_packed", None)
    if packed is not None:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        return grads
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, off = [], 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g)); off += n
        out.append(g)
    return tuple(out)


def sparse_exchange_bytes(P: int, F: int, B: int, world_size: int, touched_faces: int, touched_verts: int) -> dict:
    """Bytes one rank SENDS in ``reduce_leaves_sparse`` and in the dense alternatives (for DESIGN.md / bench lines)."""
    N = world_size
    row_f, row_v = 4 * (2 + B), 4 * 7                     # [id | dopacity | dintense(B)],  [id | dverts(3) | dcolor(3)]
    Fs, Ps = -(-F // N), -(-P // N)
    a2a = (touched_faces * row_f + touched_verts * row_v) * (N - 1) // N
    gather = (Fs * 4 * (1 + B) + Ps * 24) * (N - 1)       # the rank's reduced slice goes to N - 1 peers
    dense = 24 * P + 4 * F + 4 * B * F
    return dict(all_to_all=a2a, all_gather=gather, sparse_total=a2a + gather, dense_leaf_bytes=dense,
                dense_ring_allreduce=2 * dense * (N - 1) // N)


def reduce_leaves_sparse(dverts, dcolor, dopacity, dintense, faces, touched, group=None):
    """Sum the leaf gradients over the ranks of ``group``, moving only touched rows (module docstring).

    dverts, dcolor (P,3); dopacity (F); dintense (B,F): this rank's partial gradients; ``faces`` (F,3) integer;
    ``touched`` (F) bool: faces that may carry a non-zero partial here (a superset is fine; rows of other faces are
    NOT sent, so it must cover every face with a non-zero row or a non-zero vertex row).
    Returns new dense (dverts, dcolor, dopacity, dintense), identical on every rank (same summation order everywhere)."""
    import torch.distributed as dist
    N = dist.get_world_size(group)
    dev = dverts.device
    P, F, B = dverts.shape[0], dopacity.shape[0], dintense.shape[0]
    Fs, Ps = -(-F // N), -(-P // N)                        # slice sizes (the last slices are padded)
    f32 = torch.float32
    fidx = torch.nonzero(touched.reshape(-1), as_tuple=False).flatten()                    # ascending
    vidx = torch.unique(faces[fidx].reshape(-1).long()) if fidx.numel() else fidx            # ascending
    # rows with their global id in front (the id's bits travel in a float slot)
    frow = torch.cat([fidx.to(torch.int32).view(torch.float32).unsqueeze(1), dopacity[fidx].unsqueeze(1),
                      dintense[:, fidx].t()], dim=1).to(f32)                                # (nf, 2 + B)
    vrow = torch.cat([vidx.to(torch.int32).view(torch.float32).unsqueeze(1), dverts[vidx], dcolor[vidx]], dim=1).to(f32)   # (nv, 7)
    bounds_f = torch.arange(0, N + 1, device=dev) * Fs
    bounds_v = torch.arange(0, N + 1, device=dev) * Ps
    cf = torch.searchsorted(fidx, bounds_f)                # rows of owner d: [cf[d], cf[d+1])
    cv = torch.searchsorted(vidx, bounds_v)
    nf_send = (cf[1:] - cf[:-1]); nv_send = (cv[1:] - cv[:-1])
    cnt_send = torch.stack([nf_send, nv_send], dim=1).to(torch.int64).contiguous()         # (N, 2)
    cnt_recv = torch.empty_like(cnt_send)
    dist.all_to_all_single(cnt_recv, cnt_send, group=group)
    # split sizes must be host ints: ONE read-back of (what goes where, what comes from where)
    host = torch.cat([cnt_send.reshape(-1), cnt_recv.reshape(-1), cf.to(torch.int64), cv.to(torch.int64)]).cpu().tolist()
    cs = [host[2 * d:2 * d + 2] for d in range(N)]
    cr = [host[2 * N + 2 * d:2 * N + 2 * d + 2] for d in range(N)]
    cfh, cvh = host[4 * N:4 * N + N + 1], host[4 * N + N + 1:]
    wf, wv = 2 + B, 7
    send = torch.cat([torch.cat([frow[cfh[d]:cfh[d + 1]].reshape(-1), vrow[cvh[d]:cvh[d + 1]].reshape(-1)]) for d in range(N)])
    in_split = [cs[d][0] * wf + cs[d][1] * wv for d in range(N)]
    out_split = [cr[d][0] * wf + cr[d][1] * wv for d in range(N)]
    recv = torch.empty((sum(out_split),), dtype=f32, device=dev)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    # owner: sum what arrived (source by source, ascending: the same order on every rank) into the dense slice
    rank = dist.get_rank(group)
    slice_f = torch.zeros((Fs, 1 + B), dtype=f32, device=dev)
    slice_v = torch.zeros((Ps, 6), dtype=f32, device=dev)
    off = 0
    for sr in range(N):
        nf, nv = cr[sr]
        if nf:
            blk = recv[off:off + nf * wf].view(nf, wf)
            slice_f.index_add_(0, blk[:, 0].contiguous().view(torch.int32).long() - rank * Fs, blk[:, 1:])
        off += nf * wf
        if nv:
            blk = recv[off:off + nv * wv].view(nv, wv)
            slice_v.index_add_(0, blk[:, 0].contiguous().view(torch.int32).long() - rank * Ps, blk[:, 1:])
        off += nv * wv
    mine = torch.cat([slice_v.reshape(-1), slice_f.reshape(-1)])
    full = torch.empty((N * mine.numel(),), dtype=f32, device=dev)
    dist.all_gather_into_tensor(full, mine, group=group)
    full = full.view(N, -1)
    gv = full[:, : Ps * 6].reshape(N * Ps, 6)[:P]
    gf = full[:, Ps * 6:].reshape(N * Fs, 1 + B)[:F]
    return gv[:, :3].contiguous(), gv[:, 3:].contiguous(), gf[:, 0].contiguous(), gf[:, 1:].t().contiguous()


class DeviceExchange:
    """The same exchange with its local work on the device (csrc/dm2_exchange.hip behind ``dm2_exchange_*``): three streaming
    kernels instead of ~15 torch ops and a sort.  ``plan`` can run right after the forward (what a rank will send only depends
    on the faces its band binned), so that the one host read-back -- the all-to-all's split sizes -- overlaps the backward;
    ``reduce`` then packs, exchanges, sums and gathers.  Row order inside a segment is arbitrary; a row has a single
    contributor but for faces that straddle a band edge, so results match ``reduce_leaves_sparse`` to fp32 rounding."""

    def __init__(self, C, face_buf, faces, B, P, group=None):
        import torch.distributed as dist
        self.C, self.group = C, group
        self.N = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.B, self.P, self.F = int(B), int(P), int(faces.shape[0])
        self.flags, self.counts = C.exchange_mark(face_buf, faces, self.B, self.P, self.N)
        self.cnt_recv = torch.empty_like(self.counts)
        dist.all_to_all_single(self.cnt_recv, self.counts, group=group)          # what comes from where
        # ONE read-back of (what goes where, what comes from where): issued now, looked at in reduce()
        self._host = torch.empty((4 * self.N,), dtype=torch.int32, pin_memory=True)
        self._host.copy_(torch.cat([self.counts.reshape(-1), self.cnt_recv.reshape(-1)]), non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record()

    def reduce(self, dverts, dcolor, dopacity, dintense):
        import torch.distributed as dist
        N, B, P, F = self.N, self.B, self.P, self.F
        self._ev.synchronize()
        h = self._host.tolist()
        wf, wv = 2 + B, 7
        in_split = [h[2 * d] * wf + h[2 * d + 1] * wv for d in range(N)]
        out_split = [h[2 * N + 2 * d] * wf + h[2 * N + 2 * d + 1] * wv for d in range(N)]
        rows_in = sum(h[2 * N:])
        send = self.C.exchange_pack(self.flags, self.counts, sum(in_split), dverts, dcolor, dopacity, dintense)
        recv = torch.empty((sum(out_split),), dtype=torch.float32, device=dverts.device)
        dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=self.group)
        slice_v, slice_f = self.C.exchange_unpack(recv, [h[2 * N + 2 * d:2 * N + 2 * d + 2] for d in range(N)], rows_in, self.rank, B, P, F)
        Ps, Fs = slice_v.shape[0], slice_f.shape[0]
        mine = torch.cat([slice_v.reshape(-1), slice_f.reshape(-1)])
        full = torch.empty((N * mine.numel(),), dtype=torch.float32, device=dverts.device)
        dist.all_gather_into_tensor(full, mine, group=self.group)
        full = full.view(N, -1)
        gv = full[:, : Ps * 6].reshape(N * Ps, 6)[:P]
        gf = full[:, Ps * 6:].reshape(N * Fs, 1 + B)[:F]
        return gv[:, :3].contiguous(), gv[:, 3:].contiguous(), gf[:, 0].contiguous(), gf[:, 1:].t().contiguous()


class BandShardedOp:
    """Forward/backward of the render op on this rank's band of a full-frame call.

    ``full_args`` are the 21 boundary arguments of a FULL-frame call
    (patch_min = (0,0), patch = whole image, B arbitrary); the band's arguments
    are derived by slicing the ray tensors and moving ``patch_min``.
    """

    def __init__(self, full_args, world_size: int, rank: int, backend=None, tables_from_image=None):
        from . import _C
        self._C = backend or _C
        self.world_size, self.rank = world_size, rank
        # the product's default path (Renderer with the fused prep): the six AA tables are not handed over but built inside
        # the op's plan from verts_image (DM2_FLAG_TABLES_FROM_IMAGE): a rank's plan then reads ~50 B per face instead of ~170
        if tables_from_image is None:
            tables_from_image = hasattr(self._C, "tables_from_image") and torch.is_tensor(full_args[9]) and full_args[9].is_cuda
        self.tables_from_image = bool(tables_from_image)
        # the product backend can hand the AA-corner gradients back per vertex (DM2_FLAG_AA_GRAD_TO_VERTS; announced in the
        # forward, used by backward_leaves): no (B,F,3,2) scatter pass over all faces on every rank
        self.routed = hasattr(self._C, "aa_grad_to_verts")
        a = list(full_args)
        H = int(a[3])
        self.y0, self.rows = band_rows(H, world_size, rank)
        pm = a[1].clone()
        pm[:, 1] += self.y0
        a[1] = pm
        a[3] = self.rows
        a[19] = a[19][:, self.y0:self.y0 + self.rows].contiguous()
        a[20] = a[20][:, self.y0:self.y0 + self.rows].contiguous()
        if self.tables_from_image:
            Bv = a[8].shape[0]
            for k in range(12, 17):
                a[k] = torch.empty((Bv, 0, 3, 2), dtype=a[k].dtype, device=a[k].device)       # placeholders, never read
            a[17] = torch.empty((Bv, 0, 3), dtype=a[17].dtype, device=a[17].device)
        self.args = a
        self.fwd = None

    def _ctx(self):
        import contextlib
        return self._C.tables_from_image(True) if self.tables_from_image else contextlib.nullcontext()

    def forward(self):
        """-> (color, depth) of this rank's band: rows [y0, y0+rows) of the frame."""
        if self.rows == 0:
            a = self.args
            B, W = a[8].shape[0], int(a[2])
            dev = a[8].device
            self.fwd = None
            return torch.zeros((B, 0, W, 3), device=dev), torch.zeros((B, 0, W), device=dev)
        if self.routed:          # (the forward's packed records note the CCW reorder the routed backward needs)
            with self._C.aa_grad_to_verts(True), self._ctx():
                self.fwd = self._C.render_forward_cuda(*self.args)
        else:
            self.fwd = self._C.render_forward_cuda(*self.args)
        return self.fwd[1], self.fwd[2]

    def backward(self, dL_dcolor_band, dL_ddepth_band, group=None, reduce=True, aa_to_verts=False):
        """Band gradients -> full gradients (summed over ranks when ``reduce``).  ``aa_to_verts``: the sixth gradient is
        (B,P,2), the AA-corner gradients already scattered to the vertices (``_C.aa_grad_to_verts``)."""
        a = self.args
        if self.rows == 0:
            P, F, B = a[4].shape[0], a[5].shape[0], a[8].shape[0]
            dev = a[4].device
            # (the physical order of _C.render_backward_cuda: every rank must take the same path through the collectives)
            sizes = [P * 3, P * 3, F, B * F, B * P * 3, B * P * 2 if aa_to_verts else B * F * 6]
            packed = torch.zeros((sum(sizes),), dtype=torch.float32, device=dev)
            parts = torch.split(packed, sizes)
            grads = (parts[0].view(P, 3), parts[1].view(P, 3), parts[2].view(F), parts[4].view(B, P, 3),
                     parts[3].view(B, F), parts[5].view(B, P, 2) if aa_to_verts else parts[5].view(B, F, 3, 2))
            grads[0]._dm2_packed = packed
        else:
            f = self.fwd
            if aa_to_verts:
                with self._C.aa_grad_to_verts(True), self._ctx():
                    grads = self._C.render_backward_cuda(f[0], *a, dL_dcolor_band, dL_ddepth_band, f[7], f[8], f[9],
                                                         f[3], f[4], f[5], f[6])
            else:
                with self._ctx():
                    grads = self._C.render_backward_cuda(f[0], *a, dL_dcolor_band, dL_ddepth_band, f[7], f[8], f[9],
                                                         f[3], f[4], f[5], f[6])
        if reduce and self.world_size > 1:
            grads = allreduce_packed_grads(grads, group)
        return grads

    def touched_faces(self):
        """(F) bool: faces this rank's band binned into at least one tile of at least one view (from the forward's face
        scratch when the backend exposes it; otherwise None)."""
        if self.fwd is None:
            return torch.zeros((self.args[5].shape[0],), dtype=torch.bool, device=self.args[5].device)
        fn = getattr(self._C, "touched_faces", None)
        if fn is None:
            return None
        B, F = self.args[8].shape[0], self.args[5].shape[0]
        return fn(self.fwd[7], B, F)

    def plan_exchange(self, group=None):
        """Call after ``forward`` (product backend, exchange="sparse"): marks the rows this rank will send and starts the
        read-back of the all-to-all's split sizes, so that it is over by the time the backward has run."""
        self._xchg = None
        if self.fwd is not None and hasattr(self._C, "exchange_mark") and self.args[5].is_cuda:
            a = self.args
            self._xchg = DeviceExchange(self._C, self.fwd[7], a[5], a[8].shape[0], a[4].shape[0], group)
        return self._xchg

    def backward_leaves(self, dL_dcolor_band, dL_ddepth_band, prep_inputs, group=None, prep_backward=None, exchange="dense"):
        """Band gradients -> gradients of the leaves, summed over ranks by a dense all-reduce of 24P + 4F + 4BF bytes
        (in two parts: colour / opacity / intensity start while the host-prep backward still runs, dverts follows).

        ``prep_inputs`` = (verts, faces, mv, proj, width, height) of the host prep that produced ``verts_ndc`` /
        ``aa_face_verts`` (mv / proj of the rendered cameras, FULL image size).  Returns
        (dverts, dverts_color, dfaces_opacity, dfaces_intense); dverts includes the contribution that reaches
        ``verts`` through ``verts_ndc`` and ``aa_face_verts`` (reference: torch autograd through
        __init__.py:239-262 and pyrenderer.py:6-30; here ``_C.prepare_faces_backward``).
        """
        import torch.distributed as dist
        # with the product backend the AA-corner gradients arrive per vertex: no (B,F,3,2) scatter pass over all faces on
        # every rank (DM2_FLAG_AA_GRAD_TO_VERTS)
        routed = prep_backward is None and self.routed
        g = self.backward(dL_dcolor_band, dL_ddepth_band, reduce=False, aa_to_verts=routed)
        dverts, dcolor, dopacity, dndc, dintense, daa = g
        verts, faces, mv, proj, width, height = prep_inputs
        pb = prep_backward or self._C.prepare_faces_backward
        dense = not (exchange == "sparse" and self.world_size > 1)
        packed = getattr(dverts, "_dm2_packed", None)
        n_v, n_rest = dverts.numel(), dcolor.numel() + dopacity.numel() + dintense.numel()
        es = dverts.element_size()
        leaf_span = (packed is not None and packed.numel() >= n_v + n_rest and dverts.data_ptr() == packed.data_ptr()
                     and dcolor.data_ptr() == packed.data_ptr() + es * n_v
                     and dopacity.data_ptr() == dcolor.data_ptr() + es * dcolor.numel()
                     and dintense.data_ptr() == dopacity.data_ptr() + es * dopacity.numel())
        early = None
        if dense and leaf_span and self.world_size > 1:
            # the op's own gradients of colour / opacity / intensity are final: their all-reduce starts now and runs while
            # the host-prep backward below adds its share to dverts
            early = dist.all_reduce(packed[n_v:n_v + n_rest], op=dist.ReduceOp.SUM, group=group, async_op=True)
        if routed:
            dverts += pb(verts, faces, mv, proj, width, height, g_verts_ndc=dndc, g_verts_image=daa)
        else:
            dverts += pb(verts, faces, mv, proj, width, height, g_verts_ndc=dndc, g_aa_face_verts=daa)
        if not dense:
            if self.fwd is not None and hasattr(self._C, "exchange_mark") and faces.is_cuda and \
                    dist.get_backend(group) != "gloo":       # (gloo has no all-to-all on device tensors)
                x = getattr(self, "_xchg", None) or self.plan_exchange(group)
                self._xchg = None
                return x.reduce(dverts, dcolor, dopacity, dintense)
            touched = self.touched_faces()
            if touched is None:      # backend without the scratch accessor: any face or vertex row that is not exactly zero
                fl = faces.long()
                touched = (dopacity != 0) | (dintense != 0).any(dim=0) | (dverts[fl] != 0).any(dim=2).any(dim=1) | \
                          (dcolor[fl] != 0).any(dim=2).any(dim=1)
            return reduce_leaves_sparse(dverts, dcolor, dopacity, dintense, faces, touched, group)
        if leaf_span:
            # [dverts | dverts_color | dfaces_opacity | dfaces_intense] head the backward's packed buffer (_C.render_backward_cuda)
            if self.world_size > 1:
                dist.all_reduce(packed[:n_v], op=dist.ReduceOp.SUM, group=group)
                early.wait()
            return dverts, dcolor, dopacity, dintense
        span = torch.cat([dverts.reshape(-1), dcolor.reshape(-1), dopacity.reshape(-1), dintense.reshape(-1)])
        o1, o2, o3 = n_v, n_v + dcolor.numel(), n_v + dcolor.numel() + dopacity.numel()
        dverts, dcolor, dopacity, dintense = (span[:o1].view(dverts.shape), span[o1:o2].view(dcolor.shape),
                                              span[o2:o3].view(dopacity.shape), span[o3:].view(dintense.shape))
        if self.world_size > 1:
            dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group)
        return dverts, dcolor, dopacity, dintense
