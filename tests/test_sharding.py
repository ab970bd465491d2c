"""Tile-row band sharding (dmesh2_renderer_amd.sharding), world_size 2 over gloo on CPU.

The product has no CPU compute path, so the ranks run the sharding logic against a test double of
`_C` that is backed by the CPU oracle; what is under test is the band split, the patch arithmetic and
the single all-reduce of the packed gradients: band images must equal the rows of the full frame bit
for bit and the reduced gradients must equal the full-frame gradients (up to fp32 summation order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import scenes, soup_args, to_numpy_args
from dmesh2_renderer_amd.sharding import BandShardedOp, all_bands, band_rows


class OracleBackend:
    """Duck-typed stand-in for dmesh2_renderer_amd._C on CPU tensors (test infrastructure)."""

    def render_forward_cuda(self, *args):
        from oracle import cpu as orc
        f = orc.render_forward_cuda(*to_numpy_args(args))
        t = torch.from_numpy
        return (f.num_rendered, t(f.color), t(f.depth), t(f.buf_oarea), t(f.buf_tri_id), t(f.buf_tri_cnt), t(f.buf_doarea),
                f, torch.zeros(0), torch.zeros(0))          # slot 7 carries the oracle state object

    def render_backward_cuda(self, num_rendered, *rest):
        from oracle import cpu as orc
        dLc, dLd, fwd = rest[21], rest[22], rest[23]
        g = orc.render_backward_cuda(fwd, dLc.numpy(), dLd.numpy())
        order = ["verts", "verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"]
        flat = torch.cat([torch.from_numpy(g[k]).reshape(-1) for k in order])
        outs, off = [], 0
        for k in order:
            n = g[k].size
            outs.append(flat[off:off + n].view(g[k].shape)); off += n
        outs[0]._dm2_packed = flat
        return tuple(outs)


def oracle_prep_backward(verts, faces, mv, proj, width, height, g_verts_ndc=None, g_aa_face_verts=None):
    """CPU stand-in for _C.prepare_faces_backward (the product's is GPU only)."""
    from oracle import cpu as orc
    return torch.from_numpy(orc.prepare_faces_backward(verts, faces, mv, proj, width, height, g_ndc=g_verts_ndc, g_aa=g_aa_face_verts))


def test_band_rows_cover_the_frame():
    for H in (1080, 2160, 100, 16, 7):
        for G in (1, 2, 3, 4, 8):
            bands = all_bands(H, G)
            assert bands[0][0] == 0 and sum(r for _, r in bands) == H
            for (y0, r), (y1, _) in zip(bands, bands[1:]):
                assert y0 + r == y1 and y0 % 16 == 0
    assert [r // 16 for _, r in all_bands(2160, 8)] == [16, 17, 17, 17, 17, 17, 17, 17]      # Ty = 135
    assert band_rows(1080, 8, 7)[0] + band_rows(1080, 8, 7)[1] == 1080


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, W, H, F, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args, sc = soup_args(W, H, F, seed)
        rng = np.random.RandomState(11)
        gc = torch.from_numpy(rng.randn(1, H, W, 3).astype(np.float32)); gd = torch.from_numpy(rng.randn(1, H, W).astype(np.float32))
        op = BandShardedOp(args, world, rank, backend=OracleBackend())
        color, depth = op.forward()
        grads = op.backward(gc[:, op.y0:op.y0 + op.rows].contiguous(), gd[:, op.y0:op.y0 + op.rows].contiguous())
        grads = [g.clone() for g in grads]
        # the data-parallel variant: only the leaves' gradients cross the ranks
        leaves = op.backward_leaves(gc[:, op.y0:op.y0 + op.rows].contiguous(), gd[:, op.y0:op.y0 + op.rows].contiguous(),
                                    (sc.verts, sc.faces, sc.mv[[0]], sc.proj[[0]], W, H), prep_backward=oracle_prep_backward)
        leaves = [g.clone() for g in leaves]
        # the sparse exchange: only touched rows travel (all-to-all to the row owners, all-gather of the reduced slices)
        sparse = op.backward_leaves(gc[:, op.y0:op.y0 + op.rows].contiguous(), gd[:, op.y0:op.y0 + op.rows].contiguous(),
                                    (sc.verts, sc.faces, sc.mv[[0]], sc.proj[[0]], W, H), prep_backward=oracle_prep_backward,
                                    exchange="sparse")
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), y0=op.y0, rows=op.rows, color=color.numpy(), depth=depth.numpy(),
                 **{f"g{i}": g.numpy() for i, g in enumerate(grads)}, **{f"leaf{i}": g.numpy() for i, g in enumerate(leaves)},
                 **{f"sparse{i}": g.numpy() for i, g in enumerate(sparse)})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_sharded_render_equals_full_frame(tmp_path, world):
    from oracle import cpu as orc
    W, H, F, seed = 64, 80, 250, scenes.SEED_BASE + 31
    mp.spawn(_worker, args=(world, _free_port(), W, H, F, seed, str(tmp_path)), nprocs=world, join=True)
    args, sc = soup_args(W, H, F, seed)
    full = orc.render_forward_cuda(*to_numpy_args(args))
    rng = np.random.RandomState(11)
    gc = rng.randn(1, H, W, 3).astype(np.float32); gd = rng.randn(1, H, W).astype(np.float32)
    gfull = orc.render_backward_cuda(full, gc, gd)
    order = ["verts", "verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"]
    leaf_ref = [gfull["verts"] + orc.prepare_faces_backward(sc.verts, sc.faces, sc.mv[[0]], sc.proj[[0]], W, H,
                                                            g_ndc=gfull["verts_ndc"], g_aa=gfull["aa_face_verts"]),
                gfull["verts_color"], gfull["faces_opacity"], gfull["faces_intense"]]
    rows = 0
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        y0, n = int(d["y0"]), int(d["rows"])
        rows += n
        assert np.array_equal(d["color"].view(np.uint32), full.color[:, y0:y0 + n].view(np.uint32))
        assert np.array_equal(d["depth"].view(np.uint32), full.depth[:, y0:y0 + n].view(np.uint32))
        for i, k in enumerate(order):          # every rank holds the same, fully reduced gradients
            ref = gfull[k]
            err = np.abs(d[f"g{i}"] - ref).max() / max(np.abs(ref).max(), 1e-12)
            assert err <= 1e-5, (r, k, err)
        for i, ref in enumerate(leaf_ref):     # leaf gradients: one all-reduce of 24P + 4F + 4BF bytes
            err = np.abs(d[f"leaf{i}"] - ref).max() / max(np.abs(ref).max(), 1e-12)
            assert err <= 1e-5, (r, "leaf", i, err)
            err = np.abs(d[f"sparse{i}"] - ref).max() / max(np.abs(ref).max(), 1e-12)
            assert err <= 1e-5, (r, "sparse leaf", i, err)
            assert d[f"sparse{i}"].shape == ref.shape
            if r > 0:                           # identical on every rank: the owners sum in source order
                assert np.array_equal(d[f"sparse{i}"], np.load(tmp_path / "rank0.npz")[f"sparse{i}"])
    assert rows == H


def test_sparse_exchange_byte_model():
    from dmesh2_renderer_amd.sharding import sparse_exchange_bytes
    # BASELINE configs[4]: 3840x2160, 2 M faces (P = 3F), 8 ranks; a band touches ~1/8 of the faces plus the straddlers
    m = sparse_exchange_bytes(P=6_000_000, F=2_000_000, B=1, world_size=8, touched_faces=270_000, touched_verts=810_000)
    assert m["dense_leaf_bytes"] == 24 * 6_000_000 + 8 * 2_000_000
    assert m["sparse_total"] < 0.6 * m["dense_ring_allreduce"]
    assert m["all_to_all"] < 0.2 * m["all_gather"]
