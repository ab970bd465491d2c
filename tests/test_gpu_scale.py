"""GPU tests at the BASELINE.json configuration sizes.

The oracle cannot walk a full 1080p / 1M-triangle frame in test time, so config 4 is checked through
size-independent properties of the domain (band decomposition == full frame bit for bit, the gradient
of a sum of band losses == the full-frame gradient, linearity of the backward in the upstream gradient,
run-to-run determinism of the forward) plus an exact oracle comparison on one 48-row band of the same
frame.  Configs 2 and 3 are small enough for a direct oracle comparison at full size."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from util import ROOT, rel_linf, scenes, to_numpy_args

pytestmark = pytest.mark.gpu

GRADS = ["verts", "verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"]


def _C():
    from dmesh2_renderer_amd import _C as c
    return c


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _band(args, y0, rows):
    a = list(args)
    pm = a[1].clone(); pm[:, 1] += y0
    a[1] = pm; a[3] = rows
    a[19] = a[19][:, y0:y0 + rows].contiguous(); a[20] = a[20][:, y0:y0 + rows].contiguous()
    return a


def _fwd_bwd(args, dLc, dLd):
    C = _C()
    out = C.render_forward_cuda(*args)
    g = C.render_backward_cuda(out[0], *args, dLc, dLd, out[7], out[8], out[9], out[3], out[4], out[5], out[6])
    return out, g


@pytest.fixture(scope="module")
def cfg4():
    b = _bench()
    args, dLc, dLd, (W, H, F) = b.build_inputs("cfg4", torch.device("cuda", 0), 0, 1)
    out, g = _fwd_bwd(args, dLc, dLd)
    torch.cuda.synchronize()
    return dict(args=args, dLc=dLc, dLd=dLd, W=W, H=H, F=F, out=out, g=[x.clone() for x in g])


def test_cfg4_statistics(cfg4):
    R = cfg4["out"][0]
    assert 1.4e6 < R < 1.8e6                                    # R ~ 1.59 M at 1080p / 1 M triangles
    color, depth = cfg4["out"][1], cfg4["out"][2]
    assert torch.isfinite(color).all() and torch.isfinite(depth).all()
    assert 0.0 <= float(color.min()) and float(color.max()) <= 1.0 + 1e-4
    for g in cfg4["g"]:
        assert torch.isfinite(g).all()
    assert float(cfg4["g"][1].abs().sum()) > 0


def test_cfg4_forward_is_deterministic(cfg4):
    out2 = _C().render_forward_cuda(*cfg4["args"])
    assert out2[0] == cfg4["out"][0]
    assert torch.equal(out2[1], cfg4["out"][1]) and torch.equal(out2[2], cfg4["out"][2]) and torch.equal(out2[5], cfg4["out"][5])


def test_cfg4_band_decomposition_equals_full_frame(cfg4):
    """Tile-row bands (the multi-GPU sharding) reproduce the frame exactly; band gradients add up to the full ones."""
    from dmesh2_renderer_amd.sharding import all_bands
    H = cfg4["H"]
    gsum = [torch.zeros_like(x) for x in cfg4["g"]]
    for (y0, rows) in all_bands(H, 3):
        a = _band(cfg4["args"], y0, rows)
        out, g = _fwd_bwd(a, cfg4["dLc"][:, y0:y0 + rows].contiguous(), cfg4["dLd"][:, y0:y0 + rows].contiguous())
        assert torch.equal(out[1], cfg4["out"][1][:, y0:y0 + rows])
        assert torch.equal(out[2], cfg4["out"][2][:, y0:y0 + rows])
        for s, x in zip(gsum, g):
            s += x
    for name, s, ref in zip(GRADS, gsum, cfg4["g"]):
        err = float((s - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        assert err <= 1e-5, (name, err)


def test_cfg4_backward_is_linear_in_upstream_gradient(cfg4):
    C = _C()
    out = cfg4["out"]
    g2 = C.render_backward_cuda(out[0], *cfg4["args"], cfg4["dLc"] * 2.0, cfg4["dLd"] * 2.0, out[7], out[8], out[9],
                                out[3], out[4], out[5], out[6])
    for name, a, b in zip(GRADS, g2, cfg4["g"]):
        err = float((a - 2.0 * b).abs().max() / (2.0 * b).abs().max().clamp_min(1e-12))
        assert err <= 1e-5, (name, err)


def test_cfg4_band_matches_oracle(cfg4):
    from oracle import cpu as orc
    y0, rows = 512, 48
    a = _band(cfg4["args"], y0, rows)
    dLc = cfg4["dLc"][:, y0:y0 + rows].contiguous(); dLd = cfg4["dLd"][:, y0:y0 + rows].contiguous()
    out, g = _fwd_bwd(a, dLc, dLd)
    ref = orc.render_forward_cuda(*to_numpy_args(a), nthreads=orc.max_threads())
    assert out[0] == ref.num_rendered
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))
    assert np.array_equal(out[2].cpu().numpy().view(np.uint32), ref.depth.view(np.uint32))
    assert np.array_equal(out[5].cpu().numpy(), ref.buf_tri_cnt)
    gref = orc.render_backward_cuda(ref, dLc.cpu().numpy(), dLd.cpu().numpy())
    for name, x in zip(GRADS, g):
        assert rel_linf(x.cpu().numpy(), gref[name]) <= 1e-5, name


def test_cfg5_4k_2m_faces_band_decomposition_and_oracle_band():
    """BASELINE configs[4] (3840x2160, 2 M triangles; quoted on 8 GPUs, fits one): the 8-band decomposition the
    multi-GPU run uses reproduces the single-GPU frame bit for bit, the band gradients add up to the full ones, and
    one 32-row band agrees with the oracle."""
    from oracle import cpu as orc
    from dmesh2_renderer_amd.sharding import all_bands
    b = _bench()
    args, dLc, dLd, (W, H, F) = b.build_inputs("cfg5", torch.device("cuda", 0), 0, 1)
    out, g = _fwd_bwd(args, dLc, dLd)
    g = [x.clone() for x in g]
    assert out[0] > 3_000_000
    gsum = [torch.zeros_like(x) for x in g]
    for (y0, rows) in all_bands(H, 8):
        a = _band(args, y0, rows)
        o, gb = _fwd_bwd(a, dLc[:, y0:y0 + rows].contiguous(), dLd[:, y0:y0 + rows].contiguous())
        assert torch.equal(o[1], out[1][:, y0:y0 + rows]) and torch.equal(o[2], out[2][:, y0:y0 + rows])
        for s_, x in zip(gsum, gb):
            s_ += x
    for name, s_, ref in zip(GRADS, gsum, g):
        err = float((s_ - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        assert err <= 1e-5, (name, err)
    y0, rows = 1024, 32
    a = _band(args, y0, rows)
    gc, gd = dLc[:, y0:y0 + rows].contiguous(), dLd[:, y0:y0 + rows].contiguous()
    o, gb = _fwd_bwd(a, gc, gd)
    ref = orc.render_forward_cuda(*to_numpy_args(a), nthreads=orc.max_threads())
    assert np.array_equal(o[1].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))
    gref = orc.render_backward_cuda(ref, gc.cpu().numpy(), gd.cpu().numpy())
    for name, x in zip(GRADS, gb):
        assert rel_linf(x.cpu().numpy(), gref[name]) <= 1e-5, name


def test_cfg2_full_size_against_oracle():
    """BASELINE config 2: forward+backward 512x512, 50k triangles, AA visibility gradients on."""
    from oracle import cpu as orc
    b = _bench()
    args, dLc, dLd, _ = b.build_inputs("cfg2", torch.device("cuda", 0), 0, 1)
    out, g = _fwd_bwd(args, dLc, dLd)
    ref = orc.render_forward_cuda(*to_numpy_args(args), nthreads=orc.max_threads())
    assert out[0] == ref.num_rendered
    flipped = int((out[5].cpu().numpy() != ref.buf_tri_cnt).sum())
    assert flipped == 0
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), ref.color.view(np.uint32))
    assert np.array_equal(out[2].cpu().numpy().view(np.uint32), ref.depth.view(np.uint32))
    gref = orc.render_backward_cuda(ref, dLc.cpu().numpy(), dLd.cpu().numpy())
    for name, x in zip(GRADS, g):
        assert rel_linf(x.cpu().numpy(), gref[name]) <= 1e-5, name


def test_cfg3_layered_renderer_full_size():
    """BASELINE config 3: LayeredRenderer, 4 layers, 1024x1024, 26^3 Kuhn lattice (T = 93 750, F = 191 250)."""
    from oracle import cpu as orc
    import dmesh2_renderer_amd as dm2
    W = H = 1024
    sc = scenes.tet_lattice(W, H, 25, seed=scenes.SEED_BASE + 3)
    assert sc.tets.shape[0] == 93750 and sc.faces.shape[0] == 191250
    scd = sc.to("cuda")
    lr = dm2.LayeredRenderer(scd.mv, scd.proj, W, H, "cuda")
    layers, cnt = lr.generate([0], scd.verts, scd.faces, scd.tets, scd.face_tets, scd.tet_faces, scd.faces_existence, 4)
    ndc, img = lr.compute_verts_ndc_image(scd.verts, scd.mv[[0]], scd.proj[[0]])
    rl, rc = orc.generate_render_layers_cuda(W, H, sc.verts.numpy(), sc.faces.numpy(), sc.tets.numpy(), sc.face_tets.numpy(),
                                             sc.tet_faces.numpy(), sc.faces_existence.numpy(), ndc.cpu().numpy(), img.cpu().numpy(),
                                             lr.ray_o[[0]].cpu().numpy(), lr.ray_d[[0]].cpu().numpy(), 4, nthreads=orc.max_threads())
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert np.array_equal(layers.cpu().numpy(), rl)
    assert (rc == 4).mean() > 0.3


# ---- two ranks on the one GPU, real product path, gloo as the transport ----------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dmesh2_renderer_amd.sharding import BandShardedOp
        b = _bench()
        args, dLc, dLd, _ = b.build_inputs("cfg2", torch.device("cuda", 0), rank, world)
        op = BandShardedOp(args, world, rank)
        color, depth = op.forward()
        g = op.backward(dLc[:, op.y0:op.y0 + op.rows].contiguous(), dLd[:, op.y0:op.y0 + op.rows].contiguous())
        g = [x.clone() for x in g]
        # data-parallel variant (what bench.py --gpus N times): only the leaves' gradients are all-reduced
        sc = b._LAST["scene"]
        W, H = b.CONFIGS["cfg2"][:2]
        leaves = op.backward_leaves(dLc[:, op.y0:op.y0 + op.rows].contiguous(), dLd[:, op.y0:op.y0 + op.rows].contiguous(),
                                    (args[4], args[5], sc.mv[[0]].contiguous(), sc.proj[[0]].contiguous(), W, H))
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), y0=op.y0, rows=op.rows, color=color.cpu().numpy(),
                 **{f"g{i}": x.cpu().numpy() for i, x in enumerate(g)}, **{f"leaf{i}": x.cpu().numpy() for i, x in enumerate(leaves)})
    finally:
        dist.destroy_process_group()


def test_two_ranks_band_sharded_product_path(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.start_processes(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    b = _bench()
    args, dLc, dLd, _ = b.build_inputs("cfg2", torch.device("cuda", 0), 0, 1)
    out, g = _fwd_bwd(args, dLc, dLd)
    from dmesh2_renderer_amd import _C
    sc = b._LAST["scene"]
    W, H = b.CONFIGS["cfg2"][:2]
    extra = _C.prepare_faces_backward(args[4], args[5], sc.mv[[0]].contiguous(), sc.proj[[0]].contiguous(), W, H,
                                      g_verts_ndc=g[3], g_aa_face_verts=g[5])
    leaf_ref = [(g[0] + extra).cpu().numpy(), g[1].cpu().numpy(), g[2].cpu().numpy(), g[4].cpu().numpy()]
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        y0, rows = int(d["y0"]), int(d["rows"])
        assert np.array_equal(d["color"], out[1][:, y0:y0 + rows].cpu().numpy())
        for i, name in enumerate(GRADS):
            assert rel_linf(d[f"g{i}"], g[i].cpu().numpy()) <= 1e-5, (r, name)
        for i, ref in enumerate(leaf_ref):
            assert rel_linf(d[f"leaf{i}"], ref) <= 1e-5, (r, "leaf", i)


def test_sparse_exchange_on_rccl_single_rank():
    """The sparse leaf-gradient exchange on device tensors through RCCL (`nccl` backend) with a one-rank group: the
    collectives it uses (all_to_all_single with split sizes, all_gather_into_tensor) and the device-side packing must
    reproduce the input exactly.  (World sizes 2 and 3 are covered over gloo in tests/test_sharding.py.)"""
    import os
    import socket
    import torch.distributed as dist
    from dmesh2_renderer_amd.sharding import reduce_leaves_sparse
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = torch.Generator().manual_seed(3)
        P, F, B = 300, 100, 2
        faces = torch.arange(P, dtype=torch.int32).view(F, 3).cuda()
        touched = (torch.rand(F, generator=g) < 0.4).cuda()
        dv = torch.randn(P, 3, generator=g).cuda(); dc = torch.randn(P, 3, generator=g).cuda()
        do = torch.randn(F, generator=g).cuda(); di = torch.randn(B, F, generator=g).cuda()
        keep_f = touched.float(); keep_v = touched.repeat_interleave(3).float().unsqueeze(1)
        dv, dc, do, di = dv * keep_v, dc * keep_v, do * keep_f, di * keep_f         # rows of untouched faces are zero
        out = reduce_leaves_sparse(dv, dc, do, di, faces, touched)
        torch.cuda.synchronize()
        for a, b in zip(out, (dv, dc, do, di)):
            assert a.shape == b.shape and torch.equal(a, b)
    finally:
        dist.destroy_process_group()


def test_touched_faces_accessor_matches_the_lists():
    """_C.touched_faces (what the sparse exchange sends) == the faces that appear in this forward's tile lists."""
    from dmesh2_renderer_amd import _C
    b = _bench()
    args, dLc, dLd, _ = b.build_inputs("cfg1", torch.device("cuda", 0), 0, 1)
    a = list(args); a[3] = 64; a[19] = a[19][:, :64].contiguous(); a[20] = a[20][:, :64].contiguous()      # a band: not every face
    out = _C.render_forward_cuda(*a)
    B, F = a[8].shape[0], a[5].shape[0]
    touched = _C.touched_faces(out[7], B, F).cpu().numpy()
    N, Tn = B * 64 * 256, B * 16 * 4
    flist = _C.debug_fetch(1, N, Tn, out[0], out[8], torch.int32, out[0]).cpu().numpy()
    want = np.zeros(F, bool); want[flist] = True
    assert np.array_equal(touched, want) and 0 < touched.sum() < F


@pytest.mark.parametrize("N", [2, 3, 8])
def test_device_exchange_kernels_emulated_ranks(N):
    """dm2_exchange_mark / _pack / _unpack (csrc/dm2_exchange.hip), N ranks emulated on one GPU with the all-to-all routed by
    hand: every rank renders its band of the same frame, packs the touched rows of its partial leaf gradients per owner; every
    owner sums what it is sent; the gathered slices must equal the dense sum of the partials, and the rows a rank sends must be
    exactly the rows the torch formulation (sharding.reduce_leaves_sparse) would send."""
    from dmesh2_renderer_amd import _C
    from dmesh2_renderer_amd.sharding import BandShardedOp
    b = _bench()
    dev = torch.device("cuda", 0)
    args, dLc, dLd, (W, H, F) = b.build_inputs("cfg2", dev, 0, 1)
    faces = args[5]
    B, P = args[8].shape[0], args[4].shape[0]
    Ps, Fs = -(-P // N), -(-F // N)
    parts, sends, counts = [], [], []
    for r in range(N):
        op = BandShardedOp(args, N, r)
        op.world_size = 1
        op.forward()
        g = op.backward(dLc[:, op.y0:op.y0 + op.rows].contiguous(), dLd[:, op.y0:op.y0 + op.rows].contiguous(), reduce=False)
        leaves = (g[0], g[1], g[2], g[4])                     # dverts, dverts_color, dfaces_opacity, dfaces_intense
        parts.append([x.clone() for x in leaves])
        flags, cnt = _C.exchange_mark(op.fwd[7], faces, B, P, N)
        touched = _C.touched_faces(op.fwd[7], B, F)
        assert torch.equal(flags[:F].bool(), touched)
        vflag = torch.zeros(P, dtype=torch.bool, device=dev); vflag[faces[touched].reshape(-1).long()] = True
        assert torch.equal(flags[F:].bool(), vflag)
        ch = cnt.cpu().tolist()
        for o in range(N):                                     # rows per owner = flagged ids in the owner's range
            assert ch[o][0] == int(touched[o * Fs:(o + 1) * Fs].sum()) and ch[o][1] == int(vflag[o * Ps:(o + 1) * Ps].sum())
        total = sum(c[0] * (2 + B) + c[1] * 7 for c in ch)
        send = _C.exchange_pack(flags, cnt, total, *leaves)
        # segment by segment: the set of rows == the flagged rows of the owner's range, values included
        off = 0
        for o in range(N):
            nf, nv = ch[o]
            fr = send[off:off + nf * (2 + B)].view(nf, 2 + B); off += nf * (2 + B)
            vr = send[off:off + nv * 7].view(nv, 7); off += nv * 7
            fid = fr[:, 0].contiguous().view(torch.int32).long(); vid = vr[:, 0].contiguous().view(torch.int32).long()
            want_f = torch.nonzero(touched[o * Fs:(o + 1) * Fs]).flatten() + o * Fs
            want_v = torch.nonzero(vflag[o * Ps:(o + 1) * Ps]).flatten() + o * Ps
            assert torch.equal(torch.sort(fid).values, want_f) and torch.equal(torch.sort(vid).values, want_v)
            assert torch.equal(fr[:, 1], leaves[2][fid]) and torch.equal(fr[:, 2:], leaves[3][:, fid].t())
            assert torch.equal(vr[:, 1:4], leaves[0][vid]) and torch.equal(vr[:, 4:], leaves[1][vid])
        sends.append(send); counts.append(ch)
    # the all-to-all, by hand: owner o receives, source by source, that source's segment for o
    dense = [sum(p[i] for p in parts) for i in range(4)]
    gv, gf = [], []
    for o in range(N):
        chunks, rc = [], []
        for s_ in range(N):
            off = sum(counts[s_][k][0] * (2 + B) + counts[s_][k][1] * 7 for k in range(o))
            n = counts[s_][o][0] * (2 + B) + counts[s_][o][1] * 7
            chunks.append(sends[s_][off:off + n]); rc.append(counts[s_][o])
        recv = torch.cat(chunks) if chunks else torch.empty(0, device=dev)
        sv, sf = _C.exchange_unpack(recv, rc, sum(c[0] + c[1] for c in rc), o, B, P, F)
        gv.append(sv); gf.append(sf)
    gv = torch.cat(gv)[:P]; gf = torch.cat(gf)[:F]
    got = (gv[:, :3], gv[:, 3:], gf[:, 0], gf[:, 1:].t())
    for a, d in zip(got, dense):
        assert rel_linf(a.cpu().numpy(), d.cpu().numpy()) <= 1e-6
