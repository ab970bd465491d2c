#!/usr/bin/env python3
"""What one rank of an N-GPU run pays per step outside the collectives themselves (one GPU, RCCL group of one rank):
band forward + backward, host-prep backward, touched-face list, and the local packing / unpacking of the sparse and of the
dense exchange.  `python tools/exchange_time.py [bands]`  (bands = the N of the run being modelled; this rank takes band 0)."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dmesh2_renderer_amd import _C  # noqa: E402
from dmesh2_renderer_amd.sharding import BandShardedOp, reduce_leaves_sparse  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    bands = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    args, dLc, dLd, (W, H, F) = bench.build_inputs("cfg4", dev, 0, 1)
    sc = bench._LAST["scene"]
    op = BandShardedOp(args, bands, 0)
    dLc_b = dLc[:, op.y0:op.y0 + op.rows].contiguous(); dLd_b = dLd[:, op.y0:op.y0 + op.rows].contiguous()
    prep_inputs = (args[4], args[5], sc.mv[[0]].contiguous(), sc.proj[[0]].contiguous(), W, H)
    op.world_size = 1                     # (the collectives below run in a one-rank group)
    res = {}
    res["band forward"] = timed(op.forward)
    res["band forward + backward"] = timed(lambda: (op.forward(), op.backward(dLc_b, dLd_b, reduce=False)))
    res["... + host-prep backward (dense leaves, no collective)"] = timed(lambda: (op.forward(), op.backward_leaves(dLc_b, dLd_b, prep_inputs)))
    op.forward()
    leaves = op.backward_leaves(dLc_b, dLd_b, prep_inputs)
    leaves = [x.clone() for x in leaves]
    res["touched_faces()"] = timed(op.touched_faces)
    t = op.touched_faces()
    res["reduce_leaves_sparse (torch ops), one-rank group (local packing + two all-to-all + all-gather to self)"] = timed(lambda: reduce_leaves_sparse(*leaves, args[5], t))
    # the same exchange with its local work in HIP kernels (csrc/dm2_exchange.hip): plan (mark + count + split-size read-back;
    # overlaps the backward in a real step) and reduce (pack, all-to-all, unpack, all-gather)
    from dmesh2_renderer_amd.sharding import DeviceExchange
    B, P = args[8].shape[0], args[4].shape[0]
    res["DeviceExchange: plan (after the forward; its read-back overlaps the backward)"] = timed(lambda: DeviceExchange(_C, op.fwd[7], args[5], B, P))
    x = DeviceExchange(_C, op.fwd[7], args[5], B, P)
    res["DeviceExchange: reduce (pack + all-to-all + unpack + all-gather, one-rank group)"] = timed(lambda: x.reduce(*leaves))
    out = x.reduce(*leaves)
    ref = reduce_leaves_sparse(*leaves, args[5], t)
    assert all(torch.allclose(a, b, rtol=0, atol=1e-6 * float(b.abs().max())) for a, b in zip(out, ref)), "device exchange != torch exchange"
    flags, cnt = _C.exchange_mark(op.fwd[7], args[5], B, P, 1)
    tot = int((cnt[:, 0] * (2 + B) + cnt[:, 1] * 7).sum())
    res["  of which: dm2_exchange_mark (2 memsets + 2 kernels)"] = timed(lambda: _C.exchange_mark(op.fwd[7], args[5], B, P, 1))
    res["  of which: dm2_exchange_pack"] = timed(lambda: _C.exchange_pack(flags, cnt, tot, *leaves))
    send = _C.exchange_pack(flags, cnt, tot, *leaves)
    res["  of which: dm2_exchange_unpack (2 memsets + 1 kernel)"] = timed(lambda: _C.exchange_unpack(send, cnt.cpu(), int(cnt.sum()), 0, B, P, F))
    span = torch.cat([x.reshape(-1) for x in leaves])
    res["dense: all_reduce of the packed leaves, one-rank group"] = timed(lambda: dist.all_reduce(span))
    print(f"cfg4, band 0 of {bands} ({op.rows} rows), touched faces {int(t.sum())} of {F}")
    for k, v in res.items():
        print(f"  {v:8.3f} ms  {k}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
