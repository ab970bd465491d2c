#!/usr/bin/env python3
"""profiles/<tag>_band_costs.json: what ONE rank of an N-GPU run computes per step -- per-kernel milliseconds (the library's
stage timers: hipEvents on the op's stream) of a 1/N band of the frame at cfg4 and cfg5, N = 1, 2, 4, 8, the middle band --
and the bytes its exchange would move.  One GPU; the collectives themselves are not in these numbers.
usage: python tools/band_costs.py > profiles/r03_band_costs.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dmesh2_renderer_amd import _C  # noqa: E402
from dmesh2_renderer_amd.sharding import BandShardedOp, sparse_exchange_bytes  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    out = {"note": "per-kernel ms of one rank's band (middle band of N), medians of 10 steps, stage timers of the library; "
                   "exchange bytes per rank from sharding.sparse_exchange_bytes with the band's touched faces",
           "configs": {}}
    for cfg in ("cfg4", "cfg5"):
        args, dLc, dLd, (W, H, F) = bench.build_inputs(cfg, dev, 0, 1)
        sc = bench._LAST["scene"]
        B, P = args[8].shape[0], args[4].shape[0]
        prep_inputs = (args[4], args[5], sc.mv[[0]].contiguous(), sc.proj[[0]].contiguous(), W, H)
        rows = {}
        for N in (1, 2, 4, 8):
            op = BandShardedOp(args, N, N // 2)
            op.world_size = 1
            gc = dLc[:, op.y0:op.y0 + op.rows].contiguous(); gd = dLd[:, op.y0:op.y0 + op.rows].contiguous()
            for _ in range(3):
                op.forward(); op.backward_leaves(gc, gd, prep_inputs)
            _C.profile_enable(True)
            acc = {}
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tot, prep = [], []
            for _ in range(10):
                ev[0].record()
                op.forward()
                g = op.backward(gc, gd, reduce=False, aa_to_verts=True)
                ev[1].record()
                _C.prepare_faces_backward(*prep_inputs[:4], W, H, g_verts_ndc=g[3], g_verts_image=g[5])
                ev[2].record()
                torch.cuda.synchronize()
                for k, v in _C.profile_read().items():
                    acc.setdefault(k, []).append(v)
                tot.append(ev[0].elapsed_time(ev[2])); prep.append(ev[1].elapsed_time(ev[2]))
            _C.profile_enable(False)
            nf = int(op.touched_faces().sum().item())
            rows[f"1/{N}"] = {
                "band_rows": op.rows, "tile_face_pairs": int(op.fwd[0]), "touched_faces": nf,
                "kernel_ms": {k: round(float(np.median(v)), 4) for k, v in acc.items()},
                "host_prep_backward_ms": round(float(np.median(prep)), 4),
                "step_ms_forward_backward_prep": round(float(np.median(tot)), 4),
                "exchange_bytes_per_rank": sparse_exchange_bytes(P, F, B, N, nf, 3 * nf) if N > 1 else None,
            }
        out["configs"][cfg] = {"frame": [W, H], "faces": F, "bands": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
