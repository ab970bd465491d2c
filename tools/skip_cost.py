#!/usr/bin/env python3
"""Diagnostic: what 8160 workgroups cost when each leaves at its first line (run under rocprofv3 --kernel-trace --stats).
The backward is called without being told what the forward left (a cloned binning tensor): the library launches every
candidate kernel and those that find another mode in the device-side word return at once -- their duration is the cost of
dispatching the grid at the composite kernels' register / LDS footprint."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmesh2_renderer_amd import _C  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    args, dLc, dLd, _ = bench.build_inputs(sys.argv[1] if len(sys.argv) > 1 else "cfg4", dev, 0, 1)
    for it in range(6):
        out = _C.render_forward_cuda(*args)
        bin_buf = out[8].clone()
        _C.render_backward_cuda(out[0], *args, dLc, dLd, out[7], bin_buf, out[9], out[3], out[4], out[5], out[6])
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
