#!/usr/bin/env python3
"""Diagnostic: per-segment cycle shares of the composite kernels (needs libdm2_hip_stamps.so,
built with `make -C dmesh2_renderer_amd/csrc EXTRA=-DDM2_STAMPS BUILD=build_stamps OUT=libdm2_hip_stamps.so`).
Never used for reported timings (the stamps perturb the kernels)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmesh2_renderer_amd import _C  # noqa: E402

_C.LIB_PATH = os.path.join(ROOT, "dmesh2_renderer_amd", "csrc", os.environ.get("DM2_STAMP_LIB", "libdm2_hip_stamps.so"))
import bench  # noqa: E402

FWDQ = ["prologue", "top barrier", "stage faces", "scan+barrier+cut", "phase B1 (classify+compact)+barrier", "phase B2 (survivors)",
        "barrier + phase C (blend)", "epilogue"]
BWDM = ["prologue", "top of chunk", "-", "scan + cut + decode (every wave for itself)", "-",
        "phase B2 (pool ratio, intersection, shade)", "barrier after B2 + request of the next chunk", "phase C (replay)", "barrier after C",
        "phase D (chain + AA Jacobian + dpp + lds atomics)", "barrier after D", "flush atomics"]
BWDQ = ["prologue", "top barrier", "stage faces", "zero+scan+barrier+cut", "phase B1 (classify+compact)+barrier", "phase B2 (survivors)",
        "barrier + phase C (replay)", "-", "barrier after C", "phase D (chain+dpp+lds atomics)", "barriers before flush", "flush atomics"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    dev = torch.device("cuda", 0)
    lib = _C.load_library()
    args, dLc, dLd, _ = bench.build_inputs(cfg, dev, 0, 1)
    buf = (ctypes.c_uint64 * 32)()
    out = _C.render_forward_cuda(*args)
    torch.cuda.synchronize()
    lib.dm2_debug_stamps(buf, 32, 1)
    for it in range(2):
        out = _C.render_forward_cuda(*args)
        _C.render_backward_cuda(out[0], *args, dLc, dLd, out[7], out[8], out[9], out[3], out[4], out[5], out[6])
    torch.cuda.synchronize()
    n = lib.dm2_debug_stamps(buf, 32, 1)
    assert n == 32, n
    for name, labels, off in (("forward", FWDQ, 0), ("backward", BWDM if os.environ.get("DM2_STAMP_BWD", "mask") == "mask" else BWDQ, 16)):
        vals = [buf[off + i] for i in range(16)]
        tot = sum(vals)
        print(f"{name}: total wave-cycles {tot:.3e}")
        for i, lab in enumerate(labels):
            if vals[i]:
                print(f"   {lab:32s} {100.0 * vals[i] / tot:6.2f} %")


if __name__ == "__main__":
    main()
