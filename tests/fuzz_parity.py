#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle (both through the reference's `_C` surface):
random image sizes, triangle counts, depth complexity, temperature, K, cameras, patch windows and opacities.
Forward must be bit-exact, gradients within 1e-5 relative L_inf.  `python tests/fuzz_parity.py [seconds] [seed]`.
A development tool (needs a GPU); the fixed cases of tests/test_gpu_parity.py are the gate."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import capture_forward_args, rel_linf, scenes, to_numpy_args  # noqa: E402
from dmesh2_renderer_amd import _C  # noqa: E402
from oracle import cpu as orc  # noqa: E402

GRAD_NAMES = ["verts", "verts_color", "faces_opacity", "verts_ndc", "faces_intense", "aa_face_verts"]


def one_case(seed, idx, verbose=False):
    """Case `idx` of sweep `seed`: every random draw comes from (seed, idx), so a reported case can be replayed alone
    (`python tests/fuzz_parity.py --case SEED IDX [legacy]`)."""
    rng = np.random.default_rng([seed, idx])
    torch.manual_seed(seed * 1000003 + idx)
    W, H = int(rng.integers(8, 220)), int(rng.integers(8, 160))
    F = int(rng.choice([1, 7, 60, 400, 2500, 9000]))
    dc = float(rng.choice([0.3, 1.5, 4.0, 12.0, 40.0, 120.0]))
    temp = float(rng.choice([1.0, 1.0, 0.5, 0.25, 0.0]))
    if os.environ.get("DM2_FUZZ_TEMP0") == "1":              # a sweep of the point-sampled path only (the draw above still happens: same scenes)
        temp = 0.0
    K = int(rng.choice([0, 3, 20]))
    cams = int(rng.integers(1, 3))
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + 1000 + idx + 7919 * seed, num_cams=cams, shared_verts=bool(rng.integers(0, 2)),
                              depth_complexity=dc)
    if rng.integers(0, 3) == 0:          # opaque faces: alpha == 1 branch, early termination
        sc.faces_opacity = torch.where(torch.rand(F) < 0.5, torch.ones(F), sc.faces_opacity)
    bidx = [int(b) for b in rng.integers(0, cams, size=int(rng.integers(1, 3)))]
    pw, ph = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
    pm = [[int(rng.integers(0, W - pw + 1)), int(rng.integers(0, H - ph + 1))] for _ in bidx]
    desc = dict(W=W, H=H, F=F, dc=dc, temp=temp, K=K, cams=cams, bidx=bidx, pw=pw, ph=ph, pm=pm)
    args, _ = capture_forward_args(sc, bidx, pm, pw, ph, temp, K)
    dargs = [a.cuda() if torch.is_tensor(a) else a for a in args]
    out = _C.render_forward_cuda(*dargs)
    ref = orc.render_forward_cuda(*to_numpy_args(args), nthreads=orc.max_threads())
    ok = np.array_equal(out[1].cpu().numpy().view(np.uint32), ref.color.view(np.uint32)) and \
        np.array_equal(out[2].cpu().numpy().view(np.uint32), ref.depth.view(np.uint32)) and \
        np.array_equal(out[5].cpu().numpy(), ref.buf_tri_cnt) and int(out[0]) == int(ref.num_rendered)
    gc = rng.standard_normal(ref.color.shape).astype(np.float32)
    gd = rng.standard_normal(ref.depth.shape).astype(np.float32)
    g = _C.render_backward_cuda(out[0], *dargs, torch.from_numpy(gc).cuda(), torch.from_numpy(gd).cuda(),
                                out[7], out[8], out[9], out[3], out[4], out[5], out[6])
    gref = orc.render_backward_cuda(ref, gc, gd, nthreads=orc.max_threads())
    worst = max(rel_linf(x.cpu().numpy(), gref[n]) for x, n in zip(g, GRAD_NAMES))
    if verbose:
        print(desc, "forward ok:", ok)
        for x, n in zip(g, GRAD_NAMES):
            a, r = x.cpu().numpy(), gref[n]
            dif = np.abs(a - r)
            k = np.unravel_index(int(dif.argmax()), dif.shape) if dif.size else ()
            print(f"  {n:14s} rel L_inf {rel_linf(a, r):.3e}  max |ref| {np.abs(r).max() if r.size else 0:.4g}  worst element {k}: hip {a[k] if dif.size else 0:.7g} ref {r[k] if dif.size else 0:.7g}")
    if worst > 1e-5:
        # A face that covers thousands of pixels sums thousands of fp32 terms, and with random-sign upstream gradients
        # the sum can be orders of magnitude smaller than its terms (a single face's opacity / intensity gradient
        # is ONE such scalar): the fp32 oracle itself is then only good to 1e-5..1e-4 (order of summation).  Judge
        # both against the fp64 oracle: the HIP result must be as close to it as the fp32 oracle is, within a factor
        # that covers the luck of one summation order against another (observed ratios up to 9 in 12 000 cases).
        a64 = to_numpy_args(args)
        r64 = orc.render_forward_cuda(*a64, dtype=np.float64, nthreads=orc.max_threads())
        g64 = orc.render_backward_cuda(r64, gc.astype(np.float64), gd.astype(np.float64), nthreads=orc.max_threads())
        e_hip = max(rel_linf(x.cpu().numpy(), g64[n]) for x, n in zip(g, GRAD_NAMES))
        e_orc = max(rel_linf(gref[n], g64[n]) for n in GRAD_NAMES)
        desc = dict(desc, vs_f32_oracle=worst, hip_vs_f64=e_hip, f32_oracle_vs_f64=e_orc, idx=idx)
        # accepted when the error is summation-order noise.  Exact criterion (small patches): per element within 1e-5 of the
        # tensor's maximum + 8 eps32 * sum over pixels of |that pixel's contribution| (tests/test_gpu_coverage.py::
        # test_fuzz4_regression derives it on the one case this sweep ever flagged); large patches: the ratio to the fp32
        # oracle's own distance from fp64.  The TRUE e_hip stays in the returned worst value either way.
        accepted = e_hip <= max(1e-5, 32.0 * e_orc)
        if e_hip <= 1.5 * e_orc:
            pass                      # no farther from the fp64 result than the reference-order fp32 computation itself
        elif len(bidx) == 1 and pw * ph <= 4096:
            from test_gpu_coverage import per_pixel_terms_f64
            sabs = per_pixel_terms_f64(args, gc, gd, pm, pw, ph)
            accepted = all((np.abs(x.cpu().numpy() - g64[n]) <= 1e-5 * max(float(np.abs(g64[n]).max()), 1e-12) + 8 * 2.0 ** -24 * sabs[n]).all()
                           for x, n in zip(g, GRAD_NAMES))
        desc["accepted_as_summation_noise"] = bool(accepted)
        return ok, e_hip, desc
    return ok, worst, desc


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--case":
        seed, idx = int(sys.argv[2]), int(sys.argv[3])
        _C.set_flags(_C.DM2_FLAG_LEGACY_KERNELS if len(sys.argv) > 4 else 0)
        print(one_case(seed, idx, verbose=True))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--range":        # cases [a, b) of sweep SEED on the default kernels
        seed, a, b = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        for idx in range(a, b):
            ok, worst, desc = one_case(seed, idx)
            if not ok or worst > 1e-5:
                print(idx, ok, worst, desc, flush=True)
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0, n, worst_all, bad = time.time(), 0, 0.0, []
    families = (0,) if os.environ.get("DM2_FUZZ_DEFAULT_ONLY") == "1" else (0, _C.DM2_FLAG_LEGACY_KERNELS)     # default kernels only, or half the budget each
    for legacy in families:
        _C.set_flags(legacy)
        t1 = time.time()
        while time.time() - t1 < budget / len(families):
            ok, worst, desc = one_case(seed, n)
            n += 1
            if n % 200 == 0:          # a heartbeat: a long silent GPU run is taken to be hung
                print(f"... {n} cases, {time.time() - t0:.0f} s, worst so far {max(worst_all, worst):.2e}", flush=True)
            worst_all = max(worst_all, worst)             # the true worst error, accepted or not
            if worst > 1e-4:                              # accepted as summation noise, but large: say which case it was
                print("LARGE (accepted)", legacy, worst, dict(desc, seed=seed, idx=n - 1), flush=True)
            if not ok or (worst > 1e-5 and not desc.get("accepted_as_summation_noise", False)):
                bad.append((legacy, ok, worst, desc))
                print("MISMATCH", legacy, ok, worst, dict(desc, seed=seed, idx=n - 1), flush=True)
    _C.set_flags(0)
    print(f"{n} cases in {time.time() - t0:.0f} s, worst gradient rel L_inf {worst_all:.2e} (vs the fp64 oracle where it exceeds 1e-5 "
          f"of the fp32 one), mismatches: {len(bad)}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
