#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpixels/s forward+backward at 1920x1080 / 1M triangles.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one forward + one backward of the render op (`_C.render_forward_cuda` +
`_C.render_backward_cuda`, i.e. everything behind the reference's drop-in boundary:
binning, (tile,depth) sort, composite, gradient scatter, incl. the pair-count read-back)
over the SURVEY §8(d) synthetic scene, inputs resident in HBM.  The Python host prep
(projection, AA tables, rays) runs once outside the timed region, as §8(d) prescribes.

N > 1: the same 1080p frame is sharded by 16-pixel tile rows over the ranks
(dmesh2_renderer_amd.sharding) with ONE all-reduce of the packed gradients per step
=> "scaling": "strong".

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
`roofline` (dominant kernel, algorithmic bytes / live hipEvent duration) and
`cpu_baseline` (the CPU oracle -- a port, the reference has no CPU path -- on a bounded
band of the same frame, on this host's cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (W, H, F, cfg index for the seed)
    "cfg4": (1920, 1080, 1_000_000, 4),
    "cfg2": (512, 512, 50_000, 2),
    "cfg1": (256, 256, 2_000, 1),
    "cfg5": (3840, 2160, 2_000_000, 5),     # BASELINE configs[4] (quoted on 8 GPUs; fits one MI355X: ~1.3 GB of tensors)
}
HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6.29e12


def alg_bytes(B, P, F, N, Tn, R):
    """Algorithmic bytes per launch of the two composite kernels and per frame."""
    geom = 12 * F + 12 * P + 12 * P + 4 * F          # faces, verts, verts_color, faces_opacity
    per_view = 12 * B * P + 4 * B * F + 114 * B * F  # verts_ndc, faces_intense, six AA tables
    rays = 24 * N
    saved = 12 * N + 8 * Tn + 4 * R                  # final_T, final_prev_T, n_contrib, ranges, face_list
    fwd_kernel = geom + per_view + rays + 16 * N + saved
    grads = 24 * P + 12 * B * P + 4 * F + 28 * B * F
    bwd_kernel = geom + per_view + rays + 16 * N + saved + grads
    binning = (12 * F + 12 * B * P + 8 * B * P) + 16 * B * F + 44 * R    # preprocess inputs + face state + keys/sort/ranges
    return dict(forward_composite=fwd_kernel, backward_composite=bwd_kernel, frame=fwd_kernel + bwd_kernel + binning)


AA_TEMPERATURE = 1.0      # --aa-temperature; 1.0 is the BASELINE workload
_LAST = {}                # the scene behind the last build_inputs() (host-prep inputs of the multi-GPU step)


def build_inputs(cfg, device, rank, world):
    from dmesh2_renderer_amd import scenes
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import _C
    W, H, F, ci = CONFIGS[cfg]
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + ci).to(device)
    _LAST["scene"] = sc
    got = {}
    real = _C.render_forward_cuda

    def capture(*args):
        got["args"] = args
        B = args[8].shape[0]
        z = torch.zeros((0,), device=device)
        return 0, torch.zeros((B, H, W, 3), device=device), torch.zeros((B, H, W), device=device), z, z, z, z, z, z, z

    r = dm2.Renderer(sc.mv, sc.proj, W, H, device, aa_grad_buffer_size=20)
    _C.render_forward_cuda = capture
    try:
        with torch.no_grad():
            r([0], torch.zeros((1, 2), dtype=torch.int64, device=device), W, H, sc.verts, sc.faces, sc.verts_color,
              sc.faces_opacity, sc.faces_intense, sc.background, aa_temperature=AA_TEMPERATURE)
    finally:
        _C.render_forward_cuda = real
    args = list(got["args"])
    g = torch.Generator().manual_seed(scenes.SEED_BASE + 100 + ci)
    dLc = torch.randn((1, H, W, 3), generator=g).to(device)
    dLd = torch.randn((1, H, W), generator=g).to(device)
    return args, dLc, dLd, (W, H, F)


def host_prep_ms(cfg, device, iters=10):
    """Host prep of Renderer.forward (projection + AA tables) forward+backward, reported NEXT TO the metric
    (excluded from `value`): the reference-shaped torch ops vs the fused HIP prep (8(f) rank 1)."""
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import prep, scenes
    from dmesh2_renderer_amd.pyrenderer import Triangles
    W, H, F, ci = CONFIGS[cfg]
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + ci).to(device)
    r = dm2.Renderer(sc.mv, sc.proj, W, H, device)
    mv, proj = r.mv[[0]], r.proj[[0]]
    verts = sc.verts.clone().requires_grad_(True)
    faces32 = sc.faces.to(torch.int32)
    g_ndc = torch.randn((1, verts.shape[0], 3), device=device)
    g_aa = torch.randn((1, F, 3, 2), device=device)

    def torch_step():
        verts.grad = None
        ndc, image = r.compute_verts_ndc_image(verts, mv, proj)
        corners = image[:, sc.faces.flatten()].view(-1, 3, 2)
        tri = Triangles(corners[:, 0], corners[:, 1], corners[:, 2])
        torch.autograd.backward([ndc, tri.verts.reshape(1, F, 3, 2)], [g_ndc, g_aa])

    def fused_step():
        verts.grad = None
        out = prep.prepare(verts, faces32, mv, proj, W, H)
        torch.autograd.backward([out[0], out[2]], [g_ndc, g_aa])

    res = {}
    for name, fn in (("torch_fwd_bwd", torch_step), ("fused_hip_fwd_bwd", fused_step)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(device)
        res[name] = round((time.perf_counter() - t0) / iters * 1e3, 4)
    return res


def point_sampled_ms(cfg, device, steps=5):
    """SURVEY.md 8(d) asks for the aa_temperature = 0 figure next to the headline one: same frame, same faces,
    point-sampled coverage (no AA), forward+backward ms per step.  Reported in `config`, never in `value`."""
    global AA_TEMPERATURE
    from dmesh2_renderer_amd.sharding import BandShardedOp
    old, AA_TEMPERATURE = AA_TEMPERATURE, 0.0
    try:
        args, dLc, dLd, _ = build_inputs(cfg, device, 0, 1)
    finally:
        AA_TEMPERATURE = old
    op = BandShardedOp(args, 1, 0)

    def step():
        op.forward()
        op.backward(dLc, dLd)

    for _ in range(2):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    return round((time.perf_counter() - t0) / steps * 1e3, 4)


def cpu_baseline(args, dLc, dLd, W, H, budget_rows=None):
    """Oracle (CPU port) on a band of the frame: same faces, `rows` pixel rows in the middle."""
    from oracle import cpu as orc
    nthreads = orc.max_threads()
    rows = min(budget_rows or 128, H)
    y0 = max(((H // 2 - rows // 2) // 16) * 16, 0)
    a = [x.detach().cpu().numpy() if torch.is_tensor(x) else x for x in args]
    a[1] = a[1].copy(); a[1][:, 1] += y0
    a[3] = rows
    a[19] = np.ascontiguousarray(a[19][:, y0:y0 + rows]); a[20] = np.ascontiguousarray(a[20][:, y0:y0 + rows])
    gc = np.ascontiguousarray(dLc.cpu().numpy()[:, y0:y0 + rows]); gd = np.ascontiguousarray(dLd.cpu().numpy()[:, y0:y0 + rows])
    t0 = time.perf_counter()
    f = orc.render_forward_cuda(*a, nthreads=nthreads)
    t1 = time.perf_counter()
    orc.render_backward_cuda(f, gc, gd, nthreads=nthreads)
    t2 = time.perf_counter()
    npx = rows * W
    return {
        "value": round(npx / (t2 - t0) / 1e6, 4), "unit": "Mpixels/s", "cores": nthreads, "kind": "port",
        "sample": f"rows [{y0},{y0 + rows}) of the {W}x{H} frame ({npx} px, all {a[5].shape[0]} faces binned; "
                  f"band has {f.num_rendered} tile-face pairs); fwd {t1 - t0:.2f} s + bwd {t2 - t1:.2f} s wall, OpenMP over tiles",
        "fwd_s": round(t1 - t0, 3), "bwd_s": round(t2 - t1, 3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--aa-temperature", type=float, default=1.0,
                    help="AA temperature of the workload (1.0 = BASELINE; 0.0 = point-sampled coverage, SURVEY 8(d) asks for both)")
    ap.add_argument("--cpu-rows", type=int, default=1088,
                    help="rows of the frame the CPU baseline renders (default: the whole 1080p frame, a few seconds on a 128-thread host)")
    opt = ap.parse_args()
    global AA_TEMPERATURE
    AA_TEMPERATURE = opt.aa_temperature

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product has no CPU path)")
    # rehearsal knobs (one-GPU boxes): DM2_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0, DM2_BENCH_BACKEND=gloo
    # swaps the transport.  The driver's real runs use neither: one rank per GPU, nccl (= RCCL over xGMI).
    if os.environ.get("DM2_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DM2_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from dmesh2_renderer_amd import _C
    from dmesh2_renderer_amd.sharding import BandShardedOp
    _C.load_library()
    args, dLc, dLd, (W, H, F) = build_inputs(opt.config, device, rank, world)
    op = BandShardedOp(args, world, rank)
    dLc_b = dLc[:, op.y0:op.y0 + op.rows].contiguous()
    dLd_b = dLd[:, op.y0:op.y0 + op.rows].contiguous()

    # N > 1: "leaves" (default) all-reduces only the gradients of the leaves -- [dverts | dverts_color | dfaces_opacity
    # | dfaces_intense], 80 MB at cfg4 -- after pushing this rank's dverts_ndc / daa_face_verts partials through the fused
    # host-prep backward locally (sharding.BandShardedOp.backward_leaves); "op6" all-reduces the op's six gradient
    # tensors (140 MB, SURVEY 8(e) as written).  Both leave every rank with the full gradients of the leaves.
    reduce_mode = os.environ.get("DM2_BENCH_REDUCE", "leaves")
    sc = _LAST["scene"]
    prep_inputs = (args[4], args[5], sc.mv[[0]].contiguous(), sc.proj[[0]].contiguous(), W, H)

    def step():
        op.forward()
        if world > 1 and reduce_mode == "leaves":
            return op.backward_leaves(dLc_b, dLd_b, prep_inputs)
        return op.backward(dLc_b, dLd_b)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opt.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = dt / opt.steps * 1e3

    # per-stage hipEvent timing of the same step (separate passes so the events do not perturb `value`)
    stage_ms = {}
    _C.profile_enable(True)
    acc = {}
    reps = max(3, min(10, opt.steps))
    for _ in range(reps):
        step()
        for k, v in _C.profile_read().items():
            acc.setdefault(k, []).append(v)
    _C.profile_enable(False)
    stage_ms = {k: float(np.median(v)) for k, v in acc.items()}

    R = op.fwd[0] if op.fwd is not None else 0
    B, P = args[8].shape[0], args[4].shape[0]
    N_band = B * op.rows * W
    Tn_band = B * ((W + 15) // 16) * ((op.rows + 15) // 16)
    alg = alg_bytes(B, P, F, N_band, Tn_band, R)
    dom = max(("forward_composite", "backward_composite"), key=lambda k: stage_ms.get(k, 0.0))
    dom_ms = stage_ms.get(dom, 0.0)
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    kname = "k_render_backward" if dom == "backward_composite" else "k_render_forward"
    traffic, traffic_src, valu_util = None, None, None
    try:    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), same kernel + config only
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if tj.get("config") == opt.config and world == 1 and kname in tj:
            traffic = int((tj[kname]["fetch_kb"] + tj[kname]["write_kb"]) * 1024)
            if "sq_insts_valu" in tj[kname] and dom_ms > 0:
                # what actually bounds the kernel: share of VALU issue cycles, from the committed
                # SQ_INSTS_VALU count of this kernel and the launch time measured in this run
                valu_util = tj[kname]["sq_insts_valu"] * 4.0 / (1024 * 2.4e9 * dom_ms * 1e-3)
            traffic_src = "profiles/r01_traffic.json (FETCH_SIZE + WRITE_SIZE, raw; see its note on the gfx950 correction)"
    except (OSError, ValueError, KeyError):
        pass
    roofline = {
        "bound": "hbm", "kernel": kname,
        "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": round(achieved / (HBM_PEAK / 1e9), 5), "traffic": traffic, "traffic_source": traffic_src,
        "valu_issue_util": None if valu_util is None else round(valu_util, 4),
        "alg_bytes_per_launch": alg[dom], "avg_launch_ms": round(dom_ms, 4),
        "frame_alg_bytes": alg["frame"], "frame_frac": round(alg["frame"] / (ms_step * 1e-3) / HBM_PEAK, 5),
    }

    if rank == 0:
        tri_cnt = op.fwd[5] if op.fwd is not None else None
        out = {
            "metric": "Mpixels/s fwd+bwd @1080p/1M tris" if (opt.config == "cfg4" and AA_TEMPERATURE == 1.0) else f"Mpixels/s fwd+bwd ({opt.config})",
            "value": round(W * H / (ms_step * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"Renderer forward+backward {W}x{H}, {F} triangles (soup, P=3F), B=1, aa_temperature={AA_TEMPERATURE}, K=20, "
                            f"AA visibility gradients on ({'BASELINE configs[3]' if opt.config == 'cfg4' else opt.config})",
                "sharding": (f"tile-row bands x{world}, " + ("one all-reduce of the leaf gradients (24P+4F+4BF bytes; dverts_ndc / "
                                                             "daa_face_verts partials go through the fused prep backward locally)"
                                                             if reduce_mode == "leaves" else "one all-reduce of the six packed op gradients"))
                            if world > 1 else "single GPU",
                "num_rendered_rank0": int(R), "grad_Mtris_per_s": round(F / max(stage_ms.get("backward_composite", 0.0), 1e-9) / 1e3, 2),
                "stage_ms_rank0": {k: round(v, 4) for k, v in stage_ms.items()},
                "aa_records_per_pixel_rank0": round(float(tri_cnt.float().mean().item()), 3) if tri_cnt is not None else None,
            },
            "roofline": roofline,
        }
        if world == 1 and not opt.no_cpu:
            out["config"]["host_prep_ms_not_in_value"] = host_prep_ms(opt.config, device)
            if AA_TEMPERATURE == 1.0:
                out["config"]["ms_per_step_at_aa_temperature_0"] = point_sampled_ms(opt.config, device)
            out["cpu_baseline"] = cpu_baseline(args, dLc, dLd, W, H, opt.cpu_rows)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
