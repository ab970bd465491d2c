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
(dmesh2_renderer_amd.sharding); every rank ends the step with the full gradients of the
leaves => "scaling": "strong".  `python bench.py --gpus N` starts the N ranks itself
(a child `torch.distributed.run`, before this process touches the GPU); under an
external launcher (WORLD_SIZE set) it is one of the ranks.

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
    # the reference's own parallel axis is the batch of views (forward.cu:191 blockIdx.z, renderer.cu:457): four cameras on the
    # cfg4 scene, each rendering one 512 x 512 window of its 1080p image in ONE call (B = 4)
    "cfg4_b4": (1920, 1080, 1_000_000, 4),
    # large triangles, depth complexity 60 (SURVEY 8(d) generator with 15 x the coverage): most pairs are fully covered
    "cfg1_dc60": (256, 256, 2_000 * 15, 1),
    # the cfg4 triangles with their faces (and the soup's vertices) in the order of the 16 x 16 tile of their centroid instead of
    # at random: what a mesh whose face order follows space looks like to the memory system (a diagnostic, not a BASELINE config)
    "cfg4_sorted": (1920, 1080, 1_000_000, 4),
}
VIEWS = {"cfg4_b4": dict(cams=4, pw=512, ph=512, pm=[[0, 0], [704, 284], [1408, 568], [640, 0]])}
DEPTH_COMPLEXITY = {"cfg1_dc60": 60.0}
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


def spatially_sorted(sc):
    """The soup with its faces ordered by the tile of their centroid (row-major tiles), vertices 3 f .. 3 f + 2 following."""
    from dmesh2_renderer_amd import scenes
    F = sc.faces.shape[0]
    v = sc.verts.reshape(F, 3, 3)
    c = v.mean(1)
    depth = scenes.CAM_DIST - c[:, 2]
    t = scenes.TAN_HALF_FOV
    px = (c[:, 0] / (depth * t * sc.width / sc.height) + 1.0) * 0.5 * sc.width
    py = (c[:, 1] / (depth * t) + 1.0) * 0.5 * sc.height
    key = (py / 16).floor().clamp(0, 1e6).long() * 4096 + (px / 16).floor().clamp(0, 4095).long()
    perm = torch.argsort(key, stable=True)
    sc.verts = v[perm].reshape(-1, 3).contiguous()
    sc.verts_color = sc.verts_color.reshape(F, 3, 3)[perm].reshape(-1, 3).contiguous()
    sc.faces_opacity = sc.faces_opacity[perm].contiguous()
    sc.faces_intense = sc.faces_intense[:, perm].contiguous()
    return sc


def build_inputs(cfg, device, rank, world):
    from dmesh2_renderer_amd import scenes
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import _C
    W, H, F, ci = CONFIGS[cfg]
    view = VIEWS.get(cfg)
    kw = dict(num_cams=view["cams"]) if view else {}
    if cfg in DEPTH_COMPLEXITY:
        # (the generator sizes the triangles for a mean depth complexity of 4 at F faces: same triangles, 15 x as many)
        kw["depth_complexity"] = DEPTH_COMPLEXITY[cfg]
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + ci, **kw)
    if cfg == "cfg4_sorted":
        sc = spatially_sorted(sc)
    sc = sc.to(device)
    _LAST["scene"] = sc
    got = {}
    real = _C.render_forward_cuda

    def capture(*args):
        got["args"] = args
        B = args[8].shape[0]
        z = torch.zeros((0,), device=device)
        return 0, torch.zeros((B, H, W, 3), device=device), torch.zeros((B, H, W), device=device), z, z, z, z, z, z, z

    # (the 21 arguments of the reference's boundary as they are: the six AA tables materialised, not built inside the op)
    r = dm2.Renderer(sc.mv, sc.proj, W, H, device, aa_grad_buffer_size=20, tables_from_image=False)
    _C.render_forward_cuda = capture
    try:
        with torch.no_grad():
            if view:
                bidx = list(range(view["cams"]))
                r(bidx, torch.tensor(view["pm"], dtype=torch.int64, device=device), view["pw"], view["ph"], sc.verts, sc.faces, sc.verts_color,
                  sc.faces_opacity, sc.faces_intense[bidx], sc.background, aa_temperature=AA_TEMPERATURE)
            else:
                r([0], torch.zeros((1, 2), dtype=torch.int64, device=device), W, H, sc.verts, sc.faces, sc.verts_color,
                  sc.faces_opacity, sc.faces_intense, sc.background, aa_temperature=AA_TEMPERATURE)
    finally:
        _C.render_forward_cuda = real
    args = list(got["args"])
    g = torch.Generator().manual_seed(scenes.SEED_BASE + 100 + ci)
    Bv, ph, pw = (view["cams"], view["ph"], view["pw"]) if view else (1, H, W)
    dLc = torch.randn((Bv, ph, pw, 3), generator=g).to(device)
    dLd = torch.randn((Bv, ph, pw), generator=g).to(device)
    return args, dLc, dLd, (pw, ph, F)


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


def end_to_end_ms(cfg, device, iters=10):
    """What an unmodified caller of the module pays per training step: Renderer.forward (host prep + op) + loss.backward()
    all the way to verts / verts_color / faces_opacity / faces_intense, with the default (fused) host prep and with the
    reference-shaped torch prep.  Reported next to the metric, never in `value`."""
    import dmesh2_renderer_amd as dm2
    from dmesh2_renderer_amd import scenes
    W, H, F, ci = CONFIGS[cfg]
    sc = scenes.triangle_soup(W, H, F, scenes.SEED_BASE + ci).to(device)
    g = torch.Generator().manual_seed(scenes.SEED_BASE + 100 + ci)
    wc = torch.randn((1, H, W, 3), generator=g).to(device); wd = torch.randn((1, H, W), generator=g).to(device)
    pm = torch.zeros((1, 2), dtype=torch.int64, device=device)
    res = {}
    for name, fused in (("fused_prep", True), ("torch_prep", False)):
        r = dm2.Renderer(sc.mv, sc.proj, W, H, device, fused_prep=fused)
        leaves = [sc.verts.clone().requires_grad_(True), sc.verts_color.clone().requires_grad_(True),
                  sc.faces_opacity.clone().requires_grad_(True), sc.faces_intense.clone().requires_grad_(True)]

        def one():
            for t in leaves:
                t.grad = None
            color, depth = r([0], pm, W, H, leaves[0], sc.faces, leaves[1], leaves[2], leaves[3], sc.background, aa_temperature=AA_TEMPERATURE)
            ((color * wc).sum() + (depth * wd).sum()).backward()

        for _ in range(3):
            one()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        torch.cuda.synchronize(device)
        res[name] = round((time.perf_counter() - t0) / iters * 1e3, 4)
        del r
    return res


def point_sampled_ms(cfg, device, steps=20):
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

    for _ in range(5):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    return round((time.perf_counter() - t0) / steps * 1e3, 4)


def _cpu_leg(args, dLc, dLd, W, H, rows, nthreads):
    """Oracle (CPU port) forward + backward on `rows` pixel rows in the middle of the frame, `nthreads` OpenMP threads."""
    from oracle import cpu as orc
    rows = min(rows, H)
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
    return {"value": round(npx / (t2 - t0) / 1e6, 4), "unit": "Mpixels/s", "cores": nthreads, "fwd_s": round(t1 - t0, 3),
            "bwd_s": round(t2 - t1, 3), "rows": [y0, y0 + rows], "pixels": npx, "tile_face_pairs": int(f.num_rendered),
            "grad_Mtris_per_s": round(a[5].shape[0] / (t2 - t1) / 1e6, 4)}


def cpu_baseline(args, dLc, dLd, W, H, budget_rows=None, device=None):
    """the CPU port (the reference has no CPU path) on this host's cores, in the same run: the BASELINE
    workload on all cores and on one thread (a bounded band: one thread needs ~1 min for the whole frame), and
    BASELINE configs[1] (512x512 / 50 k) on all cores and on one thread."""
    from oracle import cpu as orc
    nthreads = orc.max_threads()
    main = _cpu_leg(args, dLc, dLd, W, H, budget_rows or 1088, nthreads)
    one = _cpu_leg(args, dLc, dLd, W, H, 32, 1)
    out = {
        "value": main["value"], "unit": "Mpixels/s", "cores": nthreads, "kind": "port",
        "sample": f"rows [{main['rows'][0]},{main['rows'][1]}) of the {W}x{H} frame ({main['pixels']} px, all {args[5].shape[0]} faces binned; "
                  f"band has {main['tile_face_pairs']} tile-face pairs); fwd {main['fwd_s']:.2f} s + bwd {main['bwd_s']:.2f} s wall, OpenMP over tiles",
        "fwd_s": main["fwd_s"], "bwd_s": main["bwd_s"], "grad_Mtris_per_s": main["grad_Mtris_per_s"],
        "one_thread": {k: one[k] for k in ("value", "unit", "cores", "fwd_s", "bwd_s", "rows", "pixels")},
    }
    if device is not None:
        try:
            a2, c2, d2, (W2, H2, F2) = build_inputs("cfg2", device, 0, 1)
            out["cfg2_512x512_50k"] = {"all_threads": _cpu_leg(a2, c2, d2, W2, H2, H2, nthreads), "one_thread": _cpu_leg(a2, c2, d2, W2, H2, H2, 1)}
        except Exception as ex:      # the baseline legs never take the bench line down
            out["cfg2_512x512_50k"] = {"error": repr(ex)[:200]}
    return out


VALU_PEAK_GINSTR_PER_S_PER_SIMD = 1.05   # measured on this part (tools/calib/valu_calib.hip, profiles/r02_valu_calibration.jsonl):
                                         # independent v_add / v_mul wave64 instructions, >= 2 waves per SIMD; = 2.3 cycles at 2.4 GHz
                                         # (one wave alone: 0.48; v_fma 0.83; DPP / v_min3 / v_bcnt 0.57; IEEE fp32 divide 0.056)


def _visible_gpus():
    """GPUs of this node as the kernel driver lists them (KFD topology nodes with SIMDs), without touching the HIP runtime;
    None when sysfs does not say (the ranks then find out themselves)."""
    import glob
    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if seen and vis:
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n if seen else None


def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources of the loaded library: counters recorded under profiles/ are only
    quoted for the very sources they were measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dmesh2_renderer_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(ROOT, "include", "dm2_hip.h")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def self_launch(opt, argv):
    """`python bench.py --gpus N` outside a launcher: start N ranks as a CHILD process group (torch.distributed.run) before
    this process has touched the GPU, pass its output through and exit with its code."""
    import socket
    import subprocess
    ndev = _visible_gpus()                                 # from sysfs: this parent never calls into HIP
    if ndev is not None and opt.gpus > ndev and os.environ.get("DM2_BENCH_SINGLE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {opt.gpus}: only {ndev} GPU(s) visible (DM2_BENCH_SINGLE_DEVICE=1 rehearses N ranks on one)")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={opt.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


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
    if opt.config in VIEWS or opt.config in DEPTH_COMPLEXITY:
        opt.no_cpu = True              # (the side legs -- host prep, end to end, CPU baseline -- belong to the BASELINE configs)
    global AA_TEMPERATURE
    AA_TEMPERATURE = opt.aa_temperature

    if "WORLD_SIZE" not in os.environ and opt.gpus > 1:
        sys.exit(self_launch(opt, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if opt.gpus != world:
        raise SystemExit(f"bench.py: --gpus {opt.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product has no CPU path)")
    # rehearsal knobs (one-GPU boxes): DM2_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0, DM2_BENCH_BACKEND=gloo
    # swaps the transport.  The driver's real runs use neither: one rank per GPU, nccl (= RCCL over xGMI).
    if os.environ.get("DM2_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    backend = os.environ.get("DM2_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from dmesh2_renderer_amd import _C
    from dmesh2_renderer_amd.sharding import BandShardedOp, sparse_exchange_bytes
    _C.load_library()
    args, dLc, dLd, (W, H, F) = build_inputs(opt.config, device, rank, world)
    op = BandShardedOp(args, world, rank)
    dLc_b = dLc[:, op.y0:op.y0 + op.rows].contiguous()
    dLd_b = dLd[:, op.y0:op.y0 + op.rows].contiguous()

    # N > 1, what crosses the links (DM2_BENCH_REDUCE):
    #   "leaves" (default)  leaf gradients, ONE dense all-reduce of 24P + 4F + 4BF bytes (80 MB at cfg4) on the backward's own
    #                       packed buffer: no local packing at all
    #   "sparse"            leaf gradients, touched rows only: all-to-all to the row owners + all-gather of the reduced slices.
    #                       Fewer bytes per link, but its device-side packing costs 0.4-0.6 ms per step and rank
    #                       (tools/exchange_time.py on one MI355X) -- more than the bytes it saves at 2..8 GPUs of one node
    #   "op6"               the op's six gradient tensors, one dense all-reduce (140 MB; SURVEY 8(e) as written)
    # All three leave every rank with the full gradients; "sparse"/"leaves" push this rank's dverts_ndc / daa_face_verts
    # partials through the fused host-prep backward locally first (sharding.BandShardedOp.backward_leaves).
    reduce_mode = os.environ.get("DM2_BENCH_REDUCE", "leaves")
    if backend != "nccl" and reduce_mode == "sparse" and world > 1 and device.type == "cuda":
        reduce_mode = "leaves"                             # gloo has no all-to-all on device tensors
    sc = _LAST["scene"]
    prep_inputs = (args[4], args[5], sc.mv[[0]].contiguous(), sc.proj[[0]].contiguous(), W, H)
    ev_x = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    exch_ms = []

    def step(timed_exchange=False):
        op.forward()
        if world == 1:
            return op.backward(dLc_b, dLd_b)
        if reduce_mode == "op6":
            g = op.backward(dLc_b, dLd_b, reduce=False)
            if timed_exchange:
                ev_x[0].record()
            from dmesh2_renderer_amd.sharding import allreduce_packed_grads
            g = allreduce_packed_grads(g)
            if timed_exchange:
                ev_x[1].record(); ev_x[1].synchronize(); exch_ms.append(ev_x[0].elapsed_time(ev_x[1]))
            return g
        if not timed_exchange:
            return op.backward_leaves(dLc_b, dLd_b, prep_inputs, exchange="sparse" if reduce_mode == "sparse" else "dense")
        # same step with events around the exchange: local part first (forward done above)
        old_ws, op.world_size = op.world_size, 1
        try:
            leaves = op.backward_leaves(dLc_b, dLd_b, prep_inputs)
        finally:
            op.world_size = old_ws
        ev_x[0].record()
        if reduce_mode == "sparse":
            from dmesh2_renderer_amd.sharding import reduce_leaves_sparse
            out = reduce_leaves_sparse(*leaves, args[5], op.touched_faces())
        else:
            span = torch.cat([t.reshape(-1) for t in leaves])
            dist.all_reduce(span)
            out = span
        ev_x[1].record(); ev_x[1].synchronize(); exch_ms.append(ev_x[0].elapsed_time(ev_x[1]))
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_check = None
    if world > 1 and reduce_mode == "sparse":
        # one untimed cross-check of the sparse exchange against the dense all-reduce on this very frame: every rank must
        # end with the same leaf gradients either way (different summation order: 1e-5 of each tensor's maximum).  All
        # ranks take the same decision (MIN over ranks); a disagreement falls back to the dense exchange and says so.
        op.forward()
        sp = op.backward_leaves(dLc_b, dLd_b, prep_inputs, exchange="sparse")
        de = op.backward_leaves(dLc_b, dLd_b, prep_inputs, exchange="dense")
        okf = 1.0
        for a, b in zip(sp, de):
            tol = 1e-5 * max(float(b.abs().max().item()), 1e-12)
            if a.shape != b.shape or float((a - b).abs().max().item()) > tol:
                okf = 0.0
        okt = torch.tensor([okf], device=device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        exchange_check = "sparse == dense on every rank" if float(okt.item()) == 1.0 else "sparse != dense: fell back to dense"
        if float(okt.item()) != 1.0:
            reduce_mode = "leaves"
        del sp, de
    for _ in range(opt.warmup):
        step()
    barrier()
    # the timed region: exactly `steps` steps between two barriers; per-step hipEvents ride along for the median
    # (ONE event per step boundary: the end of step i is the start of step i + 1; every record is a signal packet between two
    # steps' kernels, ~5 us each in a kernel trace)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(opt.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(opt.steps):
        step()
        evs[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = dt / opt.steps * 1e3
    ev_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(opt.steps))
    ms_median = ev_ms[len(ev_ms) // 2] if ev_ms else 0.0

    # per-stage hipEvent timing of the same step (separate passes so the events do not perturb `value`)
    _C.profile_enable(True)
    acc = {}
    reps = max(3, min(10, opt.steps))
    for _ in range(reps):
        step()
        for k, v in _C.profile_read().items():
            acc.setdefault(k, []).append(v)
    _C.profile_enable(False)
    stage_ms = {k: float(np.median(v)) for k, v in acc.items()}
    if world > 1:
        for _ in range(reps):
            step(timed_exchange=True)

    # the same step with all 21 arguments of the reference's boundary taken as they are (the six AA tables read from the
    # tensors the caller built, not derived inside the op's plan)
    strict_ms = None
    if world == 1 and op.tables_from_image:
        op21 = BandShardedOp(args, 1, 0, tables_from_image=False)
        for _ in range(3):
            op21.forward(); op21.backward(dLc_b, dLd_b)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(max(5, opt.steps // 2)):
            op21.forward(); op21.backward(dLc_b, dLd_b)
        torch.cuda.synchronize()
        strict_ms = (time.perf_counter() - ts) / max(5, opt.steps // 2) * 1e3
        del op21

    R = op.fwd[0] if op.fwd is not None else 0
    B, P = args[8].shape[0], args[4].shape[0]
    N_band = B * op.rows * W
    Tn_band = B * ((W + 15) // 16) * ((op.rows + 15) // 16)
    alg = alg_bytes(B, P, F, N_band, Tn_band, R)
    dom = max(("forward_composite", "backward_composite"), key=lambda k: stage_ms.get(k, 0.0))
    dom_ms = stage_ms.get(dom, 0.0)
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    fwd_mode = getattr(op.fwd[8], "_dm2_fwd_mode", None) if op.fwd is not None else None
    bwd_kernel = {_C.FWD_POOL: "k_render_backward_fast", _C.FWD_MASKS: "k_render_backward_mask"}.get(fwd_mode, "k_render_backward")
    if AA_TEMPERATURE == 0.0:
        bwd_kernel = "k_render_backward_point"
    kname = bwd_kernel if dom == "backward_composite" else ("k_render_forward_queue" if AA_TEMPERATURE > 0.0 else "k_render_forward_point")
    traffic, traffic_src, valu_util, valu_note = None, None, None, None
    try:    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), same kernel + config only
        tj = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
        # (only for the sources the counters were measured on: a changed kernel must not quote stale counters)
        if tj.get("config") == opt.config and tj.get("csrc_sha16") == csrc_digest() and world == 1 and AA_TEMPERATURE == 1.0 and kname in tj:
            # gfx950: FETCH_SIZE counts a 128-B request as 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM/rocprofv3);
            # both readings are kept in the json, the corrected one is reported here
            traffic = int(tj[kname]["hbm_bytes_corrected"])
            if "sq_insts_valu" in tj[kname] and dom_ms > 0:
                valu_util = tj[kname]["sq_insts_valu"] / (1024 * VALU_PEAK_GINSTR_PER_S_PER_SIMD * 1e9 * dom_ms * 1e-3)
                valu_note = (f"SQ_INSTS_VALU {tj[kname]['sq_insts_valu']:.4g} per launch (profiles/r03_traffic.json) / (1024 SIMDs x "
                             f"{VALU_PEAK_GINSTR_PER_S_PER_SIMD} G wave-instr/s measured by tools/calib/valu_calib.hip x launch time)")
            traffic_src = "profiles/r03_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; sources " + tj["csrc_sha16"] + ")"
    except (OSError, ValueError, KeyError):
        pass
    roofline = {
        "bound": "hbm", "kernel": kname,
        "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": round(achieved / (HBM_PEAK / 1e9), 5), "traffic": traffic, "traffic_source": traffic_src,
        "valu_issue_util": None if valu_util is None else round(valu_util, 4), "valu_issue_util_how": valu_note,
        "alg_bytes_per_launch": alg[dom], "avg_launch_ms": round(dom_ms, 4),
        "frame_alg_bytes": alg["frame"], "frame_frac": round(alg["frame"] / (ms_step * 1e-3) / HBM_PEAK, 5),
    }

    if rank == 0:
        tri_cnt = op.fwd[5] if op.fwd is not None else None
        bwd_ms = max(stage_ms.get("backward_composite", 0.0), 1e-9)
        cfg = {
            "workload": f"Renderer forward+backward {W}x{H}{' windows of 1920x1080 cameras' if opt.config in VIEWS else ''}, {F} triangles (soup, P=3F), B={B}, aa_temperature={AA_TEMPERATURE}, K=20, "
                        f"AA visibility gradients on ({'BASELINE configs[3]' if opt.config == 'cfg4' else opt.config})",
            "sharding": "single GPU" if world == 1 else f"tile-row bands x{world}, exchange={reduce_mode}",
            "op_path": ("the module's default: the op's plan builds the six AA tables from verts_image (DM2_FLAG_TABLES_FROM_IMAGE), as under "
                        "Renderer with its fused host prep" if op.tables_from_image else "all 21 arguments of the reference's boundary read as given"),
            "ms_per_step_21_argument_path": None if strict_ms is None else round(strict_ms, 4),
            "num_rendered_rank0": int(R), "grad_Mtris_per_s": round(F / bwd_ms / 1e3, 2),
            "ms_per_step_median_hipevent_rank0": round(ms_median, 4),
            "stage_ms_rank0": {k: round(v, 4) for k, v in stage_ms.items()},
            "aa_records_per_pixel_rank0": round(float(tri_cnt.float().mean().item()), 3) if tri_cnt is not None else None,
        }
        if world > 1:
            cfg["exchange_ms_rank0"] = round(float(np.median(exch_ms)), 4) if exch_ms else None
            cfg["exchange_check"] = exchange_check
            nf = int(op.touched_faces().sum().item()) if op.fwd is not None else 0
            cfg["exchange_bytes_sent_per_rank"] = sparse_exchange_bytes(P, F, B, world, nf, 3 * nf if P == 3 * F else min(P, 3 * nf))
            cfg["grad_Mtris_per_s"] = round(F / (ms_step * 1e-3) / 1e6, 2)     # whole job: every face's gradient per step
        out = {
            "metric": "Mpixels/s fwd+bwd @1080p/1M tris" if (opt.config == "cfg4" and AA_TEMPERATURE == 1.0) else f"Mpixels/s fwd+bwd ({opt.config})",
            "value": round(B * W * H / (ms_step * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg, "roofline": roofline,
        }
        if world == 1 and not opt.no_cpu:
            cfg["host_prep_ms_not_in_value"] = host_prep_ms(opt.config, device)
            cfg["end_to_end_ms_not_in_value"] = end_to_end_ms(opt.config, device)
            if AA_TEMPERATURE == 1.0:
                cfg["ms_per_step_at_aa_temperature_0"] = point_sampled_ms(opt.config, device)
            out["cpu_baseline"] = cpu_baseline(args, dLc, dLd, W, H, opt.cpu_rows, device)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
