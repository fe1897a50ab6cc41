#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native Gaussian-splatting raster path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE forward raster (preprocess -> scan -> read back R -> duplicateWithKeys -> radix sort -> tile
ranges -> blend) of ONE synthetic frame whose inputs already sit in HBM; the workload is BASELINE.json's metric
configuration: 1 M synthetic Gaussians ("cube" scene of SURVEY.md 8d, seed 0), 1024x1024, SH degree 0, fp32.
Multi-GPU: the raster of one frame does not shard (global depth sort + per-tile lists), the path shards over
SCENES: every rank renders its own frames with no data-path collective (weak scaling); value = frames of all
ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline"     -- for the dominant kernel of the step (largest hipEvent stage time): algorithmic bytes per launch
                    (SURVEY.md 8d per-unit figures, restated in DESIGN.md) / its measured average duration, vs 8 TB/s
  "cpu_baseline" -- the pure-PyTorch CPU rasterizer (oracle/torch_raster.py) on the host's cores, same workload, one full
                    frame; the single-thread C restatement (the parity checker) beside it.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (P, image size, scene kind)
    "1M_1024_cube": (1_000_000, 1024, "cube"),
    "1M_1024_shell": (1_000_000, 1024, "shell"),
    "100k_512_cube": (100_000, 512, "cube"),
    "100k_1024_cube": (100_000, 1024, "cube"),
    "1M_512_cube": (1_000_000, 512, "cube"),
    "500k_512_cube": (500_000, 512, "cube"),
}


def sort_passes(W, H):
    T = ((W + 15) // 16) * ((H + 15) // 16)
    msb = max(1, T.bit_length())  # smallest k with T >> k == 0
    return (32 + msb + 7) // 8


def algorithmic_bytes(stage: str, P: int, R: int, W: int, H: int, M: int = 1) -> float:
    """Per-launch algorithmic bytes of each pipeline stage (the terms of B_fwd / B_bwd)."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return {
        "preprocess": P * ((44 + 12 * M) + 75),
        "scan": 8 * P,
        "duplicate": 20 * P + 12 * R,
        "sort": 2 * 12 * R * sort_passes(W, H),
        "ranges": 8 * R + 8 * T,
        "blend": 40 * R + 20 * W * H,
        "blend_bwd": 40 * R + 12 * R + 28 * W * H,
        "preprocess_bwd": P * (56 + 75) + P * (12 + 8 + 24 + 12 * M + 4 + 12 + 16),
    }[stage]


def cpu_baseline(P, S, kind):
    """The north_star's CPU baseline: the pure-PyTorch CPU rasterizer (oracle/torch_raster.py) on this host's cores, on the
    SAME workload, ONE FULL FRAME (every tile: ~20 s on 32 threads -- round 2 timed every 16th tile and scaled).  The
    single-thread C restatement (oracle/libggd_oracle.so, the parity checker) is timed on the full frame as a second figure."""
    from gaussian_gan_decoder_amd.synthetic import make_scene
    from oracle import ggd_oracle as O, torch_raster as TR
    sc = make_scene(P, S, kind, seed=0)
    cam = sc.cam
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    # torch's intra-op pool on ALL cores of a 256-core host is slower than on 32 (measured: 27.7 s vs ~1 s for the
    # per-Gaussian + binning stages: the whole-array ops of this size do not scale past a few dozen threads), so the
    # baseline uses min(cores, 32) threads and says so; `cores` below is the number of threads actually used
    host_cores = os.cpu_count() or 1
    ncores = min(host_cores, 32)
    prev = torch.get_num_threads()
    torch.set_num_threads(ncores)
    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            g = TR.preprocess(sc.xyz, sc.opacities, sc.features_dc, sc.scales, sc.rotations, cam.world_view_transform,
                              cam.full_proj_transform, S, S, tanx, tany)
            b = TR.bin_and_sort(g, S, S)
            t1 = time.perf_counter()
            TR.blend(g, b, sc.bg, S, S)
            t2 = time.perf_counter()
    finally:
        torch.set_num_threads(prev)
    dt = t2 - t0
    kw = dict(means3D=sc.xyz.numpy(), opacities=sc.opacities.numpy(), shs=sc.features_dc.numpy(),
              scales=sc.scales.numpy(), rotations=sc.rotations.numpy(), viewmatrix=cam.world_view_transform.numpy(),
              projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=sc.bg.numpy(),
              W=S, H=S, tanfovx=tanx, tanfovy=tany)
    O.lib()
    t3 = time.perf_counter()
    O.forward(**kw)
    dtc = time.perf_counter() - t3
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": ncores, "kind": "port",
            "sample": f"pure-PyTorch CPU rasterizer (oracle/torch_raster.py), {ncores} torch threads, same workload "
                      f"({P} Gaussians, {S}x{S}, R = {b['num_rendered']}), one full frame: preprocess + binning + sort "
                      f"{t1 - t0:.2f} s, blend of all {g['gx'] * g['gy']} tiles {t2 - t1:.2f} s -> {dt:.2f} s/frame",
            "c_port_single_thread": {"value": 1.0 / dtc, "unit": "frames/s", "cores": 1, "kind": "port",
                                     "sample": f"1 full frame through oracle/libggd_oracle.so (gcc -O2): {dtc:.2f} s"},
            "host_cores": host_cores}


def tile_list_stats(img, W, H):
    """Mean / max length of the per-tile Gaussian lists of one rendered frame (from the `ranges` array of the image
    buffer the forward returned): SURVEY.md 8d asks for them next to every number."""
    from gaussian_gan_decoder_amd import _capi
    iv = _capi.img_view(W, H)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rg = img[iv.ranges:iv.ranges + 8 * T].view(torch.int32).reshape(T, 2).to(torch.int64)
    ln = (rg[:, 1] - rg[:, 0])
    return {"mean": float(ln.float().mean().item()), "max": int(ln.max().item()), "tiles": T}


def pmc_moved_bytes(workload):
    """Sum of the measured HBM bytes (committed PMC passes, see pmc_traffic) of every forward kernel of one frame."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "traffic*.json"))):
        try:
            doc = json.load(open(fn))
        except (OSError, ValueError):
            continue
        if doc.get("workload") != workload or "forward_kernels" not in doc:
            continue
        best = {"bytes": sum(doc["kernels"][k]["hbm_bytes"] * doc["forward_kernels"][k] for k in doc["forward_kernels"]
                             if k in doc["kernels"]), "source": os.path.relpath(fn, os.path.dirname(os.path.abspath(__file__)))}
    return best


def pmc_mlp(live_kernel_us):
    """MFMA evidence for the fused decoder MLP from the committed rocprofv3 passes (profiles/*/mlp_pmc.json, newest round
    last).  A profile whose kernel duration is more than 20 % off the time this run measures describes ANOTHER state of the
    kernel (round 5 replayed round 4's 953 us pass onto a 612 us kernel): it is refused and named in `stale`."""
    import glob
    best, stale = None, []
    root = os.path.dirname(os.path.abspath(__file__))
    for fn in sorted(glob.glob(os.path.join(root, "profiles", "*", "mlp_pmc.json"))):
        try:
            doc = json.load(open(fn))
        except (OSError, ValueError):
            continue
        doc["source"] = os.path.relpath(fn, root)
        k_us = doc.get("kernel_us")
        if not k_us or abs(k_us - live_kernel_us) > 0.2 * live_kernel_us:
            stale.append({"source": doc["source"], "kernel_us": k_us})
            continue
        best = doc
    return best, stale


def pmc_valu(workload, stage):
    """SQ_INSTS_VALU per launch of the stage's kernel from the same committed PMC pass (see pmc_traffic)."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "traffic*.json"))):
        try:
            doc = json.load(open(fn))
        except (OSError, ValueError):
            continue
        if doc.get("workload") != workload:
            continue
        k = doc.get("kernels", {}).get(doc.get("stage_to_kernel", {}).get(stage, ""))
        if k and "valu_wave_insts" in k:
            best = k["valu_wave_insts"]
    return best


def pmc_source(workload, stage):
    """The committed file pmc_traffic / pmc_valu replay their numbers from (bench.py cannot collect counters itself)."""
    import glob
    best = None
    root = os.path.dirname(os.path.abspath(__file__))
    for fn in sorted(glob.glob(os.path.join(root, "profiles", "*", "traffic*.json"))):
        try:
            doc = json.load(open(fn))
        except (OSError, ValueError):
            continue
        if doc.get("workload") == workload and doc.get("stage_to_kernel", {}).get(stage, "") in doc.get("kernels", {}):
            best = os.path.relpath(fn, root)
    return best


def roofline_entry(workload, stage, P, R, W, H, kernel_ms):
    """One roofline object (SURVEY 8d bytes of `stage` per launch / the kernel's live hipEvent time; counter traffic and VALU
    instruction count replayed from the committed PMC pass of the same workload, named in `replayed_from`)."""
    b = algorithmic_bytes(stage, P, R, W, H)
    ach = b / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": stage, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic(workload, stage), "algorithmic_bytes_per_launch": b, "kernel_ms": kernel_ms,
            "valu_wave_insts": pmc_valu(workload, stage), "replayed_from": pmc_source(workload, stage)}


def pmc_traffic(workload, stage):
    """HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    corrected as profiles/*/traffic.json states).  bench.py cannot collect counters itself; the number is the one
    measured with `rocprofv3 --pmc` on this same command (profiles/r01_final/).  None when no profile of this workload."""
    import glob
    best = None
    for fn in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "traffic*.json"))):
        try:
            doc = json.load(open(fn))
        except (OSError, ValueError):
            continue
        if doc.get("workload") != workload:
            continue
        k = doc.get("kernels", {}).get(doc.get("stage_to_kernel", {}).get(stage, ""))
        if k:
            best = k["hbm_bytes"]
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="1M_1024_cube", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backward", action="store_true", help="also time forward+backward (reported under 'extra')")
    ap.add_argument("--no-train", action="store_true", help="skip the data-parallel train-step section")
    ap.add_argument("--no-sweep", action="store_true", help="skip the forward-fps sweep over the other synthetic configs")
    ap.add_argument("--no-decode", action="store_true", help="skip the fused decode + render section (config 3)")
    ap.add_argument("--no-inflight", action="store_true", help="skip the several-frames-in-flight sections (profiling runs: their "
                    "overlapped frames stretch the kernels' durations in a rocprofv3 kernel trace)")
    ap.add_argument("--no-extra-rooflines", action="store_true", help="skip roofline_shell / roofline_backward / varying_scenes")
    ap.add_argument("--train-iters", type=int, default=8)
    ap.add_argument("--train-points", type=int, default=500_000, help="positions per scene (reference: 500 000)")
    ap.add_argument("--scenes-per-gpu", type=int, default=4)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    # one process per GPU.  (GGD_BENCH_SHARE_GPU=1 is a test hook: several ranks on the visible devices round-robin with
    # the gloo backend, to exercise the multi-rank control flow on a single-GPU box; never used for reported numbers.)
    share = os.environ.get("GGD_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if share else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from gaussian_gan_decoder_amd import _capi, rasterizer as R
    from gaussian_gan_decoder_amd.synthetic import make_scene, make_dL_dpix

    P, S, kind = WORKLOADS[args.workload]
    sc = make_scene(P, S, kind, seed=0).to(dev)  # identical scene on every rank (weak scaling: 1 frame/step/rank)
    cam = sc.cam
    empty = torch.empty(0, device=dev)
    scales, rots, opac, shs = sc.scales.contiguous(), sc.rotations.contiguous(), sc.opacities.contiguous(), \
        sc.features_dc.contiguous()
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    fargs = (sc.bg, sc.xyz, empty, opac, scales, rots, 1.0, empty, cam.world_view_transform,
             cam.full_proj_transform, tanx, tany, S, S, shs, 0, cam.camera_center, False, False)

    def step():
        return R.rasterize_gaussians_native(*fargs)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ctx = _capi.context_for(dev)

    def path_counters():
        return (ctx.get_option(_capi.STAT_MSD_FRAMES), ctx.get_option(_capi.STAT_SORT_RERUNS), ctx.capacity_retries)

    for _ in range(args.warmup):
        out = step()
    # pre-roll (untimed, whatever --warmup says): the single-call forward's speculation state -- the capacity hint, the streak that
    # arms the two-launch depth sort -- settles over the first ~10 frames of a shape; the timed region must be ONE path's steady
    # state, not a mix.  Settled = two consecutive frames on the two-launch sort, or the library says it
    # cannot be armed (option off), or 100 frames without arming (then the ordinary sort IS the steady state, and the line says so).
    preroll, settled = 0, "two_launch_sort_off"
    if ctx.get_option(_capi.OPT_MSD_SORT) and ctx.get_option(_capi.OPT_FOLD):
        settled, streak = "not_armed_after_100_frames", 0
        while preroll < 100:
            m_before = ctx.get_option(_capi.STAT_MSD_FRAMES)
            out = step()
            preroll += 1
            streak = streak + 1 if ctx.get_option(_capi.STAT_MSD_FRAMES) > m_before else 0
            if streak >= 2:
                settled = "two_launch_sort"
                break
    # ... and the device's clocks: with `--warmup 5 --steps 20` the timed region is 4 ms long and starts 2 ms after the process'
    # first kernel (measured: 4820 against 5090 frames/s for 200 steps).  GGD_BENCH_MIN_PREROLL untimed frames in all (default 128,
    # ~26 ms; 0 = only the path-settling frames above; measured with --steps 20: 0 -> 4760-4810, 64 -> 4950-5020, 256 -> 5005-5025
    # frames/s) are rendered before the timed region; the line reports how many.
    min_preroll = int(os.environ.get("GGD_BENCH_MIN_PREROLL", "128"))
    while args.warmup + preroll < min_preroll:
        out = step()
        preroll += 1
    barrier()
    c_before = path_counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    c_after = path_counters()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed_path = {"settled_as": settled, "preroll_frames": preroll, "min_untimed_frames": min_preroll,
                  "two_launch_sort_frames_timed": c_after[0] - c_before[0],
                  "ordinary_sort_frames_timed": args.steps - (c_after[0] - c_before[0]),
                  "sort_reruns_timed": c_after[1] - c_before[1], "capacity_retries_timed": c_after[2] - c_before[2]}
    # ---- the LAST TIMED frame against the exact forms, bit for bit: (a) the two-call form (ggd_forward_geometry
    #      -> host reads num_rendered -> ggd_forward_render into buffers laid out for exactly that; ordinary four-pass depth sort, no
    #      folding, no speculation) and (b) the literal duplicateWithKeys + 64-bit (tile | depth) radix sort path (debug = True), each
    #      on a FRESH context (a new stream has no capacity hint and no streak).  Any difference fails the run.
    def frames_equal(a, b):
        Rn = a[0]
        iv = _capi.img_view(S, S)
        T = ((S + 15) // 16) ** 2
        return bool(Rn == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[4][:4 * Rn], b[4][:4 * Rn])
                    and all(torch.equal(a[5][o_:o_ + n_], b[5][o_:o_ + n_])
                            for o_, n_ in ((iv.ranges, 8 * T), (iv.final_T, 4 * S * S), (iv.n_contrib, 4 * S * S))))
    verified = {}
    for name, dbg in (("verified_vs_two_call", False), ("verified_vs_radix_sort_path", True)):
        vs = torch.cuda.Stream(device=dev)
        vs.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(vs):
            _capi._contexts.pop((dev.index, int(vs.cuda_stream)), None)    # (a destroyed stream's handle may be handed out again)
            assert not _capi.context_and_stream(dev)[0].capacity_hint, "the verification context is not fresh"
            ref = R.rasterize_gaussians_native(*(fargs[:-1] + (dbg,)))
        vs.synchronize()
        torch.cuda.synchronize(dev)
        verified[name] = frames_equal(out, ref)
        _capi._contexts.pop((dev.index, int(vs.cuda_stream)), None)
        del ref, vs
    if dist is not None:
        t = torch.tensor([float(all(verified.values()))], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if float(t.item()) == 0.0:
            verified = {k: False for k in verified}
    num_rendered = out[0]
    tile_lists = tile_list_stats(out[5], S, S)
    # what the production path (depth sort of the visible Gaussians + row / column binning) has to move, per frame: visible
    # Gaussians and the (Gaussian, tile row) entries of the binning's first level, from the frame's own geometry buffer
    gv = _capi.geom_view(P)
    rect = out[3][gv.rect:gv.rect + 8 * P].view(torch.int32).reshape(P, 2)
    touched = out[3][gv.tiles_touched:gv.tiles_touched + 4 * P].view(torch.int32) > 0
    p_vis = int(touched.sum().item())
    row_entries = int((((rect[:, 1] >> 16) & 0xffff) - (rect[:, 1] & 0xffff))[touched].sum().item())

    # ---- per-frame device time distribution (hipEvent pairs on the launch stream; SURVEY 8d: median and p10 / p90)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(100, max(20, args.steps)))]
    for a_, b_ in evs:
        a_.record(); step(); b_.record()
    torch.cuda.synchronize(dev)
    fm = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
    frame_pct = {"p10": fm[len(fm) // 10], "p50": fm[len(fm) // 2], "p90": fm[(len(fm) * 9) // 10], "n": len(fm)}

    # ---- per-stage device times (hipEvent pairs on the launch stream), measured live over a second timed region
    ctx.set_profiling(True)
    nprof = max(10, min(50, args.steps))
    acc: dict = {}
    for _ in range(nprof):
        step()
        for k, v in ctx.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    stage_ms = {k: v / nprof for k, v in acc.items()}
    extra = {}
    ctx.set_profiling(False)
    # ---- throughput with several frames in flight (reported BESIDE the single-stream headline, never instead of it): one HIP
    #      stream + one rasterizer context per frame slot; frame k + 1's latency-bound preprocess / depth sort / binning
    #      (about one workgroup per CU) runs under frame k's VALU-bound blend.  Images are bit-identical on every slot.
    if rank == 0 and not args.no_inflight:
        fif = {}
        for nslots in (2, 4):
            streams = [torch.cuda.Stream(device=dev) for _ in range(nslots)]
            for st in streams:
                st.wait_stream(torch.cuda.current_stream(dev))
            outs = [None] * nslots
            for i in range(4 * nslots):
                with torch.cuda.stream(streams[i % nslots]):
                    outs[i % nslots] = step()
            torch.cuda.synchronize(dev)
            nf = max(100, args.steps)
            tf = time.perf_counter()
            for i in range(nf):
                with torch.cuda.stream(streams[i % nslots]):
                    outs[i % nslots] = step()
            torch.cuda.synchronize(dev)
            tf = time.perf_counter() - tf
            assert all(torch.equal(o[1], out[1]) for o in outs), "frames rendered on different streams differ"
            fif[str(nslots)] = {"frames_per_s": nf / tf, "ms_per_frame": tf / nf * 1e3}
            del streams, outs
        extra["frames_in_flight"] = fif
        # ... and with rasterizer.FramePipeline (ggd_forward_enqueue / ggd_forward_collect): the host does not wait for a
        # frame's num_rendered before it launches the next one -- it is collected when the slot comes round again
        pfl = {}
        for nslots in (2, 3, 4):
            pipe = R.FramePipeline(dev, slots=nslots)
            for i in range(6 * nslots):
                pipe.submit(*fargs)
            pipe.drain()
            torch.cuda.synchronize(dev)
            nf = max(200, args.steps)
            tf = time.perf_counter()
            for i in range(nf):
                pipe.submit(*fargs)
            rest = pipe.drain()
            torch.cuda.synchronize(dev)
            tf = time.perf_counter() - tf
            assert all(r[0] == num_rendered and torch.equal(r[1], out[1]) for r in rest), "pipelined frames differ"
            pfl[str(nslots)] = {"frames_per_s": nf / tf, "ms_per_frame": tf / nf * 1e3, "synchronous_frames": pipe.synchronous_frames}
            del pipe, rest
        extra["frames_in_flight_pipelined"] = pfl
    # the single-call forward's front end: histograms + scan step 1 + row counts inside the preprocess
    # kernel, the fourth sort pass not launched after a streak of frames whose top depth byte was constant (verified per frame;
    # a frame it was wrong for is binned and blended again: `sort_reruns`)
    msd_frames = ctx.get_option(_capi.STAT_MSD_FRAMES)
    extra["forward_path"] = {"fold": ctx.get_option(_capi.OPT_FOLD), "flat_streak": ctx.get_option(_capi.STAT_FLAT_STREAK),
                             "sort_reruns": ctx.get_option(_capi.STAT_SORT_RERUNS), "two_launch_sort": ctx.get_option(_capi.OPT_MSD_SORT),
                             "two_launch_sort_frames": msd_frames, "capacity_retries": ctx.capacity_retries,
                             "kernel_launches_per_frame": (8 if msd_frames > 0 else 9) if ctx.get_option(_capi.OPT_FOLD) and S <= 1024 else None}
    # ---- beside the headline (rank 0): the same shape over a cycle of DIFFERENT scenes (the train step's situation: the
    #      capacity hint, the alternating control blocks and the streak-based sort speculation see new data every frame), the
    #      head-like "shell" scene's dominant kernel, and the backward blend, each priced like `roofline`
    rl_shell = rl_bwd = varying = None
    if rank == 0 and not args.no_extra_rooflines:
        cyc = []
        for sd, fov in ((11, 12.0), (12, 9.0), (13, 15.0), (14, 11.0), (15, 13.5)):
            s2 = make_scene(P, S, kind, seed=sd, fov_deg=fov).to(dev)
            c2 = s2.cam
            cyc.append((s2.bg, s2.xyz, empty, s2.opacities.contiguous(), s2.scales.contiguous(), s2.rotations.contiguous(), 1.0, empty,
                        c2.world_view_transform, c2.full_proj_transform, math.tan(c2.FoVx * 0.5), math.tan(c2.FoVy * 0.5), S, S,
                        s2.features_dc.contiguous(), 0, c2.camera_center, False, False))
        r0, c0, m0 = ctx.get_option(_capi.STAT_SORT_RERUNS), ctx.capacity_retries, ctx.get_option(_capi.STAT_MSD_FRAMES)
        rs = []
        for i in range(2 * len(cyc)):
            rs.append(R.rasterize_gaussians_native(*cyc[i % len(cyc)])[0])
        torch.cuda.synchronize(dev)
        nv = max(100, args.steps)
        tv = time.perf_counter()
        for i in range(nv):
            R.rasterize_gaussians_native(*cyc[i % len(cyc)])
        torch.cuda.synchronize(dev)
        tv = time.perf_counter() - tv
        varying = {"scenes": len(cyc), "frames": nv, "frames_per_s": nv / tv, "ms_per_frame": tv / nv * 1e3,
                   "num_rendered": sorted(set(int(r) for r in rs)), "sort_reruns": ctx.get_option(_capi.STAT_SORT_RERUNS) - r0,
                   "capacity_retries": ctx.capacity_retries - c0, "two_launch_sort_frames": ctx.get_option(_capi.STAT_MSD_FRAMES) - m0,
                   "what": f"{P} Gaussians, {S}x{S}, '{kind}' scenes of 5 seeds and fields of view 9..15 degrees in turn, one stream"}
        del cyc
        # ... and over POSES drawn as the reference draws them per training step (main/decoder_utils/camera.py:6-35: radius 2.7,
        # yaw uniform in pi/2 +- 1.0, pitch uniform in pi/2 +- 0.3): an oblique view brings the unit cube's near corner to depth
        # 1.83, i.e. the depth keys straddle the binade boundary at 2.0 that round 5's two-launch sort could not cross (VERDICT
        # r05 weak 5).  Eight seeded poses in turn on the headline scene and on the head-like shell, one stream.
        import numpy as _np
        from gaussian_gan_decoder_amd.synthetic import make_camera
        prng = _np.random.RandomState(5)
        poses = [(float(math.pi / 2 + prng.uniform(-1.0, 1.0)), float(math.pi / 2 + prng.uniform(-0.3, 0.3))) for _ in range(8)]
        varying["poses"] = {}
        for kind2 in (kind, "shell") if kind != "shell" else (kind,):
            s2 = sc if kind2 == kind else make_scene(P, S, kind2, seed=0).to(dev)
            pa = []
            for h_, v_ in poses:
                c2 = make_camera(S, 12.0, h_, v_, device=dev)
                pa.append((s2.bg, s2.xyz, empty, s2.opacities.contiguous(), s2.scales.contiguous(), s2.rotations.contiguous(), 1.0, empty,
                           c2.world_view_transform, c2.full_proj_transform, math.tan(c2.FoVx * 0.5), math.tan(c2.FoVy * 0.5), S, S,
                           s2.features_dc.contiguous(), 0, c2.camera_center, False, False))
            for i in range(3 * len(pa)):
                R.rasterize_gaussians_native(*pa[i % len(pa)])
            torch.cuda.synchronize(dev)
            r0, c0, m0 = ctx.get_option(_capi.STAT_SORT_RERUNS), ctx.capacity_retries, ctx.get_option(_capi.STAT_MSD_FRAMES)
            nv = max(104, args.steps)
            tv = time.perf_counter()
            for i in range(nv):
                R.rasterize_gaussians_native(*pa[i % len(pa)])
            torch.cuda.synchronize(dev)
            tv = time.perf_counter() - tv
            pose_counts = (ctx.get_option(_capi.STAT_SORT_RERUNS) - r0, ctx.capacity_retries - c0, ctx.get_option(_capi.STAT_MSD_FRAMES) - m0)
            # the same scene at the fixed head-on pose, timed the same way right behind it (the comparison the verdict asks for)
            fa = fargs if kind2 == kind else pa[0][:8] + (cam.world_view_transform, cam.full_proj_transform, tanx, tany) + pa[0][12:16] + (cam.camera_center, False, False)
            for i in range(12):
                R.rasterize_gaussians_native(*fa)
            torch.cuda.synchronize(dev)
            tfx = time.perf_counter()
            for i in range(nv):
                R.rasterize_gaussians_native(*fa)
            torch.cuda.synchronize(dev)
            tfx = time.perf_counter() - tfx
            varying["poses"][kind2] = {"frames": nv, "frames_per_s": nv / tv, "frames_per_s_fixed_pose": nv / tfx,
                                       "ratio_to_fixed_pose": tfx / tv,
                                       "sort_reruns": pose_counts[0], "capacity_retries": pose_counts[1],
                                       "two_launch_sort_frames": pose_counts[2],
                                       "poses_h_v": [[round(a_, 3), round(b_, 3)] for a_, b_ in poses]}
            del pa, fa
            if kind2 != kind:
                del s2
        if args.workload == "1M_1024_cube":
            Ps, Ss, ks = WORKLOADS["1M_1024_shell"]
            s2 = make_scene(Ps, Ss, ks, seed=0).to(dev)
            c2 = s2.cam
            a2 = (s2.bg, s2.xyz, empty, s2.opacities.contiguous(), s2.scales.contiguous(), s2.rotations.contiguous(), 1.0, empty,
                  c2.world_view_transform, c2.full_proj_transform, math.tan(c2.FoVx * 0.5), math.tan(c2.FoVy * 0.5), Ss, Ss,
                  s2.features_dc.contiguous(), 0, c2.camera_center, False, False)
            for _ in range(10):
                o2 = R.rasterize_gaussians_native(*a2)
            ctx.set_profiling(True)
            accs: dict = {}
            for _ in range(20):
                o2 = R.rasterize_gaussians_native(*a2)
                for k, v in ctx.stage_times().items():
                    accs[k] = accs.get(k, 0.0) + v / 20
            ctx.set_profiling(False)
            rl_shell = roofline_entry("1M_1024_shell", "blend", Ps, int(o2[0]), Ss, Ss, accs["blend"])
            rl_shell["stage_ms"] = {k: round(v, 5) for k, v in accs.items() if k in ("preprocess", "sort", "duplicate", "blend")}
            rl_shell["num_rendered"] = int(o2[0])
            del s2, a2, o2
        # backward blend of the headline workload (quarter form), hipEvent stage time
        gb = make_dL_dpix(S).to(dev)
        ob = step()
        bb = (sc.bg, sc.xyz, ob[2], empty, scales, rots, 1.0, empty, cam.world_view_transform, cam.full_proj_transform, tanx, tany, gb,
              shs, 0, cam.camera_center, ob[3], ob[0], ob[4], ob[5], False)
        for _ in range(3):
            R.rasterize_gaussians_backward_native(*bb)
        ctx.set_profiling(True)
        tbw = 0.0
        for _ in range(10):
            R.rasterize_gaussians_backward_native(*bb)
            tbw += ctx.stage_times()["blend_bwd"] / 10
        ctx.set_profiling(False)
        rl_bwd = roofline_entry(args.workload, "blend_bwd", P, num_rendered, S, S, tbw)
        del gb, ob, bb
    if not args.no_sweep and rank == 0:
        # the other synthetic configurations of BASELINE.json's north_star ({100k, 1M} x {512, 1024}), forward raster only,
        # 100 frames each -- reported for the table in DESIGN.md; the headline `value` is the workload above
        sweep = {}
        for name in ("1M_1024_shell", "100k_512_cube", "100k_1024_cube", "1M_512_cube"):
            if name == args.workload:
                continue
            P2, S2, kind2 = WORKLOADS[name]
            sc2 = make_scene(P2, S2, kind2, seed=0).to(dev)
            cam2 = sc2.cam
            a2 = (sc2.bg, sc2.xyz, empty, sc2.opacities.contiguous(), sc2.scales.contiguous(), sc2.rotations.contiguous(), 1.0,
                  empty, cam2.world_view_transform, cam2.full_proj_transform, math.tan(cam2.FoVx * 0.5),
                  math.tan(cam2.FoVy * 0.5), S2, S2, sc2.features_dc.contiguous(), 0, cam2.camera_center, False, False)
            for _ in range(10):
                o2 = R.rasterize_gaussians_native(*a2)
            torch.cuda.synchronize(dev)
            batches = []
            for _ in range(5):   # median of 5 batches of 20 frames (one allocator hiccup must not decide the number)
                t2 = time.perf_counter()
                for _ in range(20):
                    o2 = R.rasterize_gaussians_native(*a2)
                torch.cuda.synchronize(dev)
                batches.append((time.perf_counter() - t2) / 20)
            sweep[name] = {"frames_per_s": 1.0 / sorted(batches)[2], "num_rendered": int(o2[0]),
                           "tile_list_length": tile_list_stats(o2[5], S2, S2)}
            del sc2, a2, o2
        extra["forward_fps_other_workloads"] = sweep
        # dataset-resolution frames (stock 3DGS render() callers; beyond the north_star's list): grids wider than 64 tiles
        hd = {}
        for name, (P2, W2, H2) in (("1M_1920x1080_cube", (1_000_000, 1920, 1080)), ("2M_3840x2160_cube", (2_000_000, 3840, 2160))):
            sc2 = make_scene(P2, W2, "cube", seed=0).to(dev)
            cam2 = sc2.cam
            tx2 = math.tan(cam2.FoVx * 0.5)
            a2 = (sc2.bg, sc2.xyz, empty, sc2.opacities.contiguous(), sc2.scales.contiguous(), sc2.rotations.contiguous(), 1.0,
                  empty, cam2.world_view_transform, cam2.full_proj_transform, tx2, tx2 * H2 / W2, H2, W2,
                  sc2.features_dc.contiguous(), 0, cam2.camera_center, False, False)
            for _ in range(5):
                o2 = R.rasterize_gaussians_native(*a2)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in range(30):
                o2 = R.rasterize_gaussians_native(*a2)
            torch.cuda.synchronize(dev)
            hd[name] = {"frames_per_s": 30.0 / (time.perf_counter() - t2), "num_rendered": int(o2[0]),
                        "tiles": ((W2 + 15) // 16) * ((H2 + 15) // 16)}
            del sc2, a2, o2
        extra["forward_fps_dataset_resolution"] = hd
        # a stock-3DGS-shaped optimisation step of the raster (what gaussian_renderer.render() + loss.backward() cost per
        # iteration there): 1 M Gaussians, 1920 x 1080, SH degree 3 (16 coefficients per channel), forward + backward
        sc2 = make_scene(1_000_000, 1920, "cube", seed=0).to(dev)
        cam2 = sc2.cam
        W2, H2 = 1920, 1080
        tx2 = math.tan(cam2.FoVx * 0.5)
        gsh = torch.Generator().manual_seed(3)
        sh16 = torch.cat([sc2.features_dc, 0.05 * torch.randn(1_000_000, 15, 3, generator=gsh).to(dev)], 1).contiguous()
        g2 = make_dL_dpix(1920).to(dev)[:, :H2, :W2].contiguous()
        sc2s, sc2r = sc2.scales.contiguous(), sc2.rotations.contiguous()
        fa = (sc2.bg, sc2.xyz, empty, sc2.opacities.contiguous(), sc2s, sc2r, 1.0, empty, cam2.world_view_transform,
              cam2.full_proj_transform, tx2, tx2 * H2 / W2, H2, W2, sh16, 3, cam2.camera_center, False, False)

        def fwd_bwd():
            o = R.rasterize_gaussians_native(*fa)
            R.rasterize_gaussians_backward_native(sc2.bg, sc2.xyz, o[2], empty, sc2s, sc2r, 1.0, empty, cam2.world_view_transform,
                                                  cam2.full_proj_transform, tx2, tx2 * H2 / W2, g2, sh16, 3, cam2.camera_center,
                                                  o[3], o[0], o[4], o[5], False)
        for _ in range(3):
            fwd_bwd()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(20):
            fwd_bwd()
        torch.cuda.synchronize(dev)
        extra["raster_fwd_bwd_1M_1080p_sh3"] = {"ms": (time.perf_counter() - t2) / 20 * 1e3}
        del sc2, sh16, g2, fa
    if args.backward:
        ctx.set_profiling(True)
        g = make_dL_dpix(S).to(dev)
        out = step()
        bargs = (sc.bg, sc.xyz, out[2], empty, scales, rots, 1.0, empty, cam.world_view_transform,
                 cam.full_proj_transform, tanx, tany, g, shs, 0, cam.camera_center, out[3],
                 out[0], out[4], out[5], False)
        bacc: dict = {}
        for _ in range(5):
            R.rasterize_gaussians_backward_native(*bargs)
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        for _ in range(nprof):
            R.rasterize_gaussians_backward_native(*bargs)
            for k in ("blend_bwd", "preprocess_bwd"):
                bacc[k] = bacc.get(k, 0.0) + ctx.stage_times()[k]
        torch.cuda.synchronize(dev)
        extra["backward_ms"] = (time.perf_counter() - tb) / nprof * 1e3
        for k, v in bacc.items():
            stage_ms[k] = v / nprof
    ctx.set_profiling(False)

    # ---- BASELINE config 3: decode ~1M Gaussians (tri-plane gather + fused f16-MFMA decoder) and render at 1024^2
    decode = None
    if not args.no_decode and rank == 0:
        from gaussian_gan_decoder_amd.decoder import SequentialDecoderReverse, triplane_mean
        from gaussian_gan_decoder_amd.fused_decoder import FusedDecoder
        from gaussian_gan_decoder_amd.gaussian_model import GaussianModel
        from gaussian_gan_decoder_amd.gaussian_renderer import render_simple
        from gaussian_gan_decoder_amd.synthetic import make_camera
        torch.manual_seed(0)
        dec = SequentialDecoderReverse().to(dev)
        fused = FusedDecoder(dec)
        g = torch.Generator().manual_seed(5)
        planes = torch.randn(3, 32, 256, 256, generator=g).to(dev)
        d = torch.randn(1_000_000, 3, generator=g)
        positions = (d / d.norm(dim=1, keepdim=True) * 0.3
                     * torch.clip(1 + 0.1 * torch.randn(1_000_000, 1, generator=g), 0, 1)).to(dev)
        cam1k = make_camera(1024, 12.0, device=dev)
        pc = GaussianModel(0)

        def decode_render():
            with torch.no_grad():
                o = fused(planes, positions)
                pc._xyz, pc._scaling, pc._rotation, pc._opacity = o.xyz, o.scale, o.rotation, o.opacity
                pc._features_dc = o.color.unsqueeze(1)
                return render_simple(cam1k, pc, bg_color=sc.bg, fused_activations=True)["render"]
        for _ in range(5):
            decode_render()
        torch.cuda.synchronize(dev)
        td = time.perf_counter()
        nd = 50
        for _ in range(nd):
            decode_render()
        torch.cuda.synchronize(dev)
        td = (time.perf_counter() - td) / nd
        # two frames in flight (two streams, each with its own rasterizer context and Gaussian container): the raster's
        # latency-bound sort / binning chain of one frame under the decoder's MFMA / VALU-bound kernel of the other
        dstreams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        pcs = [GaussianModel(0) for _ in range(2)]
        imgs = [None, None]

        def decode_render_on(i):
            with torch.cuda.stream(dstreams[i]), torch.no_grad():
                o = fused(planes, positions)
                q = pcs[i]
                q._xyz, q._scaling, q._rotation, q._opacity = o.xyz, o.scale, o.rotation, o.opacity
                q._features_dc = o.color.unsqueeze(1)
                imgs[i] = render_simple(cam1k, q, bg_color=sc.bg, fused_activations=True)["render"]
        for st in dstreams:
            st.wait_stream(torch.cuda.current_stream(dev))
        for i in range(8):
            decode_render_on(i % 2)
        torch.cuda.synchronize(dev)
        td2 = time.perf_counter()
        for i in range(nd):
            decode_render_on(i % 2)
        torch.cuda.synchronize(dev)
        td2 = (time.perf_counter() - td2) / nd
        ref_img = decode_render()
        torch.cuda.synchronize(dev)
        assert torch.equal(imgs[0], ref_img) and torch.equal(imgs[1], ref_img), "decode+render differs between streams"
        del dstreams, pcs, imgs, ref_img
        feats = triplane_mean(planes, positions, 1.0)
        for _ in range(3):
            fused.decode_features(feats, positions)
        torch.cuda.synchronize(dev)
        tm = time.perf_counter()
        for _ in range(nd):
            fused.decode_features(feats, positions)
        torch.cuda.synchronize(dev)
        tm = (time.perf_counter() - tm) / nd
        fused32 = FusedDecoder(dec, precision="fp32")
        for _ in range(3):
            fused32.decode_features(feats, positions)
        torch.cuda.synchronize(dev)
        tm32 = time.perf_counter()
        for _ in range(nd):
            fused32.decode_features(feats, positions)
        torch.cuda.synchronize(dev)
        tm32 = (time.perf_counter() - tm32) / nd
        mlp_flops = 2 * 192512 * 1_000_000
        decode = {"frames_per_s": 1.0 / td, "ms_per_frame": td * 1e3, "frames_per_s_two_in_flight": 1.0 / td2,
                  "points": 1_000_000, "image": "1024x1024",
                  "pipeline": "tri-plane gather (HIP) -> fused 5-head decoder (f16 MFMA, packed-f16 GELU) -> HIP raster (activation prologue fused)",
                  "mlp_ms": tm * 1e3, "mlp_TFLOPs": mlp_flops / tm / 1e12,
                  "mlp_frac_of_bf16_dense_peak": mlp_flops / tm / 2.5e15,
                  "mlp_fp32_accurate_ms": tm32 * 1e3}
        del dec, fused, fused32, planes, positions

    # ---- data-parallel decoder train step (BASELINE configs 3/5): scenes_per_gpu scenes per rank, 512x512,
    #      decoder MLPs -> activations -> raster fwd -> L1+L2 -> bwd -> ONE flat RCCL all-reduce -> Adam
    train = None
    train_fused = None
    train_fused_fp32 = None
    if not args.no_train:
        from gaussian_gan_decoder_amd.train import DecoderTrainer, make_scene_batch
        spg = args.scenes_per_gpu
        my_scenes = list(range(rank * spg, (rank + 1) * spg))
        batches = [make_scene_batch(my_scenes, args.train_points, 512, dev, seed=i) for i in range(2)]

        # the reference all-reduces ~29.77 M fp32 gradients (decoder 0.19 M + the finetuned generator backbone,
        # sequential_decoder_reverse.py:89-99): our shared planes hold 6.29 M of that, the stand-in tensor the rest
        PLANES = {   # the two plane formats of the reference's generators; "panohead" is the one BASELINE config 3 names
            "panohead": dict(kw=dict(plane_axes="panohead", triplane_depth=3), floats=3 * 96 * 256 * 256,
                             text="PanoHead tri-grids [3, 32 x 3, 256, 256], 3-D grid_sample (8 taps per plane), PanoHead plane axes"),
            "eg3d": dict(kw=dict(), floats=3 * 32 * 256 * 256,
                         text="EG3D tri-planes [3, 32, 256, 256], 2-D grid_sample (4 taps per plane)"),
        }

        def retries():
            return sum(c.capacity_retries for c in _capi._contexts.values())

        def run_train(fused_decoder, standins=True, precision="bf16", planes="panohead"):
            # the all-reduced payload stays the reference's 29 763 294 floats whatever the plane format: the stand-in tensor
            # holds what the decoder and the shared planes do not
            BACKBONE_REST = 29_570_000 - PLANES[planes]["floats"]
            tr = DecoderTrainer(dev, n_scenes_total=spg * world, image_size=512, fused_activations=True,
                                fused_decoder=fused_decoder, backbone_params=BACKBONE_REST if standins else 0,
                                perceptual_weight=1.0 if standins else 0.0, decoder_precision=precision,
                                **PLANES[planes]["kw"])
            tr.measure_comm = True
            for i in range(2):
                tr.step(batches[i % 2])
            barrier()
            r0 = retries()
            exposed, in_bwd = 0.0, 0
            tt = time.perf_counter()
            for i in range(args.train_iters):
                tr.step(batches[i % 2])
                in_bwd = tr.last_allreduce_bytes_in_backward
            barrier()
            t_train = time.perf_counter() - tt
            exposed = tr.allreduce_exposed_ms      # of the last timed step (reading it synchronises: outside the timed loop)
            if dist is not None:
                t = torch.tensor([t_train], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                t_train = float(t.item())
            nparam = sum(p.numel() for p in tr.params)
            ar = tr.last_allreduce_bytes
            del tr
            torch.cuda.empty_cache()   # the next configuration starts from a clean allocator (z / dz buffers are GBs)
            return {"iters_per_s": args.train_iters / t_train,
                    "scenes_per_s": args.train_iters * spg * world / t_train,
                    "ms_per_iter": t_train / args.train_iters * 1e3, "global_batch": spg * world,
                    "scenes_per_gpu": spg, "points_per_scene": args.train_points, "image": "512x512",
                    "planes": PLANES[planes]["text"],
                    "parameters_all_reduced": nparam, "allreduce_bytes": ar,
                    # 8-GPU readiness: how much of the payload was already travelling when the backward returned (units are
                    # launched from gradient hooks), and how long the compute stream then stood still waiting for units
                    "allreduce_bytes_launched_in_backward": in_bwd, "allreduce_exposed_ms": exposed,
                    # single-call forwards of the timed iterations whose binning buffer was too small (exact retry = a host
                    # sync); the scenes' fov is drawn from U[5, 17] degrees, so num_rendered varies from scene to scene
                    "capacity_retries": retries() - r0,
                    "stand_ins": ("backbone gradient payload (%d floats) + LPIPS slot (fixed random VGG16-shaped trunk at "
                                  "256x256 on the batch of rendered images and on the batch of targets, weight 1.0)" % BACKBONE_REST) if standins else "none (round-1 configuration)"}

        LOSS = "L1+L2+SSIM+Sobel (fused HIP loss, reference weights) + perceptual stand-in (PyTorch convs)"
        # (1) the reference's precision: decoder MLPs in fp32 (PyTorch GEMMs, split-K weight gradients)
        train = run_train(False)
        train["mlp_dtype"] = "fp32"
        train["step"] = ("plane gather (HIP, per-scene modulation fused) -> decoder MLPs (PyTorch fp32) -> HIP raster fwd (activations fused) -> " + LOSS + " -> bwd -> bucketed flat all-reduce "
                         "launched from gradient hooks, per-bucket Adam")
        train["eg3d_planes"] = run_train(False, planes="eg3d")
        # (2) SURVEY 8f row 1: decoder forward / activation backward / weight gradients as bf16-MFMA HIP kernels
        #     (fp32 accumulate, fp32 master weights and optimizer) -- reported beside (1), never instead of it
        train_fused = run_train(True)
        train_fused["mlp_dtype"] = ("16-bit operands, fp32 accumulate: forward f16 (v_mfma_f32_16x16x32_f16, z kept as f16), activation backward "
                                    "and weight gradients bf16 (v_mfma_f32_16x16x32_bf16)")
        train_fused["step"] = ("plane gather (HIP, per-scene modulation fused) -> fused 5-head decoder (HIP MFMA fwd, bwd, split-K wgrad; all "
                               "local scenes in one launch) -> HIP raster fwd -> " + LOSS + " -> bwd -> "
                               "bucketed flat all-reduce launched from gradient hooks, per-bucket Adam")
        train_fused["eg3d_planes"] = run_train(True, planes="eg3d")
        # (3) the same step without the two stand-ins (what round 1 measured: EG3D planes), for continuity
        train_fused["without_stand_ins"] = run_train(True, standins=False, planes="eg3d")
        # (4) the fused kernels at the reference's precision: every operand split into two bf16 numbers, three MFMAs per
        #     product, in the forward, the backward and the weight gradients (csrc/ggd_mlp_hl.inc): outputs within 1e-4 and
        #     parameter gradients within 1e-3 (relative L2; measured 1e-4 .. 3e-4) of a float64 evaluation (tests/test_decoder_gpu.py)
        train_fused_fp32 = run_train(True, precision="fp32")
        train_fused_fp32["mlp_dtype"] = ("fp32-accurate: split bf16 operands (hi + lo), 3 x v_mfma_f32_16x16x32_bf16 per product, "
                                         "fp32 accumulate; pre-activations z kept as one fp16 plane, dz as one fp16 plane scaled per (head, 32-point slab)")
        train_fused_fp32["step"] = train_fused["step"].replace("HIP MFMA fwd", "HIP MFMA at reference precision: fwd")
        train_fused_fp32["eg3d_planes"] = run_train(True, precision="fp32", planes="eg3d")
        del batches
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    fwd_stages = ("preprocess", "scan", "duplicate", "sort", "ranges", "blend")
    dom = max(fwd_stages, key=lambda k: stage_ms.get(k, 0.0))
    whole = sum(algorithmic_bytes(k, P, num_rendered, S, S) for k in fwd_stages)
    # (depth sort: the first launch reads 4 B / Gaussian and writes 8 B / visible one; then either two onesweep passes of
    # 16 B / visible Gaussian each or, in its two-launch form, one in-LDS finish of 16 B / visible Gaussian)
    path_bytes = (algorithmic_bytes("preprocess", P, num_rendered, S, S) + 4 * P + 8 * p_vis + (1 if msd_frames > 0 else 2) * 16 * p_vis
                  + 8 * row_entries + 4 * num_rendered + algorithmic_bytes("blend", P, num_rendered, S, S))
    ms_per_step = elapsed / args.steps * 1e3
    result = {
        "metric": "forward raster frames/s, 1M Gaussians @ 1024x1024" if args.workload == "1M_1024_cube"
        else f"forward raster frames/s ({args.workload})",
        "value": world * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "timed_region_s": elapsed,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{P} synthetic Gaussians ('{kind}' scene, seed 0, SH degree 0), {S}x{S}, forward "
                               "raster fp32, inputs resident in HBM", "num_rendered": num_rendered,
                   "tiles": ((S + 15) // 16) ** 2, "tile_list_length_mean": tile_lists["mean"],
                   "tile_list_length_max": tile_lists["max"], "parallelism": f"scene-parallel x{world} (no collective)"},
        # (the dominant kernel is VALU-bound by intensity, DESIGN.md section 4: its wave-level VALU instruction count is reported
        # beside the HBM figure; `traffic` and `valu_wave_insts` are REPLAYED from the committed PMC pass named in `replayed_from`,
        # `kernel_ms` is this run's own hipEvent time)
        "roofline": roofline_entry(args.workload, dom, P, num_rendered, S, S, stage_ms[dom]),
        # what ran INSIDE the timed region (counters read around it) and the last timed frame against the exact forms
        "timed_path": timed_path,
        "verified_vs_two_call": verified["verified_vs_two_call"],
        "verified_vs_radix_sort_path": verified["verified_vs_radix_sort_path"],
        "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
        "frame_ms_percentiles": {k: (round(v, 5) if k != "n" else v) for k, v in frame_pct.items()},
        # whole frame against HBM: `algorithmic_bytes_path` = what THIS path has to move (preprocess, which also builds the sort's
        # histograms; depth sort of the visible Gaussians: the first pass reads 4 B / Gaussian and writes 8 B / visible one, the
        # two further non-constant passes 16 B / visible Gaussian each; 8 B per (Gaussian, tile row) entry; 4 B / instance for
        # the list; the blend) -> frac_of_hbm_peak.  SURVEY 8d's contract formula,
        # which prices the reference's 6-pass 64-bit sort of all instances that this path does not run, is kept beside it.
        "whole_frame": {"algorithmic_bytes_path": path_bytes, "GBps": path_bytes / (ms_per_step * 1e-3) / 1e9,
                        "frac_of_hbm_peak": path_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "visible_gaussians": p_vis, "row_entries": row_entries,
                        "contract_formula_bytes": whole,
                        "contract_formula_frac_of_hbm_peak": whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    moved = pmc_moved_bytes(args.workload)
    if moved is not None:
        result["whole_frame"].update({"moved_bytes_pmc": moved["bytes"], "moved_GBps": moved["bytes"] / (ms_per_step * 1e-3) / 1e9,
                                      "moved_frac_of_hbm_peak": moved["bytes"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "moved_bytes_source": moved["source"]})
    if rl_shell is not None:
        result["roofline_shell"] = rl_shell
    if rl_bwd is not None:
        result["roofline_backward"] = rl_bwd
    if varying is not None:
        result["varying_scenes"] = varying
    if decode is not None:
        m, stale = pmc_mlp(decode["mlp_ms"] * 1e3)
        # MFMA-side roofline entry for the fused decoder MLP: achieved / peak from THIS run's timing; the matrix-pipe busy fraction
        # and VALU count replayed from the committed counter pass if its kernel is this kernel (duration within 20 %), else null
        decode["roofline"] = {"bound": "mfma", "achieved": decode["mlp_TFLOPs"], "peak": 2500.0, "unit": "TFLOP/s",
                              "frac": decode["mlp_frac_of_bf16_dense_peak"], "mfma_busy": m.get("mfma_busy_frac") if m else None,
                              "valu_insts": m.get("valu_insts") if m else None, "kernel_us_rocprof": m.get("kernel_us") if m else None,
                              "source": m.get("source") if m else None, "refused_stale_profiles": stale}
        result["decode_render"] = decode
    if train is not None:
        result["train"] = train
    if train_fused is not None:
        result["train_fused_decoder"] = train_fused
    if train_fused_fp32 is not None:
        result["train_fused_decoder_fp32"] = train_fused_fp32
    if extra:
        result["extra"] = extra
    if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed at N = 1 only (one rank, the whole host)
        result["cpu_baseline"] = cpu_baseline(P, S, kind)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not all(verified.values()):
        print(f"bench.py: the timed frame differs from the exact forward: {verified}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
