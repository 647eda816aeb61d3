#!/usr/bin/env python3
"""
bench.py -- forward+backward Mpixels/s of the rasterizer on BASELINE.json's headline workload:
synthetic 800x800, 1M Gaussians, SH degree 3 (config "C3").

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 runs one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N` (the driver's form) this process
is one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Started bare (`python bench.py --gpus N`, no
WORLD_SIZE) it launches its own N ranks as child processes before touching the GPU (3dgs-native_amd/launch.py), relays
rank 0's JSON line and exits with the worst child's code.

A step = one render_gaussians() + one backward() of one camera view per GPU, all inputs already
resident in HBM (device torch tensors), dL/dpixels fixed.  With N GPUs every rank holds the full
(replicated) scene and renders its own view, and the step ends with every rank holding the gradient averaged over the N
views: 11 of the 59 floats per Gaussian are all-reduced, the 48-float SH gradient -- an outer
product of the view's SH basis and a 3-float colour gradient -- is exchanged as those 3 floats (all-gather) and rebuilt
locally (3dgs-native_amd/dist.py; `--dense-exchange` all-reduces all 59 instead).  `value` = N * W * H / step time.

The JSON line also carries
  roofline     -- the dominant kernel (longest average stage) measured with HIP events recorded by the
                  library on the launch stream in a second, untimed loop of 20 steps right after the timed
                  region (so `value` pays for no stage event), priced against the 8 TB/s HBM
                  peak with the algorithmic bytes of DESIGN.md; `pipeline` = the same for the whole
                  forward+backward byte budget B = 340 N + 596 Nv + 128 D + 16 Tn + 44 P;
  cpu_baseline -- the CPU oracle (a single-thread C restatement of the reference kernels; Warp's CPU
                  backend is also one serial loop) on the same scene and view, rank 0, N=1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

# Before torch (and with it the HIP runtime) is imported: under `python -m torch.distributed.run ... bench.py` nobody else sets it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS, MAX_CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMD-32, same guide
# Cycles ONE SIMD needs per wave64 vector instruction: a SIMD-32 issues a plain VALU op over 2 cycles; DPP (and packed-f32) ops
# go at half that rate, v_exp / v_rcp / v_log / v_sqrt at a quarter.  tools/valu_rate measures 2.3 / 4.2-4.5 / 8.2 chip-wide with
# 8 waves per SIMD (profiles/r02_valu_issue_costs.txt); the roof below is priced at the architectural 2 / 4 / 8.
VALU_ISSUE_CYCLES = {"plain": 2.0, "dpp": 4.0, "transcendental": 8.0}
# Round 3 (tools/issue_costs, profiles/r03_issue_costs.txt): compares, selects, min/max, VOP3 integer ops and any op with an SGPR
# operand issue at HALF rate too (4.2-4.3 cycles), plain ops at 2.3.  The SQ counters do not split those classes, so the second
# figure below prices every vector instruction at the average of its kernel's hot loop, counted from the ISA: the forward's walk
# is 27 instructions = 19 plain + 7 compare / select / min + 1 exp = 82 cycles; the backward's pixel step 63 = 35 plain + 7
# compare / select / min + 12 DPP + 2 transcendental (+ 7 others at half rate) = 178 cycles.
AVG_ISSUE_CYCLES_MEASURED = {"blend_fwd": 82.0 / 27.0, "blend_bwd": 178.0 / 63.0}


def roofline_valu(stage, counters, avg_ms):
    """Vector-issue roof of a blend kernel: wave-instructions (rocprofv3 SQ counters of THIS build, profiles/sq_counters.json) x
    issue cycles, over the SIMD cycles the launch had (1024 SIMDs x the stage's live HIP-event time x 2.4 GHz).  DPP ops have
    no counter of their own: the backward blend executes exactly 12 (two 6-step wave scans) and 2 transcendentals (exp, rcp)
    per (bucket, pixel) step and none elsewhere on the mask-driven path bench.py runs, so DPP = 6 x transcendentals there; the
    forward has none.  Both kernels are built without v_pk_*_f32."""
    insts, trans = counters["SQ_INSTS_VALU"], counters.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    dpp = 6.0 * trans if stage == "blend_bwd" else 0.0
    plain = insts - trans - dpp
    c = VALU_ISSUE_CYCLES
    need = plain * c["plain"] + dpp * c["dpp"] + trans * c["transcendental"]
    have = SIMDS * avg_ms * 1e-3 * MAX_CLOCK_HZ
    return {"bound": "valu_issue", "kernel": counters.get("kernel"), "wave_instructions": int(insts), "transcendental": int(trans), "dpp": int(dpp),
            "scalar_instructions": int(counters.get("SQ_INSTS_SALU", 0)), "issue_cycles": int(need), "simd_cycles": int(have),
            "frac": round(need / have, 4), "frac_if_every_op_took_2_cycles": round(insts * 2.0 / have, 4),
            "frac_at_measured_issue_prices": round(insts * AVG_ISSUE_CYCLES_MEASURED.get(stage, 2.3) / have, 4), "avg_ms": round(avg_ms, 4)}


def depth_passes_needed(depths, radii):
    """The depth sort's pass count for this frame, as the library's device-side plan derives it (gsr_internal.h gsr_depth_plan):
    8-bit passes over the range of the visible Gaussians' depth bits above their minimum (low byte of the minimum cleared)."""
    vis = depths[radii > 0]
    if vis.numel() == 0:
        return 1
    bits = vis.view(__import__("torch").int32)         # positive floats order like their bit patterns
    lo, hi = int(bits.min().item()), int(bits.max().item())
    rng = hi - (lo & ~255) + 1
    return max(1, (max(1, rng).bit_length() + 7) // 8)


def stage_bytes(N, Nv, D, P, Tn, depth_passes=4):
    """Algorithmic HBM bytes per launch of each timed stage.  The per-unit figures are SURVEY.md section 8(d)'s
    (compulsory traffic: every input read once, every output written once, every list entry read once per
    tile) for the stages the reference has, and the same counting rule for the stages of our own binning
    design, priced from the item widths and pass counts the library really uses (api.hip): depth items are
    8 bytes; the first of the frame's `depth_passes` active passes reads all N items (its histogram read is the scan stage's) and
    writes the Nv visible ones, every later pass reads the Nv survivors twice (histogram, scatter) and writes them once; tile items are 4 bytes when tile bits + id bits <= 32 (C3: 12 + 20),
    else 8, ceil(tile bits / 8) passes, the first of which has no histogram read (the expansion leaves its histograms) and the last
    of which writes the 4-byte point_list instead of items; the depth-order offsets are one read of the sorted counts."""
    tb = max(1, int(np.ceil(np.log2(max(2, Tn)))))
    id_bits = max(1, int(np.ceil(np.log2(max(2, N)))))
    ib = 4 if tb + id_bits <= 32 else 8
    npass = (tb + 7) // 8
    return {
        "preprocess": 44 * N + 192 * Nv + 8 * N + 76 * Nv,
        "scan": 8 * N + 8 * N,             # + the first depth pass's histogram read (made in the scan's launch)
        "depth_sort": 8 * N + 8 * Nv + (depth_passes - 1) * 24 * Nv,
        "depth_scan": 4 * N,
        "expand": 16 * Nv + ib * D,
        "tile_sort": ((ib + 4) * D if npass == 1 else 2 * ib * D + (npass - 2) * 3 * ib * D + (2 * ib + 4) * D) + 8 * Tn,
        "ranges": 0,
        "blend_fwd": 44 * D + 8 * Tn + 24 * P,
        "bwd_prep": 64 * N,
        "blend_bwd": 40 * D + 20 * P + 44 * N,
        "geom_bwd": 4 * N + 308 * Nv + 232 * N,
    }


def measure_exchange(gsr, dist, torch, args, world, N, means, step, exchange):
    """N > 1 only, AFTER the timed region (nothing here is part of `value`): what the gradient exchange costs on this rank.
    (1) ten more steps with the exchange's own events on: the compute stream's stalls at the two collectives and the rebuild
    kernel, as the overlapped step really pays them; (2) each collective alone, synchronous, bracketed by HIP events on the
    stream it is enqueued behind (median of 5 after a barrier), to be read against DESIGN section 5's byte arithmetic."""
    dev = means.device
    rep = {"bytes_per_rank": gsr.dist.FactoredExchange.bytes_per_rank(N, world), "backend": args.backend}
    if not args.dense_exchange:
        exchange.timing = True
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        exchange.timing = False
        tm = exchange.timings_ms()
        if tm:
            rep["overlapped_step_ms"] = {k: round(float(np.median([t[k] for t in tm])), 4) for k in tm[0]}

    def alone(fn):
        ts = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return round(float(np.median(ts)), 4)

    small = torch.zeros(gsr.dist.arena_size(N, small=True), dtype=torch.float32, device=dev)
    dense = torch.zeros(gsr.dist.arena_size(N), dtype=torch.float32, device=dev)
    payload = torch.zeros(3 * N + 4, dtype=torch.float32, device=dev)
    gathered = torch.empty((world, payload.numel()), dtype=torch.float32, device=dev)
    op = dist.ReduceOp.AVG if args.backend == "nccl" else dist.ReduceOp.SUM
    rep["alone_ms"] = {
        "all_reduce_11_floats": alone(lambda: dist.all_reduce(small, op=op)),
        "all_gather_3_floats": alone(lambda: dist.all_gather_into_tensor(gathered.view(-1), payload)),
        "sh_rebuild": alone(lambda: gsr.dist.sh_gradients_from_views(means, gathered, 3, average=True)),
        "all_reduce_59_floats_dense": alone(lambda: dist.all_reduce(dense, op=op)),
    }
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C5", "C0", "C2i"], help="C3 is the headline workload; C0 / C2i are the "
                    "reference trainer's initial point set (5 000 / 100 000 Gaussians of scale 0.1), not headline configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true", help="skip the second, untimed loop that records per-stage HIP events "
                    "(the timed region never carries them)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend; nccl is RCCL (default). gloo + "
                    "--single-device rehearse the N>1 path on a one-GPU box")
    ap.add_argument("--views-per-step", type=int, default=1, help="N=1 only, not the headline workload: render this many views per "
                    "step, each on its own HIP stream, so one view's launch-bound sort chain runs under another's blend kernels")
    ap.add_argument("--dense-exchange", action="store_true", help="N>1: all-reduce the full 59-float arena instead of the factored "
                    "exchange (11 floats all-reduced + 3 all-gathered per Gaussian, SH gradient rebuilt locally)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--preheat-steps", type=int, default=100, help="untimed steps before the --warmup steps, so that a short warm-up "
                    "still starts from steady clocks (about 60 ms of load at C3)")
    ap.add_argument("--dump-steps", default=None, help="write every timed step's device time (ms, rank 0) to this file, one per line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing here has touched the GPU (torch is not even
        # imported yet); the ranks are fresh interpreters, each re-running this script with RANK/LOCAL_RANK/WORLD_SIZE set.
        from importlib import util as _ilu
        spec = _ilu.spec_from_file_location("gsr_launch", os.path.join(ROOT, "3dgs-native_amd", "launch.py"))
        launch = _ilu.module_from_spec(spec)
        spec.loader.exec_module(launch)
        rc = launch.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)
        sys.exit(rc if rc >= 0 else 128 - rc)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or bare (no WORLD_SIZE)")
    if world > 1 and args.backend == "nccl" and args.single_device:
        raise SystemExit("--single-device (every rank on cuda:0) is a rehearsal for --backend gloo: RCCL refuses two ranks on one GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    torch.cuda.set_device(local_rank if (world > 1 and not args.single_device) else 0)
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    gsr = importlib.import_module("3dgs-native_amd")
    cfg = gsr.scenes.CONFIGS[args.config]
    W, H, N = cfg["width"], cfg["height"], cfg["n"]
    if "init_scale" in cfg:     # the reference trainer's initial point set, built on the device (same on every rank)
        ip = gsr.densify.init_gaussian_params(N, cfg["init_scale"], dev)
        sc = {"means": ip["positions"].cpu().numpy(), "shs": ip["shs"].cpu().numpy().reshape(N, 16, 3), "opacities": ip["opacities"].cpu().numpy().reshape(N, 1),
              "scales": ip["scales"].cpu().numpy(), "rotations": ip["rotations"].cpu().numpy()}
    else:
        sc = gsr.scenes.synthetic_scene(N, cfg["scale_median"], cfg["scale_sigma"], cfg["seed"])  # same on every rank
    pose = gsr.scenes.LEGO_FRAME0 if world == 1 else gsr.scenes.orbit_pose(rank, world)
    cam = gsr.cameras.nerf_camera(pose, W, H, gsr.scenes.LEGO_CAMERA_ANGLE_X)
    bg = np.zeros(3, np.float32)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    means, shs, opac, scales, rots = t(sc["means"]), t(sc["shs"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"])
    dpix = t(np.random.default_rng(99).normal(0.0, 1.0, (H, W, 3)) / (H * W * 3))
    fkw = dict(background=bg, means3D=means, colors=None, opacity=opac, scales=scales, rotations=rots, scale_modifier=1.0,
               viewmatrix=cam["world_to_camera"], projmatrix=cam["full_proj_matrix"], tan_fovx=cam["tan_fovx"],
               tan_fovy=cam["tan_fovy"], image_height=H, image_width=W, sh=shs, degree=3, campos=cam["camera_center"],
               prefiltered=False, antialiasing=False, clamped=True)

    sh_out = torch.empty((N * 16, 3), dtype=torch.float32, device=dev) if world > 1 else None
    exchange = gsr.dist.FactoredExchange()
    exchange_report = None
    parity_failed = False

    def step():
        img, depth, buf = gsr.render_gaussians(**fkw)
        factored = world > 1 and not args.dense_exchange
        grads = gsr.backward(
            background=bg, means3D=means, dL_dpixels=dpix, opacity=opac, shs=shs, scales=scales, rotations=rots,
            scale_modifier=1.0, viewmatrix=fkw["viewmatrix"], projmatrix=fkw["projmatrix"], tan_fovx=fkw["tan_fovx"],
            tan_fovy=fkw["tan_fovy"], image_height=H, image_width=W, campos=fkw["campos"], radii=buf["radii"],
            means2D=buf["points_xy_image"], conic_opacity=buf["conic_opacity"], rgb=buf["colors"], cov3Ds=buf["cov3Ds"],
            clamped=buf["clamped_state"], geom_buffer=None, binning_buffer={"point_list": buf["point_list"]},
            img_buffer={"ranges": buf["ranges"], "final_Ts": buf["final_Ts"], "n_contrib": buf["n_contrib"]}, degree=3,
            sh_gradient="factored" if factored else "dense", on_payload=exchange.start_gather if factored else None)
        if factored:
            # all-reduce 11 floats per Gaussian, all-gather 3 (started inside backward, beside its per-Gaussian half), rebuild
            # the averaged SH gradient locally (beside the all-reduce)
            grads.update(exchange.finish(grads, means, 3, average=True, out=sh_out))
        elif world > 1:
            gsr.dist.reduce_gradients(grads["_arena"], world)
        return buf, grads

    vps = max(1, args.views_per_step) if world == 1 else 1
    if vps > 1:
        one_view = step
        streams = [torch.cuda.Stream(device=dev) for _ in range(vps)]

        def step():
            last = None
            main = torch.cuda.current_stream(dev)
            for st_ in streams:
                st_.wait_stream(main)
                with torch.cuda.stream(st_):
                    last = one_view()
            for st_ in streams:
                main.wait_stream(st_)
            return last

    # Everything the measurement itself needs exists BEFORE the warm-up: the library's per-stage events (allocated, paused), one
    # HIP event per step boundary (torch creates an event at its first record, so each is recorded once here), the workload's
    # D / visible count (one untimed step + two readbacks), and the interpreter's collector run and switched off (as `timeit`
    # does: a generation-2 pass stalls the launch thread for ~1 ms every ~150 steps).  Between the last warm-up step and the timed
    # region there is only the synchronize (+ barrier) the protocol asks for, and no stage event is recorded inside the region.
    use_events = not args.no_stage_events
    STAGE_STEPS = 20
    if use_events:
        gsr._lib.stage_timing(True, STAGE_STEPS, every=0)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for m in marks:
        m.record()
    buf, grads = step()
    torch.cuda.synchronize()
    D = int(buf["point_list"].shape[0])
    Nv = int((buf["radii"] > 0).sum().item())
    depth_passes = depth_passes_needed(buf["depths"].view(-1), buf["radii"].view(-1))
    del buf, grads
    import gc
    gc.collect()
    gc.disable()
    # Pre-heat: after the GPU has idled for more than about a millisecond (here: the scene build, the readbacks above) it needs
    # 10-20 ms of load to come back to its steady clocks -- the first ~20 steps run at 0.63-0.65 ms, later ones at 0.61
    # (tools/ramp_probe.py, gpurun_out/r04_b/ramp_probe.txt) -- which a 5-step warm-up (3 ms) does not cover.  So
    # --preheat-steps untimed steps (a fixed count: every rank runs the same number of collectives) precede the W warm-up steps
    # asked for, back to back with them, and the line reports them (`preheat_steps`).
    for _ in range(max(0, args.preheat_steps)):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    # one HIP event per step boundary on the launch stream (torch's current stream IS the stream handed to the library):
    # per-step device times for the median / p10 / p90; `ms_per_step` stays the wall-clock mean of the whole region
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        step()
        marks[k + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    # the per-stage table comes from a SECOND, untimed loop with the library's stage events on every step
    stages, nrec = ({}, 0)
    if use_events:
        gsr._lib.stage_sampling(1)
        for _ in range(STAGE_STEPS):
            step()
        torch.cuda.synchronize()
        stages, nrec = gsr._lib.stage_times()
        gsr._lib.stage_timing(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if world > 1:
        try:
            exchange_report = measure_exchange(gsr, dist, torch, args, world, N, means, step, exchange)
        except Exception as e:      # the line with `value` (already measured) must still go out; every rank runs the same code
            exchange_report = {"error": f"{type(e).__name__}: {e}"[:300]}

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * vps * W * H / (elapsed / args.steps) / 1e6
    per_step = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)])   # this rank's device time per step
    if args.dump_steps and rank == 0:
        np.savetxt(args.dump_steps, per_step, fmt="%.4f")
    backend_name = "RCCL" if args.backend == "nccl" else "gloo (host memory" + (", every rank on cuda:0: a rehearsal" if args.single_device else "") + ")"

    out = {
        "metric": "Mpixels/s forward+backward at 800x800, 1M Gaussians" if args.config == "C3" else f"Mpixels/s forward+backward ({args.config})",
        "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "preheat_steps": max(0, args.preheat_steps),
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {'reference initial point set (scale ' + str(cfg['init_scale']) + ')' if 'init_scale' in cfg else 'synthetic'} {W}x{H}, {N} Gaussians, SH degree 3, seed {cfg['seed']}, forward+backward, "
                               f"Lego train pose 0" + (" rotated per rank" if world > 1 else ""),
                   "width": W, "height": H, "gaussians": N, "visible": Nv, "tile_pairs_D": D, "views_per_step": world * vps, "views_per_gpu": vps,
                   "parallelism": f"dp{world}: {'one view' if vps == 1 else str(vps) + ' views on ' + str(vps) + ' streams'} per GPU, replicated Gaussians" + ((f", {backend_name} all-reduce of the 59-float gradient arena" if args.dense_exchange else
                                                                                        f", {backend_name} all-reduce of 11 floats + all-gather of 3 floats per Gaussian, SH gradient rebuilt per rank") if world > 1 else "")},
        "step_ms": {"median": round(float(np.median(per_step)), 4), "p10": round(float(np.percentile(per_step, 10)), 4),
                    "p90": round(float(np.percentile(per_step, 90)), 4), "min": round(float(per_step.min()), 4),
                    "max": round(float(per_step.max()), 4), "argmax": int(per_step.argmax()), "n": int(args.steps),
                    "first": [round(float(x), 4) for x in per_step[:8]],
                    "how": "HIP events on the launch stream between consecutive steps (rank 0); ms_per_step is the wall-clock mean"},
    }

    if exchange_report is not None:
        out["exchange"] = exchange_report
    if rank == 0:
        P, Tn = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        if stages and nrec > 0:
            sb = stage_bytes(N, Nv, D, P, Tn, depth_passes)
            timed = {k: v for k, v in stages.items() if k in sb}
            dom = max(timed, key=timed.get)
            dur_s = timed[dom] * 1e-3
            achieved = sb[dom] / dur_s / 1e9
            # HBM bytes from the PMC passes (tools/profile_round.sh) belong to ONE build of the library: they are quoted only
            # when the library loaded now hashes to the build they were measured on, else null (never a stale figure)
            traffic, build = None, gsr._lib.build_hash()
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("_build") == build:
                    traffic = tj.get(args.config, {}).get(dom)
            gpu_ms = sum(stages.values())
            B = 340 * N + 596 * Nv + 128 * D + 16 * Tn + 44 * P
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "build": build,
                               "algorithmic_bytes": int(sb[dom]), "avg_ms": round(timed[dom], 4), "steps_measured": nrec,
                               "pipeline": {"algorithmic_bytes": int(B), "gpu_ms": round(gpu_ms, 4),
                                            "achieved": round(B / (gpu_ms * 1e-3) / 1e9, 2),
                                            "frac": round(B / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                               "stage_ms": {k: round(v, 4) for k, v in stages.items()}}
            # the two blend kernels are not HBM-bound: their roof is vector-instruction issue (SURVEY.md section 8(d) "secondary
            # practical bound"), priced from the SQ counters measured on this very build (null when the build has none on file)
            qpath = os.path.join(ROOT, "profiles", "sq_counters.json")
            out["roofline_valu"] = None
            if os.path.exists(qpath):
                with open(qpath) as f:
                    qj = json.load(f)
                if qj.get("_build") == build and args.config in qj:
                    out["roofline_valu"] = {st: roofline_valu(st, cnts, timed[st]) for st, cnts in qj[args.config].items() if st in timed}
                    out["roofline_valu"]["counters"] = qj.get("_round")
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            okw = dict(fkw)
            for k_np, k_kw in [("means", "means3D"), ("opacities", "opacity"), ("scales", "scales"), ("rotations", "rotations"), ("shs", "sh")]:
                okw[k_kw] = sc[k_np]
            t0 = time.perf_counter()
            oi, od, ob = oracle.render_gaussians(**okw)
            t_f = time.perf_counter() - t0
            geom = {"radii": ob["radii"], "means2D": ob["points_xy_image"], "conic_opacity": ob["conic_opacity"], "rgb": ob["colors"],
                    "clamped_state": ob["clamped_state"]}
            t0 = time.perf_counter()
            og = oracle.backward(background=bg, means3D=sc["means"], dL_dpixels=dpix.cpu().numpy(), opacity=sc["opacities"], shs=sc["shs"],
                            scales=sc["scales"], rotations=sc["rotations"], viewmatrix=fkw["viewmatrix"], projmatrix=fkw["projmatrix"],
                            tan_fovx=fkw["tan_fovx"], tan_fovy=fkw["tan_fovy"], image_height=H, image_width=W, campos=fkw["campos"],
                            cov3Ds=ob["cov3Ds"], geom_buffer=geom, binning_buffer={"point_list": ob["point_list"]},
                            img_buffer={"ranges": ob["ranges"], "final_Ts": ob["final_Ts"], "n_contrib": ob["n_contrib"]})
            t_b = time.perf_counter() - t0
            # SURVEY section 8(d)'s courtesy figure: the same restatement over all host cores (bands of Gaussians / of tile rows in
            # threads; per-thread gradient accumulators summed afterwards).  Not how the reference's CPU path runs (Warp's CPU device
            # is one serial loop) and not the baseline: reported beside it.
            T = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))   # the cores this process may use
            t0 = time.perf_counter()
            pi, pd, pb = oracle.render_gaussians(**okw, threads=T)
            pgeom = {"radii": pb["radii"], "means2D": pb["points_xy_image"], "conic_opacity": pb["conic_opacity"], "rgb": pb["colors"],
                     "clamped_state": pb["clamped_state"]}
            oracle.backward(background=bg, means3D=sc["means"], dL_dpixels=dpix.cpu().numpy(), opacity=sc["opacities"], shs=sc["shs"],
                            scales=sc["scales"], rotations=sc["rotations"], viewmatrix=fkw["viewmatrix"], projmatrix=fkw["projmatrix"],
                            tan_fovx=fkw["tan_fovx"], tan_fovy=fkw["tan_fovy"], image_height=H, image_width=W, campos=fkw["campos"],
                            cov3Ds=pb["cov3Ds"], geom_buffer=pgeom, binning_buffer={"point_list": pb["point_list"]},
                            img_buffer={"ranges": pb["ranges"], "final_Ts": pb["final_Ts"], "n_contrib": pb["n_contrib"]}, threads=T)
            t_all = time.perf_counter() - t0
            del pi, pd, pb, pgeom
            # the same frame is also a full-size parity check of the timed path (the oracle as checker, outside the timed region)
            gb, gg = step()
            torch.cuda.synchronize()
            gi = gsr.render_gaussians(**fkw)[0].cpu().numpy()
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import parity   # the tolerances of the parity tests; here only measured and reported
            ierr = np.abs(gi.astype(np.float64) - oi).max(axis=2)
            par = {"image_max_abs_err": float(ierr.max()), "image_frac_within_2e-5": float((ierr <= parity.IMG_TIGHT).mean()),
                   "image_pixels_beyond_1e-3": int((ierr > parity.IMG_LOOSE).sum()),
                   "n_contrib_frac_equal": float((gb["n_contrib"].cpu().numpy() == ob["n_contrib"]).mean())}
            for k in ("radii", "point_offsets", "point_list", "ranges"):
                a, b = gb[k].cpu().numpy(), np.asarray(ob[k])
                par[k + "_exact"] = bool(a.shape == b.shape and (a == b).all())
            for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dshs"):
                ok, rel = parity.grad_margin(gg[k], og[k])      # |d| <= 1e-4 max|g| + 1e-3 |g|
                par[k + "_frac_within_tol"] = ok
                par[k + "_max_err_over_max"] = rel
            # the same tripwires as tests/test_gpu_parity.py::test_full_size_configs (about 10x the margins measured in rounds 1-2)
            trip = []
            if par["image_pixels_beyond_1e-3"] > parity.TRIP_FLIPS:
                trip.append("image flips")
            if par["n_contrib_frac_equal"] < parity.TRIP_NCONTRIB:
                trip.append("n_contrib")
            trip += [k for k in ("radii", "point_offsets", "point_list", "ranges") if not par[k + "_exact"]]
            trip += [k for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dshs")
                     if par[k + "_frac_within_tol"] < parity.TRIP_GRAD_FRAC or par[k + "_max_err_over_max"] > parity.TRIP_GRAD_MAX]
            par["tripwires"] = "ok" if not trip else "VIOLATED: " + ", ".join(trip)
            parity_failed = bool(trip)
            out["cpu_baseline"] = {"value": round(W * H / (t_f + t_b) / 1e6, 5), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                                   "sample": f"one full {args.config} frame (same scene and view), forward {t_f:.2f} s + backward {t_b:.2f} s, "
                                             f"single-thread C oracle (gcc -O2), host has {os.cpu_count()} cores",
                                   "all_cores": {"value": round(W * H / t_all / 1e6, 5), "unit": "Mpixels/s", "cores": T,
                                                 "kind": "port over threads (bands of Gaussians and of tile rows; the sort stays serial): NOT reference-equivalent, "
                                                         "a courtesy figure beside the single-thread baseline", "sample": f"the same frame, forward + backward {t_all:.2f} s"},
                                   "parity_full_size": {k: (round(v, 8) if isinstance(v, float) else v) for k, v in par.items()}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the timed path disagrees with the oracle beyond the tripwires (cpu_baseline.parity_full_size)")


if __name__ == "__main__":
    main()
