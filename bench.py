#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Msamples/s of the spectral path tracer on the
1M-triangle Cornell scene (config C4: 1920x1080, sobol/1024-spp schedule, `direct` integrator), tile-sharded
over N GPUs with one RCCL reduce of the float framebuffer, plus the traversal kernel's algorithmic
bandwidth against the HBM roofline and the CPU checker timed on the host cores beside it.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one iteration = one camera sample (one full path) for every pixel of the frame.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP, NTRI = 1920, 1080, 1024, 1_000_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the achievable copy rate
ROOFLINE_ITERS = 16    # iterations of the launch the roofline is measured on (and of the instrumented launch that counts its records):
                       # long enough that the fill and the drain of a launch do not colour the lane statistics


def native_oracle():
    """The CPU checker rebuilt for THIS host (`g++ -O3 -march=native`, the flags BASELINE.md section 3 promises; -ffp-contract=off kept, so
    the arithmetic is the checker's).  The shipped library is built -O2 for a generic x86-64 because it travels between machines."""
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "pr_oracle.cpp")
    flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared"]
    out = os.path.join(tempfile.mkdtemp(prefix="pr_oracle_native_"), "libpr_oracle_native.so")
    try:
        subprocess.check_call(["g++"] + flags + [src, "-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out, "g++ " + " ".join(flags[:2])
    except Exception:
        return None, "prebuilt -O2 generic x86-64 (no compiler on this host)"


def host_cpu_share():
    """Hardware threads this process may actually use: the smaller of the machine's count, the affinity mask and the cgroup CPU quota
    (a GPU box hands a container a share of its host CPUs: os.cpu_count() alone overstates it)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # cgroup v1
                q, p = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, (os.cpu_count() or 1), quota


def cpu_baseline(scene_desc, iters=4, tile=32, what="the same 1M-triangle scene"):
    """CPU checker ("port") on the host cores: `iters` full-frame iterations of the SAME workload, handed to the worker threads as
    `tile` x `tile`-pixel Z-order tiles so that every hardware thread has work (the reference's 8 x 8 grid feeds at most 64)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    cores, host_threads, quota = host_cpu_share()
    path, how = native_oracle()
    lib = oracle_binding.load_from(path) if path else None
    t0 = time.time()
    o = oracle_binding.OracleScene(scene_desc, lib=lib)
    t_build = time.time() - t0
    tx, ty = (scene_desc.width + tile - 1) // tile, (scene_desc.height + tile - 1) // tile
    o.set_tile_grid(tx, ty)
    t0 = time.time()
    o.render(iters, threads=cores)
    dt = time.time() - t0
    st = o.statistics()
    return {"value": round(st["pixel_samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "threads_busy": min(cores, tx * ty),
            "host_threads": host_threads, "cgroup_cpu_quota": quota, "kind": "port", "build": how,
            "sample": "%d iteration(s) of the full %dx%d frame of %s (%d samples, %.1f s render, %.1f s SAH BVH build excluded), "
                      "%d tiles of %dx%d pixels for %d threads; CPU restatement, not Embree"
                      % (iters, scene_desc.width, scene_desc.height, what, st["pixel_samples"], dt, t_build, tx * ty, tile, tile, cores)}


def kernel_source_sha16():
    """Fingerprint of the device sources the hot kernel is built from; a PMC summary taken from another build is not quoted."""
    import hashlib
    h = hashlib.sha256()
    for f in ("pearray_amd/csrc/device/render.hip", "pearray_amd/csrc/device/path_wave.inl", "pearray_amd/csrc/device/pr_device.h", "pearray_amd/csrc/device/bvh.hip"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def gather_ceiling():
    """Attainable rate of dependent per-lane record gathers at the traversal loop's own shape (waves per CU, lane fill, record mix, table
    size, arithmetic per step, L2 locality), measured with tools/micro/gather_ceiling.hip and committed under profiles/: G records/s."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gather_ceiling.json")), reverse=True):
        try:
            with open(path) as f:
                g = json.load(f)
            return float(g["g_records_per_s"][g["quoted"]]) * 1e9, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


PMC_EXTRA = {}   # issue / texture-path occupancy of the same launches, from the same summary (filled by pmc_traffic)


def pmc_traffic(kernel_signature):
    """HBM-side bytes per ITERATION of the dominant kernel from the newest committed rocprofv3 --pmc summary (PMC counters cannot
    be read from inside this process): (2 x FETCH_SIZE + WRITE_SIZE) KiB -- the gfx950 correction of MI355X_MICROARCH.md (HBM:
    FETCH_SIZE tallies 128-B requests at 64 B).  Only a summary of THIS kernel (same template instantiation) built from THESE
    sources counts; otherwise traffic is null and the reason is reported."""
    import glob
    reason = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):   # newest round first
        try:
            with open(path) as f:
                summ = json.load(f)
        except Exception:
            continue
        rel = os.path.relpath(path, ROOT)
        if summ.get("kernel_source_sha16") != kernel_source_sha16():
            reason = reason or "%s was taken from another build of the kernel sources" % rel
            continue
        for name, k in summ["kernels"].items():
            if kernel_signature in name.replace(" ", "") and "FETCH_SIZE_per_iteration" in k:
                PMC_EXTRA.clear()
                cycles = k.get("GRBM_GUI_ACTIVE_per_iteration", 0.0) / 8.0    # summed over the 8 XCDs
                if cycles and "SQ_INSTS_VALU_per_iteration" in k:              # what the kernel is really bound by
                    PMC_EXTRA.update({"valu_issue_busy": round(k["SQ_INSTS_VALU_per_iteration"] * 4.0 / (1024.0 * cycles), 3),  # 1024 SIMDs, 4 cycles per wave64 instruction
                                      "ta_busy": round(k.get("TA_TA_BUSY_sum_per_iteration", 0.0) / (256.0 * cycles), 3),
                                      "l2_hit": round(k["TCC_HIT_sum_per_iteration"] / (k["TCC_HIT_sum_per_iteration"] + k["TCC_MISS_sum_per_iteration"]), 3)
                                      if "TCC_HIT_sum_per_iteration" in k else None})
                return (2.0 * k["FETCH_SIZE_per_iteration"] + k["WRITE_SIZE_per_iteration"]) * 1024.0, rel, None
        reason = reason or "%s holds no counters for %s" % (rel, kernel_signature)
    return None, None, reason or "no profiles/r*_pmc_summary.json"


def roofline(ctx, rank, iters=ROOFLINE_ITERS, variant="0u", owned_fraction=1.0, pmc_workload=True):
    """Roofline of the dominant kernel, measured live on this rank: HIP events on the launch stream around every launch
    (pass 1), node/triangle record counters of the instrumented kernel variant (pass 2; same pixels, statistically identical
    iterations).  Algorithmic bytes = rays x (32 B ray + 16 B result) + 64 B x inner + 128 B x leaf BVH records fetched."""
    ctx.setTiming(True)
    ctx.render(iters)
    ctx.waitForFinish()
    fam = {k: ctx.kernelTime(k) for k in ("path", "trace_closest", "trace_any", "shade")}
    ctx.setTiming(False)
    tc0 = ctx.traceCounters()
    ctx.setInstrumentation(True)
    ctx.render(iters)
    ctx.waitForFinish()
    ctx.setInstrumentation(False)
    tc1 = ctx.traceCounters()
    d = {k: tc1[k] - tc0[k] for k in ("rays_closest", "rays_any", "nodes_closest", "leaves_closest", "nodes_any", "leaves_any",
                                      "wave_steps_closest", "wave_steps_any", "shade_batches", "shade_lanes")}
    per_ray = tc1["ray_bytes"] + tc1["hit_bytes"]
    bytes_closest = d["rays_closest"] * per_ray + tc1["node_bytes"] * d["nodes_closest"] + tc1["leaf_bytes"] * d["leaves_closest"]
    bytes_any = d["rays_any"] * per_ray + tc1["node_bytes"] * d["nodes_any"] + tc1["leaf_bytes"] * d["leaves_any"]
    persistent = fam["path"][1] > 0
    kinfo = ctx.pipelineInfo()
    if persistent:
        # one launch = `iters` iterations of everything: closest-hit and occlusion traversal + shading, fused
        name = "k_path_latency" if kinfo["kernel"] == "latency" else "k_path_persistent_occ3"
        # the instantiation the scene's features (and, for the throughput kernel, the width of its BVH records) select
        kernel, substr = name, "%s<false,%s%s>" % (name, variant, "" if kinfo["kernel"] == "latency" else (",true" if kinfo["bvh_width"] == 6 else ",false"))
        ms, n = fam["path"]
        alg_bytes = (bytes_closest + bytes_any) / n          # per launch
        iters_per_launch = iters / n
    else:
        kernel, substr = "k_trace_closest", "k_trace_closest"
        ms, n = fam["trace_closest"]
        alg_bytes = bytes_closest / n
        iters_per_launch = iters / n
    avg_ms = ms / max(n, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    t_iter, src, why_not = pmc_traffic(substr)
    if t_iter is not None and not persistent:
        t_iter, why_not = None, "lockstep summaries are per launch of one path depth"
    if t_iter is not None and not pmc_workload:
        t_iter, why_not = None, "the committed PMC summary is of the full-size workload, this run is not"
    # The PMC summary is of the FULL frame.  A rank of an N-GPU job renders `owned_fraction` of the pixels, so its launch moves about
    # that fraction of the bytes (the tiles are dealt round-robin along the Z-order curve: every rank sees the same mix of the image);
    # quoting the full-frame bytes against a share's launch time gave a figure above the chip's peak (round 4 review).
    traffic_bytes = t_iter * iters_per_launch * owned_fraction if t_iter is not None else None
    records = d["nodes_closest"] + d["leaves_closest"] + d["nodes_any"] + d["leaves_any"]
    inner = d["nodes_closest"] + d["nodes_any"]
    records_per_s = records / n / (avg_ms * 1e-3)
    ceiling, ceiling_src = gather_ceiling() if variant == "0u" else (None, None)   # (measured at the C4 loop's shape: record mix, table size, lane fill)
    # the same bytes with an inner record counted as the 48 bytes the step loads instead of the 64-byte unit the memory side moves
    # (a six-wide tree's step loads all 64)
    achieved48 = (alg_bytes - (16.0 if kinfo["bvh_width"] == 4 else 0.0) * inner / n) / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": round(traffic_bytes / (avg_ms * 1e-3) / 1e9, 1) if traffic_bytes else None,
            "traffic_bytes_per_launch": traffic_bytes, "traffic_source": src, "traffic_unavailable": why_not, "pmc": dict(PMC_EXTRA) if traffic_bytes else None,
            "traffic_scaled_to_owned_fraction": round(owned_fraction, 6) if (traffic_bytes and owned_fraction != 1.0) else None,
            "kernel": kernel, "kernel_organisation": kinfo["kernel"], "shader_waves": kinfo["shader_waves"], "shading_share": round(kinfo["shading_share"], 4),
            "grid": [kinfo["blocks"], kinfo["slots_per_block"]], "avg_launch_ms": round(avg_ms, 4),
            # children per inner record, chosen per scene by the builder from its estimate of either tree (inner records per ray through the scene's box)
            "bvh_width": kinfo["bvh_width"],
            "bvh_width_note": ("six-wide inner records: a third fewer records per ray than this scene's four-wide tree, so fewer algorithmic bytes -- across tree widths compare "
                               "Msamples/s, not frac" if kinfo["bvh_width"] == 6 else None),
            "bvh_top": ("entities in scene order", "surface-area tree over the entities", "Morton order of the entities")[kinfo["bvh_top"]], "bvh_stack_bound": kinfo["bvh_stack_bound"], "bvh_cost_estimate": {"4_wide": round(kinfo["bvh_cost_4_wide"], 2), "6_wide": round(kinfo["bvh_cost_6_wide"], 2)},
            "launches": n, "iterations_per_launch": iters_per_launch, "algorithmic_bytes_per_launch": round(alg_bytes),
            "algorithmic_bytes_per_closest_ray": round(bytes_closest / max(d["rays_closest"], 1), 1),
            "algorithmic_bytes_per_occlusion_ray": round(bytes_any / max(d["rays_any"], 1), 1),
            "nodes_per_closest_ray": round(d["nodes_closest"] / max(d["rays_closest"], 1), 2),
            "leaves_per_closest_ray": round(d["leaves_closest"] / max(d["rays_closest"], 1), 2),
            "rays_per_launch": round((d["rays_closest"] + d["rays_any"]) / n),
            # records fetched per second against the attainable rate of dependent per-lane gathers at this loop's own shape
            # (tools/micro/gather_ceiling.hip, profiles/r05_gather_ceiling.txt): what the memory pipeline allows, L2 locality included
            "records_per_s": round(records_per_s), "gather_ceiling_records_per_s": ceiling, "gather_frac": round(records_per_s / ceiling, 4) if ceiling else None,
            "gather_ceiling_source": ceiling_src,
            "frac_with_48_byte_inner_records": round(achieved48 / HBM_PEAK_GBS, 4),
            "peak_note": "8 TB/s is the HBM3E spec peak; the 125 MB working set lives in the 256 MiB Infinity Cache and FETCH_SIZE counts fabric requests "
                         "(the guide's ceiling for random gathers out of a table of this size is 7.4 - 7.9 TB/s)",
            "lane_utilisation": round(records / max(64 * (d["wave_steps_closest"] + d["wave_steps_any"]), 1), 3),
            "shade_pass_fill": round(d["shade_lanes"] / max(64 * d["shade_batches"], 1), 3) if d["shade_batches"] else None,
            "family_ms_per_iter": {k: round(v[0] / iters, 3) for k, v in fam.items() if v[1]}}


def secondary_c5(device, steps=20, warmup=8):
    """BASELINE config C5 (examples/complex.prc) beside the headline line: the same timed region and roofline on one GPU, without a CPU leg.
    (BASELINE names 8 GPUs for C5; a one-GPU figure is what a one-GPU box can measure -- tools/gpu_shares.py holds its tile-share table.)"""
    import torch
    from pearray_amd import backend, scene
    sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))
    sc.desc.settings.width, sc.desc.settings.height = W, H
    ctx = backend.RenderContext(sc, device=device)
    ctx.render(warmup)
    ctx.waitForFinish()
    before = ctx.statistics()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.render(steps)
    ctx.waitForFinish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    after = ctx.statistics()
    samples = after["pixel_samples"] - before["pixel_samples"]
    rays = sum(after[k] - before[k] for k in ("primary_rays", "bounce_rays", "shadow_rays"))
    out = {"workload": "C5: examples/complex.prc (%d triangles + 4 spheres, sky + sun, glass / rough conductor / principled materials), %dx%d, `direct` "
                       "integrator, sobol %d-spp schedule, hero wavelengths (spd CMIS), %d iterations timed on ONE GPU" % (sc.desc.n_triangles, W, H, sc.desc.settings.aa_samples, steps),
           "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
           "mrays_per_s": round(rays / dt / 1e6, 2), "mean_path_depth": round((after["camera_depth"] - before["camera_depth"]) / max(samples, 1), 3),
           "roofline": roofline(ctx, 0, variant="255u")}
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--triangles", type=int, default=NTRI)
    ap.add_argument("--workload", choices=("c4", "c5"), default="c4",
                    help="c4 (default, the BASELINE headline): 1M-triangle Cornell box; c5: examples/complex.prc (sky + sun, glass, rough conductor, "
                         "principled, spheres) from its committed array fixture, sobol 4096-spp schedule")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C5 object (`secondary`) a default one-GPU C4 run appends to its line")
    ap.add_argument("--profile-only", action="store_true", help="skip roofline/cpu passes (for rocprofv3 runs)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a one-GPU box: every rank renders on device 0 and the collectives run over gloo "
                         "(staged through host memory); exercises tile sharding + reduce end to end, the number is NOT a result")
    ap.add_argument("--check-frame", action="store_true", help="rank 0: compare the reduced frame with a one-rank render of the same iterations")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from pearray_amd import backend, distributed, scene, tiling

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.rehearse_on_one_gpu else "nccl", rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    width, height = args.width, args.height
    if args.workload == "c5":
        sc = scene.ArrayScene(os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz"))   # the sky table is rebuilt on load (prgpu_sky_table)
        sc.desc.settings.width, sc.desc.settings.height = width, height
        spp = sc.desc.settings.aa_samples
        workload = ("C5: examples/complex.prc (%d triangles + 4 spheres, sky + sun, glass / rough conductor / principled materials), %dx%d, `direct` "
                    "integrator, sobol %d-spp schedule, hero wavelengths (spd CMIS), %d iterations timed" % (sc.desc.n_triangles, width, height, spp, args.steps))
        variant, what = "255u", "the same scene (complex.prc)"
    else:
        spp = SPP
        sc = scene.cornell_soup(width, height, spp=SPP, n_triangles=args.triangles)
        workload = ("C4: Cornell box + %d-triangle soup (%d triangles), %dx%d, `direct` integrator (NEE+MIS+RR, depth 64), "
                    "sobol %d-spp schedule, hero wavelengths (spd CMIS), %d iterations timed" % (args.triangles - 32, args.triangles, width, height, SPP, args.steps))
        variant, what = "0u", "the same 1M-triangle scene"
    if args.steps + args.warmup + 2 * ROOFLINE_ITERS > spp:
        raise SystemExit("steps + warmup (+ %d roofline iterations) exceed the %d-spp schedule" % (2 * ROOFLINE_ITERS, spp))
    t0 = time.time()
    ctx = backend.RenderContext(sc, device=local)
    t_create = time.time() - t0
    # tiles dealt round-robin along the Z-order curve; the slowest rank's share decides (tools/gpu_probe_share8.py with RANK_PROBE / TILE,
    # ms per iteration, max over ranks): N = 8: 16x16 tiles 2.38, 32x32 2.47, 64x64 2.39 (rank 0 alone 2.20), 8x8 2.40;
    # N = 4: 4.28 / 4.31 / 4.39; N = 2: 7.67 / 7.53 / 7.45 -- small tiles balance better, large ones keep more coherence
    tile_px = 64 if world <= 2 else 16
    tiles = tiling.tiles_for_rank(width, height, rank, world, tile=tile_px) if world > 1 else []
    ctx.setTiles(tiles)
    xyz = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
    smp = torch.zeros((height, width), dtype=torch.int32, device=dev)
    ctx.bindFramebuffer(xyz.data_ptr(), smp.data_ptr())
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The path's one collective goes through the C ABI (prgpu_comm_* / prgpu_reduce: RCCL called from libprgpu on the scene's stream);
    # torch.distributed only launches the ranks, ships rank 0's communicator id and provides the barrier.  The one-GPU rehearsal
    # keeps the host-staged gloo reduce (RCCL refuses two ranks on one device).
    comm = None
    if not args.rehearse_on_one_gpu:
        def exchange(raw):
            if world == 1:
                return raw
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        # every rank probes RCCL on its own and the ranks agree BEFORE any call in which one of them could fail while the others block
        usable = 1 if backend.rccl_available() else 0
        if world > 1:
            flag = torch.tensor([usable], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            usable = int(flag.item())
        try:
            comm = backend.Communicator(world, rank, device=local, exchange=exchange) if usable else None
        except Exception as e:  # noqa: BLE001 -- keep the bench line: fall back to the torch.distributed reduce on every rank
            sys.stderr.write("[bench] rank %d: prgpu_comm_create failed (%s); falling back to torch.distributed.reduce\n" % (rank, e))
            comm = None
        if not usable:
            sys.stderr.write("[bench] rank %d: RCCL is not reachable through libprgpu on every rank; falling back to torch.distributed.reduce\n" % rank)
        if world > 1:           # all ranks take the same branch
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None

    def reduce_frame():
        if comm is not None:
            ctx.reduce(comm, root=0)
            ctx.waitForFinish()
        else:
            distributed.reduce_framebuffer(xyz, smp)

    def reduce_alone_ms():
        """The collective alone, once more on the finished frame (a frame may be reduced again: prgpu.h), so that a SCALE line separates
        render time, imbalance and the reduce: device time between HIP events around the RCCL group (kernel family "reduce"), or the
        host's clock around the torch.distributed fallback (which sums in place: only called when its frame is not read afterwards)."""
        if world == 1:
            return 0.0
        if comm is not None:
            ctx.setTiming(True)
            ms0 = ctx.kernelTime("reduce")[0]
            ctx.reduce(comm, root=0)
            ctx.waitForFinish()
            ms = ctx.kernelTime("reduce")[0] - ms0
            ctx.setTiming(False)
            return ms
        barrier()
        t = time.perf_counter()
        distributed.reduce_framebuffer(torch.zeros_like(xyz), torch.zeros_like(smp))
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3

    if world > 1:  # warm the collective up while the frame is still all zeros (communicator channels, buffer registration)
        reduce_frame()
    ctx.render(args.warmup)
    ctx.waitForFinish()
    before = ctx.statistics()
    barrier()
    t0 = time.perf_counter()
    ctx.render(args.steps)                      # K iterations of the hot path
    ctx.waitForFinish()
    dt_render = time.perf_counter() - t0        # this rank's share, before it waits for the others in the reduce
    reduce_frame()                              # RCCL sum onto rank 0 (a checked no-op at N=1)
    barrier()
    dt = time.perf_counter() - t0
    after = ctx.statistics()
    dt = distributed.max_scalar(dt, device=dev)
    render_ms = [x / args.steps * 1e3 for x in distributed.gather_scalars(dt_render, device=dev)]   # per rank, ms per step
    samples = distributed.sum_scalar(after["pixel_samples"] - before["pixel_samples"], device=dev)
    rays = distributed.sum_scalar(sum(after[k] - before[k] for k in ("primary_rays", "bounce_rays", "shadow_rays")), device=dev)
    depth = distributed.sum_scalar(after["camera_depth"] - before["camera_depth"], device=dev)
    own_samples = after["pixel_samples"] - before["pixel_samples"]
    reduce_ms = distributed.max_scalar(reduce_alone_ms(), device=dev)
    rccl_ranks, rccl_rank = comm.query() if comm is not None else (None, None)   # what the communicator itself says (rank 0 = the root)

    out = {
        "metric": "Msamples/s", "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "samples_per_step": int(samples / args.steps), "parallelism": "tiles%d" % world,
                   "tile_px": tile_px if world > 1 else None,
                   # a rank's own render time per step (before the reduce): max - min is imbalance, min is the per-rank latency floor
                   "rank_render_ms_per_step": {"min": round(min(render_ms), 4), "max": round(max(render_ms), 4), "slowest_rank": int(np.argmax(render_ms)),
                                               "all": [round(x, 4) for x in render_ms]},
                   "collective": "none (one rank)" if world == 1 else ("prgpu_reduce (RCCL from libprgpu)" if comm is not None else "torch.distributed.reduce"),
                   # ncclCommCount / ncclCommUserRank of the communicator the reduce ran on, read on the root: null when no RCCL communicator exists
                   # (one rank, or the host-staged rehearsal)
                   "rccl_ranks": rccl_ranks if (comm is not None and world > 1) else None, "rccl_rank_of_root": rccl_rank if (comm is not None and world > 1) else None,
                   "reduce_ms": round(reduce_ms, 4),   # the collective alone (max over ranks), repeated on the finished frame after the timed region
                   "reduce_ms_in_timed_region": round((dt - max(x * args.steps * 1e-3 for x in render_ms)) * 1e3, 4),   # wall time after the slowest rank's render
                   "mrays_per_s": round(rays / dt / 1e6, 2), "mean_path_depth": round(depth / max(samples, 1), 3),
                   "scene_create_s": round(t_create, 3)},
    }

    if args.check_frame and rank == 0:
        # the reduced frame must equal a one-rank render of the same iterations (single-tap filter: bit for bit)
        ref = backend.RenderContext(sc, device=local)
        ref.render(args.warmup + args.steps)
        ref.waitForFinish()
        rxyz, rsmp, _ = ref.output()
        # (prgpu_reduce leaves the sums in root-side planes the downloads read; the torch.distributed fallback sums into the bound tensors)
        gxyz, gsmp = (ctx.output()[:2]) if comm is not None else (xyz.cpu().numpy(), smp.cpu().numpy())
        out["frame_check"] = {"xyz_equal": bool(np.array_equal(gxyz, rxyz)), "samples_equal": bool(np.array_equal(gsmp, rsmp))}
        ref.close()
    if not args.profile_only:
        out["roofline"] = roofline(ctx, rank, variant=variant, owned_fraction=own_samples / float(args.steps * width * height),
                                   pmc_workload=(width, height) == (W, H) and (args.workload == "c5" or args.triangles == NTRI))
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, what=what)
        if rank != 0:
            out.pop("roofline", None)
        if rank == 0 and world == 1 and args.workload == "c4" and not args.no_secondary and (width, height, args.triangles) == (W, H, NTRI):
            ctx.close()   # free the C4 scene's planes first
            out["secondary"] = secondary_c5(local)

    if args.rehearse_on_one_gpu:
        out["rehearsal"] = "all ranks on device 0, gloo collectives: not a result"
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
