#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of Integrator::PathTraceBlock on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cornell|interior|dr|dr_interior|spectral|film|spectral_interior] [--spp S]

A *step* is one PathTraceBlock call over the whole frame (W*H pixels x spp passes, MIS path tracing) with the framebuffer,
RNG states and scene resident in HBM. N = 1 runs BASELINE.json configs[1] (scenes/test_035 Cornell box, 1024 x 1024, 1024 spp)
and appends ("also") the other configurations AT THEIR STATED SIZES: configs[2] (1M-triangle interior, 1920 x 1080 @ 1024 spp, wavefront
schedule, one step), configs[3] (PathTraceDR + Adam on the test_228 class, 512^2 @ 256 spp), configs[4] (1M triangles, 4096^2 x 4 albedo,
1920 x 1080 @ 512 spp, fwd + bwd + Adam, one step), the reference's spectral fixture, a thin-film fixture and the 1M-triangle interior under
spectral mode - each with its own roofline.

N > 1: one rank per GPU (the scene is replicated, no data-path collective but ONE RCCL reduce(SUM) of the framebuffer per step).
Started plainly (`python bench.py --gpus N`, WORLD_SIZE unset) the script spawns its N ranks itself through
`python -m torch.distributed.run` BEFORE touching the GPU; started by torch.distributed.run it reads RANK / WORLD_SIZE.

  --scaling strong (default)  FIXED TOTAL WORK: one frame at --spp.
        --shard pixels (default)   the north-star split: rank r renders every N-th 1024-tid chunk of the swizzled pixel order at the full
                                   --spp into a zeroed full-size frame; the reduce(SUM) assembles the frame, BIT-IDENTICAL to the single-GPU
                                   frame (the reference's result). A pixel's passes are sequential (its RNG stream continues from pass
                                   to pass), so small frames run out of independent paths per GPU ("multi-GPU").
        --shard samples            every rank renders ALL pixels with spp / N passes from its own RNG sub-streams (generators seeded as
                                   threads r*W*H.. of one big InitRandomGens call); the reduce adds the N partial frames: a statistically
                                   equivalent frame, not the reference's. Per-GPU parallelism stays at W*H pixels.
        The JSON line's `value` is the pixels split; the samples split is timed as well and reported under "also".
  --scaling weak              every rank renders the whole frame at the full --spp (N x the samples in the same time): per-GPU work fixed.

Sharded frames are verified on rank 0 after the timed region (at 8 spp; --no-verify skips it).

Prints ONE JSON line on rank 0 (contract in the task description). `roofline` names the ceiling that binds the dominant kernel: the
Cornell megakernel is VALU-issue bound (profiles/pmc_cornell.json), the wavefront trace kernel of the 1M-triangle scene is priced
against HBM both by ALGORITHMIC bytes and by the rocprofv3 counters. `cpu_baseline` times the CPU oracle (a restated
port, not the original binary) in a child process on the box's usable cores.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
SIMDS = 1024               # 256 CUs x 4 SIMDs
F_CLK_GHZ = 2.4            # max shader clock
# A wave64 VALU instruction issues over 2 cycles (32 lanes per cycle; MI355X_MICROARCH.md, execution model): 1228.8 G wave-instructions/s.
# What straight-line independent f32 code sustains in practice is about 3.1 cycles per instruction (profiles/probes/halfwave_probe.hip),
# so a frac of ~0.65 is where a VALU-only kernel tops out; the spec figure is the peak quoted.
VALU_PEAK_GINST = SIMDS * F_CLK_GHZ / 2.0


# ---- host CPU ---------------------------------------------------------------------------------------------------------------------------
def usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() is the whole machine)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(math.ceil(float(q) / float(p)))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def build_scene(workload, width, height):
    from hydracore3_amd.scene import load_hydra_xml
    if workload == "cornell":
        return load_hydra_xml(os.path.join(ROOT, "tests", "golden", "scenes", "test_035", "statex_00001.xml"), width, height)
    from hydracore3_amd.synth import interior_scene
    subdiv = int(os.environ.get("HYDRA_BENCH_SUBDIV", "4"))      # 4 = the 1M-triangle configuration; smaller values only for crossover studies
    return interior_scene(width, height, subdiv=subdiv)


def cpu_baseline_child(workload, width, height, target_s):
    """Runs in a child process (bench.py --cpu-baseline-only): the CPU oracle - a restated port of PathTraceBlock, OpenMP over pixels,
    schedule(dynamic, 64) - on a bounded sample of the same frame, twice, on the usable cores."""
    import numpy as np
    from oracle.orc import OracleIntegrator
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or usable_cores()
    sc = build_scene(workload, width, height)
    o = OracleIntegrator(sc, threads=cores)
    img = np.zeros((height, width, 4), np.float32)
    t0 = time.time(); o.path_trace_block(img, 1); t1 = time.time() - t0
    spp = int(max(1, min(64, 0.5 * target_s / max(t1, 1e-3))))
    runs = []
    for _ in range(2):
        t0 = time.time(); o.path_trace_block(img, spp); runs.append(time.time() - t0)
    rates = [width * height * spp / dt / 1e6 for dt in runs]
    out = {"value": min(rates), "unit": "Mpaths/s", "cores": cores, "kind": "port",
           "sample": f"{workload} {width}x{height} @ {spp} spp, twice ({runs[0]:.1f} s, {runs[1]:.1f} s: {rates[0]:.2f} / {rates[1]:.2f} Mpaths/s, "
                     f"{min(rates) / cores * 1e3:.1f} Kpaths/s per thread); oracle/liboracle.so = restated port of PathTraceBlock, OpenMP "
                     f"schedule(dynamic,64), OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND', 'unset')}, {cores} threads "
                     f"(affinity mask {os.environ.get('HYDRA_BENCH_AFFINITY', '?')} CPUs, cgroup quota applied, machine {os.cpu_count()}) on {cpu_model()}"}
    print("CPU_BASELINE " + json.dumps(out), flush=True)


def cpu_baseline(workload, width, height, target_s=20.0):
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="spread", OMP_PLACES="cores", HYDRA_BENCH_AFFINITY=str(len(os.sched_getaffinity(0))))
    for k in list(env):
        if k == "LD_PRELOAD" or k.startswith("ROCP") or k.startswith("HSA_TOOLS"):
            env.pop(k)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload, "--width", str(width), "--height", str(height),
                        "--cpu-seconds", str(target_s)], env=env, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    sys.stderr.write("[bench] cpu_baseline child failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    return None


# ---- roofline pieces -------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(counters, paths, spp):
    """bytes a path has to touch, from the traversal / shading counters of an instrumented launch."""
    c = counters
    trav = c["nodes"] * 64 + c["tris"] * 48 + c["instances_entered"] * 64
    surf = c["surface_hits"] * (3 * 4 + 3 * 32 + 4 + 8 + 64 + 320 + 4 * 4)
    nee = c["surface_hits"] * 320
    per_pixel = (8 + 8 + 4 + 16) * (paths / spp)
    total = trav + surf + nee + per_pixel
    return {"total": total / paths, "traversal": trav / paths, "nodes_per_ray": c["nodes"] / max(c["rays"], 1),
            "tris_per_ray": c["tris"] / max(c["rays"], 1), "rays_per_path": c["rays"] / paths}


def load_profile(name):
    f = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(f))
    except Exception:
        return None


def make_roofline(workload, integ, torch, dev, stream, N, W, H, spp, t_count, kernel_ms):
    """Counts rays / nodes / triangles with the instrumented megakernel on the same frame (fewer passes: the statistics are stationary) and
    prices the timed launch. Returns the `roofline` object."""
    import numpy as np
    sched_used, wf_rounds = integ.last_schedule()
    probe_spp = min(spp, 8)
    integ.set_schedule(1)                                     # the instrumented build is the megakernel: same rays, same triangle visits ...
    integ.set_option("stats_wide", 1 if integ.last_launch()["wide_nodes"] else 0)   # ... and the node walk of the tree the timed call used (heavy scenes: the 4-wide compressed one)
    integ.set_instrumentation(True)
    integ.InitRandomGens(N)
    integ.set_tid_interleave(0, 1)
    probe = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    integ.path_trace_block_dev(probe.data_ptr(), probe_spp, 0, N, 4, False, stream)
    torch.cuda.synchronize()
    cnt = integ.counters()
    integ.set_instrumentation(False)
    ab = algorithmic_bytes(cnt, float(N) * probe_spp, probe_spp)
    k_ms = float(np.mean(kernel_ms))
    paths = float(t_count) * spp
    hbm_alg = ab["total"] * paths / (k_ms * 1e-3) / 1e9
    extra = {"kernel": {1: "pathTraceKernel", 3: "pathTraceBlockKernel (megakernel, block-local ray repacking)"}.get(sched_used, f"wavefront: {wf_rounds} x (wfShadeKernel + wfTraceKernel)"),
             "kernel_ms": round(k_ms, 3), "algorithmic_bytes_per_path": round(ab["total"], 1), "traversal_bytes_per_path": round(ab["traversal"], 1),
             "nodes_per_ray": round(ab["nodes_per_ray"], 2), "tris_per_ray": round(ab["tris_per_ray"], 2), "rays_per_path": round(ab["rays_per_path"], 2)}
    tj = load_profile(f"traffic_{workload}.json")
    traffic, traffic_src = None, None
    if tj:
        traffic = tj["hbm_bytes_per_launch"] * paths / float(tj["paths_per_launch"])     # counter bytes per path x this launch's paths
        traffic_src = f"{tj.get('from')}, collected at commit {tj.get('commit', 'unrecorded (round 1)')}: per-path counter bytes rescaled to this launch, not measured in this run"
    pmc = load_profile(f"pmc_{workload}.json")
    if sched_used != 2 and pmc and pmc.get("valu_insts_per_path"):
        # VALU-issue roofline: wave-instructions the kernel issues (profiled count per path, a property of binary + scene) over the
        # LIVE kernel time, against 1024 SIMDs x f_clk / 2 cycles per wave64 instruction
        ginst = pmc["valu_insts_per_path"] * paths / (k_ms * 1e-3) / 1e9
        return dict({"bound": "valu", "achieved": round(ginst, 2), "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwaveinst/s",
                     "frac": round(ginst / VALU_PEAK_GINST, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "lane_utilisation": pmc.get("lane_utilisation"), "useful_lane_frac": round(ginst / VALU_PEAK_GINST * pmc.get("lane_utilisation", 0.0), 4),
                     "cycles_per_inst_per_simd": round(SIMDS * F_CLK_GHZ / ginst, 3),
                     "peak_note": "2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md); independent f32 mul / add / fma streams measure 3.1 - 3.7 on this chip "
                                  "(profiles/probes/valu_rate_probe.hip), i.e. a frac of 0.55 - 0.65 is the practical ceiling; VERDICT r1's 4-cycle pricing would read " + str(round(2.0 * ginst / VALU_PEAK_GINST, 3)),
                     "pmc_source": f"profiles/pmc_{workload}.json, collected at commit {pmc.get('commit')}: SQ_INSTS_VALU per path and SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); the kernel time is measured in this run",
                     "hbm_algorithmic": {"achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_alg / HBM_PEAK_GBS, 4),
                                         "note": "SURVEY 8d algorithmic bytes; served by L1/L2 on this scene, NOT an HBM claim"},
                     "hbm_counter_frac": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if traffic else None}, **extra)
    r = dict({"bound": "hbm", "achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_alg / HBM_PEAK_GBS, 4),
              "traffic": traffic, "traffic_source": traffic_src,
              "traversal_only_frac": round(ab["traversal"] * paths / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
              "hbm_counter_frac": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
              "note": "achieved / frac = ALGORITHMIC bytes (SURVEY 8d) over the live launch time: the top of the BVH is served by L2 / Infinity Cache, "
                      "so the counter fraction (FETCH_SIZE under the rule calibrated in profiles/r3_hbm_counter_probe.json - exact for divergent 64-B record fetches, "
                      "doubled for the shade kernel's coalesced streams - + WRITE_SIZE; Infinity-Cache hits are counted) is the memory-side figure. Random 64-B record "
                      "fetches reach 4.3 TB/s from HBM, 8.7 TB/s from the Infinity Cache and 13.6 TB/s from L2 (profiles/r3_quad_fetch_probe.log): the trace kernel runs at about 0.6 of "
                      "that for its hit mix - bound by dependent-fetch latency and instruction issue, not by bytes"}, **extra)
    if pmc:
        r["lane_utilisation"] = pmc.get("lane_utilisation"); r["pmc_source"] = f"profiles/pmc_{workload}.json, commit {pmc.get('commit')}"
    return r


# ---- forward workloads ------------------------------------------------------------------------------------------------------------------
def run_forward(args, workload, rank, world, dev, dist, backend, steps, warmup, spp, shard, with_roofline=True, warmup_spp=None):
    """Times `steps` PathTraceBlock calls; returns the result dict (rank 0) or None."""
    import numpy as np
    import torch
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd.sharding import tid_interleave
    W, H = (args.width or (1024 if workload == "cornell" else 1920)), (args.height or (1024 if workload == "cornell" else 1080))
    sc = build_scene(workload, W, H)
    integ = HipIntegrator(sc, device=dev.index, accel_layout=args.accel_layout)
    if args.blocks_per_cu:
        integ.set_launch_config(args.blocks_per_cu)
    if args.schedule or args.refill_below or args.trace_blocks_per_cu or args.groups:
        integ.set_schedule(args.schedule, args.refill_below, args.trace_blocks_per_cu, args.groups)
    if os.environ.get("HYDRA_BENCH_FORCE_FULL") == "1":          # kernel study: run the kernels with every BSDF branch on a gltf-only scene
        integ.set_option("force_full_materials", 1)
    N = W * H
    weak = args.scaling == "weak"
    share = world * args.emulate_share
    my_spp = spp
    if weak or shard == "samples":
        # sample sharding: rank r renders ALL pixels, its generators seeded as threads r*N .. (r+1)*N-1 of one big InitRandomGens call
        t_begin, t_count = 0, N
        if share > 1:
            integ.InitRandomGens(N, first_seed=rank * N)
        if not weak:
            my_spp = spp // share + (1 if rank < spp % share else 0)      # fixed total work: the frame's spp passes are dealt out over the ranks
    else:
        t_begin, t_count, chunk, stride = tid_interleave(rank, share, N)   # rank r renders every world-th 1024-tid chunk
        integ.set_tid_interleave(chunk, stride)

    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def reduce_frame(f):
        if dist is None:
            return
        if backend == "nccl":
            dist.reduce(f, dst=0, op=dist.ReduceOp.SUM)       # final RCCL reduce of the framebuffer over xGMI
        else:
            host = f.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                f.copy_(host)

    def step(n_spp=None):
        frame.zero_()
        n_spp = my_spp if n_spp is None else n_spp
        if n_spp > 0:
            integ.path_trace_block_dev(frame.data_ptr(), n_spp, t_begin, t_count, 4, False, stream)
        reduce_frame(frame)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(min(my_spp, warmup_spp) if warmup_spp else None)     # (secondary workloads warm up at fewer passes: allocations, first launches)
    if warmup_spp:
        integ.InitRandomGens(N, first_seed=(rank * N if (share > 1 and (weak or shard == "samples")) else 0))   # the timed frame starts from the seeds a fresh integrator has
    sync()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        if world == 1:
            kernel_ms.append(integ.last_kernel_ms())               # HIP events on the launch stream (syncs on the 2nd event)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kernel_ms = [integ.last_kernel_ms()]
    paths_per_step = float(N) * spp * (world if weak else 1) * (args.emulate_share if (not weak and args.emulate_share > 1) else 1) / (args.emulate_share if not weak else 1)
    paths_per_step = float(N) * spp * (world if weak else 1)      # (with --emulate-share K: the K ranks together would have finished the frame in this time)
    value = paths_per_step * steps / elapsed / 1e6
    mean_lum = float(frame[..., :3].mean().item()) / (spp * (world if weak else 1)) if rank == 0 else 0.0

    # ---- verification of the sharded frame (rank 0 re-renders the ranks' shares alone, at a few passes) ----
    verified = None
    if world > 1 and not args.no_verify and args.emulate_share == 1:
        vspp = min(8 * world, spp)
        vint = HipIntegrator(sc, device=dev.index, accel_layout=args.accel_layout)
        vf = torch.zeros_like(frame)
        mine = vspp
        if weak or shard == "samples":
            vint.InitRandomGens(N, first_seed=rank * N)
            if not weak:
                mine = vspp // world + (1 if rank < vspp % world else 0)
            if mine:
                vint.path_trace_block_dev(vf.data_ptr(), mine, 0, N, 4, False, stream)
        else:
            vint.set_tid_interleave(chunk, stride)
            vint.path_trace_block_dev(vf.data_ptr(), vspp, t_begin, t_count, 4, False, stream)
        reduce_frame(vf)
        torch.cuda.synchronize()
        if rank == 0:
            ref = torch.zeros_like(frame)
            if weak or shard == "samples":
                part = torch.zeros_like(frame)
                for r in range(world):
                    solo = HipIntegrator(sc, device=dev.index, accel_layout=args.accel_layout)
                    solo.InitRandomGens(N, first_seed=r * N)
                    n_r = vspp if weak else vspp // world + (1 if r < vspp % world else 0)
                    part.zero_()
                    if n_r:
                        solo.path_trace_block_dev(part.data_ptr(), n_r, 0, N, 4, False, stream)
                    torch.cuda.synchronize()
                    ref += part
                verified = bool(torch.allclose(ref, vf, rtol=1e-5, atol=1e-5))      # equal up to the reduce's summation order
            else:
                solo = HipIntegrator(sc, device=dev.index, accel_layout=args.accel_layout)
                solo.path_trace_block_dev(ref.data_ptr(), vspp, 0, N, 4, False, stream)
                torch.cuda.synchronize()
                verified = bool(torch.equal(ref, vf))                                 # disjoint pixels: bit-identical to the single-GPU frame
            if not verified:
                raise SystemExit(f"sharded frame ({shard}) differs from the single-GPU rendering of the same shares")
    roofline = None
    if rank == 0 and with_roofline:
        roofline = make_roofline(workload, integ, torch, dev, stream, N, W, H, my_spp if not weak else spp, t_count, kernel_ms)
    if rank != 0:
        return None
    if world == 1:
        sharding = "single GPU"
    elif weak:
        sharding = f"weak: {world} ranks x whole frame x {spp} spp each (decorrelated RNG sub-streams) + RCCL reduce(SUM)"
    elif shard == "samples":
        sharding = f"fixed work, sample sharding: {world} ranks x all pixels x {spp}/{world} spp (RNG sub-streams) + RCCL reduce(SUM)"
    else:
        sharding = (f"fixed work, pixel sharding: {world} ranks x interleaved 1024-tid chunks of one frame at {spp} spp + RCCL reduce(SUM); bit-identical to the 1-GPU frame. "
                    f"A pixel's passes are sequential (one RNG stream per pixel), so a rank has W*H/{world} independent chains: see DESIGN.md 5; the sample split is under 'also'")
    name = (f"scenes/test_035 Cornell box {W}x{H} @ {spp} spp, forward PathTraceBlock" if workload == "cornell"
            else f"synthetic 1M-triangle interior {W}x{H} @ {spp} spp, forward PathTraceBlock")
    return {"metric": "Mpaths/s (fwd PathTraceBlock, MIS path tracing)", "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "paths_per_step": int(paths_per_step), "trace_depth": sc.trace_depth, "integrator": "mispt", "sharding": sharding,
                       "mean_radiance": round(mean_lum, 5), "sharded_frame_verified": verified},
            "roofline": roofline}


# ---- differentiable rendering -----------------------------------------------------------------------------------------------------------
def dr_algorithmic_bytes(c, d, paths, spp, shade_records):
    """SURVEY.md 8d with the adjoint term, from the counters of an instrumented PathTraceDR launch (hpt_get_counters / hpt_get_dr_counters):
    traversal 64 B per node visit + 48 B per triangle test (+ 64 B per instance entered); per surface hit the shading data (one 64-byte
    shading record, or 12 B indices + 96 B vertices + 4 + 8 + 64 B through the index chain), the 320-byte Material, the 320-byte LightSource of
    the light sample and the parameter texture's 4 taps x 16 B; per pixel 36 B once per call; per sample the 16-byte reference pixel; the
    adjoint: 68-byte records written (a path's last one stays in registers) and read back by the sweep, and per record with a parameter
    texture 4 taps x 3 channels x 4 B x 2 of atomic read-modify-write."""
    trav = c["nodes"] * 64 + c["tris"] * 48 + c["instances_entered"] * 64
    surf = c["surface_hits"] * ((64 if shade_records else (12 + 96 + 4 + 8 + 64)) + 320 + 320) + d["records_with_taps"] * 4 * 16
    per_pixel = 36.0 * (paths / spp) + 16.0 * paths
    rec = (d["records_stored"] + max(d["sweep_bounces"] - (d["records"] - d["records_stored"]), 0)) * 68
    atom = d["records_with_taps"] * 4 * 3 * 4 * 2
    total = trav + surf + per_pixel + rec + atom
    return {"total": total / paths, "traversal": trav / paths, "surface": surf / paths, "records": rec / paths, "atomics": atom / paths,
            "nodes_per_ray": c["nodes"] / max(c["rays"], 1), "tris_per_ray": c["tris"] / max(c["rays"], 1), "rays_per_path": c["rays"] / paths,
            "records_per_path": d["records"] / paths}


def run_dr(args, workload, rank, world, dev, dist, backend, steps, warmup, spp_arg, warmup_spp=None):
    """IntegratorDR fwd+bwd + Adam.  dr: BASELINE.json configs[3] (test_228-class scene, 256 x 256 x 4 albedo, 512^2 @ 256 spp);
    dr_interior: configs[4] (1M-triangle interior, tex_size^2 x 4 fp32 albedo bound to its 32 gltf materials, 1920x1080).
    One step = memset(grad) + PathTraceDR (record, replay and adjoint fused, gradient atomics into HBM) [+ all_reduce(SUM) of the
    gradient and the loss over the ranks] + AdamOptimizer::step (every rank applies the identical step to its replica of a_data)."""
    import numpy as np
    import torch
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd.synth import dr_scene, interior_scene
    stream = torch.cuda.current_stream().cuda_stream
    big = workload == "dr_interior"
    if big:
        W, H = args.width or 1920, args.height or 1080
        spp = spp_arg if spp_arg != 1024 else 512                   # BASELINE configs[4]: fwd+bwd at 512 spp
        ts = int(os.environ.get("HYDRA_BENCH_TEX", "4096"))
        sc = interior_scene(W, H, tex_size=ts)                      # the generated texture is the checker the optimisation should recover
        tex_id, tw = 1, ts                                          # (texture 0 is the white dummy)
        tgt = sc
    else:
        xml = os.path.join(ROOT, "tests", "golden", "scenes", "test_228", "statex_00001.xml")
        W, H = args.width or 512, args.height or 512
        spp = spp_arg if spp_arg != 1024 else 256
        sc, tex_id = dr_scene(xml, W, H)
        tw = 256
        tgt, _ = dr_scene(xml, W, H, target=True)
    N = W * H
    weak = args.scaling == "weak"
    samples = weak or args.shard == "samples"
    # reference image: the same scene with the target (checker) albedo, a few passes on the GPU (identical on every rank)
    tgt_int = HipIntegrator(tgt, device=dev.index)
    tgt_int.set_schedule(1)                                        # (the plain megakernel: under a profiler its kernel name keeps the reference render apart from the timed PathTraceDR kernels)
    ref = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    tgt_int.path_trace_block_dev(ref.data_ptr(), 64, 0, N, 4, False, stream)
    torch.cuda.synchronize()
    ref = torch.flip(ref / 64.0, dims=[0]).contiguous()            # PixelLossPT reads the reference y-flipped (integrator_dr.cpp:1119)
    del tgt_int
    integ = HipIntegrator(sc, device=dev.index)
    integ.set_option("dr_skip_nonfinite", 1)                       # an optimisation loop: one NaN sample must not poison Adam's moments
    if args.blocks_per_cu:
        integ.set_launch_config(args.blocks_per_cu)
    if args.schedule:
        integ.set_schedule(args.schedule, args.refill_below, args.trace_blocks_per_cu, args.groups)
    off, size = integ.PutDiffTex2D(tex_id, tw, tw, 4)
    my_spp = spp
    if samples:
        t_begin, t_count = 0, N
        if world > 1:
            integ.InitRandomGens(N, first_seed=rank * N)
        if not weak:
            my_spp = spp // world + (1 if rank < spp % world else 0)
    else:
        from hydracore3_amd.sharding import tid_interleave
        t_begin, t_count, chunk, stride = tid_interleave(rank, world, N)
        integ.set_tid_interleave(chunk, stride)
    data = torch.full((size,), 0.5, dtype=torch.float32, device=dev)
    grad = torch.zeros_like(data); mom = torch.zeros_like(data); gsq = torch.zeros_like(data)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    L = integ.L
    losses = []

    def step(it, n_spp=None):
        grad.zero_(); loss.zero_(); frame.zero_()
        n_spp = my_spp if n_spp is None else n_spp
        if n_spp:
            integ._chk(L.hpt_path_trace_dr_dev(integ.h, t_begin, t_count, 4, frame.data_ptr(), n_spp, ref.data_ptr(), data.data_ptr(), grad.data_ptr(), size, loss.data_ptr(), stream))
            if samples and not weak and world > 1:
                loss.mul_(float(n_spp) / float(spp))               # PathTraceDR's loss is the sample mean over ITS passes: weight by the share
        if dist is not None:                                       # a_dataGrad: ncclAllReduce(sum), once per optimisation iteration
            if backend == "nccl":
                dist.all_reduce(grad, op=dist.ReduceOp.SUM); dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            else:
                g, l = grad.cpu(), loss.cpu()
                dist.all_reduce(g, op=dist.ReduceOp.SUM); dist.all_reduce(l, op=dist.ReduceOp.SUM)
                grad.copy_(g); loss.copy_(l)
            if weak:
                grad.div_(world); loss.div_(world)                 # mean over the ranks' independent sample sets
        integ._chk(L.hpt_adam_step_dev(integ.h, data.data_ptr(), grad.data_ptr(), mom.data_ptr(), gsq.data_ptr(), size, it, stream))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i, min(my_spp, warmup_spp) if warmup_spp else None)
    sync()
    kms = []
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
        if world == 1:
            kms.append(integ.last_kernel_ms())
        losses.append(float(loss.item()) / N)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kms = [integ.last_kernel_ms()]
    paths_per_step = N * spp * (world if weak else 1)
    value = float(paths_per_step) * steps / elapsed / 1e6
    if rank != 0:
        return None
    what = (f"synthetic 1M-triangle interior + {tw}x{tw}x4 differentiable albedo, {W}x{H} @ {spp} spp" if big
            else f"scenes/test_228 + 256x256x4 differentiable albedo, {W}x{H} @ {spp} spp")
    k_ms = float(np.mean(kms))
    sched_used, wf_rounds = integ.last_schedule()
    launch = integ.last_launch()
    wl = "dr_interior" if big else "dr"
    roof = {"kernel": {1: "pathTraceKernel<DR>", 3: "pathTraceBlockKernel<DR> (megakernel, block-local ray repacking)"}.get(sched_used, f"wavefront DR: {wf_rounds} x (wfShadeKernel<DR> + wfTraceKernel)"),
            "kernel_ms": round(k_ms, 3)}
    if world == 1:
        # the counting probe: the instrumented PathTraceDR megakernel on the same frame at a few passes (it walks the tree the timed call walked)
        probe_spp = min(spp, 8)
        integ.set_schedule(1); integ.set_option("stats_wide", 1 if launch["wide_nodes"] else 0); integ.set_instrumentation(True)
        integ.InitRandomGens(N)
        grad.zero_(); loss.zero_(); frame.zero_()
        integ._chk(L.hpt_path_trace_dr_dev(integ.h, 0, N, 4, frame.data_ptr(), probe_spp, ref.data_ptr(), data.data_ptr(), grad.data_ptr(), size, loss.data_ptr(), stream))
        torch.cuda.synchronize()
        cnt, dcnt = integ.counters(), integ.dr_counters()
        integ.set_instrumentation(False)
        ab = dr_algorithmic_bytes(cnt, dcnt, float(N) * probe_spp, probe_spp, launch["shade_records"])
        paths = float(N) * spp
        hbm_alg = ab["total"] * paths / (k_ms * 1e-3) / 1e9
        tj = load_profile(f"traffic_{wl}.json")
        traffic = tj["hbm_bytes_per_launch"] * paths / float(tj["paths_per_launch"]) if tj else None
        pmc = load_profile(f"pmc_{wl}.json")
        detail = {"algorithmic_bytes_per_path": round(ab["total"], 1), "traversal_bytes_per_path": round(ab["traversal"], 1), "surface_bytes_per_path": round(ab["surface"], 1),
                  "record_bytes_per_path": round(ab["records"], 1), "atomic_bytes_per_path": round(ab["atomics"], 1), "nodes_per_ray": round(ab["nodes_per_ray"], 2),
                  "tris_per_ray": round(ab["tris_per_ray"], 2), "rays_per_path": round(ab["rays_per_path"], 2), "records_per_path": round(ab["records_per_path"], 2),
                  "traffic": traffic, "traffic_over_algorithmic": round(traffic / (ab["total"] * paths), 3) if traffic else None,
                  "traffic_source": (f"{tj.get('from')}, collected at commit {tj.get('commit')}: per-path counter bytes ({tj.get('rule', 'FETCH_SIZE doubled + WRITE_SIZE')}) rescaled to this launch, not measured in this run" if tj else None),
                  "hbm_counter_frac": round(traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None}
        hbm = {"achieved": round(hbm_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_alg / HBM_PEAK_GBS, 4)}
        if sched_used == 2:
            # heavy scene: the wavefront DR pair is priced against HBM like the forward pair (algorithmic bytes incl. the adjoint term over the live time)
            roof.update(dict({"bound": "hbm"}, **hbm, **detail))
            roof["note"] = "achieved / frac = ALGORITHMIC bytes (SURVEY 8d incl. records and gradient atomics) over the live launch time; hbm_counter_frac is the HBM-side figure"
            if pmc:
                roof["lane_utilisation"] = pmc.get("lane_utilisation"); roof["pmc_source"] = f"profiles/pmc_{wl}.json, commit {pmc.get('commit')}"
        else:
            # light scene (the test_228 class lives in L2): VALU issue binds, as for the forward megakernel; the HBM model is carried beside it
            roof.update({"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwaveinst/s", "frac": None})
            if pmc and pmc.get("valu_insts_per_path"):
                ginst = pmc["valu_insts_per_path"] * paths / (k_ms * 1e-3) / 1e9
                roof.update({"achieved": round(ginst, 2), "frac": round(ginst / VALU_PEAK_GINST, 4), "lane_utilisation": pmc.get("lane_utilisation"),
                             "useful_lane_frac": round(ginst / VALU_PEAK_GINST * pmc.get("lane_utilisation", 0.0), 4),
                             "pmc_source": f"profiles/pmc_{wl}.json, collected at commit {pmc.get('commit')}: SQ_INSTS_VALU per path; the kernel time is measured in this run"})
            roof["hbm_algorithmic"] = dict(hbm, note="SURVEY 8d algorithmic bytes incl. the adjoint term; the scene and the 1 MB gradient live in L2, the records in L2 / Infinity Cache")
            roof.update(detail)
            roof["note"] = "forward walk + reverse sweep + gradient scatter in one kernel; float atomics execute at the memory side and are not VALU work"
    else:
        roof.update({"bound": "valu" if sched_used == 1 else "hbm", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "note": "priced on the single-GPU run only"})
    return {"metric": "Mpaths/s (fwd+bwd grad, IntegratorDR::PathTraceDR + Adam)", "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what + ", PathTraceDR fwd+bwd + Adam", "paths_per_step": int(paths_per_step), "trace_depth": sc.trace_depth, "grad_floats": int(size),
                       "sharding": "single GPU" if world == 1 else (("sample" if samples else "pixel") + f" sharding over {world} ranks ({args.scaling}) + all_reduce(SUM) of a_dataGrad and the loss"),
                       "loss_per_step": [round(v, 6) for v in losses]},
            "roofline": roof}


def run_fixture(dev, scene_name, spectral, steps=3, size=1024, spp=64, profile=None, warmup=1, warmup_spp=4):
    """An `also` entry for a fixture scene of the reference's feature branches (spectral rendering, thin films): `steps` PathTraceBlock calls on a
    device-resident frame, timed with HIP events around each launch (HipIntegrator.last_kernel_ms). `profile`: name of the committed PMC
    profile (profiles/pmc_<profile>.json, collected with `bench.py --workload <profile>` under rocprofv3) that prices the VALU roofline."""
    import numpy as np
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd.scene import load_hydra_xml
    W = H = size
    if scene_name == "interior":                                   # the 1 M-triangle interior of configs[2] under m_spectral_mode = 1 (gltf colours carried as four samples, a uniform spectrum in the table)
        W, H = 1920, 1080
        sc = build_scene("interior", W, H)
        sc.spectral_mode = 1
        sc.spec_offset_sz, sc.spec_values = [(0, 471)], np.ones(471, np.float32)
    else:
        sc = load_hydra_xml(os.path.join(ROOT, "tests", "golden", "scenes", scene_name, "statex_00001.xml"), size, size, spectral=spectral)
    integ = HipIntegrator(sc, device=dev.index)
    frame = integ.dev_array(np.zeros((H, W, 4), np.float32))
    for _ in range(warmup):
        integ.path_trace_block_dev(frame.ptr, warmup_spp)
    ms = []
    for _ in range(steps):
        integ.path_trace_block_dev(frame.ptr, spp)
        ms.append(integ.last_kernel_ms())
    mean_ms = sum(ms) / len(ms)
    paths = float(W) * H * spp
    roof = None
    pmc = load_profile(f"pmc_{profile}.json") if profile else None
    if scene_name == "interior":
        # a heavy scene: the same tree and statistically the same rays as the RGB interior, whose instrumented walk prices a path; the counters are this workload's own
        ab = (load_profile("r3i_summary.json") or {}).get("bench", {}).get("roofline", {}).get("algorithmic_bytes_per_path")
        tj = load_profile(f"traffic_{profile}.json") if profile else None
        traffic = tj["hbm_bytes_per_launch"] * paths / float(tj["paths_per_launch"]) if tj else None
        if ab:
            gbs = ab * paths / (mean_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "hbm_counter_frac": round(traffic / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "algorithmic_bytes_per_path": ab, "kernel": "wavefront: wfShadeSpecKernel + wfTraceKernel" if integ.last_launch()["schedule"] == 2 else f"schedule {integ.last_launch()['schedule']}",
                    "kernel_ms": round(mean_ms, 3),
                    "note": "algorithmic bytes per path from the RGB interior's instrumented walk (profiles/r3i_summary.json: same scene and tree, statistically the same rays); "
                            "traffic from this workload's own counter passes (profiles/traffic_spectral_interior.json), rescaled to this launch"}
    elif pmc and pmc.get("valu_insts_per_path"):
        # these scenes (16 K triangles, tables of a few hundred KB) live in L2: VALU issue binds, priced as for the Cornell megakernel
        ginst = pmc["valu_insts_per_path"] * paths / (mean_ms * 1e-3) / 1e9
        tj = load_profile(f"traffic_{profile}.json")
        traffic = tj["hbm_bytes_per_launch"] * paths / float(tj["paths_per_launch"]) if tj else None
        roof = {"bound": "valu", "achieved": round(ginst, 2), "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwaveinst/s", "frac": round(ginst / VALU_PEAK_GINST, 4),
                "traffic": traffic, "lane_utilisation": pmc.get("lane_utilisation"), "useful_lane_frac": round(ginst / VALU_PEAK_GINST * pmc.get("lane_utilisation", 0.0), 4),
                "kernel": (({3: "pathTraceBlockSpectralKernel (block-local schedule)", 2: "wavefront: wfShadeSpecKernel + wfTraceKernel"}.get(integ.last_launch()["schedule"], "pathTraceSpectralKernel")) if spectral
                           else "pathTraceKernel<MODE 4: thin films>"), "kernel_ms": round(mean_ms, 3),
                "hbm_counter_frac": round(traffic / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                "pmc_source": f"profiles/pmc_{profile}.json, collected at commit {pmc.get('commit')}: SQ_INSTS_VALU per path and lane utilisation; the kernel time is measured in this run"}
    return {"workload": (f"synthetic 1M-triangle interior {W}x{H}" if scene_name == "interior" else f"tests/golden/scenes/{scene_name} {size}x{size}") + f" @ {spp} spp, forward PathTraceBlock, " + ("m_spectral_mode = 1" if spectral else "RGB"),
            "metric": "Mpaths/s (fwd PathTraceBlock, MIS path tracing)", "value": round(paths / mean_ms / 1e3, 2), "unit": "Mpaths/s", "steps": steps,
            "ms_per_step": round(mean_ms, 3), "paths_per_step": int(paths), "sharding": "single GPU", "sharded_frame_verified": None, "roofline": roof}


FIXTURES = {"spectral": ("test_spectral", True), "film": ("thin_film", False), "spectral_interior": ("interior", True)}


def compact(r):
    """An `also` entry: the numbers of a secondary workload without the contract's boilerplate."""
    if r is None:
        return None
    return {"workload": r["config"]["workload"], "metric": r["metric"], "value": r["value"], "unit": r["unit"], "steps": r["steps"],
            "ms_per_step": r["ms_per_step"], "paths_per_step": r["config"]["paths_per_step"], "sharding": r["config"]["sharding"],
            "sharded_frame_verified": r["config"].get("sharded_frame_verified"), "roofline": r.get("roofline")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=["cornell", "interior", "dr", "dr_interior", "spectral", "film", "spectral_interior"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N = 1: skip the secondary workloads (configs[2], configs[3]) appended under \"also\"")
    ap.add_argument("--no-build", action="store_true", help="never compile (profiler runs: build first, `python __graft_entry__.py`); fail if the library is missing")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the check of the sharded frame against single-GPU renderings of the same shares")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--schedule", type=int, default=0, help="0 automatic, 1 persistent megakernel, 2 wavefront (shade + trace kernels), 3 megakernel with block-local ray repacking, 4 wavefront in one launch (block-owned slots)")
    ap.add_argument("--refill-below", type=int, default=0, help="wavefront: refill a trace wave when fewer lanes than this hold a ray")
    ap.add_argument("--trace-blocks-per-cu", type=int, default=0)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = fixed total work (one frame at --spp, split by --shard); weak = every rank renders the whole frame at --spp")
    ap.add_argument("--shard", default="pixels", choices=["pixels", "samples"],
                    help="how --scaling strong splits the frame: pixels = interleaved 1024-tid chunks at the full spp (the north-star split, bit-identical to "
                         "the single-GPU frame); samples = all pixels x spp/N passes per rank (RNG sub-streams: an equivalent frame, not the same one)")
    ap.add_argument("--emulate-share", type=int, default=1, help="study only: render rank 0's share of a K-rank job on one GPU (value = K x its rate: the K-GPU rate without the reduce)")
    ap.add_argument("--accel-layout", type=int, default=0, help="0 automatic, 1 two-level TLAS/BLAS, 2 single-level world-space BVH")
    ap.add_argument("--groups", type=int, default=0, help="wavefront: concurrent pixel groups (streams) per call, 0 = automatic")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_only:                                    # child process of cpu_baseline(): never touches the GPU
        W, H = (args.width or 1024), (args.height or 1024)
        cpu_baseline_child(args.workload if args.workload in ("cornell", "interior") else "cornell", W, H, args.cpu_seconds)
        return

    no_build = args.no_build or os.environ.get("HYDRA_BENCH_NO_BUILD") == "1"
    # ---- plain `python bench.py --gpus N`: spawn the N ranks BEFORE anything touches the GPU (never re-exec a GPU-initialised process) ----
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not no_build:
            import __graft_entry__ as g
            g.build()
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + ["--no-build"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    # ---- build and CPU baseline: before the GPU is initialised in this process --------------------------------------------------------
    import torch   # imported BEFORE libhydra_hip.so is loaded (build() loads it): the HIP runtime torch bundles must be the one the library binds to;
    #                importing torch does not initialise the GPU
    import __graft_entry__ as g
    if rank == 0 and not no_build:
        g.build()
    if no_build and not os.path.exists(g.LIB):
        raise SystemExit(f"{g.LIB} is missing and --no-build was given: run `python __graft_entry__.py` first")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("cornell", "interior"):
        W0 = args.width or (1024 if args.workload == "cornell" else 1920)
        H0 = args.height or (1024 if args.workload == "cornell" else 1080)
        cpu = cpu_baseline(args.workload, W0, H0)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # HYDRA_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the reduce is staged
    # through host memory); the real multi-GPU run uses "nccl", which is RCCL on ROCm
    backend = os.environ.get("HYDRA_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # HYDRA_BENCH_FORCE_DIST=1: initialise the process group even for one rank (exercises the RCCL code path on a one-GPU box)
    if world > 1 or os.environ.get("HYDRA_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        dist.barrier()

    if args.workload in FIXTURES:                                 # a feature-branch fixture on its own (the form the PMC profiles are collected on)
        name, spectral = FIXTURES[args.workload]
        fspp = args.spp if args.spp != 1024 else 64
        r = run_fixture(dev, name, spectral, steps=args.steps, spp=fspp, profile=args.workload, warmup=args.warmup, warmup_spp=fspp) if rank == 0 else None   # (equal launches: what profiles/summarize.py divides by)
        out = None
        if r is not None:
            out = {"metric": r["metric"], "value": r["value"], "unit": "Mpaths/s", "n_gpus": 1, "steps": r["steps"], "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                   "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": r["workload"], "paths_per_step": r["paths_per_step"], "sharding": "single GPU"},
                   "roofline": r["roofline"] or {"kernel_ms": r["ms_per_step"]}, "cpu_baseline": None}
    elif args.workload in ("dr", "dr_interior"):
        out = run_dr(args, args.workload, rank, world, dev, dist, backend, args.steps, args.warmup, args.spp)
        if out is not None:
            out["cpu_baseline"] = None
    else:
        out = run_forward(args, args.workload, rank, world, dev, dist, backend, args.steps, args.warmup, args.spp, args.shard)
        if out is not None:
            out["cpu_baseline"] = cpu
        also = []
        if world > 1 and args.scaling == "strong" and args.shard == "pixels" and not args.no_also:
            # the sample split of the same fixed work (an equivalent frame from other RNG sub-streams), timed beside the north-star split
            r = run_forward(args, args.workload, rank, world, dev, dist, backend, args.steps, args.warmup, args.spp, "samples", with_roofline=False)
            also.append(compact(r))
        if world == 1 and args.workload == "cornell" and not args.no_also and not (args.width or args.height):
            # BASELINE's other configurations at their stated sizes, timed by whoever runs this line (one long step each, warmed up at a few passes)
            r = run_forward(args, "interior", rank, world, dev, dist, backend, 1, 1, 1024, args.shard, warmup_spp=16)     # configs[2]: 1920 x 1080 @ 1024 spp
            also.append(compact(r))
            torch.cuda.empty_cache()
            r = run_dr(args, "dr", rank, world, dev, dist, backend, 3, 1, 1024)                                           # configs[3]: 512^2 @ 256 spp + Adam
            also.append(compact(r))
            r = run_dr(args, "dr_interior", rank, world, dev, dist, backend, 1, 1, 1024, warmup_spp=8)                    # configs[4]: 4096^2 x 4 albedo @ 512 spp + Adam
            also.append(compact(r))
            torch.cuda.empty_cache()
            # the reference's spectral fixture under m_spectral_mode = 1, and the thin-film fixture (RGB): short, kernel-timed
            also.append(run_fixture(dev, "test_spectral", True, profile="spectral"))
            also.append(run_fixture(dev, "thin_film", False, profile="film"))
            also.append(run_fixture(dev, "interior", True, steps=2, profile="spectral_interior", warmup_spp=8))   # a heavy scene under spectral mode: the wavefront schedule's spectral shade kernel
        if out is not None and also:
            out["also"] = [a for a in also if a is not None]

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
