#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of Integrator::PathTraceBlock on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cornell|interior] [--spp S]

A *step* is one PathTraceBlock call over the whole frame (W*H pixels x spp passes, MIS path tracing) with the
framebuffer, RNG states and scene resident in HBM. N = 1 runs BASELINE.json configs[1] (scenes/test_035 Cornell box,
1024 x 1024, 1024 spp). For N > 1 (launched through torch.distributed.run, one rank per GPU, scene replicated):

  --scaling weak (default)  sample sharding: every rank renders the WHOLE frame at --spp with its own RNG sub-streams
                            (generators seeded as threads r*W*H.. of one big InitRandomGens call) and ONE RCCL reduce(SUM)
                            per step adds the N frames on rank 0: N x the samples per pixel in (almost) the same time.
                            Per-GPU work is fixed, value = paths of all ranks / time.
  --scaling strong          pixel sharding of ONE frame at --spp: rank r renders every N-th 1024-tid chunk of the swizzled
                            pixel order into a zeroed full-size framebuffer, one RCCL reduce(SUM) assembles the frame,
                            bit-identical to the single-GPU frame (--verify). A pixel's passes are sequential (its RNG stream
                            continues from pass to pass), so this mode runs out of independent paths per GPU on small frames
                            ("multi-GPU").

Prints ONE JSON line on rank 0 (see the contract in the task description); `roofline` prices the persistent
path-tracing kernel against HBM bandwidth using ALGORITHMIC bytes measured by the library's
instrumented build on the same scene, `cpu_baseline` times the CPU oracle (a restated port, not the original binary)
on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch   # imported first: the HIP runtime torch bundles must be the one libhydra_hip.so binds to
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def build_scene(workload, width, height):
    from hydracore3_amd.scene import load_hydra_xml
    if workload == "cornell":
        return load_hydra_xml(os.path.join(ROOT, "tests", "golden", "scenes", "test_035", "statex_00001.xml"), width, height)
    from hydracore3_amd.synth import interior_scene
    subdiv = int(os.environ.get("HYDRA_BENCH_SUBDIV", "4"))      # 4 = the 1M-triangle configuration; smaller values only for crossover studies
    return interior_scene(width, height, subdiv=subdiv)


def run_dr(args, rank, world, dev, stream, dist=None, backend="nccl"):
    """IntegratorDR fwd+bwd + Adam.  --workload dr: BASELINE.json configs[3] (test_228-class scene, 256 x 256 x 4 albedo, 512^2 @ 256 spp);
    --workload dr_interior: configs[4] (1M-triangle interior, tex_size^2 x 4 fp32 albedo bound to its 32 gltf materials, 1920x1080).
    One step = memset(grad) + PathTraceDR (record, replay and adjoint fused, gradient atomics into HBM) [+ all_reduce(SUM) of the
    gradient and the loss over the ranks] + AdamOptimizer::step (every rank applies the identical step to its replica of a_data)."""
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd.synth import dr_scene, interior_scene
    big = args.workload == "dr_interior"
    if big:
        W, H = args.width or 1920, args.height or 1080
        spp = args.spp if args.spp != 1024 else 16
        ts = int(os.environ.get("HYDRA_BENCH_TEX", "4096"))
        sc = interior_scene(W, H, tex_size=ts)                      # the generated texture is the checker the optimisation should recover
        tex_id, tw = 1, ts                                          # (texture 0 is the white dummy)
        tgt = sc
    else:
        xml = os.path.join(ROOT, "tests", "golden", "scenes", "test_228", "statex_00001.xml")
        W, H = args.width or 512, args.height or 512
        spp = args.spp if args.spp != 1024 else 256
        sc, tex_id = dr_scene(xml, W, H)
        tw = 256
        tgt, _ = dr_scene(xml, W, H, target=True)
    N = W * H
    weak = args.scaling == "weak"
    # reference image: the same scene with the target (checker) albedo, a few passes on the GPU (identical on every rank)
    tgt_int = HipIntegrator(tgt, device=dev.index)
    ref = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    tgt_int.path_trace_block_dev(ref.data_ptr(), 64, 0, N, 4, False, stream)
    torch.cuda.synchronize()
    ref = torch.flip(ref / 64.0, dims=[0]).contiguous()            # PixelLossPT reads the reference y-flipped (integrator_dr.cpp:1119)
    del tgt_int
    integ = HipIntegrator(sc, device=dev.index)
    off, size = integ.PutDiffTex2D(tex_id, tw, tw, 4)
    if weak:
        t_begin, t_count = 0, N
        if world > 1:
            integ.InitRandomGens(N, first_seed=rank * N)
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

    def step(it):
        grad.zero_(); loss.zero_(); frame.zero_()
        integ._chk(L.hpt_path_trace_dr_dev(integ.h, t_begin, t_count, 4, frame.data_ptr(), spp, ref.data_ptr(), data.data_ptr(), grad.data_ptr(), size, loss.data_ptr(), stream))
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

    for i in range(args.warmup):
        step(i)
    sync()
    kms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
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
    value = float(N) * spp * args.steps * (world if weak else 1) / elapsed / 1e6
    if rank != 0:
        return
    what = (f"synthetic 1M-triangle interior + {tw}x{tw}x4 differentiable albedo, {W}x{H} @ {spp} spp" if big
            else f"scenes/test_228 + 256x256x4 differentiable albedo, {W}x{H} @ {spp} spp")
    out = {"metric": "Mpaths/s (fwd+bwd grad, IntegratorDR::PathTraceDR + Adam)", "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": what + ", PathTraceDR fwd+bwd + Adam",
                      "paths_per_step": N * spp * (world if weak else 1), "trace_depth": sc.trace_depth, "grad_floats": int(size),
                      "sharding": "single GPU" if world == 1 else (("sample" if weak else "pixel") + f" sharding over {world} ranks + all_reduce(SUM) of a_dataGrad and the loss"),
                      "loss_per_step": [round(v, 6) for v in losses]},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "kernel": "pathTraceKernel<DR>", "kernel_ms": round(float(np.mean(kms)), 3)},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)


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


def cpu_baseline(workload, width, height, target_s=12.0):
    """Time the CPU oracle (restated port of PathTraceBlock, OpenMP over pixels) on a bounded sample of the same frame."""
    from oracle.orc import OracleIntegrator
    sc = build_scene(workload, width, height)
    cores = os.cpu_count() or 1
    o = OracleIntegrator(sc, threads=cores)
    img = np.zeros((height, width, 4), np.float32)
    t0 = time.time()
    o.path_trace_block(img, 1)
    t1 = time.time() - t0
    spp = int(max(1, min(64, target_s / max(t1, 1e-3))))
    t0 = time.time()
    o.path_trace_block(img, spp)
    dt = time.time() - t0
    return {"value": width * height * spp / dt / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": f"{workload} {width}x{height} @ {spp} spp (oracle/liboracle.so, OpenMP, {cores} threads, {dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=["cornell", "interior", "dr", "dr_interior"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--schedule", type=int, default=0, help="0 automatic, 1 persistent megakernel, 2 wavefront (shade + trace kernels)")
    ap.add_argument("--refill-below", type=int, default=0, help="wavefront: refill a trace wave when fewer lanes than this hold a ray")
    ap.add_argument("--trace-blocks-per-cu", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = sample sharding (every rank renders the whole frame at --spp with its own RNG sub-streams, the frames are "
                         "summed by one RCCL reduce: N x the samples in the same time); strong = pixel sharding of ONE frame at --spp "
                         "(interleaved 1024-tid chunks, bit-identical to the single-GPU frame)")
    ap.add_argument("--emulate-share", type=int, default=1, help="study only: render rank 0's share of a K-rank job on one GPU (value = K x its rate: the K-GPU rate without the reduce)")
    ap.add_argument("--accel-layout", type=int, default=0, help="0 automatic, 1 two-level TLAS/BLAS, 2 single-level world-space BVH")
    ap.add_argument("--groups", type=int, default=0, help="wavefront: concurrent pixel groups (streams) per call, 0 = automatic")
    ap.add_argument("--verify", action="store_true", help="rank 0 re-renders the whole frame alone and checks the sharded frame is bit-identical")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
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

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if dist is not None:
        dist.barrier()
    from hydracore3_amd.api import HipIntegrator
    if args.workload in ("dr", "dr_interior"):
        run_dr(args, rank, world, dev, torch.cuda.current_stream().cuda_stream, dist, backend)
        if dist is not None:
            dist.destroy_process_group()
        return

    W, H = (args.width or (1024 if args.workload == "cornell" else 1920)), (args.height or (1024 if args.workload == "cornell" else 1080))
    spp = args.spp
    sc = build_scene(args.workload, W, H)
    integ = HipIntegrator(sc, device=dev_index, accel_layout=args.accel_layout)
    if args.blocks_per_cu:
        integ.set_launch_config(args.blocks_per_cu)
    if args.schedule or args.refill_below or args.trace_blocks_per_cu or args.groups:
        integ.set_schedule(args.schedule, args.refill_below, args.trace_blocks_per_cu, args.groups)
    if os.environ.get("HYDRA_BENCH_FORCE_FULL") == "1":          # kernel study: run the kernels with every BSDF branch on a gltf-only scene
        integ.set_option("force_full_materials", 1)
    N = W * H
    from hydracore3_amd.sharding import tid_interleave
    weak = args.scaling == "weak"
    if weak:
        # sample sharding: rank r renders ALL pixels, its generators seeded as threads r*N .. (r+1)*N-1 of one big InitRandomGens call
        t_begin, t_count = 0, N
        if world > 1:
            integ.InitRandomGens(N, first_seed=rank * N)
    else:
        t_begin, t_count, chunk, stride = tid_interleave(rank, world * args.emulate_share, N)   # rank r renders every world-th 1024-tid chunk
        integ.set_tid_interleave(chunk, stride)

    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        frame.zero_()
        integ.path_trace_block_dev(frame.data_ptr(), spp, t_begin, t_count, 4, False, stream)
        if dist is not None:
            if backend == "nccl":
                dist.reduce(frame, dst=0, op=dist.ReduceOp.SUM)   # final RCCL reduce of the framebuffer over xGMI
            else:
                host = frame.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    frame.copy_(host)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
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
    total_paths = float(N) * spp * args.steps * (world if weak else 1)     # (with --emulate-share K: the K ranks together would have finished the frame in this time)
    value = total_paths / elapsed / 1e6

    mean_lum = float(frame[..., :3].mean().item()) / (spp * (world if weak else 1)) if rank == 0 else 0.0

    verified = None
    if args.verify and rank == 0:
        integ.set_tid_interleave(0, 1)
        ref = torch.zeros_like(frame)
        if weak:
            # re-render every rank's contribution alone (same seeds, same number of calls) and sum: equal up to the reduce's summation order
            total = torch.zeros_like(frame)
            for r in range(world):
                solo = HipIntegrator(sc, device=dev_index, accel_layout=args.accel_layout)
                solo.InitRandomGens(N, first_seed=r * N)
                for _ in range(args.warmup + args.steps):
                    ref.zero_()
                    solo.path_trace_block_dev(ref.data_ptr(), spp, 0, N, 4, False, stream)
                torch.cuda.synchronize()
                total += ref
            verified = bool(torch.allclose(total, frame, rtol=1e-5, atol=1e-5))
        else:
            solo = HipIntegrator(sc, device=dev_index, accel_layout=args.accel_layout)
            for _ in range(args.warmup + args.steps):                 # the RNG streams continue from step to step
                ref.zero_()
                solo.path_trace_block_dev(ref.data_ptr(), spp, 0, N, 4, False, stream)
            torch.cuda.synchronize()
            verified = bool(torch.equal(ref, frame))
        if not verified:
            raise SystemExit("sharded frame differs from the single-GPU frame")

    roofline, cpu = None, None
    if rank == 0:
        # algorithmic bytes per path from the instrumented kernel on the same frame (fewer passes: the statistics are stationary)
        sched_used, wf_rounds = integ.last_schedule()
        probe_spp = min(spp, 8)
        integ.set_schedule(1)                                     # the instrumented build is the megakernel: same rays, same node / triangle visits
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
        paths_per_launch = float(t_count) * spp
        achieved = ab["total"] * paths_per_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj["hbm_bytes_per_launch"] * paths_per_launch / float(tj["paths_per_launch"])   # scaled to this launch's path count
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": "pathTraceKernel" if sched_used == 1 else f"wavefront: {wf_rounds} x (wfShadeKernel + wfTraceKernel)",
                    "kernel_ms": round(k_ms, 3),
                    "algorithmic_bytes_per_path": round(ab["total"], 1), "traversal_bytes_per_path": round(ab["traversal"], 1),
                    "nodes_per_ray": round(ab["nodes_per_ray"], 2), "tris_per_ray": round(ab["tris_per_ray"], 2),
                    "rays_per_path": round(ab["rays_per_path"], 2)}
        if args.workload == "cornell":
            roofline["note"] = ("36-triangle scene: the algorithmic bytes are served by L1/L2 (HBM traffic per launch is almost four orders below), so frac > 1 is "
                                "not an HBM claim; the kernel is VALU-bound: 97 % busy at 37 % lane utilisation (profiles/r1_measurements.md, pmc_mega.sh)")
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.workload, W, H)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        out = {"metric": "Mpaths/s (fwd PathTraceBlock, MIS path tracing)", "value": round(value, 2), "unit": "Mpaths/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"scenes/test_035 Cornell box {W}x{H} @ {spp} spp, forward PathTraceBlock" if args.workload == "cornell"
                          else f"synthetic 1M-triangle interior {W}x{H} @ {spp} spp, forward PathTraceBlock",
                          "paths_per_step": N * spp * (world if weak else 1), "trace_depth": sc.trace_depth, "integrator": "mispt",
                          "sharding": "single GPU" if world == 1 else
                                      (f"sample sharding: {world} ranks x whole frame x {spp} spp each (decorrelated RNG sub-streams) + RCCL reduce(SUM)" if weak
                                       else f"pixel sharding: {world} ranks x interleaved 1024-tid chunks of one frame + RCCL reduce(SUM)"),
                          "mean_radiance": round(mean_lum, 5), "sharded_frame_verified": verified},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
