"""CommitScene on the device against the host's SAH build on the 1 M-triangle interior: build times (first and repeated), tree statistics, and
the rendering rate through each tree (wavefront schedule, 4-wide nodes). Run on a GPU box: python profiles/device_build.py [spp]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth, scene as S
from hydracore3_amd.api import HipIntegrator
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = synth.interior_scene(1920, 1080, tex_size=256)
out = {}
for name, dev in (("host SAH", 0), ("device LBVH", 1)):
    os.environ["HPT_DEBUG_ACCEL"] = "1"
    g = HipIntegrator(sc)
    g.set_option("device_build", dev); g.set_option("refit", 0)
    t0 = time.time(); g.CommitScene(); t1 = time.time()
    first = g.commit_time()
    g.UpdateInstance(40, S.translate(0.3, 0.0, 0.1) @ np.asarray(sc.inst_matrices[40]))
    t2 = time.time(); g.CommitScene(); t3 = time.time()
    again = g.commit_time()
    g.UpdateInstance(40, np.asarray(sc.inst_matrices[40])); g.CommitScene()
    info = g.accel_info()
    frame = g.dev_array(np.zeros((1080, 1920, 4), np.float32))
    g.path_trace_block_dev(frame.ptr, 4)
    rates = []
    for _ in range(2):
        g.path_trace_block_dev(frame.ptr, spp); rates.append(1920 * 1080 * spp / g.last_kernel_ms() / 1e3)
    g.set_instrumentation(True); g.set_schedule(1); g.set_option("stats_wide", 1)
    probe = np.zeros((1080, 1920, 4), np.float32); g.InitRandomGens(g.N); g.PathTraceBlock(g.N, 4, probe, 2)
    c = g.counters()
    out[name] = {"first_commit_wall_ms": (t1 - t0) * 1e3, "first": first, "second_commit_wall_ms": (t3 - t2) * 1e3, "second": again, "sah_node_visits": info["sah_node_visits"],
                 "Mpaths_per_s": rates, "launch": g.last_launch(), "nodes_per_ray": c["nodes"] / c["rays"], "tris_per_ray": c["tris"] / c["rays"]}
    print(name, json.dumps(out[name]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/device_build.json", "w"), indent=1)
