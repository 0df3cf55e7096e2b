"""Throughput of the full-material kernels (every BSDF branch, blends, bump, plastic, environment) on the typed_materials fixture,
megakernel vs wavefront schedule. Usage: python profiles/full_kernel.py [size] [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.api import HipIntegrator
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in (os.environ.get("SCENES", "typed_materials,legacy_materials,env_map,test_035").split(",")):
    sc = load_hydra_xml(os.path.join(root, "tests", "golden", "scenes", name, "statex_00001.xml"), size, size)
    for sched in ((1,) if sc.inst_motion else (1, 2)):      # moving instances: megakernel only
        g = HipIntegrator(sc, accel_layout=int(os.environ.get("LAYOUT", "0"))); g.set_schedule(sched)
        if os.environ.get("NODE_MIN"):
            g.set_option("node_min", int(os.environ["NODE_MIN"])); g.LoadScene(sc)
        if os.environ.get("BPC"):
            g.set_launch_config(int(os.environ["BPC"]))
        if os.environ.get("FORCE_FULL"):
            g.set_option("force_full_materials", 1)
        g.render(4)
        out = np.zeros((size, size, 4), np.float32)
        g.PathTraceBlock(g.N, 4, out, spp)
        t = g.GetExecutionTime("PathTraceBlock")
        print(f"{name:18s} schedule {sched}: {size * size * spp / (t[0] * 1e3):8.1f} Mpaths/s ({t[0]:.1f} ms)", flush=True)
