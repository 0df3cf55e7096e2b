"""Traversal counters (nodes / triangles / instance visits per ray, wave iterations) of a fixture under the two-level and single-level
layouts, megakernel with every BSDF branch. Usage: python profiles/layout_counters.py <scene> [size] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.api import HipIntegrator
name = sys.argv[1] if len(sys.argv) > 1 else "legacy_materials"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = load_hydra_xml(os.path.join(root, "tests", "golden", "scenes", name, "statex_00001.xml"), size, size)
for layout in (1, 2):
    g = HipIntegrator(sc, accel_layout=layout); g.set_schedule(1); g.set_option("force_full_materials", 1)
    g.render(2)
    out = np.zeros((size, size, 4), np.float32)
    g.PathTraceBlock(g.N, 4, out, spp)
    t = g.GetExecutionTime("PathTraceBlock")[0]
    g.set_instrumentation(True)
    out[:] = 0
    g.PathTraceBlock(g.N, 4, out, spp)
    c = g.counters()
    rays = max(c.get("rays", 0) + c.get("shadow_rays", 0), 1)
    print(f"{name} layout {layout}: {size * size * spp / (t * 1e3):.1f} Mpaths/s; accel {g.accel_info()}")
    r = max(c["rays"] + c["shadow_rays"], 1)
    print(f"   schedule used {g.last_schedule()}; per ray: nodes {c['nodes'] / r:.2f}, tris {c['tris'] / r:.2f}, instances {c['instances_entered'] / r:.2f}; "
          f"rays per path {r / max(c['paths'], 1):.2f}; wave iterations per wave-ray: node {c['wave_node_iters'] * 64 / r:.2f}, tri {c['wave_tri_iters'] * 64 / r:.2f}")
    print("   counters:", c, flush=True)
