"""Where a fuzz seed differs: the worst pixels of HIP vs oracle with both values. Usage: python profiles/dbg_fuzz_seed.py <seed> [naive]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator
seed = int(sys.argv[1]); naive = len(sys.argv) > 2 and sys.argv[2] == "naive"
sc = synth.random_scene(seed)
g, c = HipIntegrator(sc), OracleIntegrator(sc)
print("scene:", len(sc.materials), "materials, motion", bool(sc.inst_motion), "lens", len(sc.lens_lines), "layout", g.accel_info())
for spp in (1, 1, 1, 1):
    a, b = g.render(spp, naive=naive), c.render(spp, naive=naive)
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3]).sum(-1)
    d = np.where(np.isfinite(d), d, 1e30)
    idx = np.argsort(d.reshape(-1))[::-1][:3]
    print(f"pass: gens equal {np.array_equal(g.random_gens(), c.random_gens())}, finite {np.isfinite(a).all()} / {np.isfinite(b).all()}")
    for i in idx:
        y, x = divmod(int(i), a.shape[1])
        if d[y, x] > 0: print(f"   pixel ({x},{y}): hip {a[y, x, :3]}  oracle {b[y, x, :3]}  |diff| {d[y, x]:.3e}")
