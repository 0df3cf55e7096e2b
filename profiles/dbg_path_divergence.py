import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth, scene as S
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator
sc = synth.random_scene(231)
print("materials", [(i, int(m["mtype"]) if int(m["mtype"]) < 100 else "E", m["datai"][:2].tolist(), int(m["texid"][1]) != 0xFFFFFFFF) for i, m in enumerate(sc.materials)])
for depth in (1, 2, 3, 4, 5, 6):
    sc.trace_depth = depth
    for naive in (False, True):
        g, c = HipIntegrator(sc), OracleIntegrator(sc)
        a, b = g.render(1, naive=naive), c.render(1, naive=naive)
        gg, cg = g.random_gens(), c.random_gens()
        xy = g.packed_xy()[1273]; x, y = int(xy & 0xFFFF), int(xy >> 16)
        print(depth, "naive" if naive else "mis", "gens equal", bool((gg[1273] == cg[1273]).all()), "pixel", a[y, x, :3].tolist(), b[y, x, :3].tolist(), flush=True)
# does the divergence need the motion? freeze the scene
sc.trace_depth = 6; sc.inst_motion = {}
g, c = HipIntegrator(sc), OracleIntegrator(sc)
a, b = g.render(4), c.render(4)
print("frozen scene: gens equal", np.array_equal(g.random_gens(), c.random_gens()), "max diff", float(np.abs(a - b).max()))
