"""One-off wider fuzz: seeds beyond the 12 the test suite runs, HIP vs oracle on random scenes (every material / light constructor,
blends, normal maps, plastic, environment maps, moving instances, projected textures), MIS and naive; prints the seeds that miss the bar. Usage: python profiles/fuzz_sweep.py [first] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = []
for seed in range(first, first + count):
    sc = synth.random_scene(seed)
    if os.environ.get("ONLY_MOTION") and not sc.inst_motion:
        continue
    for naive in (False, True):
        g, c = HipIntegrator(sc), OracleIntegrator(sc)
        if os.environ.get("ORC_DBG_NO_NORMAL_LERP") and sc.inst_motion:
            g.set_option("dbg_no_normal_lerp", 1)       # the same diagnostic switch on the device side
        a, b = g.render(4, naive=naive), c.render(4, naive=naive)
        d = (a[..., :3].astype(np.float64) - b[..., :3]) / 4
        l2 = float(np.sqrt(np.mean(np.sum(d * d, -1))))
        gens = np.array_equal(g.random_gens(), c.random_gens())
        fin = bool(np.isfinite(a).all()) == bool(np.isfinite(b).all())
        if l2 > 1e-3 or not gens or not fin:
            bad.append((seed, naive, l2, gens, fin))
            print("MISS", seed, naive, f"{l2:.2e}", gens, fin, flush=True)
print(f"{count} seeds from {first}: {len(bad)} misses", flush=True)
