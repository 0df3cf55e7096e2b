"""One-off wider fuzz under m_spectral_mode = 1: seeds beyond the ten the test suite runs, the spectral kernel vs the oracle on random scenes (every
material / light constructor, blends, normal maps, thin films, environment maps, moving instances, lens stacks), MIS and naive; prints the seeds
that miss the bar. Usage: python profiles/fuzz_spectral.py [first] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, diverged = [], 0
for seed in range(first, first + count):
    sc = synth.random_scene(seed, spectral=True)
    for naive in (False, True):
        g, c = HipIntegrator(sc), OracleIntegrator(sc)
        a, b = g.render(4, naive=naive), c.render(4, naive=naive)
        eq = np.all(g.random_gens() == c.random_gens(), axis=1)
        xy = c.packed_xy()
        ok = np.zeros(sc.width * sc.height, bool); ok[(xy >> 16).astype(np.int64) * sc.width + (xy & 0xFFFF).astype(np.int64)] = eq
        d = ((a[..., :3].astype(np.float64) - b[..., :3]) / 4).reshape(-1, 3)
        l2 = float(np.sqrt(np.mean(np.sum(d[ok] ** 2, -1))))
        scale = max(float(b[..., :3].mean() / 4), 1.0)
        fin = bool(np.isfinite(a).all()) == bool(np.isfinite(b).all())
        diverged += int((~eq).sum())
        if l2 > 1e-3 * scale or int((~eq).sum()) > 2 or not fin:
            bad.append((seed, naive, l2, int((~eq).sum()), fin))
            print("MISS", seed, naive, f"{l2:.2e}", int((~eq).sum()), fin, flush=True)
print(f"{count} seeds from {first}: {len(bad)} misses, {diverged} pixels with a diverged path in all", flush=True)
