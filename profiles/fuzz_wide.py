"""Random heavy interiors (seed, object count, tessellation level, a stretched / far-away instance now and then): the 4-wide compressed tree
against the BVH2 walk of the same scene - ray queries, frames and generators under both schedules - before and after a device refit.
Run on a GPU box: python profiles/fuzz_wide.py [first_seed] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from hydracore3_amd import scene as S, synth
from hydracore3_amd.api import HipIntegrator

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    objects, subdiv = int(rng.integers(30, 205)), int(rng.integers(0, 3))
    sc = synth.interior_scene(128, 80, objects=objects, subdiv=subdiv, seed=seed + 7, tex_size=16)
    if seed % 3 == 0:                                                   # odd boxes: a stretched instance, one far from the rest
        i = int(rng.integers(2, objects))
        sc.inst_matrices[i] = S.translate(*rng.uniform(-2, 2, 3)) @ np.asarray(sc.inst_matrices[i]) @ S.scale(float(rng.uniform(3, 12)), 0.05, 1.0)
        j = int(rng.integers(2, objects))
        sc.inst_matrices[j] = S.translate(3.0e3, -1.0e3, 2.0e3) @ np.asarray(sc.inst_matrices[j])
    n = rng.normal(size=(20000, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    pos = np.zeros((20000, 4), np.float32); pos[:, :3] = rng.uniform(-5, 5, (20000, 3)); pos[:, 1] = np.abs(pos[:, 1]) * 0.4
    dr = np.zeros((20000, 4), np.float32); dr[:, :3] = n; dr[:, 3] = 3.4e38
    wide, narrow = HipIntegrator(sc), HipIntegrator(sc)
    narrow.set_option("wide_nodes", 0)
    info = wide.accel_info()
    ok = np.array_equal(wide.RayQuery_NearestHit(pos, dr).view(np.uint8), narrow.RayQuery_NearestHit(pos, dr).view(np.uint8))
    for sched in (1, 2):
        wide.set_schedule(sched); narrow.set_schedule(sched)
        ok = ok and np.array_equal(wide.render(2), narrow.render(2)) and np.array_equal(wide.random_gens(), narrow.random_gens())
    k = int(rng.integers(2, objects))
    m = S.translate(*rng.uniform(-1, 1, 3)) @ np.asarray(sc.inst_matrices[k]) @ S.rotate_y(float(rng.uniform(0, 180)))
    for g in (wide, narrow):
        g.UpdateInstance(k, m); g.CommitScene()
    ok = ok and wide.commit_time()["refitted"] and np.array_equal(wide.RayQuery_NearestHit(pos, dr).view(np.uint8), narrow.RayQuery_NearestHit(pos, dr).view(np.uint8))
    print(f"seed {seed}: {objects} objects, subdiv {subdiv}, {info['inst_tris']} triangles, sah {info['sah_node_visits']:.1f}: {'ok' if ok else 'DIFFERENT'}", flush=True)
    bad += 0 if ok else 1
print(f"{count} scenes, {bad} with a difference")
sys.exit(1 if bad else 0)
