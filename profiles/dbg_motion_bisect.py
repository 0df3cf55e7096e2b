import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth, scene as S
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator
def cmp(tag, sc, naive=True, spp=1):
    g, c = HipIntegrator(sc), OracleIntegrator(sc)
    a, b = g.render(spp, naive=naive), c.render(spp, naive=naive)
    d = np.abs(a - b).reshape(-1, 4).max(axis=1)
    rel = d / np.maximum(np.abs(b).reshape(-1, 4).max(axis=1), 1e-3)
    print(f"{tag:40s} gens equal {np.array_equal(g.random_gens(), c.random_gens())}  px with rel diff > 1e-4: {int((rel > 1e-4).sum())}  max rel {rel.max():.2e}", flush=True)
for seed in (593, 567):
    print("seed", seed)
    cmp("original", synth.random_scene(seed))
    sc = synth.random_scene(seed)
    for m in sc.materials: m["texid"][1] = 0xFFFFFFFF
    cmp("no normal maps", sc)
    sc = synth.random_scene(seed); sc.set_environment((0.3, 0.3, 0.3))
    cmp("plain environment", sc)
    sc = synth.random_scene(seed)
    for i in list(sc.inst_motion): m0 = sc.inst_matrices[i]; sc.inst_motion[i] = S.translate(0.3, 0.2, 0.1) @ m0
    cmp("motion = translation only", sc)
    sc = synth.random_scene(seed)
    for i in list(sc.inst_motion): sc.inst_motion[i] = sc.inst_matrices[i].copy()
    cmp("motion matrix == matrix", sc)
    sc = synth.random_scene(seed)
    sc.materials = [S.material_lambert((0.6, 0.6, 0.6)) if int(m["mtype"]) != S.MAT_TYPE_LIGHT_SOURCE else m for m in sc.materials]
    cmp("all lambert", sc)
    sc = synth.random_scene(seed)
    sc.materials = [S.material_conductor(0.2, 3.9, 0.0, 0.0) if int(m["mtype"]) != S.MAT_TYPE_LIGHT_SOURCE else m for m in sc.materials]
    cmp("all mirrors", sc)
