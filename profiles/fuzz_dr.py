"""One-off wider check of PathTraceDR: random gltf-only scenes (synth.random_scene with every other material replaced by a gltf one, no normal
maps / environment map / motion), the scene's colour texture registered as the parameter, random parameters and reference image; loss and
gradient HIP vs oracle. Usage: python profiles/fuzz_dr.py [first] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth, scene as S
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for seed in range(first, first + count):
    sc = synth.random_scene(seed)
    r = np.random.RandomState(1000 + seed)
    mats = []
    for m in sc.materials:
        t = int(m["mtype"])
        if t in (S.MAT_TYPE_GLTF, S.MAT_TYPE_LIGHT_SOURCE):
            m = m.copy()
        else:
            m = S.material_gltf((*r.uniform(0.2, 0.9, 3), 1.0), float(r.choice([0.0, 1.0, r.uniform()])), float(r.choice([0.0, 1.0, r.uniform()])), float(r.choice([0.0, 1.0])), 1.5,
                                1 if r.uniform() < 0.6 else 0)
        m["texid"][1] = 0xFFFFFFFF
        mats.append(m)
    sc.materials = mats
    sc.set_environment(sc.env_color if sc.env_tex_id == 0xFFFFFFFF else (0.1, 0.1, 0.1))
    sc.lights = [l for l in sc.lights if int(l["geomType"]) != S.LIGHT_GEOM_ENV]
    for l in sc.lights:
        l["flags"] = int(l["flags"]) & ~2; l["texId"] = 0xFFFFFFFF
    sc.inst_motion = {}
    sc.lens_lines, sc.phys_size = np.zeros((0, 4), np.float32), (0.0, 0.0)      # (one fuzz scene in seven has a lens stack: not differentiated)
    g, c = HipIntegrator(sc), OracleIntegrator(sc)
    og, sg = g.PutDiffTex2D(1, 4, 4, 4)
    rc, oc, so = c.put_diff_tex2d(1, 4, 4, 4)
    data = r.uniform(0.2, 0.9, sg).astype(np.float32)
    ref = r.uniform(0.0, 0.5, (sc.height, sc.width, 4)).astype(np.float32)
    out_g, out_c, grad_g = np.zeros((sc.height, sc.width, 4), np.float32), np.zeros((sc.height, sc.width, 4), np.float32), np.zeros_like(data)
    spp = 4
    loss_g = g.PathTraceDR(g.N, 4, out_g, spp, ref, data, grad_g)
    loss_c, grad_c = c.path_trace_dr(out_c, spp, ref, data)
    ng = float(np.linalg.norm(grad_c))
    err = float(np.linalg.norm(grad_g - grad_c) / ng) if ng > 0 else float(np.abs(grad_g).max())
    lerr = abs(loss_g - loss_c) / max(abs(loss_c), 1e-12)
    gens = np.array_equal(g.random_gens(), c.random_gens())
    ok = err < 1e-2 and lerr < 1e-3 and gens
    bad += not ok
    print(("ok  " if ok else "MISS"), seed, f"loss {loss_c:.5f} rel {lerr:.1e}  |grad| {ng:.3e} rel err {err:.1e} gens {gens}", flush=True)
print(f"{count} seeds from {first}: {bad} misses")
