import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import scene as S, synth
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator

def build(which):
    sc = S.SceneData()
    sc.width, sc.height = 72, 48
    sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 1.8, 6.5), (0.0, 0.8, 0.0), (0.0, 1.0, 0.0)
    sc.fov, sc.trace_depth = 42.0, 5
    sc.env_color = (0.1, 0.12, 0.15, 0.0)
    chk = np.zeros((8, 8), np.uint32)
    for y in range(8):
        for x in range(8):
            chk[y, x] = 0xFFFFFFFF if (x + y) % 2 else 0xFF202020
    mask = sc.add_texture(S.Texture(chk, S.TEX_RGBA8, False, S.ADDR_WRAP, S.ADDR_WRAP, S.FILTER_LINEAR))
    M = sc.materials
    M.append(S.material_lambert((0.6, 0.6, 0.6)))
    M.append(S.material_gltf((0.8, 0.2, 0.2, 1.0), 0.0, 0.6, 1.0, 1.5))
    M.append(S.material_conductor(0.2, 3.9, 0.15, 0.15))
    M.append(S.material_diffuse((0.2, 0.3, 0.8), 0.5))
    M.append(S.material_glass((1, 1, 1), (0.9, 1.0, 0.9), 1.5))
    M.append(S.material_blend(1, 2, 0.5))
    M.append(S.material_blend(3, 1, 1.0, mask))
    M.append(S.material_blend(5, 6, 0.3))
    M.append(S.material_blend(7, 4, 0.4, mask))
    M.append(S.material_blend(0, 2, 0.0))
    p, n, t, uv, idx = synth._quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 4.0)
    sc.add_instance(sc.add_mesh(p, n, t, uv, idx, [0]), np.eye(4))
    sp = synth._sphere_mesh(2)
    ntri = sp[4].size // 3
    for i in range(6):
        gid = sc.add_mesh(sp[0], sp[1], sp[2], sp[3], sp[4], np.full(ntri, which, np.uint32))
        sc.add_instance(gid, S.translate(-3.0 + 1.2 * i, 0.6, -0.4 * (i % 2)) @ S.rotate_y(40.0 * i) @ S.scale(0.55, 0.55, 0.55))
    sc.lights.append(S.light_rect(S.translate(0.0, 4.0, 1.5), 1.0, 1.0, (1, 1, 1), 14.0))
    return sc

def l2(a, b, spp):
    d = (a[..., :3].astype(np.float64) - b[..., :3]) / spp
    return float(np.sqrt(np.mean(np.sum(d * d, -1))))


def diag(name, sc, depth, spp=1, naive=True):
    sc.trace_depth = depth
    g, c = HipIntegrator(sc), OracleIntegrator(sc)
    a, b = g.render(spp, naive=naive), c.render(spp, naive=naive)
    gg, cg = g.random_gens(), c.random_gens()
    bad = np.nonzero((gg != cg).any(axis=1))[0]
    a3, b3 = a.reshape(-1, 4), b.reshape(-1, 4)
    dif = np.nonzero(np.abs(a3 - b3).max(axis=1) > 1e-4 * spp)[0]
    print(f"{name} depth {depth} spp {spp} naive {naive}: L2 {l2(a, b, spp):.2e}; gens differ at {len(bad)} tids {bad[:8].tolist()}; colour differs at {len(dif)} px {dif[:8].tolist()}", flush=True)
    for k in list(dif[:3]):
        print("    px", int(k), "xy", int(k) % sc.width, int(k) // sc.width, "hip", a3[k].tolist(), "cpu", b3[k].tolist(), flush=True)
    for k in list(bad[:3]):
        print("    tid", int(k), "gens hip", gg[k].tolist(), "cpu", cg[k].tolist(), flush=True)


for name, which, mk in (("glass at id 4", 4, None), ("glass at id 9", 9, S.material_glass((1, 1, 1), (0.9, 1.0, 0.9), 1.5)), ("blend(1,4,0.5) at id 8", 8, S.material_blend(1, 4, 0.5)),
                        ("blend(4,4,0.5) at id 8", 8, S.material_blend(4, 4, 0.5)), ("blend(7,4,0.4,mask) at id 8", 8, None)):
    for depth in (1, 3, 5):
        for naive in (True, False):
            sc = build(which)
            if mk is not None: sc.materials[which] = mk
            diag(name, sc, depth, 4, naive)
