"""Synthetic, deterministic scenes (no files needed): analytic test set-ups and the benchmark scenes of SURVEY.md 8d."""
from __future__ import annotations

import numpy as np

from . import scene as S


def _quad(p0, ex, ey, nu=1, nv=1, uv_scale=1.0):
    """Tessellated parallelogram p0 + s*ex + t*ey, s,t in [0,1]; returns pos4, norm4, tang4, uv, idx."""
    p0, ex, ey = (np.asarray(v, np.float64) for v in (p0, ex, ey))
    n = np.cross(ex, ey)
    n /= np.linalg.norm(n)
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="xy")
    pos = p0[None, None, :] + s[..., None] * ex + t[..., None] * ey
    nvtx = (nu + 1) * (nv + 1)
    pos4 = np.concatenate([pos.reshape(nvtx, 3), np.ones((nvtx, 1))], 1).astype(np.float32)
    norm4 = np.tile(np.append(n, 0.0), (nvtx, 1)).astype(np.float32)
    tang4 = np.tile(np.append(ex / np.linalg.norm(ex), 0.0), (nvtx, 1)).astype(np.float32)
    uv = (np.stack([s, t], -1).reshape(nvtx, 2) * uv_scale).astype(np.float32)
    idx = []
    for j in range(nv):
        for i in range(nu):
            a = j * (nu + 1) + i
            idx += [a, a + 1, a + nu + 2, a, a + nu + 2, a + nu + 1]
    return pos4, norm4, tang4, uv, np.asarray(idx, np.uint32)


def _merge(parts):
    """Concatenate (pos4, norm4, tang4, uv, idx, matids) pieces into one mesh."""
    pos, nrm, tng, uv, idx, mat, base = [], [], [], [], [], [], 0
    for p, n, t, u, i, m in parts:
        pos.append(p); nrm.append(n); tng.append(t); uv.append(u)
        idx.append(i + base)
        mat.append(np.full(i.size // 3, m, np.uint32) if np.isscalar(m) else np.asarray(m, np.uint32))
        base += p.shape[0]
    return (np.concatenate(pos), np.concatenate(nrm), np.concatenate(tng), np.concatenate(uv),
            np.concatenate(idx).astype(np.uint32), np.concatenate(mat))


def plane_under_rect_light(width=64, height=64, albedo=0.5, light_h=2.0, light_half=0.5, radiance=10.0) -> S.SceneData:
    """A Lambertian floor (y = 0) under a downward-facing square light: direct lighting has a closed form."""
    sc = S.SceneData()
    sc.width, sc.height = width, height
    sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 3.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)
    sc.fov, sc.trace_depth = 35.0, 1
    sc.materials.append(S.material_lambert((albedo, albedo, albedo)))
    p, n, t, uv, idx = _quad((-20, 0, 20), (40, 0, 0), (0, 0, -40))
    g = sc.add_mesh(p, n, t, uv, idx, [0])
    sc.add_instance(g, np.eye(4))
    sc.lights.append(S.light_rect(S.translate(0, light_h, 0), light_half, light_half, (1, 1, 1), radiance))
    return sc


def furnace_plane(width=32, height=32, albedo=(0.2, 0.5, 0.9), env=(1.0, 2.0, 0.5)) -> S.SceneData:
    """A Lambertian plane filling the view under a constant environment: radiance is exactly albedo * env."""
    sc = S.SceneData()
    sc.width, sc.height = width, height
    sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 2.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, -1.0)
    sc.fov, sc.trace_depth = 40.0, 2
    sc.env_color = (*env, 0.0)
    sc.materials.append(S.material_lambert(albedo))
    p, n, t, uv, idx = _quad((-50, 0, 50), (100, 0, 0), (0, 0, -100))
    sc.add_instance(sc.add_mesh(p, n, t, uv, idx, [0]), np.eye(4))
    return sc


def icosphere(subdiv):
    """Unit icosphere; returns (verts[n,3], tris[m,3])."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v), np.asarray(f, np.uint32)


def _sphere_mesh(subdiv):
    v, f = icosphere(subdiv)
    nv = v.shape[0]
    pos4 = np.concatenate([v, np.ones((nv, 1))], 1).astype(np.float32)
    norm4 = np.concatenate([v, np.zeros((nv, 1))], 1).astype(np.float32)
    tang = np.stack([-v[:, 2], np.zeros(nv), v[:, 0]], 1)
    ln = np.linalg.norm(tang, axis=1, keepdims=True)
    tang = np.where(ln > 1e-6, tang / np.maximum(ln, 1e-6), np.array([[1.0, 0.0, 0.0]]))
    tang4 = np.concatenate([tang, np.zeros((nv, 1))], 1).astype(np.float32)
    uv = np.stack([np.arctan2(v[:, 0], v[:, 2]) / (2 * np.pi) + 0.5, np.arccos(np.clip(-v[:, 1], -1, 1)) / np.pi], 1).astype(np.float32)
    return pos4, norm4, tang4, uv, f.reshape(-1)


def interior_scene(width=1920, height=1080, objects=204, subdiv=4, seed=12345, tex_size=1024) -> S.SceneData:
    """SURVEY.md 8d 'S2 interior-1M': a closed 10 x 4 x 10 room with a ceiling rect light, filled with `objects` distinct
    tessellated icosphere meshes (subdiv 4 = 5120 triangles each; 204 x 5120 = 1 044 480 triangles, every mesh its own
    BLAS, one instance each, so ~64 MB of nodes + 48 MB of triangles are really resident). 32 gltf materials (base colour
    U[0.2,0.8]^3, metalness in {0,1} with p = 0.2, glossiness U[0,1], coat 1, IOR 1.5), one Lambert wall material, one
    emissive material, one tex_size^2 RGBA32F albedo texture bound to every gltf material. Placement / materials are drawn
    from MT19937(seed)."""
    rng = np.random.RandomState(seed)        # MT19937
    sc = S.SceneData()
    sc.width, sc.height = width, height
    sc.fov, sc.trace_depth = 60.0, 6
    sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 1.7, 4.6), (0.0, 1.2, 0.0), (0.0, 1.0, 0.0)
    tex = np.full((tex_size, tex_size, 4), 0.5, np.float32)
    yy, xx = np.mgrid[0:tex_size, 0:tex_size]
    chk = (((xx // (tex_size // 16)) + (yy // (tex_size // 16))) % 2).astype(np.float32)
    tex[..., 0] = 0.35 + 0.5 * chk
    tex[..., 1] = 0.8 - 0.4 * chk
    tex[..., 2] = 0.6
    tex[..., 3] = 1.0
    tid = sc.add_texture(S.Texture(tex, S.TEX_RGBA32F, False, S.ADDR_WRAP, S.ADDR_WRAP, S.FILTER_LINEAR))
    nmat = 32
    for _ in range(nmat):
        col = rng.uniform(0.2, 0.8, 3)
        metal = 1.0 if rng.uniform() < 0.2 else 0.0
        gloss = rng.uniform(0.0, 1.0)
        sc.materials.append(S.material_gltf((*col, 1.0), metal, gloss, 1.0, 1.5, tid))
    wall = len(sc.materials)
    sc.materials.append(S.material_lambert((0.7, 0.7, 0.7)))
    emis = len(sc.materials)
    X, Y, Z = 5.0, 4.0, 5.0
    parts = [(*_quad((-X, 0, Z), (2 * X, 0, 0), (0, 0, -2 * Z), 8, 8), wall),          # floor (normal +y)
             (*_quad((-X, Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), 8, 8), wall),          # ceiling (normal -y)
             (*_quad((-X, 0, -Z), (2 * X, 0, 0), (0, Y, 0), 8, 8), wall),              # back wall (normal +z)
             (*_quad((X, 0, Z), (-2 * X, 0, 0), (0, Y, 0), 8, 8), wall),               # front wall (normal -z)
             (*_quad((-X, 0, Z), (0, 0, -2 * Z), (0, Y, 0), 8, 8), wall),              # left wall (normal +x)
             (*_quad((X, 0, -Z), (0, 0, 2 * Z), (0, Y, 0), 8, 8), wall)]               # right wall (normal -x)
    meshes = [_merge(parts)]
    insts = [(0, np.eye(4), -1, -1)]
    lm = S.translate(0.0, Y - 0.01, 0.0)
    light_id = len(sc.lights)
    sc.lights.append(S.light_rect(lm, 1.0, 1.0, (1, 1, 1), 25.0))
    sc.materials.append(S.material_emissive((1, 1, 1), 25.0, light_id))
    sc.lights[light_id]["matId"] = emis
    lp, ln, lt, luv, lidx = _quad((-1, 0, -1), (2, 0, 0), (0, 0, 2))                    # normal -y
    meshes.append((lp, ln, lt, luv, lidx, np.full(2, emis, np.uint32)))
    insts.append((1, lm, -1, light_id))
    sp = _sphere_mesh(subdiv)
    ntri = sp[4].size // 3
    gx = int(np.ceil(np.sqrt(objects)))
    placed = 0
    for j in range(gx):
        for i in range(gx):
            if placed >= objects:
                break
            # every object is its own mesh: the unit sphere with a per-object radial bump pattern
            bump = 1.0 + 0.08 * np.sin(sp[0][:, 0:1] * rng.uniform(3, 9) + rng.uniform(0, 6.28)) * np.cos(sp[0][:, 1:2] * rng.uniform(3, 9))
            pos = sp[0].copy()
            pos[:, :3] *= bump.astype(np.float32)
            meshes.append((pos, sp[1], sp[2], sp[3], sp[4], np.full(ntri, rng.randint(0, nmat), np.uint32)))
            r = rng.uniform(0.18, 0.30)
            cx = -X + 0.5 + (2 * X - 1.0) * (i + 0.5) / gx + rng.uniform(-0.1, 0.1)
            cz = -Z + 0.5 + (2 * Z - 1.0) * (j + 0.5) / gx + rng.uniform(-0.1, 0.1)
            cy = r + rng.uniform(0.0, 2.2)
            m = S.translate(cx, cy, cz) @ S.rotate_y(rng.uniform(0, 360)) @ S.scale(r, r * rng.uniform(0.7, 1.3), r)
            insts.append((len(meshes) - 1, m, -1, -1))
            placed += 1
    sc.add_meshes(meshes)
    for g, m, rl, li in insts:
        sc.add_instance(g, m, rl, li)
    return sc


def material_zoo(width=96, height=64) -> S.SceneData:
    """Every material / light type of the hot path in one frame: gltf (Lambert, rough metal, coated plastic, mirror),
    Lambert / Oren-Nayar `diffuse`, smooth and rough conductor, smooth dielectric sphere; lit by a rect light with a
    mesh, a sphere light, a spot light, a directional light and a disc light, under a dim constant environment."""
    sc = S.SceneData()
    sc.width, sc.height = width, height
    sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 2.2, 7.5), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)
    sc.fov, sc.trace_depth = 45.0, 6
    sc.env_color = (0.03, 0.04, 0.06, 0.0)
    chk = np.zeros((8, 8), np.uint32)
    for y in range(8):
        for x in range(8):
            chk[y, x] = 0xFFE0E0E0 if (x + y) % 2 else 0xFF3050B0
    tex = sc.add_texture(S.Texture(chk, S.TEX_RGBA8, True, S.ADDR_WRAP, S.ADDR_CLAMP, S.FILTER_LINEAR))
    rough = np.zeros((4, 4, 4), np.float32)
    rough[..., 0] = np.linspace(0.3, 1.0, 16).reshape(4, 4)
    rough[..., 1] = 0.5
    rtex = sc.add_texture(S.Texture(rough, S.TEX_RGBA32F, False))
    M = sc.materials
    M.append(S.material_lambert((0.7, 0.7, 0.7), tex))                                   # 0 floor (textured)
    M.append(S.material_gltf((0.9, 0.6, 0.2, 1.0), 1.0, 0.6, 0.0, 1.5))                  # 1 rough metal
    M.append(S.material_gltf((0.2, 0.7, 0.3, 1.0), 0.0, 0.8, 1.0, 1.5))                  # 2 coated plastic
    M.append(S.material_gltf((0.9, 0.9, 0.9, 1.0), 1.0, 1.0, 0.0, 1.5))                  # 3 mirror (glossiness 1)
    M.append(S.material_diffuse((0.8, 0.3, 0.3), 0.0))                                    # 4 diffuse Lambert
    M.append(S.material_diffuse((0.3, 0.3, 0.8), 0.7))                                    # 5 diffuse Oren-Nayar
    M.append(S.material_conductor(0.2, 3.9, 0.0, 0.0))                                    # 6 smooth conductor
    M.append(S.material_conductor(1.1, 2.3, 0.25, 0.1, (1.0, 0.85, 0.6, 1.0)))            # 7 rough anisotropic conductor
    M.append(S.material_dielectric(1.5, 1.0))                                             # 8 glass
    g = S.material_gltf((0.8, 0.8, 0.8, 1.0), 0.5, 1.0, 1.0, 1.5)                         # 9 gltf with glossiness / metalness textures
    g["cflags"] |= 256
    g["texid"][2] = rtex
    g["texid"][3] = rtex
    M.append(g)
    emis = len(M)
    parts = [(*_quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 4.0), 0),
             (*_quad((-8, 0, -8), (16, 0, 0), (0, 6, 0)), 4)]
    sc.add_instance(sc.add_mesh(*_merge(parts)), np.eye(4))
    sp = _sphere_mesh(2)
    ntri = sp[4].size // 3
    for i, mat in enumerate((1, 2, 3, 5, 6, 7, 8, 9)):
        gid = sc.add_mesh(sp[0], sp[1], sp[2], sp[3], sp[4], np.full(ntri, mat, np.uint32))
        x = -4.2 + 1.2 * i
        sc.add_instance(gid, S.translate(x, 0.55 + 0.25 * (i % 2), -0.5 * (i % 3)) @ S.rotate_y(30.0 * i) @ S.scale(0.5, 0.5 + 0.1 * (i % 2), 0.5))
    # lights
    lm = S.translate(-2.0, 4.0, 0.0) @ S.rotate_x(15.0)
    L = sc.lights
    L.append(S.light_rect(lm, 0.8, 0.5, (1.0, 0.95, 0.9), 12.0))
    M.append(S.material_emissive((1.0, 0.95, 0.9), 12.0, 0))
    L[0]["matId"] = emis
    lp, ln, lt, luv, lidx = _quad((-0.5, 0, -0.8), (1.0, 0, 0), (0, 0, 1.6))
    sc.add_instance(sc.add_mesh(lp, ln, lt, luv, lidx, [emis]), lm, -1, 0)
    L.append(S.light_sphere(S.translate(2.5, 2.5, 1.0), 0.3, (0.6, 0.8, 1.0), 20.0))
    L.append(S.light_point(S.translate(0.0, 3.5, 3.0) @ S.rotate_x(-25.0), (1, 1, 1), 30.0, "spot",
                           float(np.cos(np.radians(15.0))), float(np.cos(np.radians(30.0)))))
    L.append(S.light_directional(S.rotate_x(20.0) @ S.rotate_y(40.0), (1.0, 0.9, 0.8), 0.5))
    L.append(S.light_rect(S.translate(3.5, 3.0, -3.0), 0.0, 0.0, (0.9, 0.4, 0.4), 15.0, disk_radius=0.6))
    L.append(S.light_point(S.translate(-3.5, 1.5, 2.0), (0.5, 1.0, 0.5), 6.0, "omni"))
    return sc


def dr_scene(xml_path, width=512, height=512, tex_size=256, target=False) -> S.SceneData:
    """SURVEY.md 8d 'S3 dr-228': scenes/test_228 (two 4096-triangle spheres in a box, point light) with matGray bound to a
    tex_size^2 x 4 differentiable albedo texture (drmain.cpp:185's PutDiffTex2D(1, 256, 256, 4) shape; the scene as shipped has
    no such texture, so the binding is added here). `target=True` fills the texture with the checker the optimisation should
    recover; otherwise it is the 0.5 grey starting point. Returns (scene, texId)."""
    sc = S.load_hydra_xml(xml_path, width, height)
    tex = np.full((tex_size, tex_size, 4), 0.5, np.float32)
    if target:
        yy, xx = np.mgrid[0:tex_size, 0:tex_size]
        chk = (((xx // (tex_size // 8)) + (yy // (tex_size // 8))) % 2).astype(np.float32)
        tex[..., 0] = 0.2 + 0.6 * chk
        tex[..., 1] = 0.8 - 0.5 * chk
        tex[..., 2] = 0.3 + 0.3 * chk
    tex[..., 3] = 1.0
    tid = sc.add_texture(S.Texture(tex, S.TEX_RGBA32F, False, S.ADDR_WRAP, S.ADDR_WRAP, S.FILTER_LINEAR))
    sc.materials[0]["texid"][0] = tid
    sc.materials[0]["colors"][0] = (1.0, 1.0, 1.0, 0.0)      # albedo comes from the texture
    return sc, tid


def random_scene(seed: int, width=48, height=32, spectral=False) -> S.SceneData:
    """Seeded random small scene for fuzz parity: a floor + back wall, 3..7 transformed spheres with materials drawn from every
    constructor of the hot path with parameters that include the corners (metalness 0 / 1, glossiness 0 / 1, coat 0 / 1, smooth and
    rough conductors, Oren-Nayar, glass, plastic, blends, normal maps), an optional textured material, 1..3 lights of random types, random environment, random depth."""
    r = np.random.RandomState(seed)
    sc = S.SceneData()
    sc.width, sc.height = width, height
    if spectral:                                                      # the same scene under m_spectral_mode = 1: colours carried as four samples, the loader's uniform spectrum
        sc.spectral_mode = 1
        sc.spec_offset_sz, sc.spec_values = [(0, 471)], np.ones(471, np.float32)
    sc.cam_pos = (float(r.uniform(-1, 1)), float(r.uniform(1.5, 3.0)), float(r.uniform(6.0, 8.0)))
    sc.cam_look_at, sc.cam_up = (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)
    sc.fov, sc.trace_depth = float(r.uniform(35, 60)), int(r.choice([1, 3, 5, 6, 6]))
    sc.env_color = (*[float(v) for v in r.uniform(0.0, 0.2, 3)], 0.0) if r.uniform() < 0.5 else (0.0, 0.0, 0.0, 0.0)
    img = r.randint(0, 2 ** 32, (4, 4), dtype=np.uint64).astype(np.uint32) | np.uint32(0xFF000000)
    tex = sc.add_texture(S.Texture(img, S.TEX_RGBA8, bool(r.randint(2)), int(r.choice([S.ADDR_WRAP, S.ADDR_CLAMP])), S.ADDR_WRAP,
                                   int(r.choice([S.FILTER_NEAREST, S.FILTER_LINEAR]))))
    corner = lambda: float(r.choice([0.0, 1.0, r.uniform(0, 1)]))
    col = lambda: tuple(float(v) for v in r.uniform(0.1, 0.95, 3))

    def rand_material():
        k = r.randint(8)
        if k == 7:
            return sc.material_plastic(col(), float(r.choice([0.0, 0.05, r.uniform(0.05, 0.5)])), float(r.uniform(1.3, 1.8)), 1.0, int(r.randint(2)),
                                       tex if r.uniform() < 0.3 else 0)
        if k == 6:
            return S.material_glass(col(), col(), float(r.uniform(1.2, 2.2)))
        if k == 0:
            return S.material_lambert(col(), tex if r.uniform() < 0.3 else 0)
        if k == 1:
            return S.material_gltf((*col(), 1.0), corner(), corner(), corner(), float(r.choice([1.1, 1.33, 1.5, 2.0])), tex if r.uniform() < 0.3 else 0)
        if k == 2:
            return S.material_diffuse(col(), float(r.choice([0.0, r.uniform(0.1, 1.0)])), tex if r.uniform() < 0.3 else 0)
        if k == 3:
            a = float(r.choice([0.0, r.uniform(0.05, 0.5)]))
            return S.material_conductor(float(r.uniform(0.1, 2.0)), float(r.uniform(1.0, 4.0)), a, float(r.choice([a, r.uniform(0.05, 0.5)])), (*col(), 1.0))
        if k == 4:
            return S.material_dielectric(float(r.uniform(1.2, 2.0)), 1.0)
        return S.material_gltf((*col(), 1.0), 0.0, corner(), 1.0, 1.5)
    M = sc.materials
    M.append(S.material_lambert((0.6, 0.6, 0.6), tex))
    M.append(S.material_diffuse((0.5, 0.5, 0.55), 0.3))
    nobj = int(r.randint(3, 8))
    for _ in range(nobj):
        if len(M) >= 4 and r.uniform() < 0.25:                        # a blend of two earlier materials (possibly blends themselves), sometimes masked
            a, b = int(r.randint(2, len(M))), int(r.randint(2, len(M)))
            M.append(S.material_blend(a, b, corner(), tex if r.uniform() < 0.5 else 0))
        else:
            M.append(rand_material())
    parts = [(*_quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 3.0), 0), (*_quad((-8, 0, -6), (16, 0, 0), (0, 6, 0)), 1)]
    sc.add_instance(sc.add_mesh(*_merge(parts)), np.eye(4))
    sp = _sphere_mesh(int(r.randint(1, 3)))
    ntri = sp[4].size // 3
    for i in range(nobj):
        gid = sc.add_mesh(sp[0], sp[1], sp[2], sp[3], sp[4], np.full(ntri, 2 + i, np.uint32))
        m = S.translate(float(r.uniform(-3.5, 3.5)), float(r.uniform(0.4, 1.6)), float(r.uniform(-3.0, 2.5))) @ S.rotate_y(float(r.uniform(0, 360))) @ \
            S.rotate_x(float(r.uniform(0, 90))) @ S.scale(*[float(v) for v in r.uniform(0.3, 0.8, 3)])
        sc.add_instance(gid, m)
    L = sc.lights
    for _ in range(int(r.randint(1, 4))):
        k = r.randint(6)
        pos = S.translate(float(r.uniform(-3, 3)), float(r.uniform(2.5, 4.5)), float(r.uniform(-2, 3)))
        c, mult = col(), float(r.uniform(5, 25))
        if k == 0:
            lm = pos @ S.rotate_x(float(r.uniform(-20, 20)))
            L.append(S.light_rect(lm, float(r.uniform(0.3, 0.9)), float(r.uniform(0.3, 0.9)), c, mult))
            if r.uniform() < 0.6:                                   # visible emitter mesh bound to the light
                e = len(M); M.append(S.material_emissive(c, mult, len(L) - 1)); L[-1]["matId"] = e
                hl, hw = float(L[-1]["size"][0]), float(L[-1]["size"][1])
                lp, ln, lt, luv, lidx = _quad((-hw, 0, -hl), (2 * hw, 0, 0), (0, 0, 2 * hl))
                sc.add_instance(sc.add_mesh(lp, ln, lt, luv, lidx, [e]), lm, -1, len(L) - 1)
        elif k == 1:
            L.append(S.light_sphere(pos, float(r.uniform(0.15, 0.5)), c, mult))
        elif k == 2:
            a1 = float(r.uniform(10, 25)); a2 = a1 + float(r.uniform(5, 20))
            L.append(S.light_point(pos @ S.rotate_x(float(r.uniform(-30, 30))), c, mult * 2, "spot", float(np.cos(np.radians(a1))), float(np.cos(np.radians(a2)))))
        elif k == 3:
            L.append(S.light_directional(S.rotate_x(float(r.uniform(-40, 40))) @ S.rotate_y(float(r.uniform(0, 360))), c, float(r.uniform(0.3, 1.5))))
        elif k == 4:
            L.append(S.light_rect(pos, 0.0, 0.0, c, mult, disk_radius=float(r.uniform(0.3, 0.8))))
        else:
            L.append(S.light_point(pos, c, mult, "omni"))
    # normal maps on about a quarter of the surface materials (drawn last: the scenes of the earlier seeds keep their geometry and lights)
    ny, nx = np.mgrid[0:8, 0:8]
    dx, dy = 0.45 * np.sin(nx * np.pi / 2.0 + float(r.uniform(0, 3))), 0.45 * np.cos(ny * np.pi / 2.0 + float(r.uniform(0, 3)))
    enc = lambda v: np.clip(np.rint((v * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint32)
    nz = np.clip(np.rint(np.sqrt(1.0 - dx * dx - dy * dy) * 255.0), 0, 255).astype(np.uint32)
    nmap = sc.add_texture(S.Texture(enc(dx) | (enc(dy) << 8) | (nz << 16) | np.uint32(0xFF000000), S.TEX_RGBA8, False, S.ADDR_WRAP, S.ADDR_WRAP,
                                    int(r.choice([S.FILTER_NEAREST, S.FILTER_LINEAR]))))
    for m in M:
        if int(m["mtype"]) not in (S.MAT_TYPE_BLEND, S.MAT_TYPE_LIGHT_SOURCE) and r.uniform() < 0.25:
            k = float(r.choice([1.0, 2.0, 5.0]))
            S.set_normal_map(m, nmap, bool(r.randint(2)), bool(r.randint(2)), bool(r.randint(2)), (k, 0, 0, 0), (0, k, 0, 0))
    # a third of the scenes sit under a small HDR environment map (sampled explicitly, with a shifted sampler), a quarter have a moving sphere,
    # spot lights sometimes project the colour texture (all drawn last, see above)
    if r.uniform() < 0.33:
        sky = r.uniform(0.05, 0.6, (4, 8, 4)).astype(np.float32)
        sky[int(r.randint(0, 2)), int(r.randint(0, 8)), :3] = r.uniform(5.0, 40.0, 3)
        env = sc.add_texture(S.Texture(sky, S.TEX_RGBA32F, False, S.ADDR_WRAP, S.ADDR_CLAMP, S.FILTER_LINEAR))
        sc.set_environment((1.0, 1.0, 1.0), env, 1.0, (1, 0, 0, float(r.uniform(0, 1))), (0, 1, 0, 0), cam_back=tex if r.uniform() < 0.3 else S.UINT_MAX)
    if r.uniform() < 0.25 and len(sc.inst_geom) > 2:
        i = int(r.randint(1, min(len(sc.inst_geom), 1 + nobj)))
        m0 = sc.inst_matrices[i]
        sc.inst_motion[i] = S.translate(float(r.uniform(-0.6, 0.6)), float(r.uniform(0.0, 0.5)), float(r.uniform(-0.4, 0.4))) @ m0 @ S.rotate_y(float(r.uniform(-40, 40)))
    for lt in L:
        if int(lt["distType"]) == S.LIGHT_DIST_SPOT and r.uniform() < 0.5:
            mrow = np.eye(4); mrow[:3, 3] = lt["pos"][:3]
            S.set_projective(lt, mrow, float(r.uniform(40, 80)), 0.1, 100.0, tex)
    # ... and one scene in seven is seen through a lens stack: a symmetric triplet around a stop, randomly scaled (drawn last, see above)
    if r.uniform() < 0.15:
        k = float(r.uniform(0.8, 1.2))
        rad, ap = 60.0 * k, float(r.uniform(9.0, 14.0))
        lines = [(0, rad, 4.0, 1.6, ap), (1, -rad * 2.5, 2.0, 1.0, ap), (2, 0.0, 2.0, 0.0, float(r.uniform(4.0, 8.0))), (3, rad * 2.5, 4.0, 1.6, ap),
                 (4, -rad, float(r.uniform(38.0, 46.0)), 1.0, ap)]
        sc.set_optics(lines, 0.035, 0.001, "scene_to_sensor")
    # ... and three scenes in ten have one or two of their spheres coated with a thin film (drawn last, see above): one to three films on a
    # dielectric or conducting substrate, smooth or rough (isotropic, anisotropic or textured), transparent or not, plain or mapped thickness
    if r.uniform() < 0.3:
        cand = [i for i in range(2, 2 + nobj) if int(M[i]["mtype"]) not in (S.MAT_TYPE_BLEND, S.MAT_TYPE_LIGHT_SOURCE)]
        for i in list(r.permutation(cand))[:int(r.randint(1, 3))]:
            films = [{"eta": float(r.uniform(1.2, 2.6)), "k": float(r.choice([0.0, 0.0, r.uniform(0.0, 0.05)])), "thickness": float(r.uniform(50.0, 700.0))}
                     for _ in range(int(r.choice([1, 1, 2, 3])))]
            sub = {"eta": float(r.uniform(1.3, 1.8)), "k": 0.0} if r.uniform() < 0.5 else {"eta": float(r.uniform(0.15, 2.0)), "k": float(r.uniform(1.5, 4.0))}
            kind = r.randint(4)
            alpha = 0.0 if kind == 0 else (float(r.uniform(0.05, 0.4)) if kind == 1 else (float(r.uniform(0.05, 0.4)), float(r.uniform(0.05, 0.4))))
            atex = (tex, (1, 0, 0, 0), (0, 1, 0, 0)) if kind == 3 else None
            tmap = (float(r.uniform(40.0, 200.0)), float(r.uniform(300.0, 800.0)), tex, (2, 0, 0, 0), (0, 2, 0, 0)) if r.uniform() < 0.4 else None
            M[int(i)] = sc.material_thin_film(films, sub, 0.5 if kind == 3 else alpha, float(r.choice([1.0, 1.00028])), int(r.randint(2)), tmap, atex)
    return sc
