"""CPU suite (-m "not gpu"): pins the oracle against golden integer vectors and closed forms, checks the host logic,
the ABI surface and the multi-rank sharding (gloo, world_size 2). No GPU compute here."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path
from hydracore3_amd import scene as S
from hydracore3_amd import synth
from hydracore3_amd.sharding import tid_window, tid_interleave, interleaved_tids

GOLD = os.path.join(ROOT, "tests", "golden")


# ---- golden integer vectors -----------------------------------------------------------------------------------------
def test_oracle_rng_matches_golden_kat():
    from oracle.orc import rng_kat
    kat = json.load(open(os.path.join(GOLD, "rng_kat.json")))
    for seed, v in kat.items():
        st, vals = rng_kat(int(seed), 8)
        assert list(st[:2]) == v["init"] and list(st[2:]) == v["final"]
        assert np.array_equal(vals.view(np.uint32), np.asarray(v["float4_bits"], np.uint32))
    # rndFloat4_Pseudo can return exactly 1.0f (crandom.h:52-54: (float)x1 rounds up for x1 >= 2^32-128): must be reproduced
    assert np.float32(4294967295.0) * np.float32(1.0 / 4294967296.0) == np.float32(1.0)


def test_oracle_packxy_matches_golden():
    from oracle.orc import OracleIntegrator
    g = json.load(open(os.path.join(GOLD, "packxy_16x16_t8.json")))
    sc = synth.furnace_plane(16, 16)
    assert sc.tile_size() == g["tile"]
    assert np.array_equal(OracleIntegrator(sc).packed_xy(), np.asarray(g["packedXY"], np.uint32))
    # odd sizes fall back to smaller tiles (integrator_pt.h:379-389)
    sc2 = synth.furnace_plane(12, 10)
    assert sc2.tile_size() == 2
    xy = OracleIntegrator(sc2).packed_xy()
    assert sorted((int(v) >> 16) * 12 + (int(v) & 0xFFFF) for v in xy) == list(range(120))


# ---- closed forms -----------------------------------------------------------------------------------------------------
def test_scalar_identities():
    from oracle.orc import probe
    assert abs(probe("FrDielectricPBRT", 1.0, 1.0, 1.5)[0] - 0.04) < 1e-6            # ((n-1)/(n+1))^2
    assert probe("FrDielectricPBRT", 0.0, 1.0, 1.5)[0] == pytest.approx(1.0, abs=1e-6)  # grazing
    assert probe("misWeightHeuristic", 0.37, 0.37)[0] == 0.5
    assert probe("misWeightHeuristic", 1.0, np.inf)[0] == 1.0                          # non-finite pdfs count as 0 (cglobals.h:277)
    n, k = 0.2, 3.9
    assert probe("FrComplexConductor", 1.0, n, k)[0] == pytest.approx(((n - 1) ** 2 + k * k) / ((n + 1) ** 2 + k * k), rel=1e-5)
    r = probe("FrDielectricDetailedV2", 1.0, 1.5)
    assert r[0] == pytest.approx(0.04, abs=1e-6) and r[1] == pytest.approx(-1.0) and r[3] == pytest.approx(1 / 1.5)
    # GGX: the sampling pdf integrates to one over the hemisphere of outgoing directions
    # (narrow lobe, near-normal view: the part of the lobe that falls below the horizon is negligible)
    v = np.array([0.1, 0.05, 0.99], np.float32); v /= np.linalg.norm(v)
    th = (np.arange(200) + 0.5) / 200 * (np.pi / 2)
    ph = (np.arange(400) + 0.5) / 400 * (2 * np.pi)
    acc = 0.0
    for t in th:
        for p in ph[::4]:
            l = (np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t))
            acc += probe("ggxEvalPDF", *l, *v, 0.35)[0] * np.sin(t)
    acc *= (np.pi / 2 / 200) * (2 * np.pi / 100)
    assert acc == pytest.approx(1.0, abs=0.03)
    # cosine-hemisphere sample stays on the normal's side and has unit length
    d = probe("lambertSample", 0.3, 0.7, 0.0, 0.0, 1.0)[:3]
    assert d[2] > 0 and np.linalg.norm(d) == pytest.approx(1.0, abs=1e-5)
    # Oren-Nayar with zero roughness is Lambert
    assert probe("orennayarFunc", 0.0, 0.6, 0.8, 0.6, 0.0, 0.8, 0.0)[0] == pytest.approx(1.0)


def test_furnace_plane_is_exact():
    """Lambert under a constant environment: cos * f / pdf = albedo exactly, so every sample returns albedo * env."""
    from oracle.orc import OracleIntegrator
    sc = synth.furnace_plane()
    img = OracleIntegrator(sc).render(4) / 4
    assert np.allclose(img[..., :3], np.array([0.2 * 1.0, 0.5 * 2.0, 0.9 * 0.5], np.float32), rtol=2e-5)


def test_direct_lighting_matches_form_factor():
    """Shadow-ray estimator, depth 1: E[L] = albedo/pi * Le * integral over the light of cos cos' / r^2 dA."""
    from oracle.orc import OracleIntegrator
    sc = synth.plane_under_rect_light(48, 48)
    p = sc.params(integrator=S.INTEGRATOR_SHADOW_PT)
    o = OracleIntegrator(sc, p)
    spp = 256
    img = o.render(spp)[..., 0] / spp
    # expected radiance at the floor point seen through each pixel centre
    proj_inv = np.array(p.projInv[:], np.float64).reshape(4, 4).T
    wv_inv = np.array(p.worldViewInv[:], np.float64).reshape(4, 4).T
    ys, xs = np.mgrid[0:48, 0:48]
    ndc = np.stack([2 * (xs + 0.5) / 48 - 1, 2 * (ys + 0.5) / 48 - 1, np.zeros_like(xs, float), np.ones_like(xs, float)], -1)
    cam = ndc @ proj_inv.T
    d = cam[..., :3] / cam[..., 3:4]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    dw = d @ wv_inv[:3, :3].T
    ow = wv_inv[:3, 3]
    t = -ow[1] / dw[..., 1]
    P = ow + t[..., None] * dw
    g = (np.arange(64) + 0.5) / 64 - 0.5
    lx, lz = np.meshgrid(g, g)
    L = np.stack([lx, np.full_like(lx, 2.0), lz], -1).reshape(-1, 3)      # 1 x 1 light at y = 2
    v = L[None, None] - P[..., None, :]
    r2 = np.sum(v * v, -1)
    cos_p = v[..., 1] / np.sqrt(r2)
    expect = 0.5 / np.pi * 10.0 * np.mean(cos_p * cos_p / r2, -1) * 1.0
    ok = (t > 0) & (np.abs(P[..., 0]) < 19) & (np.abs(P[..., 2]) < 19)
    rel = abs(img[ok].mean() - expect[ok].mean()) / expect[ok].mean()
    assert rel < 0.01, rel
    centre = (slice(16, 32), slice(16, 32))
    assert np.allclose(img[centre], expect[centre], rtol=0.15)


def test_naive_shadow_and_mis_estimators_agree():
    """The reference's own acceptance practice (main.cpp:156-158, testing/run_tests.py:59-64): the three integrators
    must converge to the same image. Compared on 8x8 box-filtered images, PSNR >= 30 dB is the harness's pass mark."""
    from oracle.orc import OracleIntegrator
    sc = S.load_hydra_xml(scene_path("test_035"), 64, 64)
    imgs = {}
    for name, integ, spp, naive in (("mis", S.INTEGRATOR_MIS_PT, 96, False), ("shadow", S.INTEGRATOR_SHADOW_PT, 96, False),
                                    ("naive", S.INTEGRATOR_STUPID_PT, 1024, True)):
        o = OracleIntegrator(sc, sc.params(integrator=integ))
        img = o.render(spp, naive=naive)[..., :3] / spp
        img = np.minimum(img, 1.0)                                   # the harness compares LDR images
        imgs[name] = img.reshape(8, 8, 8, 8, 3).mean(axis=(1, 3))
    def psnr(a, b):
        return 10 * np.log10(1.0 / np.mean((a - b) ** 2))
    assert psnr(imgs["mis"], imgs["shadow"]) > 35
    assert psnr(imgs["mis"], imgs["naive"]) > 30
    assert abs(imgs["mis"].mean() - imgs["naive"].mean()) / imgs["mis"].mean() < 0.03


def test_oracle_gradient_matches_finite_differences():
    from oracle.orc import OracleIntegrator
    sc = S.load_hydra_xml(scene_path("test_035"), 24, 24)
    o = OracleIntegrator(sc)
    rc, off, size = o.put_diff_tex2d(1, 256, 256, 4)
    assert rc == 0 and off == 0 and size == 256 * 256 * 4
    rng = np.random.default_rng(1)
    data = rng.uniform(0.2, 0.9, size).astype(np.float32)
    ref = rng.uniform(0, 0.5, (24, 24, 4)).astype(np.float32)
    gens = o.random_gens().copy()
    out = np.zeros((24, 24, 4), np.float32)
    loss, grad = o.path_trace_dr(out, 3, ref, data)
    idx = np.argsort(-np.abs(grad))[:12]
    o.set_random_gens(gens)
    fd = o.path_trace_dr_fd(3, ref, data, idx, h=2e-2)
    assert np.all(np.abs(grad[idx]) > 0)
    assert np.allclose(grad[idx], fd, rtol=5e-3, atol=1e-4)
    assert loss > 0 and np.all(grad.reshape(-1, 4)[:, 3] == 0)


def test_adam_step_matches_formula():
    from oracle.orc import adam_step
    rng = np.random.default_rng(0)
    x = rng.normal(size=64).astype(np.float32); g = rng.normal(size=64).astype(np.float32)
    m = np.zeros(64, np.float32); G = np.zeros(64, np.float32)
    x0 = x.copy()
    adam_step(x, g, m, G, 150)
    m_ref = 0.75 * g; G_ref = 2 * (0.5 * g * g)
    assert np.allclose(m, m_ref) and np.allclose(G, G_ref)
    assert np.allclose(x, x0 - (0.25 / 2) * m_ref / np.sqrt(G_ref + 1e-8), rtol=1e-5)


# ---- fixture loader against SURVEY Appendix B -------------------------------------------------------------------------------
def test_fixture_loader_tables():
    sc = S.load_hydra_xml(scene_path("test_035"))
    assert (sc.width, sc.height, sc.trace_depth, sc.spp) == (1024, 768, 5, 2)
    assert sc.mat_vert_offset == [(0, 0), (12, 24), (22, 44)] and sc.mat_id_by_prim.size == 24 and sc.vpos.shape[0] == 48
    assert sc.remap_inst == [(-1, -1), (-1, -1), (-1, 0)]
    assert len(sc.materials) == 8 and len(sc.lights) == 1 and len(sc.textures) == 2
    m0, m6 = sc.materials[0], sc.materials[6]
    assert m0["mtype"] == S.MAT_TYPE_GLTF and m0["cflags"] == 1 and m0["texid"][0] == 1 and m0["texid"][1] == S.UINT_MAX
    assert m0["data"][S.GLTF_FLOAT_GLOSINESS] == 1.0 and m0["data"][S.GLTF_FLOAT_IOR] == 0.0
    assert m6["mtype"] == S.MAT_TYPE_LIGHT_SOURCE and m6["data"][0] == np.float32(25.1327419) and tuple(m6["colors"][0]) == (1, 1, 1, 0)
    lt = sc.lights[0]
    assert lt["geomType"] == S.LIGHT_GEOM_RECT and lt["pdfA"] == 0.25 and tuple(lt["pos"]) == (0, np.float32(3.85), 0, 1)
    assert tuple(lt["norm"]) == (0, -1, 0, 0) and lt["matId"] == 7
    t1 = sc.textures[1]
    assert (t1.width, t1.height, t1.fmt, t1.srgb) == (256, 256, S.TEX_RGBA8, True)
    sc2 = S.load_hydra_xml(scene_path("test_228"))
    assert sc2.mat_id_by_prim.size == 8202 and sc2.vpos.shape[0] == 4310 and sc2.inst_geom == [2, 0, 1]
    l2 = sc2.lights[0]
    assert l2["geomType"] == S.LIGHT_GEOM_POINT and l2["distType"] == S.LIGHT_DIST_OMNI and l2["iesId"] == 1
    assert sc2.textures[1].data.shape == (74, 1) and sc2.textures[1].data.max() == 1.0


def test_oracle_texture_fetch_semantics():
    """Bilinear taps as Tex2DFetchAD defines them (integrator_dr.cpp:60-161): texel centres are exact, wrap by modulo."""
    from oracle.orc import OracleIntegrator
    sc = synth.furnace_plane(8, 8)
    tex = np.arange(16 * 4, dtype=np.float32).reshape(4, 4, 4)
    tid = sc.add_texture(S.Texture(tex, S.TEX_RGBA32F, False))
    o = OracleIntegrator(sc)
    uv = np.array([[(1 + 0.5) / 4, (2 + 0.5) / 4], [0.5 / 4 - 1.0, 0.5 / 4], [0.5, 0.5]], np.float32)
    r = o.tex_sample(tid, uv)
    assert np.allclose(r[0], tex[2, 1]) and np.allclose(r[1], tex[0, 0]) and np.allclose(r[2], tex[1:3, 1:3].mean((0, 1)))
    assert np.allclose(o.tex_sample(0, uv), 1.0)                      # white dummy


# ---- ABI surface -----------------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from hydracore3_amd import api
    hdr = open(os.path.join(ROOT, "include", "hydra_hip.h")).read()
    declared = set(re.findall(r"\b(hpt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.ABI), declared ^ set(api.ABI)
    lib = api.load_library()                                          # dlopen + bind: no compute, works without a GPU
    for name in declared:
        assert hasattr(lib, name)
    nm = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (hpt_[a-z0-9_]+)", nm))
    assert declared <= exported


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    with pytest.raises(HydraHipError):
        HipIntegrator(synth.furnace_plane(8, 8))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hydracore3_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("# oracle", "").lower() or f in ("scene.py",), f
    assert "import oracle" not in open(os.path.join(ROOT, "hydracore3_amd", "scene.py")).read()


# ---- sharding ---------------------------------------------------------------------------------------------------------------
def test_tid_windows_tile_the_frame():
    for n in (64, 1000, 1024 * 1024, 1920 * 1080):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                b, c = tid_window(r, world, n)
                assert b % 64 == 0 or c == 0
                cover += list(range(b, b + c))[:3] + list(range(b, b + c))[-3:]
                assert c >= 0 and b + c <= n
            assert sum(tid_window(r, world, n)[1] for r in range(world)) == n
            # interleaved chunks: disjoint, complete, and consistent with the kernel's item -> tid mapping
            seen = np.zeros(n, np.int32)
            for r in range(world):
                b, cnt, chunk, stride = tid_interleave(r, world, n)
                k = np.arange(cnt, dtype=np.int64)
                tid = b + (k // chunk) * chunk * stride + k % chunk
                assert tid.max(initial=-1) < n
                seen[tid] += 1
                runs = interleaved_tids(r, world, n)
                assert sum(c for _, c in runs) == cnt
            assert np.all(seen == 1)


_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from hydracore3_amd import synth
from hydracore3_amd.sharding import tid_window, tid_interleave, interleaved_tids
from oracle.orc import OracleIntegrator
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sc = synth.plane_under_rect_light(32, 32)
o = OracleIntegrator(sc, threads=1)
b, c = tid_window(rank, world, o.N)
img = np.zeros((32, 32, 4), np.float32)
o.path_trace_block(img, 3, tid_begin=b, tid_count=c)
t = torch.from_numpy(img)
dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    full = OracleIntegrator(sc, threads=1).render(3)
    assert np.array_equal(t.numpy(), full), "sharded frame differs from the single-rank frame"
    print("SHARD_OK")

# bench.py --scaling strong: every world-th 1024-tid chunk (here 64-tid chunks so that a 32 x 32 frame has several per rank)
o2 = OracleIntegrator(sc, threads=1)
img2 = np.zeros((32, 32, 4), np.float32)
for tid0 in range(rank * 64, o2.N, world * 64):
    o2.path_trace_block(img2, 3, tid_begin=tid0, tid_count=min(64, o2.N - tid0))
t2 = torch.from_numpy(img2)
dist.reduce(t2, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    assert np.array_equal(t2.numpy(), full), "interleaved shards differ from the single-rank frame"
    print("INTERLEAVE_OK")

# bench.py --scaling weak: sample sharding - rank r seeds its generators as threads r*N.. of one big InitRandomGens call, renders the
# whole frame, the reduce adds the frames; equals the two renders done one after the other on one rank, and the sub-streams differ
from oracle.orc import rng_kat
o3 = OracleIntegrator(sc, threads=1)
seeds = np.array([rng_kat(rank * o3.N + i, 0)[0][:2] for i in range(o3.N)], np.uint32)
o3.set_random_gens(seeds.reshape(-1))
img3 = o3.render(2)
t3 = torch.from_numpy(img3.copy())
dist.reduce(t3, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    o4 = OracleIntegrator(sc, threads=1)
    seeds1 = np.array([rng_kat(o4.N + i, 0)[0][:2] for i in range(o4.N)], np.uint32)
    o4.set_random_gens(seeds1.reshape(-1))
    other = o4.render(2)
    assert np.allclose(t3.numpy(), img3 + other, rtol=1e-6, atol=1e-6) and not np.array_equal(img3, other)
    print("SAMPLE_SHARD_OK")

# DR: all_reduce(SUM) of a_dataGrad and the loss over pixel shards == the single-rank gradient (float sums differ in order only)
from hydracore3_amd import scene as S
def dr_setup():
    scd = synth.plane_under_rect_light(16, 16)
    tid_ = scd.add_texture(S.Texture(np.full((4, 4, 4), 0.5, np.float32), S.TEX_RGBA32F, False))
    scd.materials[0]["texid"][0] = tid_
    od = OracleIntegrator(scd, threads=1)
    od.put_diff_tex2d(tid_, 4, 4, 4)
    return od
rs = np.random.default_rng(1)
data = rs.uniform(0.2, 0.9, 64).astype(np.float32)
ref = rs.uniform(0.0, 0.5, (16, 16, 4)).astype(np.float32)
od = dr_setup()
b, c = tid_window(rank, world, od.N)
loss_r, grad_r = od.path_trace_dr(np.zeros((16, 16, 4), np.float32), 3, ref, data, tid_begin=b, tid_count=c)
tg, tl = torch.from_numpy(grad_r.copy()), torch.tensor([loss_r], dtype=torch.float64)
dist.all_reduce(tg, op=dist.ReduceOp.SUM); dist.all_reduce(tl, op=dist.ReduceOp.SUM)
loss_f, grad_f = dr_setup().path_trace_dr(np.zeros((16, 16, 4), np.float32), 3, ref, data)
assert np.count_nonzero(grad_f) > 10
assert np.allclose(tg.numpy(), grad_f, rtol=1e-5, atol=1e-7) and abs(float(tl.item()) - loss_f) <= 1e-5 * abs(loss_f)
if rank == 0:
    print("DR_ALLREDUCE_OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_sharding_reassembles_the_frame(tmp_path):
    """N > 1 paths of bench.py on CPU, two gloo ranks, the oracle as the stand-in renderer: (1) disjoint tid windows and (2) interleaved
    chunks reassemble the single-rank frame bit for bit through reduce(SUM); (3) sample sharding with offset generator seeds adds two
    decorrelated renders; (4) pixel-sharded PathTraceDR + all_reduce(SUM) of a_dataGrad and the loss equals the single-rank result."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    for tag in ("SHARD_OK", "INTERLEAVE_OK", "SAMPLE_SHARD_OK", "DR_ALLREDUCE_OK"):
        assert r.returncode == 0 and tag in r.stdout, tag + "\n" + r.stdout[-2000:] + r.stderr[-2000:]


def test_texture_regulariser_gradient_against_finite_differences():
    """RegLossImage2D4f / Image2D4fRegularizer (diff_render/integrator_dr.cpp:317-367): the oracle's analytic gradient of the
    restated loss equals central finite differences, alpha gets none, it ACCUMULATES (enzyme_dup), a flat image gives zero."""
    from oracle import orc
    rng = np.random.default_rng(3)
    h, w = 9, 12
    data = rng.uniform(0.1, 0.9, (h, w, 4)).astype(np.float32)
    grad = np.full((h, w, 4), 0.25, np.float32)
    orc.image2d4f_regularizer(data, grad)
    grad -= 0.25
    assert np.all(grad[..., 3] == 0)
    eps = 1e-3
    for (y, x, c) in [(0, 0, 0), (0, 5, 1), (4, 6, 2), (1, 1, 0), (7, 10, 1), (8, 11, 2), (3, 0, 0), (4, 4, 1)]:
        d1, d2 = data.copy(), data.copy()
        d1[y, x, c] += eps; d2[y, x, c] -= eps
        fd = (orc.reg_loss_image2d4f(d1) - orc.reg_loss_image2d4f(d2)) / (2 * eps)
        assert abs(fd - grad[y, x, c]) < 2e-3 * max(1.0, abs(fd)), (y, x, c, fd, grad[y, x, c])
    flat = np.full((5, 5, 4), 0.5, np.float32)
    g = np.zeros_like(flat)
    orc.image2d4f_regularizer(flat, g)
    assert orc.reg_loss_image2d4f(flat) == 0.0 and not g.any()


def _read_blobs(path):
    raw = open(path, "rb").read()
    out, i = [], 0
    while i < len(raw):
        j = raw.index(b"\0", i)
        name = raw[i:j].decode()
        n = int.from_bytes(raw[j + 1:j + 9], "little")
        out.append((name, raw[j + 9:j + 9 + n]))
        i = j + 9 + n
    return out


@pytest.mark.parametrize("scene_name", ["test_035", "test_228", "legacy_materials", "typed_materials", "env_map", "png_textures", "jpg_textures", "test_spectral", "test_spectral+spectral", "spectral_plastic+spectral", "spectral_glass+spectral", "spectral_sky+spectral", "exr_sky", "thin_film", "thin_film+spectral", "thin_film_rough", "thin_film_rough+spectral", "spectral_textures", "spectral_textures+spectral"])
def test_cpp_scene_loader_produces_the_same_tables(scene_name, tmp_path):
    """hydracore3_amd/csrc/scene_loader.h (Hydra XML + VSGF + image4ub + IES in C++) == the Python fixture loader:
    every table byte for byte, matrices and light frames to float rounding (both invert in double)."""
    import ctypes as C
    import subprocess
    import __graft_entry__ as g
    from hydracore3_amd import scene as S
    g.build()
    tool = os.path.join(ROOT, "hydracore3_amd", "hydra_hip_render")
    out = str(tmp_path / "tables.bin")
    spectral = scene_name.endswith("+spectral")                          # m_spectral_mode = 1: the same tables, the reflectance colour of conductors unread
    scene_name = scene_name.split("+")[0]
    r = subprocess.run([tool, scene_path(scene_name), "96", "64", "1", out, "--tables"] + (["--spectral"] if spectral else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    blobs = _read_blobs(out)
    named = {k: v for k, v in blobs if not k.startswith("tex")}
    sc = S.load_hydra_xml(scene_path(scene_name), 96, 64, spectral=spectral)
    d = sc.desc()
    # spectra: resampled in float on both sides - bit for bit; the observer fit goes through exp() of two libms - to rounding
    assert named["specValues"] == np.ascontiguousarray(sc.spec_values, np.float32).tobytes()
    assert named["specOffsetSz"] == np.asarray(sc.spec_offset_sz, np.uint32).tobytes()
    assert np.allclose(np.frombuffer(named["cieXYZ"], np.float32), S.cie_xyz_fit().reshape(-1), rtol=1e-6, atol=1e-9)
    assert list(np.frombuffer(named["camResponse"], np.int32)) == [*sc.cam_response_spectrum_id, sc.cam_response_type]
    # spectra given by textures: bands and their resolved table entries
    assert named["specTexIdsWavelengths"] == np.asarray(sc.spec_tex_ids_wavelengths, np.uint32).tobytes()
    assert named["specTexOffsetSz"] == np.asarray(sc.spec_tex_offset_sz, np.uint32).tobytes()
    # thin films: the index vectors bit for bit; the tables come from the one film_precompute.h on both sides, over observer fits that differ in
    # the last place (RGB) - to rounding
    assert named["filmsThickness"] == np.asarray(sc.films_thickness, np.float32).tobytes()
    assert named["filmsSpecId"] == np.asarray(sc.films_spec_id, np.uint32).tobytes()
    assert named["filmsEtaK"] == np.asarray(sc.films_eta_k, np.float32).tobytes()
    pf = np.frombuffer(named["precompThinFilms"], np.float32)
    assert pf.size == sc.precomp_thin_films.size and np.allclose(pf, sc.precomp_thin_films, rtol=1e-5, atol=1e-6)

    def arr(ptr, dtype, count):
        return np.frombuffer((C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr), dtype=dtype, count=count).copy() if count else np.zeros(0, dtype)

    exact = {"vPos4f": sc.vpos, "vData8f": sc.vdata, "triIndices": sc.tri_indices, "matIdByPrimId": sc.mat_id_by_prim,
             "matVertOffset": np.asarray(sc.mat_vert_offset, np.uint32), "geomTriCount": np.asarray(sc.geom_tri_count, np.uint32),
             "geomVertCount": np.asarray(sc.geom_vert_count, np.uint32), "instGeomId": np.asarray(sc.inst_geom, np.uint32),
             "remapInst": np.asarray(sc.remap_inst, np.int32), "allRemapLists": sc.all_remap_lists}
    for k, ref in exact.items():
        assert named[k] == np.ascontiguousarray(ref).tobytes(), k
    ni = d.numInsts
    for k, ptr in (("instMatrices", d.instMatrices), ("normMatrices", d.normMatrices)):
        got, ref = np.frombuffer(named[k], np.float32), arr(ptr, np.float32, 16 * ni * (2 if k == "normMatrices" and sc.inst_motion else 1))
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-7), k
    mats = np.frombuffer(named["materials"], S.MATERIAL_DTYPE)
    ref_m = np.array(sc.materials, dtype=S.MATERIAL_DTYPE)
    assert mats.shape == ref_m.shape
    for f in S.MATERIAL_DTYPE.names:                                      # floats to rounding (SetMiPlastic, the metal mix weight), the rest exactly
        if S.MATERIAL_DTYPE[f].base.kind == "f":
            assert np.allclose(mats[f], ref_m[f], rtol=1e-6, atol=1e-7), f
        else:
            assert np.array_equal(mats[f], ref_m[f]), f
    lights = np.frombuffer(named["lights"], S.LIGHT_DTYPE)
    ref_l = np.array(sc.lights, dtype=S.LIGHT_DTYPE)
    assert lights.shape == ref_l.shape
    for f in S.LIGHT_DTYPE.names:
        if S.LIGHT_DTYPE[f].base.kind == "f":
            assert np.allclose(lights[f], ref_l[f], rtol=1e-6, atol=1e-7), f
        else:
            assert np.array_equal(lights[f], ref_l[f]), f
    p_ref = bytes(sc.params())
    p_got = named["params"]
    pr, pg = np.frombuffer(p_ref[:128], np.float32), np.frombuffer(p_got[:128], np.float32)       # projInv, worldViewInv
    assert np.allclose(pg, pr, rtol=1e-5, atol=1e-6)
    assert p_got[128:176] == p_ref[128:176]                                                          # window, depth, integrator, tile ...
    assert len(p_got) == len(p_ref) == 272
    assert np.allclose(np.frombuffer(p_got[176:224], np.float32), np.frombuffer(p_ref[176:224], np.float32), rtol=1e-6)   # exposure .. envColor
    assert p_got[224:240] == p_ref[224:240]                                                          # environment map ids
    assert np.allclose(np.frombuffer(p_got[240:], np.float32), np.frombuffer(p_ref[240:], np.float32), rtol=1e-6)         # its sampler rows
    assert named.get("arrays1f", b"") == np.ascontiguousarray(sc.arrays1f, np.float32).tobytes()
    assert np.allclose(np.frombuffer(named.get("lensLines", b""), np.float32), sc.lens_lines.reshape(-1), rtol=1e-6)
    assert np.allclose(np.frombuffer(named["physSize"], np.float32), sc.phys_size, rtol=1e-6)
    assert np.frombuffer(named["normMatrices2Offs"], np.uint32)[0] == d.normMatrices2Offs == (ni if sc.inst_motion else 0)
    if sc.inst_motion:
        assert np.array_equal(np.frombuffer(named["instHasMotion"], np.uint32), [1 if i in sc.inst_motion else 0 for i in range(ni)])
        assert np.allclose(np.frombuffer(named["instMatricesMotion"], np.float32), arr(d.instMatricesMotion, np.float32, 16 * ni), rtol=1e-6, atol=1e-7)
    # textures, in the order the reference's lazy loading creates them
    tex = [(np.frombuffer(h, np.uint32), b) for (kh, h), (kb, b) in zip(blobs, blobs[1:]) if kh == "texHeader" and kb == "texBytes"]
    assert len(tex) == len(sc.textures)
    for (h, b), t in zip(tex, sc.textures):
        assert tuple(int(v) for v in h) == (t.width, t.height, t.fmt, 1 if t.srgb else 0, t.addr_u, t.addr_v, t.filter)
        if t.fmt == S.TEX_RGBA8:
            assert b == t.data.tobytes()
        else:
            assert np.allclose(np.frombuffer(b, np.float32), t.data.reshape(-1), rtol=1e-6, atol=1e-7)


def test_environment_map_estimators_agree():
    """A scene lit only by a sampled HDR environment map (tests/golden/scenes/env_map): the MIS estimator (explicit SampleMap2D samples
    weighted against the implicit hits through evalMap2DPdf, integrator_pt_lgt.cpp:30-55, 175-236, integrator_pt.cpp:550-595) and the
    naive one (implicit hits only) converge to the same image - which holds only if the table pdf, the inverse sampler transform and
    the lat-long mapping are consistent. The reference's own acceptance practice (testing/run_tests.py:59-64)."""
    from hydracore3_amd import scene as S
    from oracle.orc import OracleIntegrator
    sc = S.load_hydra_xml(scene_path("env_map"), 48, 32)
    assert sc.env_enable_sam == 1 and sc.env_light_id == 0 and sc.lights[0]["geomType"] == S.LIGHT_GEOM_ENV
    t = sc.arrays1f
    assert t.size == 32 * 16 + 1 and t[0] == 0.0 and np.all(np.diff(t) > 0)
    spp = 192
    mis = OracleIntegrator(sc, sc.params(S.INTEGRATOR_MIS_PT)).render(spp) / spp
    naive = OracleIntegrator(sc, sc.params(S.INTEGRATOR_STUPID_PT)).render(spp, naive=True) / spp
    assert np.isfinite(mis).all() and np.isfinite(naive).all()
    m, n = mis[..., :3].mean(axis=(0, 1)), naive[..., :3].mean(axis=(0, 1))
    assert np.allclose(m, n, rtol=0.03), (m, n)
    # the camera back plate replaces the map for primary rays that miss: those pixels carry no Monte-Carlo noise at all
    same = np.all(np.abs(mis[..., :3] - naive[..., :3]) < 1e-6, axis=-1)
    assert same.sum() > 200 and same.sum() < 48 * 32


def test_plastic_bsdf_pdf_normalises_and_sampling_matches_eval():
    """MAT_TYPE_PLASTIC (include/cmat_plastic.h) over the table of hpt_plastic_precompute (mi_materials.cpp:377-451): the pdf plasticEval
    reports integrates to one over the hemisphere; plasticSampleAndEval's direction, value and pdf equal plasticEval at that direction;
    the table is a transmittance (0..1, rising towards normal incidence) and the internal reflectance lies in (0, 1)."""
    from hydracore3_amd import scene as S
    from oracle import orc
    sc = S.SceneData()
    for alpha, nonlinear in ((0.1, 0), (0.3, 1)):
        m = sc.material_plastic((0.6, 0.4, 0.2), alpha, 1.49, 1.000277, nonlinear)
        off = int(m["datai"][0])
        table = sc.arrays1f[off:off + 64]
        assert np.all(table > 0.0) and np.all(table < 1.0) and table[-1] > table[0] and 0.0 < float(m["data"][3]) < 1.0
        assert abs(table[-1] - (1.0 - ((1.49 / 1.000277 - 1.0) / (1.49 / 1.000277 + 1.0)) ** 2)) < 0.02          # ~ 1 - F(0) for a nearly smooth coat
        head = [*m["data"][:4], float(nonlinear), 0.6, 0.4, 0.2]
        for v in ((0.0, 0.0, 1.0), (0.6, 0.0, 0.8), (0.95, 0.1, np.sqrt(1 - 0.95 ** 2 - 0.01))):
            # midpoint quadrature of the pdf over the hemisphere in (cos theta, phi)
            n_mu, n_phi = 240, 512
            mu = (np.arange(n_mu) + 0.5) / n_mu
            phi = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
            total = 0.0
            for c in mu:
                sn = np.sqrt(1 - c * c)
                for ph in phi:
                    r = orc.probe("plasticEval", *head, *v, sn * np.cos(ph), sn * np.sin(ph), c, *table)
                    total += r[3]
            total *= (1.0 / n_mu) * (2 * np.pi / n_phi)
            assert 0.97 < total < 1.02, (alpha, v, total)       # (a little of the specular lobe falls below the horizon at grazing views)
            rng = np.random.default_rng(1)
            for _ in range(50):
                u = rng.uniform(0, 1, 3)
                smp = orc.probe("plasticSample", *head, *v, *u, *table)
                if np.all(smp[3:6] == 0.0) and np.all(smp[:3] == (0.0, 1.0, 0.0)):
                    continue                                       # the sampler's early-out: the reflected direction fell below the horizon
                ev = orc.probe("plasticEval", *head, *v, *smp[:3], *table)
                assert np.allclose(smp[3:6], ev[:3], rtol=2e-4, atol=1e-6) and np.isclose(smp[6], ev[3], rtol=2e-4), (u, smp, ev)


def _identity_scene(width=40, height=28):
    """A floor and three spheres under a constant environment (no lights): the stage of the identity checks below."""
    from hydracore3_amd import scene as S
    sc = S.SceneData()
    sc.width, sc.height = width, height
    sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 1.6, 6.0), (0.0, 0.6, 0.0), (0.0, 1.0, 0.0)
    sc.fov, sc.trace_depth = 40.0, 4
    sc.env_color = (0.8, 0.9, 1.0, 0.0)
    p, n, t, uv, idx = synth._quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 4.0)
    sc.materials.append(S.material_lambert((0.5, 0.5, 0.5)))
    sc.add_instance(sc.add_mesh(p, n, t, uv, idx, [0]), np.eye(4))
    sp = synth._sphere_mesh(2)
    ntri = sp[4].size // 3
    for i in range(3):
        gid = sc.add_mesh(sp[0], sp[1], sp[2], sp[3], sp[4], np.full(ntri, 1 + i, np.uint32))
        sc.add_instance(gid, S.translate(-1.6 + 1.6 * i, 0.7, 0.0) @ S.rotate_y(30.0 * i) @ S.scale(0.7, 0.7, 0.7))
    return sc


def test_identities_of_the_widened_branches():
    """Closed-form relations the restated branches must satisfy whatever the scene, checked on the oracle (which the GPU is held to):
    a neutral normal map changes nothing; a constant environment map equals the plain environment colour to rounding under the naive
    estimator; blend(A, A, w) has A's expectation; a clear legacy glass in a white furnace returns exactly one per path that leaves
    the scene; plastic does not create energy."""
    from hydracore3_amd import scene as S
    from oracle.orc import OracleIntegrator
    spp = 24
    base = _identity_scene()
    base.materials += [S.material_gltf((0.8, 0.3, 0.2, 1.0), 0.0, 0.7, 1.0, 1.5), S.material_conductor(0.2, 3.9, 0.2, 0.2), S.material_diffuse((0.3, 0.4, 0.8), 0.4)]
    ref = OracleIntegrator(base).render(spp)

    # 1. a normal map that says "straight up" in tangent space (float texels 0.5, 0.5, 1) on the floor, whose tangent frame is exactly
    #    orthonormal, leaves the frame unchanged (on the spheres the interpolated tangent is not exactly perpendicular to the interpolated
    #    normal, and the reference's inverse of (tan, bitan, n) then tilts even a neutral map)
    sc = _identity_scene()
    flat = sc.add_texture(S.Texture(np.tile(np.array([0.5, 0.5, 1.0, 1.0], np.float32), (2, 2, 1)), S.TEX_RGBA32F, False))
    S.set_normal_map(sc.materials[0], flat, row0=(3, 0, 0, 0), row1=(0, 3, 0, 0))
    sc.materials += base.materials[1:]
    bumped = OracleIntegrator(sc).render(spp)
    assert np.sqrt(np.mean(np.sum(((bumped - ref)[..., :3] / spp) ** 2, -1))) < 1e-6

    # 2. a constant HDR environment map: the naive estimator (implicit hits only) draws the same numbers and multiplies by the same colour
    sc = _identity_scene(); sc.materials += base.materials[1:]
    const = sc.add_texture(S.Texture(np.ones((4, 8, 4), np.float32), S.TEX_RGBA32F, False))
    sc.set_environment(base.env_color, const, sample=True)
    a = OracleIntegrator(sc, sc.params(S.INTEGRATOR_STUPID_PT)).render(spp, naive=True)
    b = OracleIntegrator(base, base.params(S.INTEGRATOR_STUPID_PT)).render(spp, naive=True)
    assert np.allclose(a, b, rtol=2e-6, atol=0)                           # (the bilinear weights of a constant map sum to one within an ulp)
    # ... and with the map sampled explicitly the MIS estimator agrees with it in the mean (uniform table pdf against the cosine lobe)
    m = OracleIntegrator(sc, sc.params(S.INTEGRATOR_MIS_PT)).render(4 * spp) / (4 * spp)
    assert np.allclose(m[..., :3].mean(axis=(0, 1)), (b / spp)[..., :3].mean(axis=(0, 1)), rtol=0.03)

    # 3. blend(A, A, w): one more generator step per hit, the same expectation
    sc = _identity_scene()
    sc.materials += [S.material_blend(4, 4, 0.3), S.material_blend(5, 5, 0.8), S.material_blend(6, 6, 0.5)] + base.materials[1:]
    blended = OracleIntegrator(sc).render(4 * spp) / (4 * spp)
    many = OracleIntegrator(base).render(4 * spp) / (4 * spp)
    assert np.allclose(blended[..., :3].mean(axis=(0, 1)), many[..., :3].mean(axis=(0, 1)), rtol=0.02)

    # 4. white furnace: clear glass (both colours one) neither absorbs nor creates light - cos * val / pdf is exactly one per event, so a path
    #    that leaves the scene carries throughput one and the pixel mean is the fraction of such paths; plastic stays at or below the furnace
    for name, mats in (("glass", [S.material_glass((1, 1, 1), (1, 1, 1), 1.5)] * 3),):
        sc = _identity_scene(); sc.env_color = (1.0, 1.0, 1.0, 0.0); sc.trace_depth = 12
        sc.materials[0] = S.material_glass((1, 1, 1), (1, 1, 1), 1.5)
        sc.materials += mats
        img = OracleIntegrator(sc, sc.params(S.INTEGRATOR_STUPID_PT)).render(spp, naive=True) / spp
        vals = np.unique(np.round(img[..., :3] * spp).astype(int))
        assert np.allclose(img[..., :3] * spp, np.round(img[..., :3] * spp), atol=2e-4), name      # every sample contributed exactly 0 or 1
        assert img[..., :3].mean() > 0.9 and img[..., :3].max() <= 1.0 + 1e-5 and vals.max() == spp
    sc = _identity_scene(); sc.env_color = (1.0, 1.0, 1.0, 0.0); sc.trace_depth = 8
    sc.materials[0] = sc.material_plastic((0.9, 0.9, 0.9), 0.15)
    sc.materials += [sc.material_plastic((1.0, 1.0, 1.0), 0.05), sc.material_plastic((0.7, 0.7, 0.7), 0.4, nonlinear=1), sc.material_plastic((0.95, 0.95, 0.95), 0.25, 1.8, 1.0)]
    img = OracleIntegrator(sc, sc.params(S.INTEGRATOR_STUPID_PT)).render(8 * spp, naive=True) / (8 * spp)
    assert img[..., :3].mean() < 1.0 and np.percentile(img[..., :3], 99) < 1.08


def test_ldr_image_files_decode_to_the_containers_texels(tmp_path):
    """.png / .bmp / .ppm textures (LoadTextureAndMakeCombined's LiteImage::LoadImage<uint32_t> branch, integrator_pt_scene_tex.cpp:24-33): the
    png_textures fixture stores test_035's 256 x 256 texture as an RGBA PNG whose rows cycle through all five filter types and the 2 x 2 one
    as a top-down 24-bit BMP; the loader returns the very texels of the .image4ub containers. A binary PPM and a bottom-up BMP round-trip too."""
    import struct
    from hydracore3_amd.scene import decode_ldr_image, load_hydra_xml
    a = load_hydra_xml(scene_path("test_035"), 32, 32)
    b = load_hydra_xml(scene_path("png_textures"), 32, 32)
    big_a = [t for t in a.textures if t.width == 256][0]
    big_b = [t for t in b.textures if t.width == 256][0]
    assert np.array_equal(big_a.data, big_b.data) and big_b.srgb and big_b.fmt == big_a.fmt
    small = [t for t in b.textures if t.width == 2 and t.height == 2]
    assert len(small) == 1 and np.all((small[0].data >> 24) == 255)
    raw0 = open(os.path.join(os.path.dirname(scene_path("test_035")), "data", "chunk_00000.image4ub"), "rb").read()
    assert np.array_equal(small[0].data & 0xFFFFFF, np.frombuffer(raw0, "<u4", 4, 8).reshape(2, 2) & 0xFFFFFF)
    rng = np.random.default_rng(1)
    px = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    ppm = tmp_path / "t.ppm"
    ppm.write_bytes(b"P6\n# a comment\n7 5\n255\n" + px.tobytes())
    want = px[..., 0].astype(np.uint32) | (px[..., 1].astype(np.uint32) << 8) | (px[..., 2].astype(np.uint32) << 16) | np.uint32(0xFF000000)
    assert np.array_equal(decode_ldr_image(str(ppm), ppm.read_bytes()), want)
    stride = (7 * 3 + 3) & ~3                                             # a bottom-up BMP: rows come out as stored
    rows = b"".join(px[y, :, ::-1].tobytes() + b"\0" * (stride - 21) for y in range(5))
    bmp = b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 7, 5, 1, 24, 0, len(rows), 2835, 2835, 0, 0) + rows
    assert np.array_equal(decode_ldr_image("x.bmp", bmp), want)
    with pytest.raises(NotImplementedError):
        decode_ldr_image("x.jpg", b"")


def test_jpeg_reader_returns_libjpegs_texels():
    """.jpg / .jpeg textures (the same LiteImage branch): csrc/jpeg_decode.h against PIL's libjpeg, texel for texel - baseline and progressive,
    4:4:4 / 4:2:2 / 4:2:0 chroma, greyscale, optimised Huffman tables, restart markers, sizes that are no multiple of the MCU - and the
    jpg_textures fixture's two files through the scene loader."""
    import io
    PILImage = pytest.importorskip("PIL.Image")
    from hydracore3_amd.scene import decode_ldr_image, load_hydra_xml
    rng = np.random.RandomState(3)

    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos((x + y) / 9.0), 60 + (x * 3 + y * 2) % 190], -1) + rng.randint(-20, 20, (h, w, 3))
        return np.clip(a, 0, 255).astype(np.uint8)

    variants = (("4:4:4", dict(subsampling=0)), ("4:2:2", dict(subsampling=1)), ("4:2:0", dict(subsampling=2)), ("4:2:0 progressive", dict(subsampling=2, progressive=True)),
                ("4:4:4 progressive", dict(subsampling=0, progressive=True)), ("grey", dict()), ("4:2:0 optimised", dict(subsampling=2, optimize=True)),
                ("4:2:0 restart", dict(subsampling=2, restart_marker_rows=1)), ("4:2:2 restart blocks", dict(subsampling=1, restart_marker_blocks=3)))
    for w, h in ((64, 48), (37, 29), (17, 9), (3, 5), (1, 1)):
        for name, kw in variants:
            pic = picture(w, h)
            im = PILImage.fromarray(pic[..., 0] if name == "grey" else pic)
            buf = io.BytesIO(); im.save(buf, "JPEG", quality=85, **kw)
            ref = np.asarray(PILImage.open(io.BytesIO(buf.getvalue())).convert("RGB")).astype(np.uint32)
            mine = decode_ldr_image("x.jpg", buf.getvalue())
            assert mine.shape == (h, w) and np.all((mine >> 24) == 255)
            assert np.array_equal(mine & 0xFFFFFF, ref[..., 0] | (ref[..., 1] << 8) | (ref[..., 2] << 16)), (w, h, name)
    sc = load_hydra_xml(scene_path("jpg_textures"), 32, 32)
    folder = os.path.join(os.path.dirname(scene_path("jpg_textures")), "data")
    for t in sc.textures[1:]:
        f = "texture1.jpg" if t.width == 256 else "texture0.jpeg"
        ref = np.asarray(PILImage.open(os.path.join(folder, f)).convert("RGB")).astype(np.uint32)
        assert np.array_equal(t.data & 0xFFFFFF, ref[..., 0] | (ref[..., 1] << 8) | (ref[..., 2] << 16))
    with pytest.raises(NotImplementedError):
        decode_ldr_image("x.jpg", b"not a jpeg")


# ---- spectral rendering: tables and the oracle's restatement --------------------------------------------------------------------
def test_spectrum_resampling_and_wavelength_placement():
    """Spectrum::ResampleUniform (471 samples at 1 nm from 360 nm: linear between the tabulated points, zero outside them) and
    SampleWavelengths (spectrum.h:58-75: one offset, three rotations by a quarter of the range, wrapped into [a, b])."""
    from oracle.orc import probe
    u = S.resample_uniform([400.0, 500.0, 700.0], [1.0, 3.0, 1.0])
    assert u.shape == (471,) and u.dtype == np.float32
    assert u[0] == 0.0 and u[39] == 0.0 and u[40] == 1.0 and u[140] == 3.0 and u[340] == 1.0 and u[341] == 0.0
    assert u[90] == pytest.approx(2.0) and u[240] == pytest.approx(2.0)
    w = probe("SampleWavelengths", 0.5, 360.0, 830.0)[:4]
    assert list(w) == [595.0, 712.5, 830.0, 477.5]                       # 830 is not > b: kept; the next one wraps to a + 117.5
    w = probe("SampleWavelengths", 0.0, 360.0, 830.0)[:4]
    assert list(w) == [360.0, 477.5, 595.0, 712.5]
    # XYZToRGB: the D65 white point (0.9505, 1, 1.089) maps to about (1, 1, 1)
    assert np.allclose(probe("XYZToRGB", 0.9505, 1.0, 1.089)[:3], 1.0, atol=2e-3)


def test_cie_observer_fit_is_close_to_the_tabulated_one():
    """The fixture loaders' m_cie_xyz: the multi-lobe fit integrates to the reference's CIE_Y_integral (106.856895, spectrum.h:154) within
    0.5 % and peaks where the photopic curve does."""
    cie = S.cie_xyz_fit()
    assert cie.shape == (471, 4) and cie.dtype == np.float32
    assert cie[:, 1].sum() == pytest.approx(106.856895, rel=5e-3)
    assert 553 <= 360 + int(np.argmax(cie[:, 1])) <= 557
    assert cie[:, 0].sum() == pytest.approx(cie[:, 1].sum(), rel=0.01)    # equal-energy white: X = Y = Z
    assert cie[:, 2].sum() == pytest.approx(cie[:, 1].sum(), rel=0.01)


def test_spectral_fixture_tables():
    """LoadSceneSpectrumData on the reference's spectral fixture: seven spectra of 471 samples each, ids wired into the materials and the
    light as the XML names them."""
    sc = S.load_hydra_xml(scene_path("test_spectral"), 32, 32, spectral=True)
    assert sc.spectral_mode == 1 and sc.params().spectralMode == 1
    assert [tuple(v) for v in sc.spec_offset_sz] == [(471 * i, 471) for i in range(7)]
    assert sc.spec_values.shape == (7 * 471,) and np.isfinite(sc.spec_values).all() and sc.spec_values.max() > 0
    assert int(sc.lights[0]["specId"]) == 4
    cond = [m for m in sc.materials if int(m["mtype"]) == S.MAT_TYPE_CONDUCTOR]
    assert len(cond) == 1 and list(cond[0]["spdid"][:2]) == [5, 6]
    diff = [int(m["spdid"][0]) for m in sc.materials if int(m["mtype"]) == S.MAT_TYPE_DIFFUSE]
    assert sorted(diff) == [1, 2, 3]
    d = sc.desc()
    assert d.numSpectra == 7 and d.numSpecValues == 7 * 471 and d.numCieXYZ == 471
    rgb = S.load_hydra_xml(scene_path("test_spectral"), 32, 32)
    assert rgb.spectral_mode == 0 and rgb.params().spectralMode == 0


def test_oracle_spectral_render_of_grey_scene_keeps_the_luminance():
    """A consistency property of the restated spectral path: with every spectrum removed the fixture is grey (reflectances and emission
    are single values splat over the four wavelengths), each path carries the same radiance L at its four wavelengths, and the CIE
    observer turns that into Y = L * mean(ybar) * (830 - 360) / 106.857 = L in expectation - so the spectral image's luminance equals
    the RGB image of the same scene up to Monte-Carlo noise."""
    from oracle.orc import OracleIntegrator
    imgs = {}
    for spectral in (False, True):
        sc = S.load_hydra_xml(scene_path("test_spectral"), 48, 48, spectral=spectral)
        for m in sc.materials: m["spdid"] = 0xFFFFFFFF
        for l in sc.lights: l["specId"] = 0xFFFFFFFF
        img = OracleIntegrator(sc).render(32)[..., :3] / 32
        imgs[spectral] = img
    rgb = imgs[False]
    assert np.allclose(rgb[..., 0], rgb[..., 1]) and rgb.mean() > 1e-3
    M = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    y_spec = (imgs[True].reshape(-1, 3) @ np.linalg.inv(M).T)[:, 1].mean()
    print(f"mean luminance: spectral {y_spec:.5f}, rgb {rgb[..., 1].mean():.5f}")
    assert y_spec == pytest.approx(rgb[..., 1].mean(), rel=0.03)


def test_wide_node_quantiser_is_conservative(tmp_path):
    """quantizeNode4 (csrc/hpt_types.h), the host / device routine behind the 4-wide compressed nodes of the heavy-scene trace kernel: two million
    random child bounds - flat, point-sized, far from the origin, tiny next to huge - decode (fma(byte, 2^(b - 127), org), as the kernel does it) to
    boxes that contain the originals and are at most two grid steps looser (tests/cpp/quantize_test.cpp, plain g++)."""
    exe = str(tmp_path / "quantize_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "quantize_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "all conservative" in r.stdout, r.stdout + r.stderr


def test_folded_box_tests_are_conservative(tmp_path):
    """The trace kernels' box tests compute a plane's distance as plane * id - origin * id with a finite reciprocal direction, and in the 4-wide node
    with the decode folded in (csrc/hpt_device.h: slabRay, nodeSlabs, wideNodeStep). Boxes only cull, so what must hold is one-sided: 1.25 million
    rays that EXACTLY meet an unpadded box (long double) - aimed at its inside, faces, edges and corners, from near and very far, with tiny and zero
    direction components, the reciprocal an ulp off, `best` at the exact entry distance - are all answered "hit" by the float tests on the padded
    (and quantised) box (tests/cpp/slab_fold_test.cpp, plain g++; without Aabb::pad a quarter of them would be missed)."""
    exe = str(tmp_path / "slab_fold_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "slab_fold_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "all conservative" in r.stdout, r.stdout + r.stderr


def test_exr_decoders_agree_and_read_what_was_written(tmp_path):
    """OpenEXR textures (LoadImage4fFromEXR / LoadImage1fFromEXR through tinyexr in the reference, imageutils.cpp:317-392): the Python and the C++
    decoder read tests/golden/exr/*.exr (written by make_exr_scene.py: ZIPS + HALF RGB, NONE + FLOAT single channel with an infinity, ZIP + HALF
    RGBA over three 16-line blocks) to the arrays that went in, and the exr_sky fixture's map equals the env_map fixture's .image4f one."""
    import __graft_entry__ as g
    g.build()
    tool = os.path.join(ROOT, "hydracore3_amd", "hydra_hip_render")
    rng = np.random.default_rng(7)
    a = rng.uniform(0, 4, (5, 7, 3)).astype(np.float32)
    y = rng.uniform(0, 1e5, (9, 6)).astype(np.float32); y[2, 3] = np.inf
    b = rng.uniform(0, 2, (40, 12, 4)).astype(np.float32)
    h16 = lambda v: v.astype(np.float16).astype(np.float32)
    expect = {"zips_half_rgb": np.concatenate([h16(a), np.ones((5, 7, 1), np.float32)], -1),
              "none_float_y": np.repeat(y[..., None], 4, -1),
              "zip_half_rgba": h16(b)}
    for name, ref in expect.items():
        path = os.path.join(GOLD, "exr", name + ".exr")
        img = S.decode_exr(open(path, "rb").read())
        assert img.shape == ref.shape and np.array_equal(img, ref), name
        out = str(tmp_path / (name + ".bin"))
        r = subprocess.run([tool, "--exr", path, out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = open(out, "rb").read()
        w, h = np.frombuffer(raw, np.uint32, 2)
        assert (h, w) == ref.shape[:2]
        assert np.array_equal(np.frombuffer(raw, np.float32, offset=8).reshape(h, w, 4), ref), name
    sky_a = S.load_hydra_xml(scene_path("env_map"), 32, 32)
    sky_b = S.load_hydra_xml(scene_path("exr_sky"), 32, 32)
    assert all(np.array_equal(p.data, q.data) and p.fmt == q.fmt for p, q in zip(sky_a.textures, sky_b.textures))
    assert sky_b.env_enable_sam == 1 and np.array_equal(sky_a.arrays1f, sky_b.arrays1f)


def test_parallel_bvh_build_equals_the_sequential_one(tmp_path):
    """Bvh2Builder's thread pool (csrc/bvh_build.h): the top of the tree by the calling thread, subtrees of disjoint primitive ranges by worker
    threads, appended afterwards - same primitive order, depth and, node for node, the same child boxes and leaves as the sequential build,
    on random boxes and on a regular grid with many equal centroids; and collapseToWide's 4-wide tree of each: every BVH2 leaf reachable exactly
    once, two to four children per node, every decoded child box around its BVH2 source box (tests/cpp/bvh_build_test.cpp, plain g++)."""
    exe = str(tmp_path / "bvh_build_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "bvh_build_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("equal the sequential one") == 3 and r.stdout.count("every leaf once") == 3, r.stdout + r.stderr


def test_thin_film_airy_identities_and_tables():
    """FrFilm / FrFilmRefl (include/airy_reflectance.h:9-106) in the oracle against the closed forms of a single film, and the product's table
    builder (csrc/film_precompute.h through hpt_film_precompute, written apart from the oracle) against the oracle at the tables' nodes."""
    import ctypes as C
    from hydracore3_amd import scene as S
    from oracle import orc
    lam = 550.0
    # (1) a film of the substrate's own index is no film: one Fresnel interface
    for cos_i in (1.0, 0.8, 0.3):
        r = orc.probe("FrFilm", cos_i, 1.0, 0.0, 1.5, 0.0, 1.5, 0.0, 200.0, lam)
        s2 = (1 - cos_i * cos_i) / 1.5 ** 2; ct = np.sqrt(1 - s2)
        rs = ((cos_i - 1.5 * ct) / (cos_i + 1.5 * ct)) ** 2; rp = ((1.5 * cos_i - ct) / (1.5 * cos_i + ct)) ** 2
        assert abs(r[0] - 0.5 * (rs + rp)) < 2e-6 and abs(r[0] + r[1] - 1.0) < 2e-6 and abs(r[2] - r[0]) < 2e-6
    # (2) normal incidence: the Airy formula; (3) a quarter-wave layer of index sqrt(n0 nT) cancels the reflection
    for nf, nt, d in ((1.38, 1.52, 99.6), (2.2, 1.5, 140.0), (1.33, 1.0, 320.0)):
        r = orc.probe("FrFilm", 1.0, 1.0, 0.0, nf, 0.0, nt, 0.0, d, lam)
        r1, r2 = (1.0 - nf) / (1.0 + nf), (nf - nt) / (nf + nt)
        delta = 4 * np.pi * nf * d / lam
        airy = (r1 * r1 + r2 * r2 + 2 * r1 * r2 * np.cos(delta)) / (1 + r1 * r1 * r2 * r2 + 2 * r1 * r2 * np.cos(delta))
        assert abs(r[0] - airy) < 5e-6 and abs(r[0] + r[1] - 1.0) < 5e-6, (nf, nt, d, r, airy)
    nf = np.sqrt(1.0 * 1.69)
    assert orc.probe("FrFilm", 1.0, 1.0, 0.0, nf, 0.0, 1.69, 0.0, lam / (4 * nf), lam)[0] < 1e-6
    # (4) an absorbing substrate transmits nothing into it that FrFilm counts, and reflects more than the bare dielectric film
    r = orc.probe("FrFilm", 0.7, 1.0, 0.0, 2.0, 0.0, 0.2, 3.0, 100.0, lam)
    assert 0.5 < r[0] < 1.0 and abs(r[2] - r[0]) < 2e-6
    # (5) the product's spectral table at its nodes == the oracle's FrFilm; a second film of the substrate's index changes nothing (multFrFilm == FrFilm)
    sc = S.SceneData(); sc.spectral_mode = 1
    sc.spec_offset_sz = [(0, 471)]; sc.spec_values = np.ones(471, np.float32)
    sc.material_thin_film([{"eta": 1.7, "k": 0.02, "thickness": 260.0}], {"eta": 1.45, "k": 0.0}, alpha=0.0, transparent=1)
    one = sc.precomp_thin_films.copy().reshape(4, 94, 180)
    sc.material_thin_film([{"eta": 1.7, "k": 0.02, "thickness": 260.0}, {"eta": 1.45, "k": 0.0, "thickness": 75.0}], {"eta": 1.45, "k": 0.0}, alpha=0.0, transparent=1)
    two = sc.precomp_thin_films[one.size:].reshape(4, 94, 180)
    assert np.allclose(one[:, :, :170], two[:, :, :170], rtol=0, atol=2e-5)   # (at grazing angles inside the substrate the two routines guard a vanishing denominator differently)
    for i in (0, 40, 93):
        w = np.float32(469.0) / np.float32(93.0) * np.float32(i) + np.float32(360.0)
        for j in (0, 60, 150, 179):
            c = float(np.clip(np.cos(np.float32(np.pi / 2 / 179.0 * j)), 1e-3, 1.0))
            f = orc.probe("FrFilm", c, 1.00028, 0.0, 1.7, 0.02, 1.45, 0.0, 260.0, float(w))
            b = orc.probe("FrFilm", c, 1.45, 0.0, 1.7, 0.02, 1.00028, 0.0, 260.0, float(w))
            assert np.allclose([one[0, i, j], one[1, i, j], one[2, i, j], one[3, i, j]], [f[0], f[1], b[0], b[1]], rtol=0, atol=2e-5), (i, j)
    # (6) RGB tables: a lossless film keeps the white point (R + T = 1 in every channel up to the observer fit), from both sides
    rgb = S.SceneData()
    rgb.material_thin_film([{"eta": 1.33, "thickness": 300.0}], {"eta": 1.5}, alpha=0.0, transparent=1)
    t = rgb.precomp_thin_films.reshape(4, 180, 3)
    assert np.all(np.abs(t[0] + t[1] - 1.0) < 2e-3) and np.all(np.abs(t[2][:100] + t[3][:100] - 1.0) < 2e-3) and t[0].min() >= 0.0 and t[0].max() <= 1.0
    assert np.ptp(t[0, 0]) > 0.01                                        # ... and it is coloured: interference


def test_jpeg_reader_refuses_what_it_does_not_read(tmp_path):
    """Truncated, oversized and non-Huffman files come back as an error code through the C ABI, never as an exception or a crash."""
    import ctypes as C
    import struct
    from hydracore3_amd.api import load_library
    lib = load_library()
    w, h = C.c_uint32(0), C.c_uint32(0)
    def rc(raw):
        buf = np.frombuffer(raw, np.uint8) if len(raw) else np.zeros(1, np.uint8)
        return lib.hpt_decode_jpeg(buf.ctypes.data, len(raw), C.byref(w), C.byref(h), None, 0)
    assert rc(b"") != 0 and rc(b"\xff\xd8\xff\xd9") != 0
    sof = lambda marker, hh, ww: b"\xff" + bytes([marker]) + struct.pack(">HBHHB", 11, 8, hh, ww, 1) + b"\x01\x11\x00"
    assert rc(b"\xff\xd8" + sof(0xC0, 65535, 65535) + b"\xff\xd9") != 0          # 2^32 pixels: refused before anything is allocated
    assert rc(b"\xff\xd8" + sof(0xC9, 8, 8) + b"\xff\xd9") != 0                  # arithmetic coding
    assert rc(b"\xff\xd8" + sof(0xC0, 8, 8) + b"\xff\xda\x00\x08\x01\x01\x00\x00\x3f\x00" + b"\x00" * 8 + b"\xff\xd9") != 0   # a scan without tables


def test_crafted_exr_with_wrapping_chunk_offset_is_refused(tmp_path):
    """csrc/scene_loader.h: decodeExr checks a chunk offset as `off > size - 8`, not `off + 8 > size` (an offset near 2^64 wraps and the memcpy behind
    the check reads out of bounds), and refuses a header whose data window the file cannot hold before it allocates the planes. A valid file
    with its first chunk offset overwritten by 2^64 - 4 must come back as an error, not a crash."""
    import __graft_entry__ as g
    g.build()
    tool = os.path.join(ROOT, "hydracore3_amd", "hydra_hip_render")
    good = open(os.path.join(GOLD, "exr", "none_float_y.exr"), "rb").read()
    # the offset table follows the header's terminating zero byte: find it by decoding once with the Python reader's layout knowledge -
    # the first chunk's offset is the smallest table entry and points at its own scanline header (y, size)
    import struct
    # header: magic, version, attributes (name\0 type\0 size data) ..., \0
    p = 8
    while good[p] != 0:
        e = good.index(b"\0", p); p = e + 1
        e = good.index(b"\0", p); p = e + 1
        size = struct.unpack_from("<i", good, p)[0]; p += 4 + size
    p += 1
    bad = bytearray(good); struct.pack_into("<Q", bad, p, 0xFFFFFFFFFFFFFFFC)
    path = tmp_path / "wrap.exr"; path.write_bytes(bytes(bad))
    r = subprocess.run([tool, "--exr", str(path), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and r.returncode > 0 and "exr" in (r.stdout + r.stderr).lower(), (r.returncode, r.stdout, r.stderr)   # an error message, not a signal
    short = tmp_path / "short.exr"; short.write_bytes(good[:p + 4])
    r = subprocess.run([tool, "--exr", str(short), str(tmp_path / "o2.bin")], capture_output=True, text=True)
    assert r.returncode > 0, (r.returncode, r.stdout, r.stderr)
