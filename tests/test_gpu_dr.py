"""PathTraceDR on the GPU (hand-derived adjoint + HBM atomics) against the oracle (forward-mode duals on the CPU)."""
import numpy as np
import pytest

from conftest import scene_path
from hydracore3_amd.scene import load_hydra_xml

pytestmark = pytest.mark.gpu


def setup(width=64, height=64, depth=None):
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    sc = load_hydra_xml(scene_path("test_035"), width, height)
    if depth:
        sc.trace_depth = depth
    gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc)
    # drmain.cpp:185: PutDiffTex2D(1, 256, 256, 4): texture 1 is the cube / floor albedo of test_035
    og, sg = gpu.PutDiffTex2D(1, 256, 256, 4)
    rc, oc, scn = cpu.put_diff_tex2d(1, 256, 256, 4)
    assert (og, sg) == (oc, scn) == (0, 256 * 256 * 4)
    rng = np.random.default_rng(5)
    data = rng.uniform(0.2, 0.9, sg).astype(np.float32)
    ref = rng.uniform(0.0, 0.5, (height, width, 4)).astype(np.float32)
    return sc, gpu, cpu, data, ref


def test_put_diff_tex2d_bad_id():
    _, gpu, cpu, _, _ = setup(16, 16)
    off, size = gpu.PutDiffTex2D(77, 8, 8, 4)        # reference: message + (size_t(-1), 0) (integrator_dr.cpp:35-39)
    assert size == 0 and off == 0xFFFFFFFFFFFFFFFF


def test_gradient_matches_oracle():
    sc, gpu, cpu, data, ref = setup()
    spp = 4
    out_g = np.zeros((sc.height, sc.width, 4), np.float32)
    out_c = np.zeros_like(out_g)
    grad_g = np.zeros_like(data)
    loss_g = gpu.PathTraceDR(gpu.N, 4, out_g, spp, ref, data, grad_g)
    loss_c, grad_c = cpu.path_trace_dr(out_c, spp, ref, data)
    print(f"loss gpu={loss_g:.6f} cpu={loss_c:.6f}; |grad| gpu={np.abs(grad_g).sum():.4f} cpu={np.abs(grad_c).sum():.4f}; nnz={np.count_nonzero(grad_c)}")
    assert np.count_nonzero(grad_c) > 1000
    assert abs(loss_g - loss_c) <= 1e-4 * abs(loss_c)
    # rendered colour accumulated by the replay
    d = (out_g[..., :3] - out_c[..., :3]) / spp
    assert np.sqrt(np.mean(np.sum(d * d, -1))) < 1e-3
    # gradient buffers: rtol 1e-2 (north_star), judged on the buffer norm and element-wise where the gradient is not tiny
    err = np.linalg.norm(grad_g - grad_c) / np.linalg.norm(grad_c)
    print(f"relative gradient error = {err:.3e}")
    assert err < 1e-2
    big = np.abs(grad_c) > 1e-3 * np.abs(grad_c).max()
    assert np.allclose(grad_g[big], grad_c[big], rtol=1e-2, atol=1e-5 * np.abs(grad_c).max())
    # alpha texels get no gradient (loss uses rgb only, integrator_dr.cpp:1128-1130)
    assert np.all(grad_g.reshape(-1, 4)[:, 3] == 0)
    # dataGrad is overwritten, not accumulated (memset at integrator_dr.cpp:1139)
    grad2 = np.full_like(data, 7.0)
    gpu2 = setup()[1]
    gpu2.PathTraceDR(gpu2.N, 4, np.zeros_like(out_g), spp, ref, data, grad2)
    assert np.allclose(grad2, grad_g, rtol=1e-4, atol=1e-6 * np.abs(grad_g).max())


def test_rng_streams_advance_like_forward():
    """The record pass is an ordinary PathTrace: after PathTraceDR the generators are where PathTraceBlock leaves them."""
    sc, gpu, cpu, data, ref = setup(32, 32)
    out = np.zeros((32, 32, 4), np.float32)
    gpu.PathTraceDR(gpu.N, 4, out, 3, ref, data, np.zeros_like(data))
    from hydracore3_amd.api import HipIntegrator
    fwd = HipIntegrator(sc)
    fwd.render(3)
    assert np.array_equal(gpu.random_gens(), fwd.random_gens())


def test_gradient_matches_oracle_on_metal_and_coated_gltf():
    """BASELINE configs[4] in miniature: the interior scene's 32 gltf materials (metalness 0 / 1, glossiness U[0,1], coat 1) all
    bound to ONE differentiable texture; the hand-derived adjoint (d val / d baseColor through Lambert, metal Fresnel and the coat
    term) against the oracle's forward-mode duals."""
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd import synth
    from oracle.orc import OracleIntegrator
    sc = synth.interior_scene(64, 48, objects=10, subdiv=1, tex_size=16)
    gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc)
    og, sg = gpu.PutDiffTex2D(1, 16, 16, 4)
    rc, oc, scn = cpu.put_diff_tex2d(1, 16, 16, 4)
    assert (og, sg) == (oc, scn) == (0, 16 * 16 * 4)
    rng = np.random.default_rng(11)
    data = rng.uniform(0.2, 0.9, sg).astype(np.float32)
    ref = rng.uniform(0.0, 0.5, (sc.height, sc.width, 4)).astype(np.float32)
    spp = 4
    out_g, out_c = np.zeros((sc.height, sc.width, 4), np.float32), np.zeros((sc.height, sc.width, 4), np.float32)
    grad_g = np.zeros_like(data)
    loss_g = gpu.PathTraceDR(gpu.N, 4, out_g, spp, ref, data, grad_g)
    loss_c, grad_c = cpu.path_trace_dr(out_c, spp, ref, data)
    err = np.linalg.norm(grad_g - grad_c) / np.linalg.norm(grad_c)
    print(f"loss gpu={loss_g:.6f} cpu={loss_c:.6f}; relative gradient error = {err:.3e}; nnz = {np.count_nonzero(grad_c)}")
    assert np.count_nonzero(grad_c) > 500
    assert abs(loss_g - loss_c) <= 1e-4 * abs(loss_c)
    assert err < 1e-2


def test_texture_regulariser_matches_oracle():
    """hpt_image2d4f_regularizer_dev (gather form, no atomics) == the oracle's Image2D4fRegularizer, accumulating into grad."""
    from hydracore3_amd.api import HipIntegrator
    from oracle import orc
    sc = load_hydra_xml(scene_path("test_035"), 16, 16)
    gpu = HipIntegrator(sc)
    rng = np.random.default_rng(9)
    for (h, w) in ((256, 256), (7, 33), (3, 3), (2, 8)):
        data = rng.uniform(0.0, 1.0, (h, w, 4)).astype(np.float32)
        data[h // 2:, : w // 2] = 0.5                                   # a flat patch: S == 0 terms
        g0 = rng.uniform(-1.0, 1.0, (h, w, 4)).astype(np.float32)
        gc = g0.copy()
        orc.image2d4f_regularizer(data, gc)
        gg = g0.copy()
        gpu.Image2D4fRegularizer(data, gg)
        assert np.allclose(gg, gc, rtol=1e-4, atol=1e-4), (h, w, float(np.abs(gg - gc).max()))
        assert np.array_equal(gg[..., 3], g0[..., 3])


def test_dr_under_the_wavefront_schedule_equals_megakernel():
    """PathTraceDR scheduled as shade / trace kernel pairs (records per pool slot, occluded light samples cleared when their shadow
    ray returns, reverse sweep in the shade pass) == the DR megakernel: same colours and generators bit for bit, loss and gradient up
    to the order of the float atomics; and == the oracle within the gradient bar."""
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd import synth
    from oracle.orc import OracleIntegrator
    sc = synth.interior_scene(96, 64, objects=12, subdiv=1, tex_size=16)
    mega, wf, cpu = HipIntegrator(sc), HipIntegrator(sc), OracleIntegrator(sc)
    mega.set_schedule(1); wf.set_schedule(2, 56, 0, 2)
    for integ in (mega, wf):
        assert integ.PutDiffTex2D(1, 16, 16, 4) == (0, 16 * 16 * 4)
    cpu.put_diff_tex2d(1, 16, 16, 4)
    rng = np.random.default_rng(4)
    data = rng.uniform(0.2, 0.9, 16 * 16 * 4).astype(np.float32)
    ref = rng.uniform(0.0, 0.5, (sc.height, sc.width, 4)).astype(np.float32)
    spp = 5
    outs, grads, losses = [], [], []
    for integ in (mega, wf):
        out = np.zeros((sc.height, sc.width, 4), np.float32); g = np.zeros_like(data)
        losses.append(integ.PathTraceDR(integ.N, 4, out, spp, ref, data, g))
        outs.append(out); grads.append(g)
    assert mega.last_schedule()[0] == 1 and wf.last_schedule()[0] == 2
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(mega.random_gens(), wf.random_gens())
    assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[0])
    assert np.linalg.norm(grads[0] - grads[1]) <= 1e-5 * np.linalg.norm(grads[0])
    out_c = np.zeros_like(outs[0])
    loss_c, grad_c = cpu.path_trace_dr(out_c, spp, ref, data)
    assert abs(losses[1] - loss_c) <= 1e-4 * abs(loss_c)
    assert np.linalg.norm(grads[1] - grad_c) / np.linalg.norm(grad_c) < 1e-2


def test_drmain_loop_tool_optimises_the_texture(tmp_path):
    """tests/cpp/hydra_hip_dr.cpp: the optimisation loop of diff_render/drmain.cpp:174-261 (texture := 1, PutDiffTex2D(1, 256, 256, 4),
    PathTraceDR -> Adam per iteration) in C++ on the C ABI, all arrays device-resident. The first loss equals the one the Python host
    gets for the same inputs, the loss falls, and the texels that receive gradient move from 1.0 towards the scene's albedo."""
    import os
    import subprocess
    from conftest import ROOT
    from hydracore3_amd.api import HipIntegrator
    tool = os.path.join(ROOT, "hydracore3_amd", "hydra_hip_dr")
    prefix = str(tmp_path / "opt")
    W = H = 96
    spp, ref_spp, iters = 16, 64, 16
    r = subprocess.run([tool, scene_path("test_035"), str(W), str(H), str(spp), str(iters), "1", "256", "256", prefix, "--ref-spp", str(ref_spp),
                        "--dump-every", str(iters - 1)], capture_output=True, text=True)
    print(r.stdout[-600:])
    assert r.returncode == 0, r.stdout + r.stderr
    losses = np.loadtxt(prefix + "_loss.txt")
    assert losses.shape == (iters,) and np.isfinite(losses).all()
    # PixelLossPT is a per-sample loss (variance included), too noisy at test sizes to demand a monotone fall: the bias of the rendered
    # frame against the reference must shrink instead
    # the same first iteration through the Python host
    sc = load_hydra_xml(scene_path("test_035"), W, H)
    fwd = HipIntegrator(sc)
    ref = (fwd.render(ref_spp) / ref_spp)[::-1].copy()
    dr = HipIntegrator(sc)
    off, size = dr.PutDiffTex2D(1, 256, 256, 4)
    data, grad = np.ones(size, np.float32), np.zeros(size, np.float32)
    loss0 = dr.PathTraceDR(dr.N, 4, np.zeros((H, W, 4), np.float32), spp, ref, data, grad)
    assert abs(loss0 - losses[0]) <= 2e-3 * abs(loss0), (loss0, losses[0])
    tex = np.fromfile(prefix + "_tex.bin", np.float32).reshape(256, 256, 4)
    touched = np.abs(grad.reshape(256, 256, 4)[..., :3]).sum(-1) > 0
    assert touched.sum() > 1000
    truth = sc.textures[1]
    assert truth.width == 256 and truth.height == 256
    rgb8 = np.stack([(truth.data >> s) & 0xFF for s in (0, 8, 16)], -1).astype(np.float32) / 255.0
    albedo = rgb8 ** 2.2 if truth.srgb else rgb8
    before = np.abs(1.0 - albedo[touched]).mean()
    after = np.abs(tex[..., :3][touched] - albedo[touched]).mean()
    print(f"mean |texel - albedo| over {int(touched.sum())} touched texels: {before:.4f} -> {after:.4f}; loss {losses[0]:.5f} -> {losses[-1]:.5f}")
    assert after < before
    frame = np.fromfile(prefix + "_frame.bin", np.float32).reshape(H, W, 4)
    assert np.isfinite(frame).all() and frame[..., :3].mean() > 0.0
    first = np.fromfile(prefix + "_00.bin", np.float32).reshape(H, W, 4)
    last = np.fromfile(prefix + f"_{iters - 1:02d}.bin", np.float32).reshape(H, W, 4)
    assert np.array_equal(last, frame)
    mse = lambda f: float(np.mean((np.clip(f[::-1, :, :3], 0, 2) - np.clip(ref[..., :3], 0, 2)) ** 2))
    print(f"frame vs reference (clipped MSE): {mse(first):.5f} -> {mse(last):.5f}")
    assert mse(last) < 0.8 * mse(first)
