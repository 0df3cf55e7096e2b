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


# ---- BASELINE configs[3] (C4): scenes/test_228 + a differentiable albedo on matGray, IES point light ------------------------------------
def _c4_setup(width=64, height=64, seed=21):
    from hydracore3_amd import synth
    sc, tex_id = synth.dr_scene(scene_path("test_228"), width, height)
    rng = np.random.default_rng(seed)
    data = rng.uniform(0.2, 0.9, 256 * 256 * 4).astype(np.float32)
    ref = rng.uniform(0.0, 0.5, (height, width, 4)).astype(np.float32)
    return sc, tex_id, data, ref


@pytest.mark.parametrize("schedule", [1, 2, 3])
def test_c4_dr_test228_matches_oracle(schedule):
    """BASELINE.json configs[3] on its own scene: test_228 (8 202 triangles, two spheres in a box) with matGray bound to the 256 x 256 x 4
    differentiable albedo of drmain.cpp:185, lit by the scene's POINT light with an IES profile - the misWeight = 1 branch of
    kernel_SampleLightSource (integrator_pt.cpp:405-411) and LightIntensity's IES lookup (integrator_pt_lgt.cpp:129-139) inside the
    adjoint. Loss to 1e-4, colours to the image bar, gradient rtol 1e-2 against the oracle's forward-mode duals; both schedules."""
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    sc, tex_id, data, ref = _c4_setup()
    gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc)
    if schedule == 2:
        gpu.set_schedule(2, 56, 0, 1)
    else:
        gpu.set_schedule(schedule)                                         # 1: megakernel, 3: megakernel with block-local ray repacking (hpt_block.hip)
    assert gpu.PutDiffTex2D(tex_id, 256, 256, 4) == (0, 256 * 256 * 4)
    assert cpu.put_diff_tex2d(tex_id, 256, 256, 4)[1:] == (0, 256 * 256 * 4)
    spp = 4
    out_g, out_c = np.zeros((sc.height, sc.width, 4), np.float32), np.zeros((sc.height, sc.width, 4), np.float32)
    grad_g = np.zeros_like(data)
    loss_g = gpu.PathTraceDR(gpu.N, 4, out_g, spp, ref, data, grad_g)
    loss_c, grad_c = cpu.path_trace_dr(out_c, spp, ref, data)
    assert gpu.last_schedule()[0] == schedule
    err = np.linalg.norm(grad_g - grad_c) / np.linalg.norm(grad_c)
    d = (out_g[..., :3] - out_c[..., :3]) / spp
    l2 = float(np.sqrt(np.sum(d * d, -1)).max())                          # the per-pixel bar: the worst pixel, not a mean over the frame
    same = int(np.sum(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
    print(f"schedule {schedule}: loss gpu={loss_g:.6f} cpu={loss_c:.6f}; relative gradient error = {err:.3e}; nnz = {np.count_nonzero(grad_c)}; "
          f"L2 = {l2:.2e}; identical generators {same} / {gpu.N}")
    assert np.count_nonzero(grad_c) > 2000
    assert abs(loss_g - loss_c) <= 1e-4 * abs(loss_c)
    assert l2 < 1e-3
    assert err < 1e-2
    big = np.abs(grad_c) > 1e-3 * np.abs(grad_c).max()
    assert np.allclose(grad_g[big], grad_c[big], rtol=1e-2, atol=1e-5 * np.abs(grad_c).max())
    assert same >= gpu.N - 2


# ---- a19: AdamOptimizer<float>::step ---------------------------------------------------------------------------------------------------
def test_adam_step_dev_matches_oracle():
    """hpt_adam_step_dev against the oracle's restatement of adam.h:43-62, bit for bit (same formula, IEEE sqrt and division, no contraction
    on either side), at iterations on both sides of the iter / 100 steps of the learning-rate schedule; moments carried from call to call."""
    from hydracore3_amd.api import HipIntegrator
    from oracle import orc
    gpu = HipIntegrator(load_hydra_xml(scene_path("test_035"), 16, 16))
    rng = np.random.default_rng(17)
    n = 100003                                                           # not a multiple of the block size
    x = rng.normal(0.5, 0.3, n).astype(np.float32)
    m = np.zeros(n, np.float32); G = np.zeros(n, np.float32)
    dx, dm, dG = gpu.dev_array(x), gpu.dev_array(m), gpu.dev_array(G)
    for it in (0, 1, 99, 100, 199, 250, 1000):
        g = (rng.normal(0.0, 1.0, n) * 10.0 ** rng.uniform(-9, 1, n)).astype(np.float32)
        g[::97] = 0.0                                                    # texels without gradient
        dg = gpu.dev_array(g)
        gpu.AdamStep_dev(dx, dg, dm, dG, it)
        orc.adam_step(x, g, m, G, it)
        for name, a, b in (("state", dx.download(), x), ("momentum", dm.download(), m), ("gsquare", dG.download(), G)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (it, name, float(np.abs(a - b).max()))
        dg.free()
    # the learning rate really is 0.25 / (iter / 100 + 1) with an INTEGER division (adam.h:60)
    x2 = np.ones(4, np.float32); d2 = [gpu.dev_array(v) for v in (x2, np.full(4, 2.0, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32))]
    gpu.AdamStep_dev(*d2, 199)
    step = 1.0 - d2[0].download()[0]
    assert abs(step - (0.25 / 2.0) * 1.5 / np.sqrt(4.0 + 1e-8)) < 1e-6, step


def test_three_optimisation_iterations_match_oracle():
    """drmain's loop (diff_render/drmain.cpp:196-246) for three iterations, device-resident on the GPU (PathTraceDR_dev -> AdamStep_dev)
    against the oracle (orc_path_trace_dr -> orc_adam_step): the generators continue from iteration to iteration on both sides and the
    losses agree to 1e-4. The texture: Adam's step gamma * m / sqrt(G + 1e-8) is a steep function of the gradient around |g| ~ 1e-4
    (slope ~ 700), where a gradient that differs in its last bits (float atomics / the reference's per-thread partial sums add in
    another order) moves the texel by up to a few 1e-3; away from that knee (|g| >= 1e-2: slope < 2e-3) the texels agree to 1e-5."""
    from hydracore3_amd.api import HipIntegrator
    from oracle import orc
    from oracle.orc import OracleIntegrator
    sc, tex_id, data0, ref = _c4_setup(48, 48, seed=5)
    gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc)
    gpu.set_schedule(1)
    _, size = gpu.PutDiffTex2D(tex_id, 256, 256, 4); cpu.put_diff_tex2d(tex_id, 256, 256, 4)
    spp = 4
    x = data0.copy(); m = np.zeros_like(x); G = np.zeros_like(x)
    dx, dm, dG, dgrad = gpu.dev_array(x), gpu.dev_array(m), gpu.dev_array(G), gpu.dev_array(np.zeros_like(x))
    dref, dloss = gpu.dev_array(ref), gpu.dev_array(np.zeros(1, np.float32))
    dframe = gpu.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
    strong = np.zeros(x.size, bool)
    for it in range(3):
        gpu.PathTraceDR_dev(dframe, spp, dref, dx, dgrad, dloss)
        loss_g = float(dloss.download()[0]) / gpu.N
        grad_g = dgrad.download()
        x_before = dx.download()
        gpu.AdamStep_dev(dx, dgrad, dm, dG, it)
        out_c = np.zeros((sc.height, sc.width, 4), np.float32)
        loss_c, grad_c = cpu.path_trace_dr(out_c, spp, ref, x)
        orc.adam_step(x, grad_c, m, G, it)
        xg = dx.download()
        diff = np.abs(xg - x)
        strong |= np.abs(grad_c) >= 1e-2
        moved = float(np.mean(np.abs(x - data0) > 1e-3))
        gerr = float(np.linalg.norm(grad_g - grad_c) / np.linalg.norm(grad_c))
        print(f"iteration {it}: loss gpu={loss_g:.6f} cpu={loss_c:.6f}; gradient error {gerr:.2e}; texels within 1e-5: all {np.mean(diff <= 1e-5) * 100:.3f} %, "
              f"|g| >= 1e-2 ({int(strong.sum())}) {np.mean(diff[strong] <= 1e-5) * 100:.3f} %; max {diff.max():.2e}, rms {np.sqrt(np.mean(diff * diff)):.2e}; moved {moved * 100:.1f} %")
        assert abs(loss_g - loss_c) <= 1e-4 * abs(loss_c)
        assert gerr < 1e-2
        assert np.mean(diff <= 1e-5) >= 0.90                             # measured 98.5 / 96.8 / 94.9 %: the differences compound through m and G
        assert np.sqrt(np.mean(diff * diff)) < 2e-4                      # measured 1.5e-5 / 2.9e-5 / 5.6e-5
        if it == 0:
            assert strong.sum() > 1000 and np.mean(diff[strong] <= 1e-5) >= 0.999   # away from the knee: measured 100 % of 18 116 texels
    assert moved > 0.005                                                 # the optimiser really moved the texels the camera and the bounces see
    assert np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)) > 0.999


# ---- a17: PixelLossPT with samples that are not finite ------------------------------------------------------------------------------------
def _poisoned(width=40, height=40):
    """test_035 with a block of NaN texels in the differentiable albedo: every path that fetches one carries a NaN radiance into
    PixelLossPT (what a 0/0 inside a BSDF does on the big scenes, forced here)."""
    sc = load_hydra_xml(scene_path("test_035"), width, height)
    rng = np.random.default_rng(8)
    data = rng.uniform(0.2, 0.9, (256, 256, 4)).astype(np.float32)
    data[96:160, 96:160, :3] = np.nan
    ref = rng.uniform(0.0, 0.5, (height, width, 4)).astype(np.float32)
    return sc, data.reshape(-1), ref


@pytest.mark.parametrize("schedule", [1, 2])
def test_nonfinite_samples_follow_the_reference_by_default(schedule):
    """PixelLossPT (integrator_dr.cpp:1103-1132) has no guard: a NaN sample goes into out_color, the loss and the gradient. That is the
    default here as well, identical to the oracle: same NaN pixels, NaN loss, the same texels poisoned. hpt_set_option("dr_skip_nonfinite", 1)
    is the opt-in deviation for optimisation loops (such a sample contributes nothing); the oracle mirrors it and both agree there too."""
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    sc, data, ref = _poisoned()
    spp = 3
    for skip in (0, 1):
        gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc)
        gpu.set_schedule(schedule, 56, 0, 1)
        gpu.PutDiffTex2D(1, 256, 256, 4); cpu.put_diff_tex2d(1, 256, 256, 4)
        gpu.set_option("dr_skip_nonfinite", skip); cpu.set_option("dr_skip_nonfinite", skip)
        out_g, out_c = np.zeros((sc.height, sc.width, 4), np.float32), np.zeros((sc.height, sc.width, 4), np.float32)
        grad_g = np.zeros_like(data)
        loss_g = gpu.PathTraceDR(gpu.N, 4, out_g, spp, ref, data, grad_g)
        loss_c, grad_c = cpu.path_trace_dr(out_c, spp, ref, data)
        assert gpu.last_schedule()[0] == schedule
        nan_g, nan_c = np.isnan(out_g[..., :3]).any(-1), np.isnan(out_c[..., :3]).any(-1)
        gn_g, gn_c = np.isnan(grad_g), np.isnan(grad_c)
        print(f"skip={skip}: loss gpu={loss_g} cpu={loss_c}; NaN pixels gpu={int(nan_g.sum())} cpu={int(nan_c.sum())}; NaN texels gpu={int(gn_g.sum())} cpu={int(gn_c.sum())}")
        assert np.array_equal(gpu.random_gens(), cpu.random_gens())
        if skip == 0:
            assert np.isnan(loss_g) and np.isnan(loss_c)
            assert nan_c.sum() > 20 and np.array_equal(nan_g, nan_c)
            assert gn_c.sum() > 100 and np.array_equal(gn_g, gn_c)
        else:
            assert np.isfinite(loss_g) and abs(loss_g - loss_c) <= 1e-4 * abs(loss_c)
            assert not nan_g.any() and not nan_c.any() and not gn_g.any() and not gn_c.any()
        ok = ~nan_c
        d = (out_g[..., :3][ok] - out_c[..., :3][ok]) / spp
        assert np.sqrt(np.mean(np.sum(d * d, -1))) < 1e-3
        fin = ~gn_c & ~gn_g
        assert np.linalg.norm(grad_g[fin] - grad_c[fin]) <= 1e-2 * np.linalg.norm(grad_c[fin])


def test_dr_refuses_environment_maps():
    """The DR kernels add the constant m_envColor; the reference's replay evaluates EnvironmentColor() with the map and its MIS weight
    (integrator_dr.cpp:1077-1098). Scenes with a map, its sampling or a back plate are refused instead of differentiated differently."""
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    sc = load_hydra_xml(scene_path("env_map"), 32, 32)
    gpu = HipIntegrator(sc)
    img = np.zeros((32, 32, 4), np.float32)
    with pytest.raises(HydraHipError, match="not differentiated|gltf"):
        gpu.PathTraceDR(gpu.N, 4, img, 1, img, np.zeros(4, np.float32), np.zeros(4, np.float32))


# ---- BASELINE configs[4] (C5) at its full size: properties ----------------------------------------------------------------------------------
def test_c5_full_size_properties():
    """1 M triangles, ONE 4096 x 4096 x 4 fp32 albedo (67 M parameters) bound to the 32 gltf materials, 1920 x 1080, 2 spp: far beyond what
    the oracle finishes in seconds, so size-independent properties instead - the DR wavefront schedule and the DR megakernel give the same
    colours and generators bit for bit, the gradient is finite, and the two gradients (float atomics in different orders) agree to 1e-5 of
    their norm; so do the losses."""
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd import synth
    W, H, ts, spp = 1920, 1080, 4096, 2
    sc = synth.interior_scene(W, H, tex_size=ts)
    rng = np.random.default_rng(2)
    ref = rng.uniform(0.0, 0.5, (H, W, 4)).astype(np.float32)
    data = sc.textures[1].data.reshape(-1).astype(np.float32).copy()            # the generated checker: colours vary over the texture
    res = []
    for schedule in (1, 2):
        gpu = HipIntegrator(sc)
        gpu.set_schedule(schedule)
        off, size = gpu.PutDiffTex2D(1, ts, ts, 4)
        assert (off, size) == (0, ts * ts * 4) and data.size == size
        d = [gpu.dev_array(a) for a in (np.zeros((H, W, 4), np.float32), ref, data)]
        dgrad, dloss = gpu.dev_array(np.zeros(size, np.float32)), gpu.dev_array(np.zeros(1, np.float32))
        gpu.PathTraceDR_dev(d[0], spp, d[1], d[2], dgrad, dloss)
        res.append((d[0].download(), dgrad.download(), float(dloss.download()[0]) / (W * H), gpu.random_gens(), gpu.last_schedule()[0]))
        del gpu
    (f1, g1, l1, r1, s1), (f2, g2, l2, r2, s2) = res
    assert (s1, s2) == (1, 2)
    assert np.isfinite(g1).all() and np.isfinite(g2).all() and np.isfinite(f1).all()
    assert np.array_equal(f1, f2)
    assert np.array_equal(r1, r2)
    n1 = float(np.linalg.norm(g1.astype(np.float64)))
    dn = float(np.linalg.norm(g1.astype(np.float64) - g2.astype(np.float64)))
    print(f"loss mega={l1:.6f} wavefront={l2:.6f}; |g| = {n1:.4e}, |g_wf - g_mega| = {dn:.3e} ({dn / n1:.2e}); touched texels {np.count_nonzero(g1) // 3}")
    assert np.count_nonzero(g1) > 1_000_000
    assert dn <= 1e-5 * n1
    assert abs(l1 - l2) <= 1e-5 * abs(l1)
    assert np.all(g1.reshape(-1, 4)[:, 3] == 0)
