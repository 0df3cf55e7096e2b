"""Spectral rendering (m_spectral_mode = 1) of the HIP path against the CPU oracle: the reference's own spectral fixture
(scenes/test_spectral/spectral_cornell_conductor.xml: reflectance spectra on the walls, a rough conductor with eta / k spectra, an area
light with an emission spectrum), its variants, the three framebuffer forms of kernel_ContributeToImage and the refusals."""
import os

import numpy as np
import pytest

from conftest import scene_path, pixel_errors, assert_pixel_parity
from hydracore3_amd.scene import load_hydra_xml, MAT_TYPE_CONDUCTOR

pytestmark = pytest.mark.gpu

SPECTRAL_XML = scene_path("test_spectral")


def _l2(a, b, spp):
    """the per-pixel bar: the LARGEST L2 norm of a pixel's RGB difference between the spp-normalised frames (conftest.pixel_errors)"""
    return float(pixel_errors(a, b, spp).max())


def _pair(sc, **kw):
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    return HipIntegrator(sc, **kw), OracleIntegrator(sc)


@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_spectral_fixture_matches_oracle(layout):
    """Four wavelengths per path, spectra looked up per wavelength, CIE observer at the end: per-pixel L2 < 1e-3 of the mean radiance
    scale, most pixels bit-identical, generator streams identical wherever the paths did not diverge - in every acceleration layout."""
    sc = load_hydra_xml(SPECTRAL_XML, 96, 96, spectral=True)
    gpu, cpu = _pair(sc, accel_layout=layout)
    spp = 16
    a, b = gpu.render(spp), cpu.render(spp)
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    l2 = _l2(a, b, spp)
    same = float(np.mean(np.all(a[..., :3] == b[..., :3], axis=-1)))
    same_rng = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
    print(f"layout {layout}: per-pixel L2 = {l2:.3e} (mean {b[..., :3].mean() / spp:.4f}), bit-identical pixels {same * 100:.2f} %, identical generators {same_rng * 100:.2f} %")
    assert l2 < 1e-3
    assert same > 0.2
    assert same_rng > 0.99


def test_spectral_draws_one_more_random_number_per_path():
    """GetRandomNumbersSpec (integrator_pt.cpp:116-118): with a trace depth of 0 a path is the lens draw plus the wavelength draw, so the
    generators after one pass differ from the RGB mode's and equal the oracle's bit for bit."""
    sc = load_hydra_xml(SPECTRAL_XML, 32, 32, spectral=True)
    sc.trace_depth = 0
    gpu, cpu = _pair(sc)
    gpu.render(1); cpu.render(1)
    assert np.array_equal(gpu.random_gens(), cpu.random_gens())
    rgb = load_hydra_xml(SPECTRAL_XML, 32, 32, spectral=False)
    rgb.trace_depth = 0
    g2, _ = _pair(rgb)
    g2.render(1)
    assert not np.array_equal(g2.random_gens(), gpu.random_gens())


def test_monochrome_and_wavelength_layer_framebuffers():
    """kernel_ContributeToImage's other two forms (integrator_pt.cpp:631-654): channels == 1 adds the first wavelength's sample, channels > 4
    adds every sample to the layer of its wavelength bin ([channels][H][W])."""
    sc = load_hydra_xml(SPECTRAL_XML, 64, 64, spectral=True)
    spp = 8
    for channels in (1, 3, 16):
        gpu, cpu = _pair(sc)
        a = np.zeros(64 * 64 * channels, np.float32); b = a.copy()
        gpu.PathTraceBlock(gpu.N, channels, a, spp)
        cpu.path_trace_block(b, spp, channels=channels)
        assert np.isfinite(a).all() and a.sum() > 0
        if channels == 16:
            la, lb = a.reshape(16, 64, 64), b.reshape(16, 64, 64)
            assert (la.sum(axis=(1, 2)) > 0).all()                    # every wavelength bin received samples
            d = (la.astype(np.float64) - lb) / spp
            err = float(np.sqrt(np.mean(np.sum(d * d, axis=0))))
            scale = float(np.sum(lb, axis=0).mean() / spp)
        else:
            ia, ib = a.reshape(64, 64, channels), b.reshape(64, 64, channels)
            d = (ia.astype(np.float64) - ib) / spp
            err = float(np.sqrt(np.mean(np.sum(d * d, axis=-1))))
            scale = float(ib.sum(axis=-1).mean() / spp)
        print(f"channels {channels}: per-pixel L2 {err:.3e}, scale {scale:.4f}, bit-identical values {np.mean(a == b) * 100:.2f} %")
        assert err < 1e-3 * max(scale, 1.0)
        assert np.mean(a == b) > 0.5
        assert np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)) > 0.99


def test_spectral_accumulates_and_tid_windows_compose():
    sc = load_hydra_xml(SPECTRAL_XML, 64, 64, spectral=True)
    from hydracore3_amd.api import HipIntegrator
    one, two = HipIntegrator(sc), HipIntegrator(sc)
    full = one.render(6)
    img = np.zeros_like(full)
    half = (two.N // 2 // 64) * 64
    for _ in range(2):                                                  # 2 x 3 passes over two windows == 6 passes over the frame
        two.PathTraceBlock(half, 4, img, 3, tid_begin=0)
        two.PathTraceBlock(two.N - half, 4, img, 3, tid_begin=half)
    assert np.array_equal(img, full)
    assert np.array_equal(one.random_gens(), two.random_gens())


def test_smooth_conductor_and_constant_parameters():
    """The fixture's conductor made mirror-smooth (conductorSmoothSampleAndEval with eta / k per wavelength) and a conductor without spectra
    (SampleMatParamSpectrum falls back to the scalar eta / k)."""
    for variant in ("smooth", "no-spectra"):
        sc = load_hydra_xml(SPECTRAL_XML, 64, 64, spectral=True)
        n = 0
        for m in sc.materials:
            if int(m["mtype"]) == MAT_TYPE_CONDUCTOR:
                n += 1
                if variant == "smooth": m["data"][0] = m["data"][1] = 0.0
                else: m["spdid"][0] = m["spdid"][1] = 0xFFFFFFFF
        assert n == 1
        gpu, cpu = _pair(sc)
        a, b = gpu.render(8), cpu.render(8)
        l2 = _l2(a, b, 8)
        print(f"{variant}: per-pixel L2 = {l2:.3e}, identical generators {np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)) * 100:.2f} %")
        assert l2 < 1e-3


def test_spectral_plastic_matches_oracle():
    """The `plastic` node in spectral mode (integrator_pt_mat.cpp:264-275, 486-500; mi::fresnel_coat_precompute's spectral branch,
    mi_materials.cpp:383-404): a rough plastic sphere with a reflectance spectrum and the nonlinear colour shift, smooth grey plastic walls
    without a spectrum (tests/golden/scenes/spectral_plastic, derived from the reference's spectral fixture by make_spectral_plastic_scene.py).
    The kernel runs the RGB routine twice per vertex - (x, y, z), then w - which is the float4 arithmetic channel by channel."""
    xml = scene_path("spectral_plastic")
    for spectral in (True, False):
        sc = load_hydra_xml(xml, 96, 96, spectral=spectral)
        gpu, cpu = _pair(sc)
        a, b = gpu.render(12), cpu.render(12)
        l2 = _l2(a, b, 12)
        same = float(np.mean(np.all(a[..., :3] == b[..., :3], axis=-1)))
        gens = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
        print(f"spectral={spectral}: per-pixel L2 = {l2:.3e} (mean {b[..., :3].mean() / 12:.4f}), bit-identical pixels {same * 100:.2f} %, identical generators {gens * 100:.2f} %")
        assert np.isfinite(a).all() and a[..., :3].mean() > 0
        assert l2 < 1e-3 and same > 0.1 and gens > 0.99
    # the sampling weight of the spectral precomputation: 1 / (mean of the reflectance spectrum + 1), 1 / 1.5 without a spectrum
    sc = load_hydra_xml(xml, 32, 32, spectral=True)
    w = sorted(float(m["data"][2]) for m in sc.materials if int(m["mtype"]) == 5)
    assert w[0] == pytest.approx(1.0 / 1.5) and 0.8 < w[1] < 0.95


def test_dispersive_dielectric_matches_oracle():
    """A smooth dielectric whose interior IOR is a spectrum (cmat_dielectric.h:8-56, integrator_pt_mat.cpp:277-287): the first wavelength's IOR
    bends the ray, the path is marked RAY_FLAG_WAVES_DIVERGED and SpectrumToXYZ keeps that wavelength only, at four times the weight
    (spectrum.h:157-170). Own fixture (make_spectral_plastic_scene.py): the reference's spectral Cornell box with a flint-like glass sphere."""
    xml = scene_path("spectral_glass")
    for spectral, channels in ((True, 4), (True, 16), (False, 4)):
        sc = load_hydra_xml(xml, 96, 96, spectral=spectral)
        gpu, cpu = _pair(sc)
        a = np.zeros(96 * 96 * channels, np.float32); b = a.copy()
        gpu.PathTraceBlock(gpu.N, channels, a, 12)
        cpu.path_trace_block(b, 12, channels=channels)
        gens = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
        if channels == 4:
            ia, ib = a.reshape(96, 96, 4), b.reshape(96, 96, 4)
            l2 = _l2(ia, ib, 12)
        else:
            d = (a.reshape(16, 96, 96).astype(np.float64) - b.reshape(16, 96, 96)) / 12
            l2 = float(np.sqrt(np.mean(np.sum(d * d, axis=0))))
        print(f"spectral={spectral}, channels={channels}: per-pixel L2 = {l2:.3e}, bit-identical values {np.mean(a == b) * 100:.2f} %, identical generators {gens * 100:.2f} %")
        assert np.isfinite(a).all() and a.sum() > 0
        assert l2 < 1e-3 and np.mean(a == b) > 0.3 and gens > 0.99
    # dispersion is there: the sphere's pixels differ between the dispersive glass and the same glass with a constant IOR
    sc = load_hydra_xml(xml, 64, 64, spectral=True)
    from hydracore3_amd.api import HipIntegrator
    disp = HipIntegrator(sc).render(16)
    for m in sc.materials:
        if int(m["mtype"]) == 7: m["spdid"][0] = 0xFFFFFFFF
    flat = HipIntegrator(sc).render(16)
    assert not np.array_equal(disp, flat)


def test_environment_spectrum_matches_oracle():
    """A `sky` light whose colour carries a spectrum (m_envSpecId / m_envSpecMult, integrator_pt_scene.cpp:456-457): rays that leave the open
    Cornell box pick up SampleUniformSpectrum(envSpec) * mult / 106.856895 (integrator_pt_lgt.cpp:181-188). Own fixture spectral_sky."""
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    xml = scene_path("spectral_sky")
    for spectral in (True, False):
        sc = load_hydra_xml(xml, 96, 96, spectral=spectral)
        assert sc.env_spec_id == 7 and sc.params().envSpecIdPlus1 == 8 and sc.params().envSpecMult == pytest.approx(0.8)
        gpu, cpu = _pair(sc)
        a, b = gpu.render(12), cpu.render(12)
        l2 = _l2(a, b, 12)
        gens = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
        print(f"spectral={spectral}: per-pixel L2 = {l2:.3e} (mean {b[..., :3].mean() / 12:.4f}), identical generators {gens * 100:.2f} %")
        assert l2 < 1e-3 and gens > 0.99 and np.mean(np.all(a[..., :3] == b[..., :3], axis=-1)) > 0.2
    # the sky is seen: brighter than the same scene without its spectrum id
    sc = load_hydra_xml(xml, 64, 64, spectral=True)
    with_sky = HipIntegrator(sc).render(8)[..., :3].mean()
    sc.env_spec_id = 0xFFFFFFFF; sc.env_color = (0.0, 0.0, 0.0, 0.0)
    assert with_sky > 1.2 * HipIntegrator(sc).render(8)[..., :3].mean()
    sc.env_spec_id = 99
    with pytest.raises(HydraHipError, match="m_envSpecId"):
        HipIntegrator(sc)


def test_camera_response_spectra():
    """SpectralCamRespoceToRGB with m_camResponseSpectrumId set (integrator_spectrum.cpp:76-121): the response spectra replace the CIE
    observer; both response types (0 = CAM_RESPONCE_XYZ: through XYZToRGB, 1 = CAM_RESPONCE_RGB: taken as it is)."""
    for ids, rtype in (((1, 2, 3), 0), ((1, -1, -1), 1)):
        sc = load_hydra_xml(SPECTRAL_XML, 48, 48, spectral=True)
        sc.cam_response_spectrum_id = ids
        sc.cam_response_type = rtype
        gpu, cpu = _pair(sc)
        a, b = gpu.render(8), cpu.render(8)
        scale = max(float(np.abs(b[..., :3]).mean() / 8), 1.0)
        assert _l2(a, b, 8) < 1e-3 * scale
        assert np.mean(np.all(a[..., :3] == b[..., :3], axis=-1)) > 0.2


def test_same_scene_in_rgb_mode_still_matches():
    sc = load_hydra_xml(SPECTRAL_XML, 64, 64, spectral=False)
    gpu, cpu = _pair(sc)
    a, b = gpu.render(8), cpu.render(8)
    assert _l2(a, b, 8) < 1e-3


def test_spectral_mode_refusals():
    """What the spectral kernel does not cover is refused by name (never rendered in RGB instead)."""
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    # a scene that came without spectral tables
    sc = load_hydra_xml(SPECTRAL_XML, 32, 32, spectral=True)
    sc.spec_offset_sz = []
    with pytest.raises(HydraHipError, match="spectral"):
        HipIntegrator(sc)
    # the input-ray integrator with three channels (kernel_CopyColorToOutput knows one and four)
    sc = load_hydra_xml(SPECTRAL_XML, 32, 32, spectral=True)
    gpu = HipIntegrator(sc)
    rays = np.zeros((gpu.N, 4), np.float32); rays[:, 2] = -1.0
    with pytest.raises(HydraHipError, match="spectral"):
        gpu.PathTraceFromInputRaysBlock(gpu.N, 3, np.zeros((gpu.N, 4), np.float32), rays, np.zeros((gpu.N, 3), np.float32), 1)
    # more than four channels only in spectral mode
    rgb = HipIntegrator(load_hydra_xml(SPECTRAL_XML, 32, 32, spectral=False))
    with pytest.raises(HydraHipError, match="channels"):
        rgb.PathTraceBlock(rgb.N, 16, np.zeros(32 * 32 * 16, np.float32), 1)
    # a spectrum id past the table
    sc = load_hydra_xml(SPECTRAL_XML, 32, 32, spectral=True)
    sc.lights[0]["specId"] = 1000
    with pytest.raises(HydraHipError, match="spectrum"):
        HipIntegrator(sc)


@pytest.mark.parametrize("name", ["test_035", "test_228"])
def test_gltf_scenes_in_spectral_mode(name):
    """Legacy hydra_material / gltf surfaces under m_spectral_mode = 1 (integrator_pt_mat.cpp:170-176, 385-394): the base colour times the texel is
    carried as four spectral samples as it is (GetColorFromNode warns, integrator_pt_scene_mat.cpp:133-136), lights without a spectrum keep
    their colour - the reference's Cornell box and its IES scene through the spectral kernel, against the oracle."""
    sc = load_hydra_xml(scene_path(name), 96, 96, spectral=True)
    gpu, cpu = _pair(sc)
    spp = 16
    a, b = gpu.render(spp), cpu.render(spp)
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    l2 = _l2(a, b, spp)
    same_rng = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
    print(f"{name}: per-pixel L2 = {l2:.3e} (mean {b[..., :3].mean() / spp:.4f}), identical generators {same_rng * 100:.2f} %")
    assert l2 < 1e-3 * max(float(b[..., :3].mean() / spp), 1.0)
    assert same_rng > 0.99


@pytest.mark.parametrize("name", ["legacy_materials", "typed_materials", "env_map"])
def test_every_material_type_in_spectral_mode(name):
    """The own fixtures with every material constructor - legacy glass, Lambert / metal mixes, coated plastic, gltf with four textures, rough
    conductors, Oren-Nayar, dielectric, plastic, BLENDS (nested, texture-masked) and NORMAL MAPS, a MOVING instance (legacy_materials), a sampled
    HDR environment map with a camera back plate seen through a LENS stack (env_map) - under m_spectral_mode = 1: the spectral kernel walks blend
    trees, bends normals, refracts through the legacy glass, interpolates instance transforms and weighs the map as the RGB kernels do, on four
    samples per path."""
    sc = load_hydra_xml(scene_path(name), 96, 64, spectral=True)
    assert name != "legacy_materials" or sc.inst_motion
    gpu, cpu = _pair(sc)
    spp = 16
    a, b = gpu.render(spp), cpu.render(spp)
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    d = np.sqrt(np.sum(((a[..., :3].astype(np.float64) - b[..., :3]) / spp) ** 2, axis=-1))
    scale = max(float(b[..., :3].mean() / spp), 1.0)
    apart = d > 1e-3 * scale
    same_rng = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
    print(f"{name}: per-pixel L2 = {np.sqrt(np.mean(d * d)):.3e} (mean {b[..., :3].mean() / spp:.4f}), {int(apart.sum())} pixels apart, L2 of the others {np.sqrt(np.mean(d[~apart] ** 2)):.3e}, identical generators {same_rng * 100:.2f} %")
    assert int(apart.sum()) <= 6 and np.sqrt(np.mean(d[~apart] ** 2)) < 5e-5 * scale
    assert same_rng > 0.99


def test_lens_stack_in_spectral_mode():
    """m_enableOpticSim under m_spectral_mode (integrator_pt.cpp:79-118): the film point goes through the lens stack, then the wavelength draw."""
    sc = load_hydra_xml(SPECTRAL_XML, 64, 64, spectral=True)
    rad, ap = 60.0, 12.0
    sc.set_optics([(0, rad, 4.0, 1.6, ap), (1, -rad * 2.5, 2.0, 1.0, ap), (2, 0.0, 2.0, 0.0, 6.0), (3, rad * 2.5, 4.0, 1.6, ap), (4, -rad, 42.0, 1.0, ap)], 0.035, 0.001, "scene_to_sensor")
    gpu, cpu = _pair(sc)
    a, b = gpu.render(16), cpu.render(16)
    l2 = _l2(a, b, 16)
    print(f"lens, spectral: per-pixel L2 = {l2:.3e} (mean {b[..., :3].mean() / 16:.4f})")
    assert np.isfinite(a).all() and a[..., :3].mean() > 0 and l2 < 1e-3
    assert np.array_equal(gpu.random_gens(), cpu.random_gens())
    plain = load_hydra_xml(SPECTRAL_XML, 64, 64, spectral=True)
    from hydracore3_amd.api import HipIntegrator
    assert _l2(HipIntegrator(plain).render(16), a, 16) > 1e-2           # ... and the lens is in the frame


@pytest.mark.parametrize("seed", list(range(10)))
def test_fuzzed_scenes_in_spectral_mode(seed):
    """synth.random_scene under m_spectral_mode = 1: every material constructor at its corners, blends, normal maps, films, every light type,
    projected textures, HDR environment maps, moving instances, lens stacks, random depth and integrator - the spectral kernel against the oracle."""
    from hydracore3_amd import synth
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    from hydracore3_amd.scene import INTEGRATOR_MIS_PT, INTEGRATOR_SHADOW_PT, INTEGRATOR_STUPID_PT
    sc = synth.random_scene(seed, spectral=True)
    p = sc.params([INTEGRATOR_MIS_PT, INTEGRATOR_SHADOW_PT, INTEGRATOR_MIS_PT, INTEGRATOR_STUPID_PT][seed % 4])
    gpu, cpu = HipIntegrator(sc, p), OracleIntegrator(sc, p)
    spp = 4
    a, b = gpu.render(spp), cpu.render(spp)
    assert np.isfinite(b).all() and np.isfinite(a).all()
    # a path that takes another branch somewhere (device vs glibc sinf / cosf / powf: see test_fuzzed_scenes_match_oracle) leaves its pixel's
    # generator in another state: such pixels are counted and left out of the norm (a single sample of 4 may carry a light's worth per wavelength)
    eq = np.all(gpu.random_gens() == cpu.random_gens(), axis=1)
    xy = cpu.packed_xy()
    ok = np.zeros(sc.width * sc.height, bool); ok[(xy >> 16).astype(np.int64) * sc.width + (xy & 0xFFFF).astype(np.int64)] = eq
    d = ((a[..., :3].astype(np.float64) - b[..., :3]) / spp).reshape(-1, 3)
    l2 = float(np.sqrt(np.mean(np.sum(d[ok] ** 2, axis=-1))))
    differ = int((~eq).sum())
    print(f"seed {seed}: depth {sc.trace_depth}, {len(sc.lights)} lights, L2 {l2:.2e} over the pixels with equal generators, pixels with a divergent path: {differ} of {gpu.N}")
    assert l2 < 1e-3 * max(float(b[..., :3].mean() / spp), 1.0)
    assert differ <= 2


def test_naive_path_trace_in_spectral_mode():
    """NaivePathTraceBlock under m_spectral_mode = 1 (integrator_pt.cpp:681-717 on four wavelengths): no light sampling, one more bounce; the
    reference's fixture and the material fixture against the oracle, and the naive and MIS estimators converge to the same frame."""
    from hydracore3_amd.scene import INTEGRATOR_STUPID_PT
    for name, size in (("test_spectral", 64), ("typed_materials", 64)):
        sc = load_hydra_xml(scene_path(name), size, size, spectral=True)
        p = sc.params(integrator=INTEGRATOR_STUPID_PT)
        from hydracore3_amd.api import HipIntegrator
        from oracle.orc import OracleIntegrator
        gpu, cpu = HipIntegrator(sc, p), OracleIntegrator(sc, p)
        a, b = gpu.render(16, naive=True), cpu.render(16, naive=True)
        l2 = _l2(a, b, 16)
        same_rng = float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1)))
        print(f"{name}, naive, spectral: per-pixel L2 = {l2:.3e} (mean {b[..., :3].mean() / 16:.4f}), identical generators {same_rng * 100:.2f} %")
        assert np.isfinite(a).all() and a[..., :3].mean() > 0 and same_rng > 0.99
        assert l2 < 1e-3 * max(float(b[..., :3].mean() / 16), 1.0)


def test_input_rays_in_spectral_mode():
    """PathTraceFromInputRaysBlock under m_spectral_mode = 1 (integrator_pt.cpp:159-199, 659-676): the cam plugin's rays carry their wavelength
    (RayPosAndW::wave), all four samples of a path sit at it, and kernel_CopyColorToOutput adds the four raw samples - no generator step for the
    lens, the time or the wavelength."""
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    for name in ("test_spectral", "typed_materials"):
        sc = load_hydra_xml(scene_path(name), 64, 64, spectral=True)
        gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc)
        n = 4096
        rng = np.random.default_rng(5)
        pos = np.zeros((n, 4), np.float32); dr = np.zeros((n, 4), np.float32)
        pos[:, :2] = rng.uniform(-0.05, 0.05, (n, 2)); pos[:, 3] = rng.uniform(380.0, 780.0, n)
        d = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), -np.ones(n)], 1)
        dr[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
        for channels in (4, 1):
            gpu.InitRandomGens(gpu.N); cpu.set_random_gens(gpu.random_gens())
            og, oc = np.full((n, channels), 0.25, np.float32), np.full((n, channels), 0.25, np.float32)
            gpu.PathTraceFromInputRaysBlock(n, channels, pos, dr, og, 6)
            cpu.path_trace_from_input_rays_block(pos, dr, oc, 6, channels)
            dlt = (og.astype(np.float64) - oc) / 6
            l2 = float(np.sqrt(np.mean(np.sum(dlt * dlt, -1))))
            print(f"{name}, input rays, {channels} channels: per-ray L2 = {l2:.3e} (mean {oc.mean() - 0.25:.4f})")
            assert np.isfinite(og).all() and l2 < 1e-3 * max(float(np.abs(oc).mean()), 1.0) and float(np.mean(og[:, 0] - 0.25)) > 0.0
            assert np.mean(np.all(gpu.random_gens()[:n] == cpu.random_gens()[:n], axis=1)) > 0.995


@pytest.mark.parametrize("layout", [0, 2])
def test_spectra_given_by_textures(layout):
    """KSPEC_SPD_TEX (SampleMatColorSpectrumTexture, integrator_spectrum.cpp:128-180; LoadSceneSpectrumData's lambda_ref_ids): the walls' reflectance
    from five maps at 400 ... 720 nm (sampler attributes on the <spectrum> node), the plastic sphere's from two; wavelengths outside the bands read
    zero; the interval search keeps its probes inside the table (spectrum.h:42-55 does not). The fixture against the oracle, and a scene loaded
    for RGB rendering (bands unresolved) refused in spectral mode."""
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    sc = load_hydra_xml(scene_path("spectral_textures"), 96, 96, spectral=True)
    assert len(sc.spec_tex_ids_wavelengths) == 7 and sc.spec_tex_offset_sz[7] == (0, 5) and sc.spec_tex_offset_sz[8] == (5, 2)
    gpu, cpu = _pair(sc, accel_layout=layout)
    spp = 16
    a, b = gpu.render(spp), cpu.render(spp)
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    assert_pixel_parity(a, b, spp, gpu, cpu, max_divergent=4, max_over=4, what=f"spectral textures, layout {layout}: ")
    plain = load_hydra_xml(SPECTRAL_XML, 96, 96, spectral=True)
    assert _l2(HipIntegrator(plain).render(spp), a, spp) > 1e-2          # ... and the maps are in the frame
    rgb = load_hydra_xml(scene_path("spectral_textures"), 32, 32, spectral=False)
    HipIntegrator(rgb).render(1)                                         # RGB rendering never looks at the bands
    rgb.spectral_mode = 1
    with pytest.raises(HydraHipError, match="spectral"):
        HipIntegrator(rgb)


def test_offsets_update_onto_a_glass_material_widens_the_spectral_kernel():
    """The spectral kernel's scope is chosen from the materials a hit can REACH at CommitDeviceData. Update_m_matIdOffsets (integrator_pt.h:470)
    re-points a mesh at other triangle ranges - here at those of a twin mesh nothing instances, whose triangles carry a legacy GLASS material -
    so afterwards a hit reaches a material type the chosen scope (diffuse only: no gltf, no glass branch) does not hold. The update must widen
    the scope: the frame equals the oracle's rendering of a scene BUILT with those offsets, and differs from the frame before the update."""
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    from hydracore3_amd import scene as S, synth

    def build(swapped):
        sc = S.SceneData()
        sc.width, sc.height = 64, 48
        sc.cam_pos, sc.cam_look_at, sc.cam_up = (0.0, 1.6, 6.0), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0)
        sc.fov, sc.trace_depth = 40.0, 6
        sc.env_color = (0.2, 0.25, 0.3, 0.0)
        sc.spectral_mode = 1
        sc.spec_offset_sz, sc.spec_values = [(0, 471)], np.ones(471, np.float32)     # the loader's uniform spectrum (integrator_pt_scene.cpp:406-418)
        sc.materials.append(S.material_diffuse((0.7, 0.6, 0.5)))
        sc.materials.append(S.material_diffuse((0.2, 0.5, 0.8)))
        sc.materials.append(S.material_glass((1.0, 1.0, 1.0), (0.9, 1.0, 0.9), 1.5))
        p, n, t, uv, idx = synth._quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2)
        sc.add_instance(sc.add_mesh(p, n, t, uv, idx, [0]), np.eye(4))
        sp = synth._sphere_mesh(2)
        ntri = sp[4].size // 3
        a = sc.add_mesh(sp[0], sp[1], sp[2], sp[3], sp[4], np.full(ntri, 1, np.uint32))        # the instanced sphere: diffuse
        sc.add_mesh(sp[0], sp[1], sp[2], sp[3], sp[4], np.full(ntri, 2, np.uint32))            # its twin, never instanced: glass
        sc.add_instance(a, S.translate(0.0, 1.0, 0.0))
        sc.lights.append(S.light_rect(S.translate(0.0, 4.0, 1.0), 0.8, 0.8, (1, 1, 1), 15.0))
        mvo = np.asarray(sc.mat_vert_offset, np.uint32).reshape(-1, 2).copy()
        if swapped:
            mvo[1] = mvo[2]
            sc.mat_vert_offset = [tuple(int(v) for v in r) for r in mvo]
        return sc, mvo

    sc, mvo = build(False)
    gpu = HipIntegrator(sc)
    spp = 8
    before = gpu.render(spp)
    upd = HipIntegrator(sc)
    swapped = mvo.copy(); swapped[1] = mvo[2]                    # mesh 1 now reads the twin's triangle range: the glass ids
    upd.Update_m_matIdOffsets(swapped)
    after = upd.render(spp)
    sc2, _ = build(True)
    ref = OracleIntegrator(sc2).render(spp)
    built = HipIntegrator(sc2).render(spp)
    assert np.isfinite(after).all() and _l2(after, before, spp) > 1e-2            # the sphere turned to glass
    assert np.array_equal(after, built)                                          # the update == a scene built that way, bit for bit
    assert _l2(after, ref, spp) < 1e-3


@pytest.mark.parametrize("name", ["test_spectral", "spectral_plastic", "spectral_glass", "spectral_sky", "spectral_textures", "typed_materials", "env_map", "thin_film"])
def test_spectral_block_local_schedule_equals_the_plain_kernel(name):
    """m_spectral_mode = 1 under the block-local schedule (pathTraceBlockSpectralKernel: persistent blocks on the work queue, the lanes' rays
    pooled in LDS and drained with ray replacement) and under the wavefront schedule (wfShadeSpecKernel + wfTraceKernel) calls the same
    shadeVertexSpec in the same order as the one-thread-per-pixel kernel: frames and generators are bit-identical in every scope, both
    layouts, for MIS and (block-local) naive path tracing and the three framebuffer forms."""
    from hydracore3_amd.api import HipIntegrator
    sc = load_hydra_xml(scene_path(name), 96, 64, spectral=True)
    plain = HipIntegrator(sc); plain.set_schedule(1)
    ref = plain.render(6)
    assert plain.last_launch()["schedule"] == 1 and np.isfinite(ref).all() and ref[..., :3].mean() > 0
    for layout in (1, 2):
        blk = HipIntegrator(sc, accel_layout=layout); blk.set_schedule(3)
        img = blk.render(6)
        assert blk.last_launch()["schedule"] == 3
        assert np.array_equal(img, ref), layout
        assert np.array_equal(blk.random_gens(), plain.random_gens())
        # ... and under the wavefront schedule (wfShadeSpecKernel + the shared trace kernel; a small trace grid so that rays are suspended and resumed too)
        for trace_blocks in (0, 1):
            wf = HipIntegrator(sc, accel_layout=layout); wf.set_schedule(2, 56, trace_blocks, 1)
            img = wf.render(6)
            assert wf.last_launch()["schedule"] == 2
            assert np.array_equal(img, ref), (layout, trace_blocks)
            assert np.array_equal(wf.random_gens(), plain.random_gens())
    a, b = HipIntegrator(sc), HipIntegrator(sc)
    a.set_schedule(1); b.set_schedule(3)
    assert np.array_equal(a.render(3, naive=True), b.render(3, naive=True)) and np.array_equal(a.random_gens(), b.random_gens())
    for channels in (1, 12):
        fa = np.zeros(96 * 64 * channels, np.float32); fb = fa.copy(); fc = fa.copy()
        a2, b2, c2 = HipIntegrator(sc), HipIntegrator(sc), HipIntegrator(sc)
        a2.set_schedule(1); b2.set_schedule(3); c2.set_schedule(2)
        a2.PathTraceBlock(a2.N, channels, fa, 3); b2.PathTraceBlock(b2.N, channels, fb, 3); c2.PathTraceBlock(c2.N, channels, fc, 3)
        assert np.array_equal(fa, fb) and np.array_equal(fa, fc) and fa.sum() > 0, channels


def test_interior_under_spectral_mode_matches_oracle():
    """A heavy scene under m_spectral_mode = 1: the 17 K-triangle miniature of the interior (SAH estimate ~40, gltf materials carried as four
    samples, the loader's uniform spectrum) takes the block-local spectral kernel on the 4-wide compressed tree by the automatic choice, and
    its frame and generators equal the CPU oracle's (per pixel) and the plain kernel's (bit for bit)."""
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd import synth
    from oracle.orc import OracleIntegrator
    sc = synth.interior_scene(160, 96, subdiv=1, tex_size=16)
    sc.spectral_mode = 1
    sc.spec_offset_sz, sc.spec_values = [(0, 471)], np.ones(471, np.float32)
    gpu, cpu = HipIntegrator(sc), OracleIntegrator(sc, threads=len(os.sched_getaffinity(0)))
    spp = 8
    a, b = gpu.render(spp), cpu.render(spp)
    ll = gpu.last_launch()
    assert ll["schedule"] == 3 and ll["wide_nodes"], ll
    assert np.isfinite(a).all() and a[..., :3].mean() / spp > 0.02
    assert_pixel_parity(a, b, spp, gpu, cpu, max_divergent=2, max_over=8, what="interior 160x96 @ 8 spp under spectral mode, block-local kernel + 4-wide tree: ")
    plain = HipIntegrator(sc); plain.set_schedule(1)
    assert np.array_equal(plain.render(spp), a) and np.array_equal(plain.random_gens(), gpu.random_gens())
    wf = HipIntegrator(sc); wf.set_schedule(2)                  # what a full-size call of this scene takes: wfShadeSpecKernel + the 4-wide trace kernel
    assert np.array_equal(wf.render(spp), a) and np.array_equal(wf.random_gens(), gpu.random_gens())
    assert wf.last_launch()["schedule"] == 2 and wf.last_launch()["wide_nodes"]
