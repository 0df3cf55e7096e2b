"""Thin films (MAT_TYPE_THIN_FILM: include/cmat_film.h, include/airy_reflectance.h; dispatch integrator_pt_mat.cpp:197-249, 422-470) of the HIP
path against the CPU oracle, in RGB mode (the loader's thickness x angle / angle tables in the MODE 4 / 5 / 6 kernels) and in spectral mode
(wavelength x angle tables, or the Airy summation per vertex for one film with a thickness map). The reference ships no film scene: the
fixtures are tests/golden/make_film_scene.py's, built on the reference's spectral Cornell box."""
import numpy as np
import pytest

from conftest import scene_path, pixel_errors, assert_pixel_parity
from hydracore3_amd.scene import load_hydra_xml, MAT_TYPE_THIN_FILM, FILM_PRECOMP_FLAG, FILM_TRANSPARENT

pytestmark = pytest.mark.gpu


def _l2(a, b, spp):
    """the per-pixel bar: the LARGEST L2 norm of a pixel's RGB difference between the spp-normalised frames (conftest.pixel_errors)"""
    return float(pixel_errors(a, b, spp).max())


def _compare(tag, a, b, spp, max_apart):
    """The smooth films are trig-free and agree to float rounding in every pixel. A rough film draws its micro-normal through sinf / cosf,
    the device's and glibc's differ in the last place, and two refractions through a sphere amplify that: about 3 paths in 100 000 end up on the
    other side of a silhouette and change their pixel by a light's worth (profiles/dbg_film.py lists them). Those pixels are counted; the
    others must agree closely."""
    d = np.sqrt(np.sum(((a[..., :3].astype(np.float64) - b[..., :3]) / spp) ** 2, axis=-1))
    scale = max(float(b[..., :3].mean() / spp), 1.0)
    apart = d > 1e-3 * scale
    rest = float(np.sqrt(np.mean(d[~apart] ** 2)))
    print(f"{tag}: per-pixel L2 {np.sqrt(np.mean(d * d)):.3e} (mean radiance {b[..., :3].mean() / spp:.4f}); {int(apart.sum())} of {d.size} pixels apart, L2 of the others {rest:.3e}, "
          f"bit-identical pixels {np.mean(np.all(a[..., :3] == b[..., :3], axis=-1)) * 100:.2f} %")
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    assert int(apart.sum()) <= max_apart
    assert rest < 5e-5 * scale


def _pair(sc, **kw):
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    return HipIntegrator(sc, **kw), OracleIntegrator(sc)


def _brighten(sc, k):
    """the fixture's light is sized for its emission spectrum; in RGB mode its plain colour leaves the frame at 0.007: scale it up so that the
    tolerance below means something"""
    for L in sc.lights:
        L["intensity"] = L["intensity"] * np.float32(k)
    for m in sc.materials:
        if int(m["mtype"]) == 0xEFFFFFFF:
            m["colors"][0] = m["colors"][0] * np.float32(k)


@pytest.mark.parametrize("name", ["thin_film", "thin_film_rough"])
@pytest.mark.parametrize("spectral", [False, True])
@pytest.mark.parametrize("layout", [0, 1, 2])
def test_film_fixtures_match_oracle(name, spectral, layout):
    sc = load_hydra_xml(scene_path(name), 96, 96, spectral=spectral)
    films = [m for m in sc.materials if int(m["mtype"]) == MAT_TYPE_THIN_FILM]
    assert len(films) == 3
    if not spectral:
        _brighten(sc, 60.0)
        assert all(int(m["data"][FILM_PRECOMP_FLAG].view(np.uint32)) == 1 for m in films)            # RGB: every film reads a table
    elif name == "thin_film":
        assert sorted(int(m["data"][FILM_PRECOMP_FLAG].view(np.uint32)) for m in films) == [0, 1, 1]  # one film + thickness map: no table
    gpu, cpu = _pair(sc, accel_layout=layout)
    spp = 16
    a, b = gpu.render(spp), cpu.render(spp)
    _compare(f"{name} spectral={spectral} layout {layout}", a, b, spp, 0 if name == "thin_film" else 30)
    assert float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1))) > 0.995


def test_naive_and_other_integrators_with_films():
    """NaivePathTrace (MODE 5) and the other m_intergatorType values of PathTrace run the same film code."""
    from hydracore3_amd.scene import INTEGRATOR_SHADOW_PT, INTEGRATOR_STUPID_PT
    from hydracore3_amd.api import HipIntegrator
    from oracle.orc import OracleIntegrator
    sc = load_hydra_xml(scene_path("thin_film"), 64, 64, spectral=False)
    _brighten(sc, 60.0)
    for params, naive in ((sc.params(integrator=INTEGRATOR_STUPID_PT), True), (sc.params(integrator=INTEGRATOR_SHADOW_PT), False), (sc.params(integrator=INTEGRATOR_STUPID_PT), False)):
        gpu, cpu = HipIntegrator(sc, params), OracleIntegrator(sc, params)
        a, b = gpu.render(16, naive=naive), cpu.render(16, naive=naive)
        _compare(f"integrator {params.integratorType} naive={naive}", a, b, 16, 2)     # (one path of 65 536 takes another turn in the naive run)
    rough = load_hydra_xml(scene_path("thin_film_rough"), 64, 64, spectral=False)
    _brighten(rough, 60.0)
    p = rough.params(integrator=INTEGRATOR_STUPID_PT)
    gpu, cpu = HipIntegrator(rough, p), OracleIntegrator(rough, p)
    _compare("rough films, naive", gpu.render(16, naive=True), cpu.render(16, naive=True), 16, 12)


def test_film_scene_with_other_mode_tables_is_refused():
    """LoadScene sizes m_precomp_thin_films by m_spectral_mode: a scene loaded for RGB rendering and switched to spectral mode (or back) would
    read tables of the other shape - refused with a message instead."""
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    sc = load_hydra_xml(scene_path("thin_film"), 32, 32, spectral=False)
    sc.spectral_mode = 1
    with pytest.raises(HydraHipError, match="thin film"):
        HipIntegrator(sc).render(1)
    sc = load_hydra_xml(scene_path("thin_film_rough"), 32, 32, spectral=True)
    sc.spectral_mode = 0
    with pytest.raises(HydraHipError, match="thin film"):
        HipIntegrator(sc).render(1)


def test_film_indices_are_validated():
    from hydracore3_amd.api import HipIntegrator, HydraHipError
    sc = load_hydra_xml(scene_path("thin_film"), 32, 32, spectral=False)
    sc.films_eta_k = sc.films_eta_k[:-1]                               # the last film's k offset now reaches past the vector
    with pytest.raises(HydraHipError, match="thin film"):
        HipIntegrator(sc)


def test_opaque_and_transparent_switch():
    """FILM_TRANSPARENT = 0 on a dielectric substrate reflects everything (cmat_film.h:145-153); still equal to the oracle"""
    sc = load_hydra_xml(scene_path("thin_film"), 64, 64, spectral=False)
    _brighten(sc, 60.0)
    for m in sc.materials:
        if int(m["mtype"]) == MAT_TYPE_THIN_FILM:
            m["data"][FILM_TRANSPARENT] = np.uint32(0).view(np.float32)
    gpu, cpu = _pair(sc)
    _compare("opaque films", gpu.render(8), cpu.render(8), 8, 0)


def test_films_as_blend_leaves():
    """A film under MAT_TYPE_BLEND: the descent draws its generator step, the film leaf samples (integrator_pt_mat.cpp:123-130, 197-249) and
    MaterialEval's tree walk reaches filmRoughEval through the (id, weight) stack (:316-333, 422-470)."""
    from hydracore3_amd import scene as S
    sc = load_hydra_xml(scene_path("thin_film"), 64, 64, spectral=False)
    _brighten(sc, 60.0)
    ids = [i for i, m in enumerate(sc.materials) if int(m["mtype"]) == MAT_TYPE_THIN_FILM]
    white = next(i for i, m in enumerate(sc.materials) if int(m["mtype"]) == S.MAT_TYPE_DIFFUSE)
    for i in ids[:2]:                                                    # the two wall films become film / diffuse mixes
        sc.materials.append(sc.materials[i].copy())
        sc.materials[i] = S.material_blend(white, len(sc.materials) - 1, 0.6)
    gpu, cpu = _pair(sc)
    _compare("films under blends", gpu.render(16), cpu.render(16), 16, 8)
    assert float(np.mean(np.all(gpu.random_gens() == cpu.random_gens(), axis=1))) > 0.995


def test_films_under_the_wavefront_schedule_and_with_a_moving_instance():
    """wfShadeKernel<.., FILM> renders the frame the megakernel renders, bit for bit, in both layouts; the sphere given a second key matrix
    (AddInstanceMotion) runs the FILM + MOTION variants of both schedules, against the oracle."""
    from hydracore3_amd.api import HipIntegrator
    from hydracore3_amd import scene as S
    sc = load_hydra_xml(scene_path("thin_film_rough"), 80, 80, spectral=False)
    _brighten(sc, 60.0)
    for layout in (1, 2):
        mega, wf = HipIntegrator(sc, accel_layout=layout), HipIntegrator(sc, accel_layout=layout)
        mega.set_schedule(1); wf.set_schedule(2)
        a, b = mega.render(8), wf.render(8)
        assert mega.last_schedule()[0] == 1 and wf.last_schedule()[0] == 2
        assert np.array_equal(a, b) and np.array_equal(mega.random_gens(), wf.random_gens())
    sphere = max(range(len(sc.inst_geom)), key=lambda i: sc.geom_tri_count[sc.inst_geom[i]])
    m1 = np.asarray(sc.inst_matrices[sphere], np.float64).reshape(4, 4)
    sc.inst_motion = {sphere: S.translate(1.2, 0.4, 0.0) @ m1}
    gpu, cpu = _pair(sc)
    a, b = gpu.render(16), cpu.render(16)
    _compare("film sphere in motion", a, b, 16, 30)
    for layout in (1, 2):
        w2 = HipIntegrator(sc, accel_layout=layout); w2.set_schedule(2)
        assert np.array_equal(w2.render(16), a)
    still = load_hydra_xml(scene_path("thin_film_rough"), 80, 80, spectral=False)
    _brighten(still, 60.0)
    assert _l2(HipIntegrator(still).render(16), a, 16) > 1e-2            # ... and the motion is in the frame
