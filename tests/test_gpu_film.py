"""Thin films (MAT_TYPE_THIN_FILM: include/cmat_film.h, include/airy_reflectance.h; dispatch integrator_pt_mat.cpp:197-249, 422-470) of the HIP
path against the CPU oracle, in RGB mode (the loader's thickness x angle / angle tables in the MODE 4 / 5 / 6 kernels) and in spectral mode
(wavelength x angle tables, or the Airy summation per vertex for one film with a thickness map). The reference ships no film scene: the
fixtures are tests/golden/make_film_scene.py's, built on the reference's spectral Cornell box."""
import numpy as np
import pytest

from conftest import scene_path
from hydracore3_amd.scene import load_hydra_xml, MAT_TYPE_THIN_FILM, FILM_PRECOMP_FLAG, FILM_TRANSPARENT

pytestmark = pytest.mark.gpu


def _l2(a, b, spp):
    d = (a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) / spp
    return float(np.sqrt(np.mean(np.sum(d * d, axis=-1))))


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
    assert np.isfinite(a).all() and a[..., :3].mean() > 0
    scale = float(b[..., :3].mean() / spp)
    l2 = _l2(a, b, spp)
    same = float(np.mean(np.all(a[..., :3] == b[..., :3], axis=-1)))
    eq = np.all(gpu.random_gens() == cpu.random_gens(), axis=1)
    same_rng = float(np.mean(eq))
    # a path that takes another branch on the device (its sinf / cosf / acosf / expf differ from glibc's in the last place: a sample on the other
    # side of the reflect / refract choice or of a wo.z test) changes its pixel by a whole light's worth; such pixels are counted, the others
    # must agree closely. The generator of a pixel tells: it has advanced differently where a path took another turn.
    xy = cpu.packed_xy()
    px = (xy >> 16).astype(np.int64) * sc.width + (xy & 0xFFFF).astype(np.int64)
    ok = np.zeros(sc.width * sc.height, bool); ok[px] = eq
    d = ((a[..., :3].astype(np.float64) - b[..., :3]) / spp).reshape(-1, 3)
    l2_same = float(np.sqrt(np.mean(np.sum(d[ok] ** 2, axis=-1))))
    print(f"{name} spectral={spectral} layout {layout}: per-pixel L2 = {l2:.3e} (mean {scale:.4f}; {l2_same:.3e} over the pixels with identical generators), "
          f"bit-identical pixels {same * 100:.2f} %, identical generators {same_rng * 100:.2f} % ({int((~eq).sum())} pixels apart)")
    assert l2_same < 2e-4 * max(scale, 1.0)
    assert l2 < 3e-2 * max(scale, 1.0)
    assert same_rng > 0.995


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
        l2, scale = _l2(a, b, 16), float(b[..., :3].mean() / 16)
        print(f"integrator {params.integratorType} naive={naive}: per-pixel L2 = {l2:.3e}, mean {scale:.4f}")
        assert np.isfinite(a).all() and a[..., :3].mean() > 0 and l2 < 2e-3 * max(scale, 1.0)


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
    a, b = gpu.render(8), cpu.render(8)
    assert _l2(a, b, 8) < 2e-3 * max(float(b[..., :3].mean() / 8), 1.0)
