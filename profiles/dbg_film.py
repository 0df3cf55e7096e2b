"""Where the HIP path and the oracle differ on the thin-film fixtures: pixels apart by more than a threshold, the largest ones listed.
Usage (GPU box): python profiles/dbg_film.py"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import scene_path
from hydracore3_amd.scene import load_hydra_xml, INTEGRATOR_STUPID_PT
from hydracore3_amd.api import HipIntegrator
from oracle.orc import OracleIntegrator


def brighten(sc, k):
    for L in sc.lights:
        L["intensity"] = L["intensity"] * np.float32(k)
    for m in sc.materials:
        if int(m["mtype"]) == 0xEFFFFFFF:
            m["colors"][0] = m["colors"][0] * np.float32(k)


def report(tag, a, b, spp):
    d = np.sqrt(np.sum(((a[..., :3].astype(np.float64) - b[..., :3]) / spp) ** 2, axis=-1))
    scale = float(b[..., :3].mean() / spp)
    bad = d > 1e-3 * max(scale, 1.0)
    print(f"{tag}: scale {scale:.4f}, L2 {np.sqrt(np.mean(d * d)):.3e}, pixels apart {int(bad.sum())} of {d.size}, L2 of the rest {np.sqrt(np.mean(d[~bad] ** 2)):.3e}, max of the rest {d[~bad].max():.3e}")
    for i in np.argsort(d.reshape(-1))[::-1][:6]:
        y, x = divmod(int(i), d.shape[1])
        print(f"   ({x:3d},{y:3d}) hip {a[y, x, :3] / spp} oracle {b[y, x, :3] / spp}")


for name, spectral in (("thin_film", False), ("thin_film_rough", False), ("thin_film_rough", True)):
    sc = load_hydra_xml(scene_path(name), 96, 96, spectral=spectral)
    if not spectral:
        brighten(sc, 60.0)
    for spp in (16, 64):
        g, c = HipIntegrator(sc), OracleIntegrator(sc)
        report(f"{name} spectral={spectral} spp={spp}", g.render(spp), c.render(spp), spp)
    if not spectral:
        p = sc.params(integrator=INTEGRATOR_STUPID_PT)
        g, c = HipIntegrator(sc, p), OracleIntegrator(sc, p)
        report(f"{name} naive", g.render(16, naive=True), c.render(16, naive=True), 16)
        g, c = HipIntegrator(sc, p), OracleIntegrator(sc, p)
        report(f"{name} stupid", g.render(16), c.render(16), 16)
