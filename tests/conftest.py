import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both native libraries are built before any test; the HIP one cross-compiles without a GPU."""
    import __graft_entry__ as g
    g.build()


def scene_path(name):
    return os.path.join(SCENES, name, "statex_00001.xml")


# ---- the image bar, per pixel ---------------------------------------------------------------------------------------------------------------
# north_star: "rendered images match the CPU integrator within per-pixel L2 < 1e-3 at fixed seeds". The bar is held PER PIXEL (the L2 norm
# of a pixel's RGB difference of the spp-normalised frames), not as a mean over the frame: a mean over 10^5 pixels hides single pixels
# far above the bar. Two kinds of pixel can legitimately sit above it, both from last-bit differences between the device's and glibc's
# sinf / cosf / powf / acosf amplified by a DISCRETE decision of the path: (1) a path that took another branch and drew another number of
# randoms - its pixel's generator ends in another state; (2) a path whose shadow ray (or grazing closest hit) came out the other way - one
# light sample more or less, the same draws, the same generator state. Neither is tolerated silently: both are COUNTED against bounds
# written in the test (a handful per 10^5 .. 10^7 paths), every other pixel must meet the bar, and the RMS over those other pixels must
# sit an order of magnitude below it.
def pixel_errors(a, b, spp):
    import numpy as np
    d = (np.asarray(a)[..., :3].astype(np.float64) - np.asarray(b)[..., :3].astype(np.float64)) / spp
    return np.sqrt(np.sum(d * d, axis=-1))


def assert_pixel_parity(a, b, spp, gpu=None, cpu=None, tol=1e-3, max_divergent=0, max_over=None, scale=1.0, rest_rms=1e-4, what=""):
    """a, b: (H, W, >=3) frames (sums over spp passes); gpu / cpu: the integrators (their random_gens() tell which pixels' generators differ).
    Asserts: pixels with ||drgb|| / spp >= tol * scale: at most max_over (default: max_divergent); pixels whose generators differ: at most
    max_divergent; RMS of the per-pixel errors of all pixels under the bar < rest_rms * scale.
    Returns (worst pixel under the bar, RMS of the pixels under the bar, pixels over the bar, pixels with differing generators)."""
    import numpy as np
    e = pixel_errors(a, b, spp)
    n_div, div_img = 0, np.zeros(e.shape, bool)
    if gpu is not None and cpu is not None:
        same = np.all(gpu.random_gens() == cpu.random_gens(), axis=1)
        xy = gpu.packed_xy()
        n = min(len(xy), len(same))
        div_img[(xy[:n] >> 16) & 0xFFFF, xy[:n] & 0xFFFF] = ~same[:n]
        n_div = int(np.sum(~same))
    max_over = max_divergent if max_over is None else max_over
    bar = tol * scale
    over_img = e >= bar
    over = int(np.sum(over_img))
    rest = e[~over_img]
    rms = float(np.sqrt(np.mean(rest * rest))) if rest.size else 0.0
    worst = float(rest.max()) if rest.size else 0.0
    print(f"{what}per-pixel L2 (bar {bar:.1e}): {over} of {e.size} pixels over it (worst {float(e.max()):.3e}; {int(np.sum(over_img & div_img))} of them with a divergent "
          f"generator), {n_div} pixels with a divergent generator; the other pixels: worst {worst:.3e}, RMS {rms:.3e}")
    assert over <= max_over, f"{what}{over} pixels at or above {bar:.1e} (bound {max_over})"
    assert n_div <= max_divergent, f"{what}{n_div} pixels with a divergent path (bound {max_divergent})"
    assert rms < rest_rms * scale, f"{what}RMS of the pixels under the bar {rms:.3e} >= {rest_rms * scale:.1e}"
    return worst, rms, over, n_div
