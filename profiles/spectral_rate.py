"""Throughput of the spectral kernel on the reference's spectral Cornell fixture next to the RGB kernels on the same scene; SCENES=a,b,.. names
other fixtures (the thin-film ones: SCENES=test_spectral,thin_film,thin_film_rough).
usage (GPU box): python profiles/spectral_rate.py [size] [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (before the library: it brings the ROCm runtime up)
from hydracore3_amd.api import HipIntegrator  # noqa: E402
from hydracore3_amd.scene import load_hydra_xml  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
for name, spectral in [(n, sp) for n in os.environ.get("SCENES", "test_spectral").split(",") for sp in (False, True)]:
    xml = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", name, "statex_00001.xml")
    sc = load_hydra_xml(xml, size, size, spectral=spectral)
    gpu = HipIntegrator(sc)
    frame = gpu.dev_array(np.zeros((size, size, 4), np.float32))
    gpu.path_trace_block_dev(frame.ptr, 4)                       # warm-up
    best = 1e30
    for _ in range(3):
        gpu.path_trace_block_dev(frame.ptr, spp)
        best = min(best, gpu.last_kernel_ms())               # HIP events around the launch
    print(f"{name:16s} {'spectral' if spectral else 'rgb     '}: {size}x{size} x {spp} spp in {best:.2f} ms = {size * size * spp / best / 1e3:.1f} Mpaths/s ({gpu.accel_info()['layout']})")
