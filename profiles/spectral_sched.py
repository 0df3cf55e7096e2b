"""Spectral rendering: the one-thread-per-pixel kernel (schedule 1) against the block-local one (3), and the RGB kernels on the same scenes. python profiles/spectral_sched.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd import synth
def rate(sc, sched, spp=64):
    g = HipIntegrator(sc); g.set_schedule(sched)
    fr = g.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
    g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, spp)
    return sc.width * sc.height * spp / g.last_kernel_ms() / 1e3, g.last_launch()
for name in ('test_spectral', 'thin_film', 'typed_materials', 'legacy_materials'):
    for spectral in (True, False):
        sc = load_hydra_xml(f'tests/golden/scenes/{name}/statex_00001.xml', 1024, 1024, spectral=spectral)
        print(name, 'spectral' if spectral else 'RGB', {s: round(rate(sc, s)[0], 1) for s in (0, 1, 3)}, rate(sc, 0)[1], flush=True)
sc = synth.interior_scene(1920, 1080, tex_size=256)
print('interior 1M RGB', {s: round(rate(sc, s, 16)[0], 1) for s in (0, 3)}, flush=True)
sc.spectral_mode = 1; sc.spec_offset_sz, sc.spec_values = [(0, 471)], np.ones(471, np.float32)
print('interior 1M spectral', {s: round(rate(sc, s, 16)[0], 1) for s in (1, 3)}, flush=True)
