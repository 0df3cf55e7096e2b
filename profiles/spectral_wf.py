"""A 1 M-triangle interior under m_spectral_mode = 1: plain kernel (schedule 1), block-local (3), wavefront (2: wfShadeSpecKernel + the shared trace kernel), automatic. python profiles/spectral_wf.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd import synth
def rate(sc, sched, spp=32):
    g = HipIntegrator(sc); g.set_schedule(sched)
    fr = g.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
    g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, spp)
    return round(sc.width * sc.height * spp / g.last_kernel_ms() / 1e3, 1), g.last_launch()["schedule"]
sc = synth.interior_scene(1920, 1080, tex_size=256)
sc.spectral_mode = 1; sc.spec_offset_sz, sc.spec_values = [(0, 471)], np.ones(471, np.float32)
for s in (3, 2, 0, 1):
    print('interior 1M spectral schedule', s, rate(sc, s), flush=True)
