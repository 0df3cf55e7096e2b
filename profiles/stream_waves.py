"""Schedule 4 built for 3 / 4 / 5 waves per SIMD (HYDRA_HIP_LIB selects the build): the 1 M-triangle interior at 64 spp. python profiles/stream_waves.py <blocks per CU ...>"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd import synth
sc = synth.interior_scene(1920, 1080, tex_size=256)
def rate(sched, spp=64, tb=0, refill=0):
    g = HipIntegrator(sc); g.set_schedule(sched, refill, tb)
    fr = g.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
    g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, spp)
    return round(sc.width * sc.height * spp / g.last_kernel_ms() / 1e3, 1)
for tb in [int(x) for x in sys.argv[1:]] or [0]:
    print(os.environ.get("HYDRA_HIP_LIB", "product"), "streaming, blocks per CU", tb, rate(4, tb=tb), flush=True)
