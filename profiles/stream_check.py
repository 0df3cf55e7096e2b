"""Schedule 4 (block-owned streaming, hpt_stream.hip) against the wavefront schedule: bit-identical frames on a miniature, then the 1 M-triangle interior's rate. python profiles/stream_check.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd import synth
sc = synth.interior_scene(384, 320, objects=24, subdiv=2, tex_size=64)
a = HipIntegrator(sc, accel_layout=2); a.set_schedule(2); ra = a.render(6)
b = HipIntegrator(sc, accel_layout=2); b.set_schedule(4); rb = b.render(6)
print("miniature: schedule", b.last_launch()["schedule"], "equal frames", np.array_equal(ra, rb), "equal generators", np.array_equal(a.random_gens(), b.random_gens()), "mean", float(rb[..., :3].mean()), flush=True)
if not np.array_equal(ra, rb): sys.exit(1)
sc = synth.interior_scene(1920, 1080, tex_size=256)
def rate(sched, spp=64, tb=0):
    g = HipIntegrator(sc); g.set_schedule(sched, 0, tb)
    fr = g.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
    g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, spp)
    return round(sc.width * sc.height * spp / g.last_kernel_ms() / 1e3, 1), g.last_launch()["schedule"]
print("interior 1M wavefront", rate(2), flush=True)
for tb in (0, 4, 3, 6):
    print("interior 1M streaming, blocks per CU", tb, rate(4, tb=tb), flush=True)
