"""Heavy scene, small frames (a multi-GPU pixel share): megakernel (4-wide tree) vs wavefront vs block-local (BVH2). python profiles/bw_heavy.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd import synth
for (W, H) in ((640, 360), (960, 540)):
    sc = synth.interior_scene(W, H, tex_size=256)
    for sched in (1, 2, 3):
        g = HipIntegrator(sc); g.set_schedule(sched)
        fr = g.dev_array(np.zeros((H, W, 4), np.float32))
        g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, 64)
        print(f'interior 1M tris {W}x{H} schedule {sched}: {W * H * 64 / g.last_kernel_ms() / 1e3:.1f} Mpaths/s  {g.last_launch()}', flush=True)
