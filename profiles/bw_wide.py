"""Block-local schedule on the test_228 class: BVH2 vs the 4-wide compressed tree (hpt_set_option("bw_wide")), forward. python profiles/bw_wide.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.scene import load_hydra_xml
xml = 'tests/golden/scenes/test_228/statex_00001.xml'
for W in (512, 1024):
    sc = load_hydra_xml(xml, W, W)
    for wide in (0, 1):
        for nm in (4, 16, 32):
            g = HipIntegrator(sc); g.set_schedule(3); g.set_option('bw_wide', wide); g.set_option('bw_node_min', nm)
            fr = g.dev_array(np.zeros((W, W, 4), np.float32))
            g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, 64)
            print(f'test_228 {W}^2 bw_wide {wide} node_min {nm}: {W * W * 64 / g.last_kernel_ms() / 1e3:.1f} Mpaths/s {g.last_launch()}', flush=True)
