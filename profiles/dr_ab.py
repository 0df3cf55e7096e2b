"""Diagnostic: forward-only vs fwd+bwd throughput on the DR scene (test_228 + differentiable albedo). Run on a GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.synth import dr_scene
xml = "tests/golden/scenes/test_228/statex_00001.xml"
sc, tid = dr_scene(xml, 512, 512)
spp = 32
g = HipIntegrator(sc)
img = np.zeros((512, 512, 4), np.float32)
g.PathTraceBlock(g.N, 4, img, spp); g.PathTraceBlock(g.N, 4, img, spp)
print("forward  ", g.N * spp / g.GetExecutionTime("PathTraceBlock")[0] / 1e3, "Mpaths/s")
for bpc in (2, 3, 4):
    d = HipIntegrator(sc); d.set_launch_config(bpc)
    off, size = d.PutDiffTex2D(tid, 256, 256, 4)
    data = np.full(size, 0.5, np.float32); grad = np.zeros_like(data); ref = np.zeros((512, 512, 4), np.float32)
    d.PathTraceDR(d.N, 4, img, spp, ref, data, grad); d.PathTraceDR(d.N, 4, img, spp, ref, data, grad)
    print("fwd+bwd bpc", bpc, d.N * spp / d.GetExecutionTime("PathTraceDR")[0] / 1e3, "Mpaths/s")
