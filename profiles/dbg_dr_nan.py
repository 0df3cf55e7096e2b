import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
W, H, ts = 1920, 1080, 1024
sc = synth.interior_scene(W, H, tex_size=ts)
gpu = HipIntegrator(sc)
off, size = gpu.PutDiffTex2D(1, ts, ts, 4)
data = np.full(size, 0.5, np.float32)
ref = np.zeros((H, W, 4), np.float32)
out = np.zeros((H, W, 4), np.float32)
grad = np.zeros_like(data)
loss = gpu.PathTraceDR(gpu.N, 4, out, 64, ref, data, grad)
bad = ~np.isfinite(grad)
print("loss", loss, "non-finite grads", int(bad.sum()), "max |grad| finite", float(np.abs(grad[~bad]).max()), "non-finite out", int((~np.isfinite(out)).sum()))
idx = np.argwhere(bad).ravel()[:10]
print("bad idx", idx.tolist(), [ (int(i)//4 % ts, int(i)//4 // ts, int(i) % 4) for i in idx])
big = np.argsort(-np.abs(np.where(bad, 0, grad)))[:5]
print("largest finite", [(int(i), float(grad[i])) for i in big])
