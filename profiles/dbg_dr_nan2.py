import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
W, H, ts, spp = 1920, 1080, 1024, 64
sc = synth.interior_scene(W, H, tex_size=ts)
dev = torch.device("cuda", 0)
tgt = HipIntegrator(sc)
ref = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
tgt.path_trace_block_dev(ref.data_ptr(), 64, 0, W * H, 4, False, None); torch.cuda.synchronize()
ref = torch.flip(ref / 64.0, dims=[0]).contiguous()
print("ref finite", bool(torch.isfinite(ref).all()), float(ref.max()))
integ = HipIntegrator(sc)
off, size = integ.PutDiffTex2D(1, ts, ts, 4)
data = torch.full((size,), 0.5, dtype=torch.float32, device=dev)
grad = torch.zeros_like(data); mom = torch.zeros_like(data); gsq = torch.zeros_like(data)
loss = torch.zeros(1, dtype=torch.float32, device=dev); frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
L = integ.L
for it in range(3):
    grad.zero_(); loss.zero_(); frame.zero_()
    integ._chk(L.hpt_path_trace_dr_dev(integ.h, 0, W * H, 4, frame.data_ptr(), spp, ref.data_ptr(), data.data_ptr(), grad.data_ptr(), size, loss.data_ptr(), None))
    torch.cuda.synchronize()
    print(it, "loss", float(loss.item()) / (W * H), "grad nonfinite", int((~torch.isfinite(grad)).sum()), "max|grad|", float(grad[torch.isfinite(grad)].abs().max()),
          "frame nonfinite", int((~torch.isfinite(frame)).sum()), "data range", float(data.min()), float(data.max()), "data nonfinite", int((~torch.isfinite(data)).sum()))
    integ._chk(L.hpt_adam_step_dev(integ.h, data.data_ptr(), grad.data_ptr(), mom.data_ptr(), gsq.data_ptr(), size, it, None))
    torch.cuda.synchronize()
    print("   after adam: data range", float(data[torch.isfinite(data)].min()), float(data[torch.isfinite(data)].max()), "nonfinite", int((~torch.isfinite(data)).sum()))
