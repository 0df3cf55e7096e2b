"""PathTraceDR on the 1M-triangle interior: megakernel vs wavefront schedule (profiles/dr_sched.py <spp> <tex>)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
spp, ts = int(sys.argv[1]), int(sys.argv[2])
W, H = 1920, 1080
sc = synth.interior_scene(W, H, tex_size=ts)
dev = torch.device("cuda", 0)
ref = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
for sched in (1, 2, 1, 2):
    integ = HipIntegrator(sc); integ.set_schedule(sched)
    off, size = integ.PutDiffTex2D(1, ts, ts, 4)
    data = torch.full((size,), 0.5, dtype=torch.float32, device=dev); grad = torch.zeros_like(data)
    loss = torch.zeros(1, dtype=torch.float32, device=dev); frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    for rep in range(2):
        grad.zero_(); loss.zero_(); frame.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        integ._chk(integ.L.hpt_path_trace_dr_dev(integ.h, 0, W * H, 4, frame.data_ptr(), spp, ref.data_ptr(), data.data_ptr(), grad.data_ptr(), size, loss.data_ptr(), None))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"schedule {sched} (used {integ.last_schedule()}): {W * H * spp / dt / 1e6:.1f} Mpaths/s  loss {float(loss.item()) / (W * H):.5f}  |grad| {float(grad.abs().sum()):.4e}", flush=True)
    del integ
