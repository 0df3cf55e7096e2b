"""Lane utilisation of the wavefront trace kernel (instrumented build): profiles/wf_stats.py <workload> <spp> [refill ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from bench import build_scene
from hydracore3_amd.api import HipIntegrator

wl, spp = sys.argv[1], int(sys.argv[2])
refills = [int(x) for x in sys.argv[3:]] or [48]
W, H = (1024, 1024) if wl == "cornell" else (1920, 1080)
sc = build_scene(wl, W, H)
for layout in (1, 2):
    integ = HipIntegrator(sc, accel_layout=layout)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for rb in refills:
        integ.set_schedule(2, rb)
        integ.set_instrumentation(True)
        integ.InitRandomGens(W * H)
        integ.path_trace_block_dev(frame.data_ptr(), spp, 0, W * H, 4, False, None)
        torch.cuda.synchronize()
        c = integ.counters()
        vals = [int(c[k]) for k in list(c)]
        nodeLane, nodeWave, triLane, triWave, nref = vals[:5]
        ph = vals[5:9]; tot = float(sum(ph)) or 1.0
        tail, wtot = vals[10], max(vals[11], 1)
        integ.set_instrumentation(False)
        integ.InitRandomGens(W * H)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        integ.path_trace_block_dev(frame.data_ptr(), spp, 0, W * H, 4, False, None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{wl} layout={layout} refill<{rb}: node util {nodeLane / (64.0 * nodeWave):.3f}  tri util {triLane / (64.0 * triWave):.3f}  "
              f"wave node iters {nodeWave:.3e} tri iters {triWave:.3e} refills {nref:.3e}  iters {integ.last_schedule()[1]}  {W * H * spp / dt / 1e6:.1f} Mpaths/s\n"
              f"    wave time: refill {ph[0] / tot:.3f}  node loop {ph[1] / tot:.3f}  leaves {ph[2] / tot:.3f}  ray end+vote {ph[3] / tot:.3f}   cycles per wave node iter {ph[1] / max(nodeWave, 1):.0f}  per wave tri iter {ph[2] / max(triWave, 1):.0f}   wave time after the queue ran dry {tail / wtot:.3f}", flush=True)
