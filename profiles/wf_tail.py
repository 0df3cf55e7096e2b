"""Tail of the wavefront trace passes vs the suspension grace: profiles/wf_tail.py <spp> <grace...>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_scene
from hydracore3_amd.api import HipIntegrator
spp = int(sys.argv[1]); graces = [int(x) for x in sys.argv[2:]] or [0, 8]
W, H = 1920, 1080
sc = build_scene("interior", W, H)
integ = HipIntegrator(sc)
frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for groups in (1, 2):
    for g in graces:
        integ.set_schedule(2, 56, 0, groups); integ.set_option("wf_grace", g)
        integ.set_instrumentation(True); integ.InitRandomGens(W * H)
        integ.path_trace_block_dev(frame.data_ptr(), spp, 0, W * H, 4, False, None); torch.cuda.synchronize()
        v = list(integ.counters().values())
        integ.set_instrumentation(False); integ.InitRandomGens(W * H)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        integ.path_trace_block_dev(frame.data_ptr(), spp, 0, W * H, 4, False, None); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"groups {groups} grace {g:3d}: rounds {integ.last_schedule()[1]:4d}  suspended rays {v[12]:.3e}  node util {v[0] / (64.0 * v[1]):.3f}  "
              f"dry wave time {v[10] / max(v[11], 1):.3f}  {W * H * spp / dt / 1e6:.1f} Mpaths/s", flush=True)
