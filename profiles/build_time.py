"""CommitScene on the 1M-triangle interior: host build of either layout, and UpdateInstance + CommitScene as a device refit against a host rebuild.
Run on a GPU box: python profiles/build_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd import synth, scene as S
from hydracore3_amd.api import HipIntegrator
t0 = time.time(); sc = synth.interior_scene(1920, 1080, tex_size=256); t1 = time.time()
print(f"scene synth {t1 - t0:.2f} s", flush=True)
for layout in (1, 2):
    t2 = time.time(); g = HipIntegrator(sc, accel_layout=layout); t3 = time.time()
    print(f"layout {layout}: LoadScene {t3 - t2:.2f} s; CommitScene {g.commit_time()}", flush=True)
    if layout == 2:
        for refit in (1, 0):
            g.set_option("refit", refit)
            m = S.translate(0.3, 0.0, 0.1) @ np.asarray(sc.inst_matrices[40])
            t4 = time.time(); g.UpdateInstance(40, m); g.CommitScene(); t5 = time.time()
            print(f"   UpdateInstance + CommitScene, refit={refit}: {(t5 - t4) * 1e3:.1f} ms wall; {g.commit_time()}", flush=True)
# host threads of CommitScene (single-level layout, the collapse to the 4-wide tree included)
for n in (1, 2, 4, 8, 16):
    g = HipIntegrator(sc, accel_layout=2)
    g.set_option("build_threads", n); g.set_option("refit", 0)
    g.UpdateInstance(40, np.asarray(sc.inst_matrices[40])); g.CommitScene()
    print(f"   build_threads {n}: CommitScene {g.commit_time()}", flush=True)
