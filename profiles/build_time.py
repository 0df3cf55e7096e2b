import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydracore3_amd import synth
from hydracore3_amd.api import HipIntegrator
t0 = time.time(); sc = synth.interior_scene(1920, 1080, tex_size=256); t1 = time.time()
for layout in (1, 2):
    t2 = time.time(); g = HipIntegrator(sc, accel_layout=layout); t3 = time.time()
    print(f"layout {layout}: scene synth {t1 - t0:.2f} s, LoadScene (BVH build + upload) {t3 - t2:.2f} s", flush=True)
