import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd import synth
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("test_035", "test_228", "interior"):
    if name == "interior":
        sc = synth.interior_scene(160, 96, objects=24, subdiv=2, tex_size=64)
    else:
        sc = load_hydra_xml(os.path.join(root, "tests", "golden", "scenes", name, "statex_00001.xml"), 96, 96)
    mega = HipIntegrator(sc, accel_layout=1); mega.set_schedule(1)
    ref = mega.render(5)
    for layout in (1, 2):
        wf = HipIntegrator(sc, accel_layout=layout); wf.set_schedule(2, 48, 0, 1)
        img = wf.render(5)
        d = np.abs(img - ref).max(axis=-1)
        bad = np.argwhere(d > 0)
        print(name, "layout", layout, "rounds", wf.last_schedule()[1], "differing pixels", len(bad), "max diff", float(d.max()), "gens equal", bool(np.array_equal(wf.random_gens(), mega.random_gens())), bad[:5].tolist(), flush=True)
