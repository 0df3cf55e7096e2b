"""SAH estimate of node visits per ray for the scenes whose best schedule was measured (policy calibration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydracore3_amd import synth
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.api import HipIntegrator
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scenes = [(n, load_hydra_xml(os.path.join(root, "tests", "golden", "scenes", n, "statex_00001.xml"), 256, 256)) for n in ("test_035", "test_228", "typed_materials", "legacy_materials", "env_map")]
scenes += [(f"interior subdiv {sd}", synth.interior_scene(256, 144, subdiv=sd, tex_size=64)) for sd in (0, 1, 2, 3, 4)]
for name, sc in scenes:
    for layout in ((1,) if sc.inst_motion else (1, 2)):
        g = HipIntegrator(sc, accel_layout=layout)
        print(f"{name:22s} layout {layout}: {g.accel_info()}", flush=True)
