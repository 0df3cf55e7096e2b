import os, sys
sys.path.insert(0, os.getcwd())
import torch
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd import synth
for n in ("legacy_materials", "typed_materials", "env_map", "test_228"):
    for L in (1, 2):
        print(n, L, flush=True)
        HipIntegrator(load_hydra_xml(f'tests/golden/scenes/{n}/statex_00001.xml', 64, 64), accel_layout=L)
print("interior", flush=True)
HipIntegrator(synth.interior_scene(256, 144, subdiv=4, tex_size=64))
