import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd import synth
def check(name, sc, spp=4):
    a = HipIntegrator(sc); a.set_schedule(1); ia = a.render(spp)
    b = HipIntegrator(sc); b.set_schedule(3); ib = b.render(spp)
    print(name, 'schedule', b.last_launch(), 'identical frame', np.array_equal(ia, ib), 'gens', np.array_equal(a.random_gens(), b.random_gens()), 'max diff', float(np.abs(ia - ib).max()), flush=True)
check('cornell two-level', load_hydra_xml('tests/golden/scenes/test_035/statex_00001.xml', 128, 128))
check('test_228', load_hydra_xml('tests/golden/scenes/test_228/statex_00001.xml', 128, 128))
check('interior mini', synth.interior_scene(160, 96, subdiv=1, tex_size=16))
check('typed_materials (full kernel)', load_hydra_xml('tests/golden/scenes/typed_materials/statex_00001.xml', 96, 64))
# rates
for name, sc in (('test_228 512^2', load_hydra_xml('tests/golden/scenes/test_228/statex_00001.xml', 512, 512)),
                 ('test_228 1024^2', load_hydra_xml('tests/golden/scenes/test_228/statex_00001.xml', 1024, 1024)),
                 ('typed_materials 1024^2', load_hydra_xml('tests/golden/scenes/typed_materials/statex_00001.xml', 1024, 1024))):
    for sched in (1, 2, 3):
        g = HipIntegrator(sc); g.set_schedule(sched)
        fr = g.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
        g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, 64)
        print(f'{name} schedule {sched}: {sc.width * sc.height * 64 / g.last_kernel_ms() / 1e3:.1f} Mpaths/s', flush=True)
