"""Diagnostic: share of wave-cycles per phase of the bounce loop (instrumented STATS build, s_memtime stamps). Run on a GPU box: python profiles/phases.py"""
import sys; sys.path.insert(0,'.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.synth import interior_scene
for name in (sys.argv[1:] or ('cornell','interior')):
    sc = load_hydra_xml('tests/golden/scenes/test_035/statex_00001.xml',1024,1024) if name=='cornell' else interior_scene(1920,1080)
    for layout in (1, 2):
      g = HipIntegrator(sc, accel_layout=layout); g.set_instrumentation(True); print('layout', layout)
      img = np.zeros((sc.height, sc.width,4),np.float32); g.PathTraceBlock(g.N,4,img,8)
      c = g.counters(); tot = sum(c[k] for k in c if k.startswith('cyc_'))
      print(name, {k: round(c[k]/tot,3) for k in c if k.startswith('cyc_')}, 'trips/wave-path', c['loop_trips']*64/c['paths'], 'rays/path', c['rays']/c['paths'])
      print('   nodes/ray', c['nodes']/c['rays'], 'tris/ray', c['tris']/c['rays'], 'lane util: node loop', c['nodes']/(64*c['wave_node_iters']), 'tri loop', c['tris']/(64*c['wave_tri_iters']), 'wave node iters per wave-ray', c['wave_node_iters']*64/c['rays'])
