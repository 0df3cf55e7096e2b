"""Block-local schedule (hpt_block.hip): refill threshold x node-loop vote on the test_228 class, forward and PathTraceDR. python profiles/bw_sweep.py"""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.scene import load_hydra_xml
from hydracore3_amd.synth import dr_scene
xml = 'tests/golden/scenes/test_228/statex_00001.xml'
def fwd(sc, sched, refill=None, nm=None, spp=64):
    g = HipIntegrator(sc); g.set_schedule(sched)
    if refill is not None: g.set_option('bw_refill_below', refill); g.set_option('bw_node_min', nm)
    fr = g.dev_array(np.zeros((sc.height, sc.width, 4), np.float32))
    g.path_trace_block_dev(fr.ptr, 4); g.path_trace_block_dev(fr.ptr, spp)
    return sc.width * sc.height * spp / g.last_kernel_ms() / 1e3
def dr(sched, refill=None, nm=None, spp=64, W=512):
    sc, tex_id = dr_scene(xml, W, W); tgt, _ = dr_scene(xml, W, W, target=True)
    ref = np.ascontiguousarray((HipIntegrator(tgt).render(16) / 16.0)[::-1])
    g = HipIntegrator(sc); g.set_schedule(sched); g.set_option('dr_skip_nonfinite', 1)
    if refill is not None: g.set_option('bw_refill_below', refill); g.set_option('bw_node_min', nm)
    off, size = g.PutDiffTex2D(tex_id, 256, 256, 4)
    data = g.dev_array(np.full(size, 0.5, np.float32)); grad = g.dev_array(np.zeros(size, np.float32))
    frame = g.dev_array(np.zeros((W, W, 4), np.float32)); refd = g.dev_array(ref); loss = g.dev_array(np.zeros(1, np.float32))
    g.PathTraceDR_dev(frame, 4, refd, data, grad, loss); g.PathTraceDR_dev(frame, spp, refd, data, grad, loss)
    return W * W * spp / g.last_kernel_ms() / 1e3, grad.download() if hasattr(grad, 'download') else None
sc5, sc10 = load_hydra_xml(xml, 512, 512), load_hydra_xml(xml, 1024, 1024)
print('forward megakernel 512 / 1024: %.1f / %.1f' % (fwd(sc5, 1), fwd(sc10, 1)), flush=True)
for refill in (32, 48, 56, 64):
    for nm in (0, 4, 8, 16):
        print(f'forward block-local refill {refill} node_min {nm}: 512^2 {fwd(sc5, 3, refill, nm):.1f}  1024^2 {fwd(sc10, 3, refill, nm):.1f}', flush=True)
print('DR megakernel: %.1f' % dr(1)[0], flush=True)
for refill in (32, 48, 64):
    for nm in (0, 4, 16):
        print(f'DR block-local refill {refill} node_min {nm}: {dr(3, refill, nm)[0]:.1f}', flush=True)
