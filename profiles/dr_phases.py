"""Diagnostic: where PathTraceDR's wave cycles go (instrumented STATS + DR megakernel: s_memtime stamps per phase, record / sweep / atomic counts).
Run on a GPU box:  python profiles/dr_phases.py [dr|dr_interior] [spp]"""
import os, sys, json, time
sys.path.insert(0, '.')
import numpy as np
from hydracore3_amd.api import HipIntegrator
from hydracore3_amd.synth import dr_scene, interior_scene

what = sys.argv[1] if len(sys.argv) > 1 else 'dr'
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if what == 'dr':
    xml = 'tests/golden/scenes/test_228/statex_00001.xml'
    W = H = 512
    sc, tex_id = dr_scene(xml, W, H); tw = 256
    tgt, _ = dr_scene(xml, W, H, target=True)
else:
    W, H = 1920, 1080
    tw = int(os.environ.get('HYDRA_BENCH_TEX', '1024'))
    sc = interior_scene(W, H, tex_size=tw); tex_id = 1; tgt = sc
ref = HipIntegrator(tgt).render(16) / 16.0
ref = np.ascontiguousarray(ref[::-1])
out = {}
for instrumented in (False, True):
    g = HipIntegrator(sc)
    g.set_schedule(1)
    g.set_option('dr_skip_nonfinite', 1)
    off, size = g.PutDiffTex2D(tex_id, tw, tw, 4)
    data = g.dev_array(np.full(size, 0.5, np.float32)); grad = g.dev_array(np.zeros(size, np.float32))
    frame = g.dev_array(np.zeros((H, W, 4), np.float32)); refd = g.dev_array(ref); loss = g.dev_array(np.zeros(1, np.float32))
    g.set_instrumentation(instrumented)
    g.PathTraceDR_dev(frame, 4, refd, data, grad, loss)       # warm-up
    g.PathTraceDR_dev(frame, spp, refd, data, grad, loss)
    ms = g.last_kernel_ms()
    paths = W * H * spp
    print(f"{what} {W}x{H} @ {spp} spp, instrumented={instrumented}: kernel {ms:.2f} ms = {paths / ms / 1e3:.1f} Mpaths/s, launch {g.last_launch()}")
    if instrumented:
        c = g.counters(); d = g.dr_counters()
        cyc = {k: c[k] for k in c if k.startswith('cyc_')}
        cyc['cyc_record_store'] = d['cyc_record_store']; cyc['cyc_sweep'] = d['cyc_sweep']
        tot = float(sum(cyc.values()))
        print('  phase shares of wave cycles:', {k[4:]: round(v / tot, 4) for k, v in cyc.items()})
        print('  per path: rays %.2f, nodes/ray %.1f, tris/ray %.2f, surface hits %.2f, records %.2f (with taps %.2f)' % (
            c['rays'] / c['paths'], c['nodes'] / c['rays'], c['tris'] / c['rays'], c['surface_hits'] / c['paths'], d['records'] / c['paths'], d['records_with_taps'] / c['paths']))
        print('  loop trips per wave-path %.3f; lane util node loop %.3f, tri loop %.3f' % (c['loop_trips'] * 64 / c['paths'], c['nodes'] / (64 * max(c['wave_node_iters'], 1)), c['tris'] / (64 * max(c['wave_tri_iters'], 1))))
        print('  sweeps: %.3f of the trips run one; lanes per sweep %.1f; bounces per swept path %.2f; atomic wave-instructions per path %.3f (per sweep %.1f)' % (
            d['sweep_wave_trips'] / max(c['loop_trips'], 1), d['sweep_lanes'] / max(d['sweep_wave_trips'], 1), d['sweep_bounces'] / max(d['sweep_lanes'], 1),
            d['atomic_wave_insts'] / c['paths'], d['atomic_wave_insts'] / max(d['sweep_wave_trips'], 1)))
        out = {'workload': what, 'spp': spp, 'counters': c, 'dr_counters': d, 'kernel_ms_instrumented': ms}
    else:
        out_ms = ms
out['kernel_ms'] = out_ms
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open(f'gpurun_out/dr_phases_{what}.json', 'w'), indent=1)

# the forward kernel (instrumented: every BSDF branch) on the same scene, for the per-path cycle comparison
g = HipIntegrator(sc); g.set_schedule(1); g.set_instrumentation(True)
img = np.zeros((H, W, 4), np.float32); g.PathTraceBlock(g.N, 4, img, 8)
c = g.counters(); tot = float(sum(c[k] for k in c if k.startswith('cyc_')))
print('forward (instrumented, same scene): phase shares', {k[4:]: round(c[k] / tot, 4) for k in c if k.startswith('cyc_')})
print('  wave cycles per path: forward', {k[4:]: round(c[k] / c['paths'], 1) for k in c if k.startswith('cyc_')})
cd = out['counters']; dd = out['dr_counters']
print('  wave cycles per path: DR     ', dict({k[4:]: round(cd[k] / cd['paths'], 1) for k in cd if k.startswith('cyc_')}, record_store=round(dd['cyc_record_store'] / cd['paths'], 1), sweep=round(dd['cyc_sweep'] / cd['paths'], 1)))
g2 = HipIntegrator(sc); g2.set_schedule(1); fr = g2.dev_array(np.zeros((H, W, 4), np.float32)); g2.path_trace_block_dev(fr.ptr, 4); g2.path_trace_block_dev(fr.ptr, spp)
print(f'forward lean kernel: {W * H * spp / g2.last_kernel_ms() / 1e3:.1f} Mpaths/s')
