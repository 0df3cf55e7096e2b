"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (the CPU restatement of
HydraCore3's PathTraceBlock / PathTraceDR).  PARITY UNPINNED, see oracle/README.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing in
hydracore3_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from hydracore3_amd.scene import HIT_DTYPE, Params, SceneData, SceneDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_pack_xy.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_init_random_gens.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_get_random_gens.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_set_random_gens.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        for f in (L.orc_path_trace_block, L.orc_naive_path_trace_block):
            f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_ray_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_ray_any.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_set_optics.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float]
        L.orc_ray_nearest_motion.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_int]
        L.orc_ray_any_motion.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_int]
        L.orc_put_diff_tex2d.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_put_diff_tex2d.restype = C.c_int
        L.orc_path_trace_dr.restype = C.c_float
        L.orc_path_trace_dr.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_path_trace_dr_fd.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_path_trace_from_input_rays_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.orc_rng_kat.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_tex_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_probe.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        L.orc_probe.restype = C.c_int
        L.orc_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.orc_reg_loss_image2d4f.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_reg_loss_image2d4f.restype = C.c_double
        L.orc_image2d4f_regularizer.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class OracleIntegrator:
    """Mirrors the part of the reference's Integrator surface the path needs (integrator_pt.h:123-703)."""

    def __init__(self, scene: SceneData, params: Params = None, threads: int = 0):
        self.L = lib()
        self.scene = scene
        self.params = params if params is not None else scene.params()
        self._desc = scene.desc()
        self.h = self.L.orc_create(C.byref(self._desc), C.byref(self.params))
        self.W, self.H = self.params.winWidth, self.params.winHeight
        self.N = self.W * self.H
        self.L.orc_set_threads(threads)
        lines = np.ascontiguousarray(scene.lens_lines, np.float32).reshape(-1, 4)
        self.L.orc_set_optics(self.h, lines.ctypes.data if lines.size else None, lines.shape[0], C.c_float(scene.phys_size[0]), C.c_float(scene.phys_size[1]))
        self.L.orc_pack_xy(self.h, None)
        self.L.orc_init_random_gens(self.h, self.N)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def set_params(self, params: Params):
        self.params = params
        self.L.orc_set_params(self.h, C.byref(params))

    def packed_xy(self):
        out = np.zeros(self.N, np.uint32)
        self.L.orc_pack_xy(self.h, out.ctypes.data)
        return out

    def random_gens(self):
        out = np.zeros((self.N, 2), np.uint32)
        self.L.orc_get_random_gens(self.h, out.ctypes.data, self.N)
        return out

    def set_random_gens(self, gens):
        gens = np.ascontiguousarray(gens, np.uint32).reshape(-1, 2)      # one uint2 per thread
        self.L.orc_set_random_gens(self.h, gens.ctypes.data, gens.shape[0])

    def path_trace_block(self, out_color, pass_num, tid_begin=0, tid_count=None, channels=4, naive=False):
        tid_count = self.N - tid_begin if tid_count is None else tid_count
        fn = self.L.orc_naive_path_trace_block if naive else self.L.orc_path_trace_block
        fn(self.h, tid_begin, tid_count, channels, out_color.ctypes.data, pass_num)
        return out_color

    def path_trace_from_input_rays_block(self, ray_pos, ray_dir, out_color, pass_num, channels=4):
        n = ray_pos.shape[0]
        self.L.orc_path_trace_from_input_rays_block(self.h, n, channels, ray_pos.ctypes.data, ray_dir.ctypes.data, out_color.ctypes.data, pass_num)

    def render(self, spp, channels=4, naive=False):
        img = np.zeros((self.H, self.W, channels), np.float32)
        self.path_trace_block(img, spp, channels=channels, naive=naive)
        return img

    def ray_nearest(self, pos_near, dir_far, brute=False):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], HIT_DTYPE)
        self.L.orc_ray_nearest(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], out.ctypes.data, int(brute))
        return out

    def ray_any(self, pos_near, dir_far, brute=False):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], np.uint32)
        self.L.orc_ray_any(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], out.ctypes.data, int(brute))
        return out

    def ray_nearest_motion(self, pos_near, dir_far, time, brute=False):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], HIT_DTYPE)
        self.L.orc_ray_nearest_motion(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], C.c_float(time), out.ctypes.data, int(brute))
        return out

    def ray_any_motion(self, pos_near, dir_far, time, brute=False):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], np.uint32)
        self.L.orc_ray_any_motion(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], C.c_float(time), out.ctypes.data, int(brute))
        return out

    def set_option(self, name, value):
        if self.L.orc_set_option(self.h, name.encode(), int(value)) != 0:
            raise KeyError(name)

    def put_diff_tex2d(self, tex_id, w, h, channels):
        off, size = C.c_uint64(0), C.c_uint64(0)
        rc = self.L.orc_put_diff_tex2d(self.h, tex_id, w, h, channels, C.byref(off), C.byref(size))
        return rc, off.value, size.value

    def path_trace_dr(self, out_color, pass_num, ref_img, data, tid_begin=0, tid_count=None, channels=4):
        tid_count = self.N - tid_begin if tid_count is None else tid_count
        data = np.ascontiguousarray(data, np.float32)
        ref_img = np.ascontiguousarray(ref_img, np.float32)
        grad = np.zeros_like(data)
        loss = self.L.orc_path_trace_dr(self.h, tid_begin, tid_count, channels, out_color.ctypes.data, pass_num,
                                        ref_img.ctypes.data, data.ctypes.data, grad.ctypes.data, data.size)
        return float(loss), grad

    def path_trace_dr_fd(self, pass_num, ref_img, data, idx, h=1e-2, tid_begin=0, tid_count=None, channels=4):
        tid_count = self.N - tid_begin if tid_count is None else tid_count
        data = np.ascontiguousarray(data, np.float32)
        ref_img = np.ascontiguousarray(ref_img, np.float32)
        idx = np.ascontiguousarray(idx, np.uint64)
        out = np.zeros(idx.size, np.float64)
        self.L.orc_path_trace_dr_fd(self.h, tid_begin, tid_count, channels, pass_num, ref_img.ctypes.data,
                                    data.ctypes.data, data.size, idx.ctypes.data, idx.size, h, out.ctypes.data)
        return out

    def tex_sample(self, tex_id, uv):
        uv = np.ascontiguousarray(uv, np.float32)
        out = np.zeros((uv.shape[0], 4), np.float32)
        self.L.orc_tex_sample(self.h, tex_id, uv.ctypes.data, uv.shape[0], out.ctypes.data)
        return out


def rng_kat(seed, n_draws):
    state = np.zeros(4, np.uint32)
    vals = np.zeros((n_draws, 4), np.float32)
    lib().orc_rng_kat(seed, n_draws, state.ctypes.data, vals.ctypes.data)
    return state, vals


def adam_step(state, grad, momentum, gsquare, it):
    lib().orc_adam_step(state.ctypes.data, grad.ctypes.data, momentum.ctypes.data, gsquare.ctypes.data, state.size, it)


def reg_loss_image2d4f(data):
    """data: float32 [h, w, 4]."""
    return lib().orc_reg_loss_image2d4f(data.shape[1], data.shape[0], data.ctypes.data)


def image2d4f_regularizer(data, grad):
    lib().orc_image2d4f_regularizer(data.shape[1], data.shape[0], data.ctypes.data, grad.ctypes.data)


def probe(name, *args):
    a = np.zeros(max(8, len(args)), np.float32)
    a[:len(args)] = args
    out = np.zeros(8, np.float32)
    rc = lib().orc_probe(name.encode(), a.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise KeyError(name)
    return out
