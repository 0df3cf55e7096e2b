"""ctypes binding of the C ABI in include/hydra_hip.h (hydracore3_amd/libhydra_hip.so).

`HipIntegrator` mirrors the slice of the reference's `Integrator` / `IntegratorDR` class surface that the hot path
needs (integrator_pt.h:123-703, diff_render/integrator_dr.h:27-136): same method names, same argument meaning, same
ownership rules (the caller owns ``out_color`` and it is accumulated into; ``m_randomGens`` lives with the integrator).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .scene import HIT_DTYPE, Params, SceneData, SceneDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HYDRA_HIP_LIB", os.path.join(_HERE, "libhydra_hip.so"))   # override: A/B builds of the kernels

# every symbol include/hydra_hip.h declares: name -> (restype, argtypes)
_vp, _u32, _u64, _f, _i, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_size_t
ABI = {
    "hpt_create": (_i, [_i, C.POINTER(_vp)]),
    "hpt_destroy": (None, [_vp]),
    "hpt_last_error": (C.c_char_p, [_vp]),
    "hpt_device_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.c_char_p, _sz]),
    "hpt_set_optics": (_i, [_vp, _vp, _u32, _f, _f]),
    "hpt_plastic_precompute": (_i, [_f, _f, _f, _vp, _vp, _vp, C.POINTER(_f), C.POINTER(_f)]),
    "hpt_decode_jpeg": (_i, [_vp, _u64, C.POINTER(_u32), C.POINTER(_u32), _vp, _u64]),
    "hpt_film_precompute": (_i, [_vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_i)]),
    "hpt_device_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "hpt_device_free": (_i, [_vp, _vp]),
    "hpt_device_copy": (_i, [_vp, _vp, _vp, _sz, _i]),
    "hpt_device_memset": (_i, [_vp, _vp, _i, _sz]),
    "hpt_clear_geom": (_i, [_vp]),
    "hpt_add_geom_triangles3f": (_u32, [_vp, _vp, _sz, _vp, _sz, _u32, _sz]),
    "hpt_update_geom_triangles3f": (_i, [_vp, _u32, _vp, _sz, _vp, _sz, _u32, _sz]),
    "hpt_clear_scene": (_i, [_vp]),
    "hpt_add_instance": (_u32, [_vp, _u32, _vp]),
    "hpt_update_instance": (_i, [_vp, _u32, _vp]),
    "hpt_commit_scene": (_i, [_vp, _u32]),
    "hpt_ray_query_nearest": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "hpt_ray_query_any": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "hpt_ray_query_nearest_motion": (_i, [_vp, _vp, _vp, _u32, _f, _vp]),
    "hpt_ray_query_any_motion": (_i, [_vp, _vp, _vp, _u32, _f, _vp]),
    "hpt_add_instance_motion": (_u32, [_vp, _u32, _vp, _u32]),
    "hpt_upload_scene": (_i, [_vp, C.POINTER(SceneDesc)]),
    "hpt_update_params": (_i, [_vp, C.POINTER(Params)]),
    "hpt_update_materials": (_i, [_vp, _sz, _sz, _vp]),
    "hpt_update_lights": (_i, [_vp, _sz, _sz, _vp]),
    "hpt_update_mat_id_offsets": (_i, [_vp, _vp, _sz]),
    "hpt_pack_xy": (_i, [_vp, _u32, _u32]),
    "hpt_get_packed_xy": (_i, [_vp, _vp, _u32]),
    "hpt_init_random_gens": (_i, [_vp, _u32]),
    "hpt_init_random_gens_from": (_i, [_vp, _u32, _u32]),
    "hpt_get_random_gens": (_i, [_vp, _vp, _u32]),
    "hpt_set_random_gens": (_i, [_vp, _vp, _u32]),
    "hpt_path_trace_block": (_i, [_vp, _u32, _u32, _u32, _vp, _u32]),
    "hpt_naive_path_trace_block": (_i, [_vp, _u32, _u32, _u32, _vp, _u32]),
    "hpt_path_trace_block_dev": (_i, [_vp, _u32, _u32, _u32, _vp, _u32, _i, _vp]),
    "hpt_path_trace_from_input_rays_block": (_i, [_vp, _u32, _u32, _vp, _vp, _vp, _u32]),
    "hpt_path_trace_from_input_rays_block_dev": (_i, [_vp, _u32, _u32, _vp, _vp, _vp, _u32, _vp]),
    "hpt_set_tid_interleave": (_i, [_vp, _u32, _u32]),
    "hpt_put_diff_tex2d": (_i, [_vp, _u32, _u32, _u32, _u32, C.POINTER(_u64), C.POINTER(_u64)]),
    "hpt_reset_diff_tex": (_i, [_vp]),
    "hpt_path_trace_dr": (_i, [_vp, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp, _sz, C.POINTER(_f)]),
    "hpt_path_trace_dr_dev": (_i, [_vp, _u32, _u32, _u32, _vp, _u32, _vp, _vp, _vp, _sz, _vp, _vp]),
    "hpt_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "hpt_image2d4f_regularizer_dev": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "hpt_image2d4f_regularizer": (_i, [_vp, _i, _i, _vp, _vp]),
    "hpt_get_execution_time": (_i, [_vp, C.c_char_p, C.POINTER(_f)]),
    "hpt_set_instrumentation": (_i, [_vp, _i]),
    "hpt_get_counters": (_i, [_vp, C.POINTER(_u64)]),
    "hpt_get_dr_counters": (_i, [_vp, C.POINTER(_u64)]),
    "hpt_set_launch_config": (_i, [_vp, _i]),
    "hpt_set_accel_layout": (_i, [_vp, _i]),
    "hpt_set_schedule": (_i, [_vp, _i, _i, _i, _i]),
    "hpt_comm_get_unique_id": (_i, [_vp, _vp]),
    "hpt_comm_init": (_i, [_vp, _i, _i, _vp]),
    "hpt_comm_destroy": (_i, [_vp]),
    "hpt_reduce_framebuffer": (_i, [_vp, _vp, _sz, _i, _vp]),
    "hpt_allreduce_grad": (_i, [_vp, _vp, _sz, _vp]),
    "hpt_get_schedule": (_i, [_vp, C.POINTER(_i), C.POINTER(_u32)]),
    "hpt_get_last_launch": (_i, [_vp, C.POINTER(_u32)]),
    "hpt_get_accel_info": (_i, [_vp, C.POINTER(_f)]),
    "hpt_set_option": (_i, [_vp, C.c_char_p, _i]),
    "hpt_get_commit_time": (_i, [_vp, C.POINTER(_f)]),
    "hpt_last_kernel_ms": (_i, [_vp, C.POINTER(_f)]),
}

_LIB = None


class HydraHipError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree HIP library and bind every ABI symbol; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HydraHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _LIB = lib
    return _LIB


COUNTER_NAMES = ("rays", "nodes", "tris", "surface_hits", "shadow_rays", "paths", "instances_entered", "tex_fetches",
                 "cyc_queue_regen", "cyc_trace_nearest", "cyc_shade", "cyc_trace_shadow", "cyc_path_end", "loop_trips", "wave_node_iters", "wave_tri_iters")


class DevArray:
    """A float32 array in HBM owned through the C ABI's hpt_device_* helpers (tests and tools that do not link the HIP runtime)."""

    def __init__(self, integ, ptr, shape):
        self.integ, self.ptr, self.shape = integ, ptr, tuple(shape)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * 4

    def upload(self, host):
        host = np.ascontiguousarray(host, np.float32)
        assert host.size == self.size
        self.integ._chk(self.integ.L.hpt_device_copy(self.integ.h, self.ptr, host.ctypes.data, self.nbytes, 1))

    def download(self):
        out = np.zeros(self.shape, np.float32)
        self.integ._chk(self.integ.L.hpt_device_copy(self.integ.h, out.ctypes.data, self.ptr, self.nbytes, 2))
        return out

    def free(self):
        if self.ptr:
            self.integ._chk(self.integ.L.hpt_device_free(self.integ.h, self.ptr))
            self.ptr = None


class HipIntegrator:
    """Integrator-shaped front end of the HIP core. One instance = one hpt_ctx = one GPU."""

    def __init__(self, scene: SceneData = None, params: Params = None, device: int = 0, accel_layout: int = 0):
        self.L = load_library()
        h = _vp()
        rc = self.L.hpt_create(device, C.byref(h))
        if rc != 0:
            raise HydraHipError(f"hpt_create(device={device}) failed with code {rc}: no usable MI355X / HIP device")
        self.h = h
        self.scene = None
        self.params = None
        self.W = self.H = self.N = 0
        if accel_layout:
            self._chk(self.L.hpt_set_accel_layout(self.h, accel_layout))   # 1 = two-level TLAS/BLAS, 2 = single-level, 3 = triangle sweep (tiny scenes)
        if scene is not None:
            self.LoadScene(scene, params)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.hpt_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise HydraHipError(f"hydra_hip error {rc}: {self.L.hpt_last_error(self.h).decode()}")

    # ---- LoadScene / CommitDeviceData / UpdateMembersPlainData / PackXYBlock (main.cpp:249-267) ---------------------
    def LoadScene(self, scene: SceneData, params: Params = None):
        self.scene = scene
        self._desc = scene.desc()
        self.CommitDeviceData()
        self.UpdateMembersPlainData(params if params is not None else scene.params())
        lines = np.ascontiguousarray(scene.lens_lines, np.float32).reshape(-1, 4)       # m_lines / m_physSize (lens simulation; none: off)
        self._chk(self.L.hpt_set_optics(self.h, lines.ctypes.data if lines.size else None, lines.shape[0], float(scene.phys_size[0]), float(scene.phys_size[1])))
        self.PackXYBlock(self.W, self.H, 1)
        self.InitRandomGens(self.N)

    def CommitDeviceData(self):
        self._chk(self.L.hpt_upload_scene(self.h, C.byref(self._desc)))

    def UpdateMembersPlainData(self, params: Params):
        self.params = params
        self._chk(self.L.hpt_update_params(self.h, C.byref(params)))
        self.W, self.H = params.winWidth, params.winHeight
        self.N = self.W * self.H

    def PackXYBlock(self, tidX, tidY, a_passNum=1):
        self._chk(self.L.hpt_pack_xy(self.h, tidX, tidY))

    def InitRandomGens(self, a_maxThreads, first_seed=0):
        """Integrator::InitRandomGens; first_seed != 0 seeds the generators as threads first_seed.. of one larger call (sample sharding)."""
        self._chk(self.L.hpt_init_random_gens_from(self.h, a_maxThreads, first_seed))

    def packed_xy(self):
        out = np.zeros(self.N, np.uint32)
        self._chk(self.L.hpt_get_packed_xy(self.h, out.ctypes.data, self.N))
        return out

    def random_gens(self):
        out = np.zeros((self.N, 2), np.uint32)
        self._chk(self.L.hpt_get_random_gens(self.h, out.ctypes.data, self.N))
        return out

    def set_random_gens(self, gens):
        gens = np.ascontiguousarray(gens, np.uint32).reshape(-1, 2)      # one uint2 per thread
        self._chk(self.L.hpt_set_random_gens(self.h, gens.ctypes.data, gens.shape[0]))

    def Update_m_materials(self, first, mats):
        mats = np.ascontiguousarray(mats)
        self._chk(self.L.hpt_update_materials(self.h, first, mats.size, mats.ctypes.data))

    def Update_m_lights(self, first, lights):
        lights = np.ascontiguousarray(lights)
        self._chk(self.L.hpt_update_lights(self.h, first, lights.size, lights.ctypes.data))

    def Update_m_matIdOffsets(self, mat_vert_offset):
        """Integrator::Update_m_matIdOffsets (integrator_pt.h:470): re-upload m_matVertOffset (uint32 [numGeoms, 2])."""
        mvo = np.ascontiguousarray(mat_vert_offset, np.uint32)
        self._chk(self.L.hpt_update_mat_id_offsets(self.h, mvo.ctypes.data, mvo.size // 2))

    # ---- the hot path -----------------------------------------------------------------------------------------------
    def PathTraceBlock(self, tid, channels, out_color, a_passNum, tid_begin=0):
        """Integrator::PathTraceBlock(tid, channels, out_color, a_passNum); `tid` = number of threads (pixels)."""
        assert out_color.dtype == np.float32 and out_color.flags["C_CONTIGUOUS"]
        self._chk(self.L.hpt_path_trace_block(self.h, tid_begin, tid, channels, out_color.ctypes.data, a_passNum))

    def NaivePathTraceBlock(self, tid, channels, out_color, a_passNum, tid_begin=0):
        assert out_color.dtype == np.float32 and out_color.flags["C_CONTIGUOUS"]
        self._chk(self.L.hpt_naive_path_trace_block(self.h, tid_begin, tid, channels, out_color.ctypes.data, a_passNum))

    def path_trace_block_dev(self, dev_ptr, pass_num, tid_begin=0, tid_count=None, channels=4, naive=False, stream=None):
        tid_count = self.N - tid_begin if tid_count is None else tid_count
        self._chk(self.L.hpt_path_trace_block_dev(self.h, tid_begin, tid_count, channels, dev_ptr, pass_num, int(naive), stream))

    def PathTraceFromInputRaysBlock(self, tid, channels, in_rayPosAndNear, in_rayDirAndFar, out_color, a_passNum):
        """Integrator::PathTraceFromInputRaysBlock; rays: float32 [tid, 4] (RayPosAndW / RayDirAndT), camera space."""
        for a in (in_rayPosAndNear, in_rayDirAndFar, out_color):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        self._chk(self.L.hpt_path_trace_from_input_rays_block(self.h, tid, channels, in_rayPosAndNear.ctypes.data, in_rayDirAndFar.ctypes.data, out_color.ctypes.data, a_passNum))

    def render(self, spp, channels=4, naive=False):
        img = np.zeros((self.H, self.W, channels), np.float32)
        (self.NaivePathTraceBlock if naive else self.PathTraceBlock)(self.N, channels, img, spp)
        return img

    def GetExecutionTime(self, name):
        out = (C.c_float * 4)(0, 0, 0, 0)
        self._chk(self.L.hpt_get_execution_time(self.h, name.encode(), out))
        return list(out)

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._chk(self.L.hpt_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- ISceneObject queries -----------------------------------------------------------------------------------------
    def RayQuery_NearestHitMotion(self, pos_near, dir_far, time):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], HIT_DTYPE)
        self._chk(self.L.hpt_ray_query_nearest_motion(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], float(time), out.ctypes.data))
        return out

    def RayQuery_AnyHitMotion(self, pos_near, dir_far, time):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], np.uint32)
        self._chk(self.L.hpt_ray_query_any_motion(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], float(time), out.ctypes.data))
        return out

    def RayQuery_NearestHit(self, pos_near, dir_far):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], HIT_DTYPE)
        self._chk(self.L.hpt_ray_query_nearest(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], out.ctypes.data))
        return out

    def RayQuery_AnyHit(self, pos_near, dir_far):
        pos_near = np.ascontiguousarray(pos_near, np.float32)
        dir_far = np.ascontiguousarray(dir_far, np.float32)
        out = np.zeros(pos_near.shape[0], np.uint32)
        self._chk(self.L.hpt_ray_query_any(self.h, pos_near.ctypes.data, dir_far.ctypes.data, pos_near.shape[0], out.ctypes.data))
        return out

    # ---- IntegratorDR -----------------------------------------------------------------------------------------------------
    def PutDiffTex2D(self, texId, width, height, channels):
        off, size = _u64(0), _u64(0)
        rc = self.L.hpt_put_diff_tex2d(self.h, texId, width, height, channels, C.byref(off), C.byref(size))
        if rc != 0 and size.value == 0 and off.value == 0xFFFFFFFFFFFFFFFF:
            return (off.value, 0)          # reference behaviour for a bad id: message + (size_t(-1), 0)
        self._chk(rc)
        return (off.value, size.value)

    def PathTraceDR(self, tid, channels, out_color, a_passNum, a_refImg, a_data, a_dataGrad, tid_begin=0):
        a_refImg = np.ascontiguousarray(a_refImg, np.float32)
        a_data = np.ascontiguousarray(a_data, np.float32)
        assert a_dataGrad.dtype == np.float32 and a_dataGrad.size == a_data.size
        loss = C.c_float(0)
        self._chk(self.L.hpt_path_trace_dr(self.h, tid_begin, tid, channels, out_color.ctypes.data, a_passNum, a_refImg.ctypes.data,
                                           a_data.ctypes.data, a_dataGrad.ctypes.data, a_data.size, C.byref(loss)))
        return loss.value

    # ---- device-resident arrays for the *_dev entry points (hpt_device_malloc / copy / free) ---------------------------------
    def dev_array(self, host: np.ndarray):
        """Upload a float32 array; returns a DevArray (free()d with the integrator or explicitly)."""
        host = np.ascontiguousarray(host, np.float32)
        p = _vp()
        self._chk(self.L.hpt_device_malloc(self.h, host.nbytes, C.byref(p)))
        a = DevArray(self, p.value, host.shape)
        a.upload(host)
        return a

    def PathTraceDR_dev(self, out: "DevArray", a_passNum, ref: "DevArray", data: "DevArray", grad: "DevArray", loss: "DevArray", tid_begin=0, tid=None, channels=4):
        """PathTraceDR with every array resident in HBM (drmain's loop): memset(grad), memset(loss), launch. The loss word holds the
        sum over pixels of the per-sample losses / a_passNum (divide by W*H for PathTraceDR's return value)."""
        tid = self.N - tid_begin if tid is None else tid
        self._chk(self.L.hpt_device_memset(self.h, grad.ptr, 0, grad.nbytes))
        self._chk(self.L.hpt_device_memset(self.h, loss.ptr, 0, 4))
        self._chk(self.L.hpt_path_trace_dr_dev(self.h, tid_begin, tid, channels, out.ptr, a_passNum, ref.ptr, data.ptr, grad.ptr, data.size, loss.ptr, None))

    def AdamStep_dev(self, state: "DevArray", grad: "DevArray", momentum: "DevArray", gsq: "DevArray", it: int):
        """AdamOptimizer<float>::step(state, grad, iter) (diff_render/adam.h:43-62) on device arrays."""
        self._chk(self.L.hpt_adam_step_dev(self.h, state.ptr, grad.ptr, momentum.ptr, gsq.ptr, state.size, int(it), None))

    # ---- instrumentation ----------------------------------------------------------------------------------------------------
    def set_instrumentation(self, enabled: bool):
        self._chk(self.L.hpt_set_instrumentation(self.h, int(enabled)))

    def counters(self):
        out = (_u64 * 16)()
        self._chk(self.L.hpt_get_counters(self.h, out))
        return dict(zip(COUNTER_NAMES, [int(v) for v in out]))

    def dr_counters(self):
        out = (_u64 * 16)()
        self._chk(self.L.hpt_get_dr_counters(self.h, out))
        return dict(zip(("records", "records_with_taps", "cyc_record_store", "cyc_sweep", "sweep_wave_trips", "sweep_lanes", "atomic_wave_insts", "sweep_bounces", "records_stored"), [int(v) for v in out]))

    def set_tid_interleave(self, chunk: int, stride: int):
        self._chk(self.L.hpt_set_tid_interleave(self.h, chunk, stride))

    def set_schedule(self, schedule: int, refill_below: int = 0, trace_blocks_per_cu: int = 0, groups: int = 0):
        """0 automatic, 1 persistent megakernel, 2 wavefront (shade kernel + trace kernel with ray replacement), 3 megakernel with block-local ray
        repacking, 4 the wavefront schedule in one launch (block-owned slots; heavy gltf / emissive scenes, elsewhere the automatic choice). hpt_set_schedule."""
        self._chk(self.L.hpt_set_schedule(self.h, schedule, refill_below, trace_blocks_per_cu, groups))

    def Image2D4fRegularizer(self, data, grad):
        """grad += d RegLossImage2D4f / d data (diff_render/integrator_dr.cpp:361-367); float32 [h, w, 4] arrays."""
        assert data.dtype == np.float32 and grad.dtype == np.float32 and data.shape == grad.shape and data.shape[-1] == 4
        self._chk(self.L.hpt_image2d4f_regularizer(self.h, data.shape[1], data.shape[0], data.ctypes.data, grad.ctypes.data))

    def set_option(self, name: str, value: int):
        self._chk(self.L.hpt_set_option(self.h, name.encode(), value))

    def accel_info(self):
        out = (C.c_float * 4)()
        self._chk(self.L.hpt_get_accel_info(self.h, out))
        return {"sah_node_visits": out[0], "inst_tris": int(out[1]), "instances": int(out[2]), "flat": int(out[3]) == 1,
                "layout": ("two-level", "flat", "sweep")[int(out[3])]}

    def commit_time(self):
        """The last CommitScene: host ms, upload ms, device refit ms, refitted (bool)."""
        out = (C.c_float * 4)()
        self._chk(self.L.hpt_get_commit_time(self.h, out))
        return {"host_ms": out[0], "upload_ms": out[1], "refit_ms": out[2], "device_ms": out[2], "refitted": int(out[3]) == 1, "device_built": int(out[3]) == 2}

    def UpdateInstance(self, inst_id, matrix_rowmajor):
        """ISceneObject::UpdateInstance (CrossRT.h:134); takes effect at the next CommitScene."""
        from .scene import colmajor
        cm = colmajor(np.asarray(matrix_rowmajor))
        self._chk(self.L.hpt_update_instance(self.h, inst_id, cm.ctypes.data))

    def CommitScene(self, options=4):
        """ISceneObject::CommitScene (CrossRT.h:109); options as BuildOptions (:8-14): 1 BUILD_LOW, 2 BUILD_MEDIUM (the single-level tree is built on the
        device), 4 BUILD_HIGH (the host's SAH build)."""
        self._chk(self.L.hpt_commit_scene(self.h, options))

    def last_launch(self):
        """What the last PathTrace* call walked: schedule, 4-wide compressed tree, 64-byte shading records, HBM part of the stacks."""
        out = (C.c_uint32 * 4)()
        self._chk(self.L.hpt_get_last_launch(self.h, out))
        return {"schedule": int(out[0]), "wide_nodes": bool(out[1]), "shade_records": bool(out[2]), "deep_stack": bool(out[3])}

    def last_schedule(self):
        s, it = C.c_int(0), C.c_uint32(0)
        self._chk(self.L.hpt_get_schedule(self.h, C.byref(s), C.byref(it)))
        return s.value, it.value

    def set_launch_config(self, blocks_per_cu: int):
        self._chk(self.L.hpt_set_launch_config(self.h, blocks_per_cu))

    def device_info(self):
        cu, wf = C.c_int(0), C.c_int(0)
        name = C.create_string_buffer(128)
        self._chk(self.L.hpt_device_info(self.h, C.byref(cu), C.byref(wf), name, 128))
        return {"cus": cu.value, "wavefront": wf.value, "arch": name.value.decode()}
