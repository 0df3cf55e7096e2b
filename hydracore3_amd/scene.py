"""Host-side scene tables for the HIP path tracer (and, in tests, for the CPU oracle).

This module produces exactly the flat buffers HydraCore3's ``Integrator::LoadScene`` fills
(``integrator_pt.h:472-500``): ``m_materials`` (320-byte records, ``include/cmaterial.h:187-203``),
``m_lights`` (320-byte records, ``include/clight.h:19-56``), ``m_vData8f``, ``m_triIndices``,
``m_matIdByPrimId``, ``m_matVertOffset``, ``m_normMatrices``, ``m_remapInst`` and the camera matrices.

Two producers:
  * :func:`load_hydra_xml` -- a small reader of the Hydra scene XML + VSGF + image4ub files for the scenes the
    reference ships (``scenes/test_035``, ``scenes/test_228``), following ``integrator_pt_scene.cpp:421-943``,
    ``integrator_pt_scene_mat.cpp:280-450``, ``integrator_pt_scene_lgt.cpp:5-222``, ``integrator_pt_scene_tex.cpp:53-93``
    and ``external/LiteScene/cmesh4.cpp:140-167``.  It is a fixture loader, not a replacement for the reference's
    scene loader (which stays in front of the HIP core in a real integration).
  * :class:`SceneBuilder` -- programmatic construction for the synthetic benchmark scenes (bench/scenes.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

# ---------------------------------------------------------------------------------------------------------------------
# record layouts
# ---------------------------------------------------------------------------------------------------------------------
MATERIAL_DTYPE = np.dtype([
    ("mtype", "<u4"), ("cflags", "<u4"), ("lightId", "<u4"), ("nonlinear", "<u4"),
    ("texid", "<u4", 4), ("spdid", "<u4", 4), ("datai", "<u4", 4),
    ("colors", "<f4", (4, 4)), ("row0", "<f4", (4, 4)), ("row1", "<f4", (4, 4)),
    ("data", "<f4", 16),
])
assert MATERIAL_DTYPE.itemsize == 320

LIGHT_DTYPE = np.dtype([
    ("matrix", "<f4", 16), ("iesMatrix", "<f4", 16),
    ("samplerRow0", "<f4", 4), ("samplerRow1", "<f4", 4), ("samplerRow0Inv", "<f4", 4), ("samplerRow1Inv", "<f4", 4),
    ("pos", "<f4", 4), ("intensity", "<f4", 4), ("norm", "<f4", 4),
    ("size", "<f4", 2), ("pdfA", "<f4"), ("geomType", "<u4"),
    ("distType", "<u4"), ("flags", "<u4"), ("pdfTableOffset", "<u4"), ("pdfTableSize", "<u4"),
    ("specId", "<u4"), ("texId", "<u4"), ("iesId", "<u4"), ("mult", "<f4"),
    ("pdfTableSizeX", "<u4"), ("pdfTableSizeY", "<u4"), ("camBackTexId", "<u4"), ("lightCos1", "<f4"),
    ("lightCos2", "<f4"), ("matId", "<u4"), ("dummy2", "<f4"), ("dummy3", "<f4"),
])
assert LIGHT_DTYPE.itemsize == 320

# include/cmaterial.h:26-46
GLTF_COMPONENT_LAMBERT, GLTF_COMPONENT_COAT, GLTF_COMPONENT_METAL, GLTF_COMPONENT_ORENNAYAR = 1, 2, 4, 16
FLAG_FOUR_TEXTURES, FLAG_PACK_FOUR_PARAMS_IN_TEXTURE, FLAG_INVERT_GLOSINESS = 256, 512, 1024
FLAG_NMAP_INVERT_X, FLAG_NMAP_INVERT_Y, FLAG_NMAP_SWAP_XY = 32, 64, 128
MAT_TYPE_GLTF, MAT_TYPE_CONDUCTOR, MAT_TYPE_DIFFUSE, MAT_TYPE_DIELECTRIC = 1, 3, 4, 7
MAT_TYPE_GLASS = 2
MAT_TYPE_BLEND = 6
MAT_TYPE_PLASTIC = 5
MAT_TYPE_LIGHT_SOURCE = 0xEFFFFFFF
# include/cmaterial.h:67-147 (slots in Material::colors / Material::data)
GLTF_COLOR_BASE, GLTF_COLOR_COAT, GLTF_COLOR_METAL = 0, 1, 2
GLTF_FLOAT_MI_FDR_INT, GLTF_FLOAT_MI_FDR_EXT, GLTF_FLOAT_MI_SSW, GLTF_FLOAT_ALPHA = 0, 1, 2, 3
GLTF_FLOAT_GLOSINESS, GLTF_FLOAT_IOR, GLTF_FLOAT_ROUGH_ORENNAYAR, GLTF_FLOAT_REFL_COAT = 4, 5, 6, 7
EMISSION_COLOR, EMISSION_MULT = 0, 0
# thin film (include/cmaterial.h:45, 164-179)
MAT_TYPE_THIN_FILM = 8
(FILM_ROUGH_U, FILM_ROUGH_V, FILM_PRECOMP_FLAG, FILM_PRECOMP_OFFSET, FILM_ETA_OFFSET, FILM_K_OFFSET, FILM_ETA_SPECID_OFFSET, FILM_K_SPECID_OFFSET, FILM_ETA_EXT,
 FILM_THICKNESS_OFFSET, FILM_THICKNESS_MIN, FILM_THICKNESS_MAX, FILM_THICKNESS_MAP, FILM_THICKNESS, FILM_LAYERS_COUNT, FILM_TRANSPARENT) = range(16)
# include/clight.h:5-17
LIGHT_GEOM_RECT, LIGHT_GEOM_DISC, LIGHT_GEOM_SPHERE, LIGHT_GEOM_DIRECT, LIGHT_GEOM_POINT, LIGHT_GEOM_ENV = 1, 2, 3, 4, 5, 6
LIGHT_DIST_LAMBERT, LIGHT_DIST_OMNI, LIGHT_DIST_SPOT = 0, 1, 2
LIGHT_FLAG_POINT_AREA, LIGHT_FLAG_PROJECTIVE = 1, 2
# LiteImage::Sampler numbering
ADDR_WRAP, ADDR_CLAMP = 0, 2
FILTER_NEAREST, FILTER_LINEAR = 0, 1
TEX_RGBA8, TEX_RGBA32F, TEX_R32F = 0, 1, 2
# integrator_pt.h:330-332
INTEGRATOR_STUPID_PT, INTEGRATOR_SHADOW_PT, INTEGRATOR_MIS_PT = 0, 1, 2
UINT_MAX = 0xFFFFFFFF


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("flags", C.c_uint32),
                ("addressU", C.c_uint32), ("addressV", C.c_uint32), ("filter", C.c_uint32), ("reserved", C.c_uint32),
                ("data", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("numGeoms", C.c_uint32), ("numInsts", C.c_uint32), ("numVerts", C.c_uint32), ("numTris", C.c_uint32),
                ("vPos4f", C.c_void_p), ("vData8f", C.c_void_p), ("triIndices", C.c_void_p), ("matIdByPrimId", C.c_void_p),
                ("matVertOffset", C.c_void_p), ("geomTriCount", C.c_void_p), ("geomVertCount", C.c_void_p),
                ("instGeomId", C.c_void_p), ("instMatrices", C.c_void_p), ("normMatrices", C.c_void_p),
                ("remapInst", C.c_void_p), ("allRemapLists", C.c_void_p),
                ("allRemapListsLen", C.c_uint32), ("allRemapListsSize", C.c_uint32),
                ("materials", C.c_void_p), ("numMaterials", C.c_uint32), ("numLights", C.c_uint32),
                ("lights", C.c_void_p), ("textures", C.POINTER(TextureDesc)),
                ("numTextures", C.c_uint32), ("numArrays1f", C.c_uint32), ("arrays1f", C.c_void_p),
                ("instMatricesMotion", C.c_void_p), ("instHasMotion", C.c_void_p), ("normMatrices2Offs", C.c_uint32), ("reserved", C.c_uint32),
                # spectral rendering: m_spec_values, m_spec_offset_sz, m_cie_xyz, m_camResponseSpectrumId / m_camResponseType
                ("specValues", C.c_void_p), ("specOffsetSz", C.c_void_p), ("numSpecValues", C.c_uint32), ("numSpectra", C.c_uint32),
                ("cieXYZ", C.c_void_p), ("numCieXYZ", C.c_uint32), ("camResponseSpectrumId", C.c_int32 * 3),
                ("camResponseType", C.c_uint32), ("reserved2", C.c_uint32),
                # thin films: m_films_thickness_vec, m_films_spec_id_vec, m_films_eta_k_vec, m_precomp_thin_films
                ("filmsThickness", C.c_void_p), ("filmsSpecId", C.c_void_p), ("filmsEtaK", C.c_void_p), ("precompThinFilms", C.c_void_p),
                ("numFilmsThickness", C.c_uint32), ("numFilmsSpecId", C.c_uint32), ("numFilmsEtaK", C.c_uint32), ("numPrecompThinFilms", C.c_uint32),
                # spectra given by textures: m_spec_tex_ids_wavelengths, m_spec_tex_offset_sz
                ("specTexIdsWavelengths", C.c_void_p), ("specTexOffsetSz", C.c_void_p), ("numSpecTexBands", C.c_uint32), ("reserved3", C.c_uint32)]


class FilmParams(C.Structure):
    """hpt_film_params (include/hydra_hip.h)."""
    _fields_ = [("spectralMode", C.c_int32), ("extIOR", C.c_float), ("layers", C.c_uint32), ("thicknessMap", C.c_int32),
                ("thicknessMin", C.c_float), ("thicknessMax", C.c_float), ("numSpectra", C.c_uint32), ("reserved", C.c_uint32),
                ("eta", C.c_void_p), ("k", C.c_void_p), ("etaSpecId", C.c_void_p), ("kSpecId", C.c_void_p), ("thickness", C.c_void_p),
                ("specValues", C.c_void_p), ("specOffsetSz", C.c_void_p), ("cieXYZ", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("projInv", C.c_float * 16), ("worldViewInv", C.c_float * 16),
                ("winStartX", C.c_int32), ("winStartY", C.c_int32), ("winWidth", C.c_int32), ("winHeight", C.c_int32),
                ("fbWidth", C.c_int32), ("fbHeight", C.c_int32),
                ("traceDepth", C.c_uint32), ("integratorType", C.c_uint32), ("renderLayer", C.c_uint32),
                ("tileSize", C.c_uint32), ("spectralMode", C.c_uint32), ("envSpecIdPlus1", C.c_uint32),
                ("exposureMult", C.c_float), ("camLensRadius", C.c_float), ("camTargetDist", C.c_float), ("envSpecMult", C.c_float),
                ("camRespoceRGB", C.c_float * 4), ("envColor", C.c_float * 4),
                ("envTexId", C.c_uint32), ("envLightId", C.c_uint32), ("envCamBackId", C.c_uint32), ("envEnableSam", C.c_uint32),
                ("envSamRow0", C.c_float * 4), ("envSamRow1", C.c_float * 4)]


class Hit(C.Structure):     # CRT_Hit, external/CrossRT/CrossRT.h:23-30
    _fields_ = [("t", C.c_float), ("primId", C.c_uint32), ("instId", C.c_uint32), ("geomId", C.c_uint32),
                ("coords", C.c_float * 4)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("primId", "<u4"), ("instId", "<u4"), ("geomId", "<u4"), ("coords", "<f4", 4)])


# ---------------------------------------------------------------------------------------------------------------------
# LiteMath-convention matrices (column-major storage: flat[col*4+row])
# ---------------------------------------------------------------------------------------------------------------------
def colmajor(m_rowmajor_4x4: np.ndarray) -> np.ndarray:
    """4x4 array indexed [row, col] -> 16 floats in LiteMath's column-major order."""
    return np.asarray(m_rowmajor_4x4, dtype=np.float64).T.reshape(16).astype(np.float32)


def perspective_matrix(fovy_deg: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    ymax = z_near * math.tan(fovy_deg * math.pi / 360.0)
    xmax = ymax * aspect
    left, right, bottom, top = -xmax, xmax, -ymax, ymax
    t, t2, t3, t4 = 2.0 * z_near, right - left, top - bottom, z_far - z_near
    m = np.zeros((4, 4))
    m[0, 0] = t / t2
    m[1, 1] = t / t3
    m[0, 2] = (right + left) / t2
    m[1, 2] = (top + bottom) / t3
    m[2, 2] = (-z_far - z_near) / t4
    m[3, 2] = -1.0
    m[2, 3] = (-t * z_far) / t4
    return m


def look_at(eye, center, up) -> np.ndarray:
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up / np.linalg.norm(up))
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return m


def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def scale(x, y, z):
    return np.diag([x, y, z, 1.0])


def rotate_y(deg):
    a = math.radians(deg)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
    return m


def rotate_x(deg):
    a = math.radians(deg)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
    return m


# ---------------------------------------------------------------------------------------------------------------------
# material / light constructors
# ---------------------------------------------------------------------------------------------------------------------
def _blank_material() -> np.ndarray:
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["lightId"] = UINT_MAX
    m["texid"] = (0, UINT_MAX, 0, 0)      # texid[1] = 0xFFFFFFFF: no normal map (integrator_pt_scene.cpp:604)
    m["spdid"] = UINT_MAX
    for i in range(4):
        m["row0"][i] = (1, 0, 0, 0)
        m["row1"][i] = (0, 1, 0, 0)
    return m


def fresnel_diffuse_reflectance(eta: float) -> float:
    """mi::fresnel_diffuse_reflectance (mi_materials.cpp:105-130), float arithmetic."""
    f = np.float32
    eta = f(eta)
    inv_eta = f(1.0) / eta
    approx_1 = f(0.0636) * inv_eta + (eta * (eta * f(-1.4399) + f(0.7099)) + f(0.6681))
    acc = f(0.0)
    for c in reversed([0.919317, -3.4793, 6.75335, -7.80989, 4.98554, -1.36881]):
        acc = acc * inv_eta + f(c)
    return float(approx_1 if eta < 1.0 else acc)


def set_mi_plastic(m, int_ior, ext_ior, diffuse, specular):
    """SetMiPlastic (mi_materials.cpp:455-469)."""
    m["colors"][GLTF_COLOR_BASE] = diffuse
    m["colors"][GLTF_COLOR_COAT] = specular
    eta = np.float32(int_ior) / np.float32(ext_ior)
    m["data"][GLTF_FLOAT_IOR] = eta
    m["data"][GLTF_FLOAT_MI_FDR_INT] = fresnel_diffuse_reflectance(1.0 / float(eta))
    m["data"][GLTF_FLOAT_MI_FDR_EXT] = fresnel_diffuse_reflectance(float(eta))
    d_mean = 0.3333333 * (diffuse[0] + diffuse[1] + diffuse[2])
    s_mean = 0.3333333 * (specular[0] + specular[1] + specular[2])
    m["data"][GLTF_FLOAT_MI_SSW] = s_mean / (d_mean + s_mean)


def material_lambert(color, tex_id=0) -> np.ndarray:
    """Diffuse-only branch of ConvertOldHydraMaterial (integrator_pt_scene_mat.cpp:410-419, 446-447)."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_GLTF
    m["cflags"] = GLTF_COMPONENT_LAMBERT
    m["colors"][GLTF_COLOR_BASE] = (*color[:3], 0.0)
    m["data"][GLTF_FLOAT_GLOSINESS] = 1.0
    m["data"][GLTF_FLOAT_IOR] = 0.0
    m["texid"][0] = tex_id
    return m


def material_gltf(base_color, metalness=0.0, glossiness=1.0, coat=1.0, ior=1.5, tex_id=0) -> np.ndarray:
    """ConvertGLTFMaterial (integrator_pt_scene_mat.cpp:176-278)."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_GLTF
    m["cflags"] = GLTF_COMPONENT_LAMBERT | GLTF_COMPONENT_COAT
    m["data"][GLTF_FLOAT_REFL_COAT] = coat
    bc = (*base_color[:3], base_color[3] if len(base_color) > 3 else 0.0)
    m["colors"][GLTF_COLOR_METAL] = 1.0
    m["data"][GLTF_FLOAT_ALPHA] = metalness
    m["data"][GLTF_FLOAT_GLOSINESS] = glossiness
    set_mi_plastic(m, ior, 1.0, bc, (1.0, 1.0, 1.0, 1.0))
    m["colors"][GLTF_COLOR_COAT] = 1.0
    m["texid"][0] = tex_id
    return m


def material_diffuse(color, roughness=0.0, tex_id=0) -> np.ndarray:
    """type="diffuse" material (integrator_pt_scene_mat.cpp:516-572): Lambert or Oren-Nayar."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_DIFFUSE
    m["cflags"] = GLTF_COMPONENT_LAMBERT
    m["colors"][0] = (*color[:3], 0.0)
    if roughness > 0.0:
        m["cflags"] = GLTF_COMPONENT_ORENNAYAR
        m["data"][0] = roughness
    m["texid"][0] = tex_id
    return m


def material_conductor(eta, k, alpha_u=0.0, alpha_v=0.0, reflectance=(1, 1, 1, 1)) -> np.ndarray:
    """type="rough_conductor" material (integrator_pt_scene_mat.cpp:452-514), RGB mode: scalar eta / k."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_CONDUCTOR
    m["colors"][0] = reflectance
    m["data"][0], m["data"][1], m["data"][2], m["data"][3] = alpha_u, alpha_v, eta, k
    return m


def material_dielectric(int_ior=1.5, ext_ior=1.00028) -> np.ndarray:
    """type="dielectric" material (integrator_pt_scene_mat.cpp:574-617)."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_DIELECTRIC
    m["colors"][0] = 1.0
    m["colors"][1] = 1.0
    m["data"][0], m["data"][1] = ext_ior, int_ior
    return m


def material_glass(color_reflect=(1.0, 1.0, 1.0), color_transp=(1.0, 1.0, 1.0), ior=1.5) -> np.ndarray:
    """MAT_TYPE_GLASS, the legacy Hydra glass (include/cmat_glass.h:236-277; slots cmaterial.h:85-92): specular reflection or
    refraction chosen by the Fresnel term, never lit by shadow rays."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_GLASS
    m["colors"][0] = (*color_reflect[:3], 0.0)
    m["colors"][1] = (*color_transp[:3], 0.0)
    m["data"][2] = ior
    return m


def material_blend(mat_id1: int, mat_id2: int, weight: float, mask_tex=0) -> np.ndarray:
    """MAT_TYPE_BLEND (include/cmaterial.h:43,155; integrator_pt_mat.cpp:23-77): picks child 2 with probability weight * mask.x, child 1
    otherwise; children are material ids (they may be blends themselves, up to BLEND_STACK_SIZE = 4 levels)."""
    m = _blank_material()
    m["mtype"] = MAT_TYPE_BLEND
    m["data"][0] = weight
    m["datai"][0], m["datai"][1] = mat_id1, mat_id2
    m["texid"][0] = mask_tex
    return m


def set_normal_map(m, tex_id: int, invert_x=False, invert_y=False, swap_xy=False, row0=(1, 0, 0, 0), row1=(0, 1, 0, 0)):
    """Normal-map bump of any material (integrator_pt_scene.cpp:603-638, integrator_pt_mat.cpp:94-107): texid[1] + the NMAP flags."""
    m["texid"][1] = tex_id
    m["row0"][1] = row0
    m["row1"][1] = row1
    m["cflags"] = int(m["cflags"]) | (FLAG_NMAP_INVERT_X if invert_x else 0) | (FLAG_NMAP_INVERT_Y if invert_y else 0) | (FLAG_NMAP_SWAP_XY if swap_xy else 0)
    return m


def material_emissive(color, mult=1.0, light_id=UINT_MAX, tex_id=0) -> np.ndarray:
    m = _blank_material()
    m["mtype"] = MAT_TYPE_LIGHT_SOURCE
    m["colors"][EMISSION_COLOR] = (*color[:3], 0.0)
    m["data"][EMISSION_MULT] = mult
    m["lightId"] = light_id
    m["texid"][0] = tex_id
    return m


def _blank_light() -> np.ndarray:
    lt = np.zeros((), dtype=LIGHT_DTYPE)
    lt["distType"] = LIGHT_DIST_LAMBERT
    lt["iesId"] = lt["texId"] = lt["specId"] = lt["camBackTexId"] = lt["matId"] = UINT_MAX
    lt["samplerRow0"] = (1, 0, 0, 0)
    lt["samplerRow1"] = (0, 1, 0, 0)
    lt["mult"] = 1.0
    return lt


def light_rect(matrix, half_length, half_width, color, mult, disk_radius=None) -> np.ndarray:
    """rect / disk branch of LoadLightSourceFromNode (integrator_pt_scene_lgt.cpp:69-91)."""
    m = np.asarray(matrix, dtype=np.float64)
    lt = _blank_light()
    lt["pos"] = (m @ np.array([0, 0, 0, 1.0])).astype(np.float32)
    n = m @ np.array([0, -1.0, 0, 0])
    lt["norm"] = (n / np.linalg.norm(n)).astype(np.float32)
    lt["intensity"] = (*color[:3], 0.0)
    lt["mult"] = mult
    sc = [np.linalg.norm(m[:3, i]) for i in range(3)]
    rot = m.copy()
    rot[:, 3] = (0, 0, 0, 1)
    lt["matrix"] = colmajor(rot)
    lt["size"] = (half_length, half_width)
    if disk_radius is not None:
        lt["geomType"] = LIGHT_GEOM_DISC
        lt["size"][0] = disk_radius
        lt["pdfA"] = 1.0 / (math.pi * disk_radius * disk_radius * sc[0] * sc[2])
    else:
        lt["geomType"] = LIGHT_GEOM_RECT
        lt["pdfA"] = 1.0 / (4.0 * half_length * half_width * sc[0] * sc[2])
    return lt


def light_sphere(matrix, radius, color, mult) -> np.ndarray:
    m = np.asarray(matrix, dtype=np.float64)
    lt = _blank_light()
    r = radius * np.linalg.norm(m[:3, 0])
    lt["pos"] = (m @ np.array([0, 0, 0, 1.0])).astype(np.float32)
    lt["norm"] = (0, -1, 0, 0)
    lt["intensity"] = (*color[:3], 0.0)
    lt["mult"] = mult
    lt["geomType"] = LIGHT_GEOM_SPHERE
    lt["size"] = (r, r)
    lt["pdfA"] = 1.0 / (4.0 * math.pi * r * r)
    return lt


def light_point(matrix, color, mult, dist="omni", cos1=0.0, cos2=0.0) -> np.ndarray:
    m = np.asarray(matrix, dtype=np.float64)
    lt = _blank_light()
    lt["pos"] = (m @ np.array([0, 0, 0, 1.0])).astype(np.float32)
    n = m @ np.array([0, -1.0, 0, 0])
    lt["norm"] = (n / np.linalg.norm(n)).astype(np.float32)
    lt["intensity"] = (*color[:3], 0.0)
    lt["mult"] = mult
    lt["geomType"] = LIGHT_GEOM_POINT
    lt["distType"] = {"omni": LIGHT_DIST_OMNI, "uniform": LIGHT_DIST_OMNI, "ies": LIGHT_DIST_OMNI,
                      "spot": LIGHT_DIST_SPOT}.get(dist, LIGHT_DIST_LAMBERT)
    lt["pdfA"] = 1.0
    lt["lightCos1"], lt["lightCos2"] = cos1, cos2
    return lt


def pdf_table_from_image(tex):
    """PdfTableFromImage + PrefixSumm (integrator_pt_scene_lgt.cpp:219-270): luminance max(r, g, b) at the texel centres, floored at a tenth
    of the mean, prefix-summed in double; w*h + 1 floats."""
    assert tex.fmt == TEX_RGBA32F, "only HDR maps are sampled explicitly (integrator_pt_scene.cpp:461-463)"
    lum = tex.data[..., :3].max(axis=-1).astype(np.float32)
    row_sums = np.zeros(tex.height, np.float32)
    for x in range(tex.width):                                    # avgInRow += lum, in float, left to right
        row_sums = (row_sums + lum[:, x]).astype(np.float32)
    avg = np.float32(0.0)
    for y in range(tex.height):
        avg = np.float32(avg + row_sums[y])
    avg = np.float32(avg / np.float32(tex.width * tex.height))
    lum = np.maximum(lum, np.float32(0.1) * avg).reshape(-1)
    acc = np.concatenate([[0.0], np.cumsum(lum.astype(np.float64))])
    return acc.astype(np.float32), tex.width, tex.height


def set_projective(lt, matrix, fov, z_near, z_far, tex_id=UINT_MAX):
    """The <projective> node of a spot light (integrator_pt_scene_lgt.cpp:136-159): iesMatrix = perspective(fov, 1, near, far) * lookAt(pos,
    M * (0, -1, 0), rot(M) * (0, 0, 1)); with a texture the light becomes a slide projector (LIGHT_FLAG_PROJECTIVE)."""
    m = np.asarray(matrix, dtype=np.float64)
    rot = m.copy(); rot[:, 3] = (0, 0, 0, 1)
    pos = (m @ np.array([0, 0, 0, 1.0]))[:3]
    look_at_t = (m @ np.array([0, -1.0, 0, 1.0]))[:3]
    up_t = (rot @ np.array([0, 0, 1.0, 1.0]))[:3]
    lt["iesMatrix"] = colmajor(perspective_matrix(fov, 1.0, z_near, z_far) @ look_at(pos, look_at_t, up_t))
    if tex_id != UINT_MAX:
        lt["flags"] = int(lt["flags"]) | LIGHT_FLAG_PROJECTIVE
        lt["texId"] = tex_id
    return lt


def light_directional(matrix, color, mult) -> np.ndarray:
    m = np.asarray(matrix, dtype=np.float64)
    lt = _blank_light()
    lt["pos"] = (m @ np.array([0, 0, 0, 1.0])).astype(np.float32)
    n = m @ np.array([0, -1.0, 0, 0])
    lt["norm"] = (n / np.linalg.norm(n)).astype(np.float32)
    lt["intensity"] = (*color[:3], 0.0)
    lt["mult"] = mult
    lt["geomType"] = LIGHT_GEOM_DIRECT
    return lt


# ---------------------------------------------------------------------------------------------------------------------
class Texture:
    def __init__(self, data: np.ndarray, fmt: int, srgb: bool, addr_u=ADDR_WRAP, addr_v=ADDR_WRAP, filt=FILTER_LINEAR):
        self.fmt, self.srgb, self.addr_u, self.addr_v, self.filter = fmt, srgb, addr_u, addr_v, filt
        if fmt == TEX_RGBA8:
            self.data = np.ascontiguousarray(data, dtype=np.uint32)
            self.height, self.width = self.data.shape
        elif fmt == TEX_RGBA32F:
            self.data = np.ascontiguousarray(data, dtype=np.float32)
            self.height, self.width = self.data.shape[:2]
        else:
            self.data = np.ascontiguousarray(data, dtype=np.float32)
            self.height, self.width = self.data.shape


def white_dummy_texture() -> Texture:
    """m_textures[0]: 1x1 white, NEAREST / CLAMP (integrator_pt_scene_tex.cpp:7-16)."""
    return Texture(np.full((1, 1), 0xFFFFFFFF, dtype=np.uint32), TEX_RGBA8, False, ADDR_CLAMP, ADDR_CLAMP, FILTER_NEAREST)


class SceneData:
    """Flat scene buffers + camera/settings, convertible to the C ABI structures."""

    def __init__(self):
        self.vpos = np.zeros((0, 4), np.float32)
        self.vdata = np.zeros((0, 8), np.float32)
        self.tri_indices = np.zeros((0,), np.uint32)
        self.mat_id_by_prim = np.zeros((0,), np.uint32)
        self.mat_vert_offset = []       # (triOffset, vertOffset) per geom
        self.geom_tri_count, self.geom_vert_count = [], []
        self.inst_geom, self.inst_matrices, self.remap_inst = [], [], []
        self.lens_lines, self.phys_size = np.zeros((0, 4), np.float32), (0.0, 0.0)   # lens simulation: m_lines {radius, thickness, ior, aperture}, m_physSize
        self.inst_motion = {}                                 # instance id -> matrix at the end of the motion (hydraxml.h:170-176 <motion matrix=..>)
        # spectral rendering (m_spectral_mode): spectra resampled at 1 nm from LAMBDA_MIN (spectrum.cpp: ResampleUniform), {offset, size} per id,
        # the CIE 1931 observer (the reference embeds the tabulated one; without a copy of it in this image: the analytic fit, see cie_xyz_fit)
        self.spectral_mode = 0
        self.spec_values = np.zeros(0, np.float32)
        self.spec_offset_sz = []
        self.cie_xyz = None
        self.cam_response_spectrum_id = (-1, -1, -1)
        self.cam_response_type = 0                            # CAM_RESPONCE_XYZ = 0, CAM_RESPONCE_RGB = 1 (integrator_pt.h:531-534)
        self.cam_respoce_rgb = (1.0, 1.0, 1.0, 1.0)           # m_camRespoceRGB
        self.env_spec_id, self.env_spec_mult = UINT_MAX, 1.0  # m_envSpecId, m_envSpecMult: the sky light's spectrum and multiplier (integrator_pt_scene.cpp:456-457)
        # thin films (integrator_pt.h:587-590): thickness per film, eta then k per layer and their spectrum ids, the precomputed tables
        self.spec_tex_ids_wavelengths, self.spec_tex_offset_sz = [], []   # spectra given by textures: (texture, wavelength) per band, (first band, bands) per spectrum id
        self.films_thickness, self.films_spec_id, self.films_eta_k = [], [], []
        self.precomp_thin_films = np.zeros(0, np.float32)
        self.all_remap_lists = np.zeros((1,), np.int32)     # no lists: just the trailing offset 0
        self.all_remap_lists_size = 0
        self.materials, self.lights = [], []
        self.textures = [white_dummy_texture()]
        # camera / settings
        self.width, self.height = 512, 512
        self.fov, self.near, self.far = 45.0, 0.01, 100.0
        self.cam_pos, self.cam_look_at, self.cam_up = (0, 0, 15), (0, 0, 0), (0, 1, 0)
        self.trace_depth, self.spp = 6, 1
        self.env_color = (0.0, 0.0, 0.0, 0.0)
        # the environment map of LoadSceneLights (integrator_pt_scene.cpp:441-478): texture, light entry when sampled explicitly, camera back
        self.env_tex_id, self.env_light_id, self.env_cam_back_id, self.env_enable_sam = UINT_MAX, UINT_MAX, UINT_MAX, 0
        self.env_sam_row0, self.env_sam_row1 = (1.0, 0.0, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0)
        self.arrays1f = np.zeros(0, np.float32)               # m_arrays1f: pdf tables
        self.exposure_mult = 1.0
        self.cam_lens_radius = 0.0
        self._keep = []

    # -- geometry ----------------------------------------------------------------------------------------------------
    def add_mesh(self, pos4, norm4, tang4, uv2, indices, mat_ids) -> int:
        """LoadSceneGeometry (integrator_pt_scene.cpp:727-837): returns geomId."""
        pos4 = np.asarray(pos4, np.float32).reshape(-1, 4)
        nv = pos4.shape[0]
        vd = np.zeros((nv, 8), np.float32)
        vd[:, 0:3] = np.asarray(norm4, np.float32).reshape(nv, -1)[:, :3]
        vd[:, 4:7] = np.asarray(tang4, np.float32).reshape(nv, -1)[:, :3]
        uv2 = np.asarray(uv2, np.float32).reshape(nv, 2)
        vd[:, 3], vd[:, 7] = uv2[:, 0], uv2[:, 1]
        indices = np.asarray(indices, np.uint32).reshape(-1)
        nt = indices.size // 3
        mat_ids = np.asarray(mat_ids, np.uint32).reshape(-1)
        if mat_ids.size == 1:
            mat_ids = np.full((nt,), mat_ids[0], np.uint32)
        self.mat_vert_offset.append((self.mat_id_by_prim.size, self.vpos.shape[0]))
        self.geom_tri_count.append(nt)
        self.geom_vert_count.append(nv)
        self.vpos = np.concatenate([self.vpos, pos4])
        self.vdata = np.concatenate([self.vdata, vd])
        self.tri_indices = np.concatenate([self.tri_indices, indices])
        self.mat_id_by_prim = np.concatenate([self.mat_id_by_prim, mat_ids])
        return len(self.geom_tri_count) - 1

    def add_meshes(self, meshes):
        """Bulk add_mesh for many meshes (one concatenation instead of one per mesh)."""
        vp, vd, ti, mi = [self.vpos], [self.vdata], [self.tri_indices], [self.mat_id_by_prim]
        tri_off, vert_off = self.mat_id_by_prim.size, self.vpos.shape[0]
        first = len(self.geom_tri_count)
        for pos4, norm4, tang4, uv2, indices, mat_ids in meshes:
            pos4 = np.asarray(pos4, np.float32).reshape(-1, 4)
            nv = pos4.shape[0]
            d = np.zeros((nv, 8), np.float32)
            d[:, 0:3] = np.asarray(norm4, np.float32).reshape(nv, -1)[:, :3]
            d[:, 4:7] = np.asarray(tang4, np.float32).reshape(nv, -1)[:, :3]
            uv2 = np.asarray(uv2, np.float32).reshape(nv, 2)
            d[:, 3], d[:, 7] = uv2[:, 0], uv2[:, 1]
            indices = np.asarray(indices, np.uint32).reshape(-1)
            nt = indices.size // 3
            mat_ids = np.asarray(mat_ids, np.uint32).reshape(-1)
            if mat_ids.size == 1:
                mat_ids = np.full((nt,), mat_ids[0], np.uint32)
            self.mat_vert_offset.append((tri_off, vert_off))
            self.geom_tri_count.append(nt)
            self.geom_vert_count.append(nv)
            vp.append(pos4); vd.append(d); ti.append(indices); mi.append(mat_ids)
            tri_off += nt
            vert_off += nv
        self.vpos, self.vdata = np.concatenate(vp), np.concatenate(vd)
        self.tri_indices, self.mat_id_by_prim = np.concatenate(ti), np.concatenate(mi)
        return first

    def add_instance(self, geom_id, matrix_rowmajor, remap_list=-1, light_id=-1, motion_matrix=None) -> int:
        """AddInstance, or AddInstanceMotion(geomId, {matrix, matrix_motion}, 2) when motion_matrix is given (integrator_pt_scene.cpp:852-885)."""
        self.inst_geom.append(geom_id)
        self.inst_matrices.append(np.asarray(matrix_rowmajor, np.float64).reshape(4, 4))
        self.remap_inst.append((remap_list, light_id))
        if motion_matrix is not None:
            self.inst_motion[len(self.inst_geom) - 1] = np.asarray(motion_matrix, np.float64).reshape(4, 4)
        return len(self.inst_geom) - 1

    def set_remap_lists(self, lists):
        """LoadSceneRemapLists (integrator_pt_scene.cpp:909-924): lists of (from,to) pairs sorted by `from`."""
        flat, offs = [], []
        for lst in lists:
            offs.append(len(flat))
            flat.extend(int(v) for v in lst)
        offs.append(len(flat))
        self.all_remap_lists_size = len(flat)
        self.all_remap_lists = np.asarray(flat + offs, np.int32)

    def add_texture(self, tex: Texture) -> int:
        self.textures.append(tex)
        return len(self.textures) - 1

    def material_plastic(self, color, alpha=0.1, int_ior=1.49, ext_ior=1.000277, nonlinear=0, tex_id=0, row0=(1, 0, 0, 0), row1=(0, 1, 0, 0)):
        """LoadPlasticMaterial (integrator_pt_scene_mat.cpp:675-757): MAT_TYPE_PLASTIC (include/cmat_plastic.h), a rough dielectric coat over a
        diffuse base; its 64-entry transmittance table goes to m_arrays1f (offset in datai[0]). The table and the two scalars come from
        hpt_plastic_precompute = mi::fresnel_coat_precompute (mi_materials.cpp:377-451), the one implementation both loaders share."""
        from .api import load_library
        m = np.zeros((), dtype=MATERIAL_DTYPE)
        m["mtype"], m["lightId"], m["nonlinear"] = MAT_TYPE_PLASTIC, UINT_MAX, nonlinear
        m["spdid"] = UINT_MAX
        m["texid"] = (tex_id, UINT_MAX, 0, 0)
        for k in range(4):
            m["row0"][k] = (1, 0, 0, 0); m["row1"][k] = (0, 1, 0, 0)
        m["row0"][0], m["row1"][0] = row0, row1
        c4 = np.array([*color[:3], color[3] if len(color) > 3 else 0.0], np.float32)
        m["colors"][0] = c4
        m["data"][1] = np.float32(int_ior) / np.float32(ext_ior)              # PLASTIC_IOR_RATIO
        a = np.float32(alpha) if alpha != 0.0 else np.float32(1e-6)           # "dirty hack" (:723-727)
        m["data"][0] = a                                                      # PLASTIC_ROUGHNESS
        table = np.zeros(64, np.float32)
        refl, weight = C.c_float(0), C.c_float(0)
        spec = np.ones(4, np.float32)
        rc = load_library().hpt_plastic_precompute(float(a), float(np.float32(int_ior)), float(np.float32(ext_ior)), c4.ctypes.data, spec.ctypes.data,
                                                   table.ctypes.data, C.byref(refl), C.byref(weight))
        if rc != 0:
            raise ValueError("hpt_plastic_precompute rejected the parameters")
        m["data"][3], m["data"][2] = refl.value, weight.value                 # PLASTIC_PRECOMP_REFLECTANCE, PLASTIC_SPEC_SAMPLE_WEIGHT
        m["datai"][0] = self.arrays1f.size
        self.arrays1f = np.concatenate([self.arrays1f, table]).astype(np.float32)
        return m

    def material_thin_film(self, layers, substrate=None, alpha=0.0, ext_ior=1.00028, transparent=0, thickness_map=None, alpha_tex=None):
        """LoadThinFilmMaterial (integrator_pt_scene_mat.cpp:1020-1193): MAT_TYPE_THIN_FILM (include/cmat_film.h). `layers`: the films, outermost first,
        as dicts {eta, k, thickness, eta_spec, k_spec}; `substrate`: {eta, k, eta_spec, k_spec} or None (the last film then stands for it, as in
        the reference, where FILM_LAYERS_COUNT counts <layers> children plus the material's own <eta>). `alpha`: a number or (alpha_u, alpha_v);
        thickness_map = (min, max, tex_id, row0, row1); alpha_tex = (tex_id, row0, row1). The reflectance / transmittance tables come from
        hpt_film_precompute (csrc/film_precompute.h), the one implementation both loaders share, for the scene's CURRENT spectral_mode."""
        from .api import load_library
        m = np.zeros((), dtype=MATERIAL_DTYPE)
        m["mtype"], m["lightId"] = MAT_TYPE_THIN_FILM, UINT_MAX
        m["colors"][0] = (1, 1, 1, 0)
        m["spdid"] = UINT_MAX
        m["texid"] = (0, UINT_MAX, 0, 0)                                      # (the reference zeroes all four; slot 1 is the normal map: none)
        for k in range(4):
            m["row0"][k] = (1, 0, 0, 0); m["row1"][k] = (0, 1, 0, 0)
        au, av = (alpha, alpha) if np.isscalar(alpha) else alpha
        if alpha_tex is not None:
            m["texid"][0], m["row0"][0], m["row1"][0] = alpha_tex
            if alpha_tex[0] != 0:
                au = av = 1.0
        m["data"][FILM_ROUGH_U], m["data"][FILM_ROUGH_V] = au, av
        fbits = lambda u: np.uint32(u).view(np.float32)
        tmap = thickness_map is not None
        if tmap:
            m["data"][FILM_THICKNESS_MIN], m["data"][FILM_THICKNESS_MAX] = thickness_map[0], thickness_map[1]
            m["texid"][2], m["row0"][2], m["row1"][2] = thickness_map[2], thickness_map[3], thickness_map[4]
        m["data"][FILM_THICKNESS_MAP] = fbits(1 if tmap else 0)
        m["data"][FILM_ETA_EXT] = np.float32(ext_ior)
        t_off, s_off, e_off = len(self.films_thickness), len(self.films_spec_id), len(self.films_eta_k)
        m["data"][FILM_THICKNESS_OFFSET], m["data"][FILM_ETA_SPECID_OFFSET], m["data"][FILM_ETA_OFFSET] = fbits(t_off), fbits(s_off), fbits(e_off)
        stack = list(layers) + ([substrate] if substrate is not None else [])
        for l in layers:
            if "thickness" in l:
                self.films_thickness.append(np.float32(l["thickness"]))
        for l in stack:
            self.films_eta_k.append(np.float32(l.get("eta", 0.0))); self.films_spec_id.append(int(l.get("eta_spec", UINT_MAX)))
        if len(self.films_thickness) <= t_off:
            raise ValueError("thin film: no layer carries a thickness (the reference reads m_films_thickness_vec past its end)")
        n = len(stack)
        m["data"][FILM_THICKNESS] = self.films_thickness[t_off]
        m["data"][FILM_LAYERS_COUNT] = fbits(n)
        m["data"][FILM_K_SPECID_OFFSET], m["data"][FILM_K_OFFSET] = fbits(len(self.films_spec_id)), fbits(len(self.films_eta_k))
        for l in stack:
            self.films_eta_k.append(np.float32(l.get("k", 0.0))); self.films_spec_id.append(int(l.get("k_spec", UINT_MAX)))
        m["data"][FILM_TRANSPARENT] = fbits(int(transparent))
        # the tables
        keep = []
        def arr(a, dt):
            a = np.ascontiguousarray(a, dt); keep.append(a); return a.ctypes.data
        fp = FilmParams()
        fp.spectralMode, fp.extIOR, fp.layers, fp.thicknessMap = int(self.spectral_mode), float(np.float32(ext_ior)), n, int(tmap)
        fp.thicknessMin, fp.thicknessMax = float(m["data"][FILM_THICKNESS_MIN]), float(m["data"][FILM_THICKNESS_MAX])
        fp.eta, fp.k = arr(self.films_eta_k[e_off:e_off + n], np.float32), arr(self.films_eta_k[e_off + n:e_off + 2 * n], np.float32)
        fp.etaSpecId, fp.kSpecId = arr(self.films_spec_id[s_off:s_off + n], np.uint32), arr(self.films_spec_id[s_off + n:s_off + 2 * n], np.uint32)
        fp.thickness = arr(self.films_thickness[t_off:], np.float32)
        fp.numSpectra = len(self.spec_offset_sz)
        fp.specValues = arr(self.spec_values, np.float32) if len(self.spec_offset_sz) else None
        fp.specOffsetSz = arr(np.asarray(self.spec_offset_sz, np.uint32).reshape(-1, 2), np.uint32) if len(self.spec_offset_sz) else None
        fp.cieXYZ = arr(self.cie_xyz if self.cie_xyz is not None else cie_xyz_fit(), np.float32)
        lib = load_library()
        count, pre = C.c_uint64(0), C.c_int(0)
        if lib.hpt_film_precompute(C.byref(fp), None, 0, C.byref(count), C.byref(pre)) != 0:
            raise ValueError("hpt_film_precompute rejected the parameters")
        m["data"][FILM_PRECOMP_FLAG] = fbits(pre.value)
        m["data"][FILM_PRECOMP_OFFSET] = fbits(self.precomp_thin_films.size if pre.value else 0)
        if pre.value:
            table = np.zeros(count.value, np.float32)
            if lib.hpt_film_precompute(C.byref(fp), table.ctypes.data, table.size, C.byref(count), C.byref(pre)) != 0:
                raise ValueError("hpt_film_precompute failed")
            self.precomp_thin_films = np.concatenate([self.precomp_thin_films, table]).astype(np.float32)
        return m

    def set_optics(self, lines, sensor_diagonal=0.035, scale=1.0, order="sensor_to_scene"):
        """LoadOpticsFromNode (integrator_pt_scene.cpp:1078-1141): `lines` = (id, curvature_radius, thickness, ior, semi_diameter) tuples, sorted by id
        (descending for order == "scene_to_sensor"); m_physSize from the sensor diagonal. The reference reads m_aspect there before anything
        has set it; height / width of the frame is taken here (the film's aspect in pbrt's RealisticCamera, which this code follows)."""
        ids = sorted(lines, key=lambda l: l[0], reverse=(order == "scene_to_sensor"))
        self.lens_lines = np.array([[scale * l[1], scale * l[2], l[3], scale * l[4]] for l in ids], np.float32).reshape(-1, 4)
        aspect = np.float32(self.height) / np.float32(self.width)
        d = np.float32(sensor_diagonal)
        px = np.float32(2.0) * np.sqrt(d * d / (np.float32(1.0) + aspect * aspect), dtype=np.float32)
        self.phys_size = (float(px), float(aspect * px))

    def set_environment(self, color, tex_id=UINT_MAX, mult=1.0, row0=(1, 0, 0, 0), row1=(0, 1, 0, 0), cam_back=UINT_MAX, sample=None):
        """The LIGHT_GEOM_ENV branch of LoadLightSourceFromNode + LoadSceneLights (integrator_pt_scene_lgt.cpp:36-59,
        integrator_pt_scene.cpp:441-486): a plain colour, or a lat-long map that is sampled explicitly when it is HDR (`sample` overrides
        the reference's rule 'EXR file or more than 4 bytes per pixel'). Returns the light id of the sampled map, or -1."""
        self.env_color = (*[float(v) for v in color[:3]], float(color[3]) if len(color) > 3 else 0.0)
        self.env_sam_row0, self.env_sam_row1 = tuple(float(v) for v in row0), tuple(float(v) for v in row1)
        self.env_tex_id, self.env_light_id, self.env_cam_back_id, self.env_enable_sam = tex_id, UINT_MAX, cam_back, 0
        if tex_id == UINT_MAX:
            return -1
        tex = self.textures[tex_id]
        if sample is None:
            sample = tex.fmt == TEX_RGBA32F
        self.env_enable_sam = 1 if sample else 0
        if not sample:
            return -1
        lt = _blank_light()
        lt["intensity"] = self.env_color
        lt["mult"] = mult
        lt["geomType"], lt["distType"] = LIGHT_GEOM_ENV, LIGHT_DIST_OMNI
        lt["texId"], lt["camBackTexId"] = tex_id, cam_back
        lt["samplerRow0"], lt["samplerRow1"] = row0, row1
        t = np.eye(4)
        t[0, :], t[1, :] = row0, row1
        ti = np.linalg.inv(t)
        lt["samplerRow0Inv"], lt["samplerRow1Inv"] = ti[0, :].astype(np.float32), ti[1, :].astype(np.float32)
        table, w, h = pdf_table_from_image(tex)
        lt["pdfTableOffset"], lt["pdfTableSize"], lt["pdfTableSizeX"], lt["pdfTableSizeY"] = self.arrays1f.size, table.size, w, h
        self.arrays1f = np.concatenate([self.arrays1f, table]).astype(np.float32)
        self.env_light_id = len(self.lights)
        self.lights.append(lt)
        return self.env_light_id

    # -- C structures ------------------------------------------------------------------------------------------------
    def tile_size(self) -> int:
        """SetViewport (integrator_pt.h:379-389)."""
        for ts in (8, 4, 2):
            if self.width % ts == 0 and self.height % ts == 0:
                return ts
        return 1

    def params(self, integrator=INTEGRATOR_MIS_PT, render_layer=0, trace_depth=None) -> Params:
        p = Params()
        aspect = float(self.width) / float(self.height)
        proj = perspective_matrix(self.fov, aspect, self.near, self.far)
        wv = look_at(self.cam_pos, self.cam_look_at, self.cam_up)
        p.projInv[:] = colmajor(np.linalg.inv(proj)).tolist()
        p.worldViewInv[:] = colmajor(np.linalg.inv(wv)).tolist()
        p.winStartX = p.winStartY = 0
        p.winWidth = p.fbWidth = self.width
        p.winHeight = p.fbHeight = self.height
        p.traceDepth = self.trace_depth if trace_depth is None else trace_depth
        p.integratorType = integrator
        p.renderLayer = render_layer
        p.tileSize = self.tile_size()
        p.spectralMode = int(self.spectral_mode)
        p.envSpecIdPlus1 = (int(self.env_spec_id) + 1) & UINT_MAX
        p.envSpecMult = float(self.env_spec_mult)
        p.exposureMult = self.exposure_mult
        p.camLensRadius = self.cam_lens_radius
        p.camTargetDist = float(np.linalg.norm(np.asarray(self.cam_look_at, float) - np.asarray(self.cam_pos, float)))
        p.camRespoceRGB[:] = [float(v) for v in self.cam_respoce_rgb]
        p.envColor[:] = list(self.env_color)
        p.envTexId, p.envLightId, p.envCamBackId, p.envEnableSam = self.env_tex_id, self.env_light_id, self.env_cam_back_id, self.env_enable_sam
        p.envSamRow0[:] = list(self.env_sam_row0)
        p.envSamRow1[:] = list(self.env_sam_row1)
        return p

    def desc(self) -> SceneDesc:
        d = SceneDesc()
        k = self._keep = []

        def ptr(a):
            a = np.ascontiguousarray(a)
            k.append(a)
            return a.ctypes.data

        ni = len(self.inst_geom)
        d.numGeoms, d.numInsts = len(self.geom_tri_count), ni
        d.numVerts, d.numTris = self.vpos.shape[0], self.mat_id_by_prim.size
        d.vPos4f, d.vData8f = ptr(self.vpos), ptr(self.vdata)
        d.triIndices, d.matIdByPrimId = ptr(self.tri_indices), ptr(self.mat_id_by_prim)
        d.matVertOffset = ptr(np.asarray(self.mat_vert_offset, np.uint32).reshape(-1, 2))
        d.geomTriCount = ptr(np.asarray(self.geom_tri_count, np.uint32))
        d.geomVertCount = ptr(np.asarray(self.geom_vert_count, np.uint32))
        d.instGeomId = ptr(np.asarray(self.inst_geom, np.uint32))
        mats = np.stack([colmajor(m) for m in self.inst_matrices]) if ni else np.zeros((0, 16), np.float32)
        # m_normMatrices[i] = transpose(inverse4x4(M_i))  (integrator_pt_scene.cpp:877)
        nm = np.stack([colmajor(np.linalg.inv(m).T) for m in self.inst_matrices]) if ni else np.zeros((0, 16), np.float32)
        # motion blur (integrator_pt_scene.cpp:848-897): once one instance moves, m_normMatrices carries a second half for the end of the motion
        # (the same matrix again for the instances that stay put) and m_normMatrices2Offs = the instance count
        d.instMatricesMotion, d.instHasMotion, d.normMatrices2Offs = None, None, 0
        if self.inst_motion and ni:
            end = [self.inst_motion.get(i, self.inst_matrices[i]) for i in range(ni)]
            nm = np.concatenate([nm, np.stack([colmajor(np.linalg.inv(m).T) for m in end])])
            d.instMatricesMotion = ptr(np.stack([colmajor(m) for m in end]))
            d.instHasMotion = ptr(np.asarray([1 if i in self.inst_motion else 0 for i in range(ni)], np.uint32))
            d.normMatrices2Offs = ni
        d.instMatrices, d.normMatrices = ptr(mats), ptr(nm)
        d.remapInst = ptr(np.asarray(self.remap_inst, np.int32).reshape(-1, 2))
        d.allRemapLists = ptr(self.all_remap_lists)
        d.allRemapListsLen, d.allRemapListsSize = self.all_remap_lists.size, self.all_remap_lists_size
        marr = np.array(self.materials, dtype=MATERIAL_DTYPE) if self.materials else np.zeros((0,), MATERIAL_DTYPE)
        larr = np.array(self.lights, dtype=LIGHT_DTYPE) if self.lights else np.zeros((0,), LIGHT_DTYPE)
        d.materials, d.numMaterials = ptr(marr), marr.size
        d.lights, d.numLights = ptr(larr) if larr.size else None, larr.size
        tarr = (TextureDesc * len(self.textures))()
        for i, t in enumerate(self.textures):
            tarr[i].width, tarr[i].height, tarr[i].format = t.width, t.height, t.fmt
            tarr[i].flags = 1 if t.srgb else 0
            tarr[i].addressU, tarr[i].addressV, tarr[i].filter = t.addr_u, t.addr_v, t.filter
            tarr[i].data = ptr(t.data)
        k.append(tarr)
        d.textures, d.numTextures = tarr, len(self.textures)
        d.arrays1f, d.numArrays1f = (ptr(self.arrays1f.astype(np.float32)), int(self.arrays1f.size)) if self.arrays1f.size else (None, 0)
        d.specValues, d.specOffsetSz, d.numSpecValues, d.numSpectra, d.cieXYZ, d.numCieXYZ = None, None, 0, 0, None, 0
        d.camResponseSpectrumId = (C.c_int32 * 3)(-1, -1, -1); d.camResponseType = 0
        if self.spec_offset_sz:
            d.specValues = ptr(np.asarray(self.spec_values, np.float32)); d.numSpecValues = int(np.asarray(self.spec_values).size)
            d.specOffsetSz = ptr(np.asarray(self.spec_offset_sz, np.uint32).reshape(-1, 2)); d.numSpectra = len(self.spec_offset_sz)
            cie = self.cie_xyz if self.cie_xyz is not None else cie_xyz_fit()
            d.cieXYZ = ptr(np.asarray(cie, np.float32).reshape(-1, 4)); d.numCieXYZ = int(np.asarray(cie).reshape(-1, 4).shape[0])
            d.camResponseSpectrumId = (C.c_int32 * 3)(*[int(v) for v in self.cam_response_spectrum_id]); d.camResponseType = int(self.cam_response_type)
        d.specTexIdsWavelengths, d.specTexOffsetSz, d.numSpecTexBands = None, None, 0
        if self.spec_tex_ids_wavelengths and len(self.spec_tex_offset_sz) == len(self.spec_offset_sz):
            d.specTexIdsWavelengths, d.numSpecTexBands = ptr(np.asarray(self.spec_tex_ids_wavelengths, np.uint32).reshape(-1, 2)), len(self.spec_tex_ids_wavelengths)
            d.specTexOffsetSz = ptr(np.asarray(self.spec_tex_offset_sz, np.uint32).reshape(-1, 2))
        d.filmsThickness, d.filmsSpecId, d.filmsEtaK, d.precompThinFilms = None, None, None, None
        d.numFilmsThickness = d.numFilmsSpecId = d.numFilmsEtaK = d.numPrecompThinFilms = 0
        if self.films_eta_k:
            d.filmsThickness, d.numFilmsThickness = ptr(np.asarray(self.films_thickness, np.float32)), len(self.films_thickness)
            d.filmsSpecId, d.numFilmsSpecId = ptr(np.asarray(self.films_spec_id, np.uint32)), len(self.films_spec_id)
            d.filmsEtaK, d.numFilmsEtaK = ptr(np.asarray(self.films_eta_k, np.float32)), len(self.films_eta_k)
            if self.precomp_thin_films.size:
                d.precompThinFilms, d.numPrecompThinFilms = ptr(self.precomp_thin_films.astype(np.float32)), int(self.precomp_thin_films.size)
        return d


# ---------------------------------------------------------------------------------------------------------------------
# fixture loader for the scenes the reference ships
# ---------------------------------------------------------------------------------------------------------------------
def load_vsgf(path):
    """cmesh4::LoadMeshFromVSGF (external/LiteScene/cmesh4.cpp:140-167, header cmesh4.h:18-32)."""
    raw = open(path, "rb").read()
    _size, nv, ni, _nm, flags = struct.unpack_from("<QIIII", raw, 0)
    off = 24
    pos = np.frombuffer(raw, "<f4", nv * 4, off).reshape(nv, 4); off += nv * 16
    if not (flags & 8):
        norm = np.frombuffer(raw, "<f4", nv * 4, off).reshape(nv, 4); off += nv * 16
    else:
        norm = np.zeros((nv, 4), np.float32)
    if flags & 1:
        tang = np.frombuffer(raw, "<f4", nv * 4, off).reshape(nv, 4); off += nv * 16
    else:
        tang = np.zeros((nv, 4), np.float32)
    uv = np.frombuffer(raw, "<f4", nv * 2, off).reshape(nv, 2); off += nv * 8
    idx = np.frombuffer(raw, "<u4", ni, off); off += ni * 4
    mats = np.frombuffer(raw, "<u4", ni // 3, off)
    return pos, norm, tang, uv, idx, mats


def load_image4ub(path):
    """integrator_pt_scene_tex.cpp:53-93: 8-byte {w,h} header, then RGBA8 texels."""
    raw = open(path, "rb").read()
    w, h = struct.unpack_from("<II", raw, 0)
    return np.frombuffer(raw, "<u4", w * h, 8).reshape(h, w)


LAMBDA_MIN, LAMBDA_MAX = 360.0, 830.0            # include/cglobals.h:22-23


def resample_uniform(wavelengths, values):
    """Spectrum::ResampleUniform over Spectrum::Sample (spectrum.cpp:7-48): 471 float32 values at LAMBDA_MIN + c nm; zero outside the tabulated
    range, linear in between, in float arithmetic as the reference evaluates it."""
    w = np.asarray(wavelengths, np.float32); v = np.asarray(values, np.float32)
    out = np.zeros(int(LAMBDA_MAX - LAMBDA_MIN + 1), np.float32)
    if w.size == 0:
        return out
    for c in range(out.size):
        lam = np.float32(LAMBDA_MIN + float(c))
        if lam < w[0] or lam > w[-1]:
            continue
        # BinarySearch (spectrum.h:26-40): the last index o in [0, n - 2] with w[o] <= lam
        last, first = w.size - 2, 1
        while last > 0:
            half = last >> 1
            middle = first + half
            if w[middle] <= lam:
                first = middle + 1; last = last - (half + 1)
            else:
                last = half
        o = min(max(first - 1, 0), w.size - 2)
        t = (lam - w[o]) / (w[o + 1] - w[o])
        out[c] = v[o] + t * (v[o + 1] - v[o])                          # LiteMath lerp(a, b, t) = a + t (b - a)
    return out


def load_spd(path):
    """LoadSPDFromFile (spectrum.cpp:50-71): 'lambda value' per line, '#' comments."""
    w, v = [], []
    for line in open(path):
        line = line.rstrip("\n")
        if not line or line[0] == "#":
            continue
        a, b = line.split(" ", 1)
        w.append(float(a)); v.append(float(b))
    return w, v


def spectrum_mean(values):
    """mi::spectrum_mean (mi_materials.cpp:331-374): trapezoid integral of the resampled spectrum over [LAMBDA_MIN, LAMBDA_MAX] in double,
    divided by the range in float."""
    v = np.asarray(values, np.float32).astype(np.float64)
    if v.size < 2 or (v < 0).any():
        raise ValueError("spectrum_mean: a spectrum needs two non-negative samples at least")
    interval = (float(LAMBDA_MAX) - float(LAMBDA_MIN)) / (v.size - 1)
    integral = 0.0
    for i in range(v.size - 1):                                           # summed in the reference's order
        integral += 0.5 * interval * (v[i] + v[i + 1])
    if integral <= 0.0:
        raise ValueError("spectrum_mean: no probability mass")
    return np.float32(np.float32(integral) / np.float32(LAMBDA_MAX - LAMBDA_MIN))


def cie_xyz_fit():
    """The CIE 1931 2-degree observer at 360..830 nm, float32 [471, 4] = {x, y, z, 0}. The reference carries the tabulated functions in its
    source (spectrum.cpp:103-396) and hands them to the integrator as m_cie_xyz. Reference SOURCE is not copied into this repository, so the
    fixture loaders use the analytic multi-lobe fit of Wyman, Sloan and Shirley (JCGT 2013; within about 1 % of the tables): spectral frames
    and RGB thin-film tables made by these loaders differ from the reference renderer's by that much, which the HIP-vs-oracle tests cannot
    see (both sides get the same fit). A HydraCore3 host passes its own table through hpt_scene_desc::cieXYZ - the kernels only read what they are given."""
    lam = np.arange(int(LAMBDA_MAX - LAMBDA_MIN + 1), dtype=np.float64) + LAMBDA_MIN

    def g(mu, s1, s2):
        t = (lam - mu) / np.where(lam < mu, s1, s2)
        return np.exp(-0.5 * t * t)
    x = 1.056 * g(599.8, 37.9, 31.0) + 0.362 * g(442.0, 16.0, 26.7) - 0.065 * g(501.1, 20.4, 26.2)
    y = 0.821 * g(568.8, 46.9, 40.5) + 0.286 * g(530.9, 16.3, 31.1)
    z = 1.217 * g(437.0, 11.8, 36.0) + 0.681 * g(459.0, 26.0, 13.8)
    return np.stack([x, y, z, np.zeros_like(x)], 1).astype(np.float32)


def decode_exr(raw):
    """OpenEXR, the subset tinyexr's LoadEXR is used for by the reference (imageutils.cpp:317-392): single-part scanline files, NONE / ZIPS / ZIP
    compression, HALF / FLOAT / UINT channels. Returns float32 [h, w, 4] in FILE order (top scanline first) the way LoadEXR hands it out:
    R, G, B (, A = 1 when absent); a single-channel file fills all four components with its value."""
    import zlib
    if raw[:4] != b"\x76\x2f\x31\x01":
        raise ValueError("not an OpenEXR file")
    version, = struct.unpack_from("<I", raw, 4)
    if version & 0x1A00:                                              # 0x200 tiled, 0x800 deep, 0x1000 multi-part (0x400 = long names: fine)
        raise NotImplementedError("EXR: tiled / multi-part / deep files are not read")
    p, attrs = 8, {}
    while raw[p] != 0:
        e = raw.index(b"\0", p); name = raw[p:e].decode(); p = e + 1
        e = raw.index(b"\0", p); typ = raw[p:e].decode(); p = e + 1
        size, = struct.unpack_from("<i", raw, p); p += 4
        attrs[name] = (typ, raw[p:p + size]); p += size
    p += 1
    chans, q, cd = [], 0, attrs["channels"][1]
    while cd[q] != 0:
        e = cd.index(b"\0", q); cname = cd[q:e].decode(); q = e + 1
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", cd, q); q += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError("EXR: subsampled channels are not read")
        chans.append((cname, ptype))
    comp = attrs["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    lines_per_block = {0: 1, 2: 1, 3: 16}.get(comp)
    if lines_per_block is None:
        raise NotImplementedError(f"EXR: compression {comp} is not read (NONE, ZIPS, ZIP are)")
    nblocks = (h + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from(f"<{nblocks}Q", raw, p)
    bytes_per = {0: 4, 1: 2, 2: 4}
    line_bytes = sum(bytes_per[t] * w for _n, t in chans)
    planes = {n: np.zeros((h, w), np.float32) for n, _t in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", raw, off)
        data = raw[off + 8:off + 8 + size]
        nl = min(lines_per_block, y1 - y + 1)
        if comp != 0 and size < nl * line_bytes:                       # stored raw when compression does not shrink the block
            t = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int32)
            t = np.cumsum(np.concatenate([t[:1], t[1:] - 128])) & 0xFF       # predictor: d[i] = d[i - 1] + t[i] - 128
            t = t.astype(np.uint8)
            half = (t.size + 1) // 2
            out = np.empty(t.size, np.uint8); out[0::2] = t[:half]; out[1::2] = t[half:]   # de-interleave the two byte halves
            data = out.tobytes()
        q = 0
        for l in range(nl):
            for n, t in chans:                                           # channels are stored in alphabetical order, one run per scanline
                if t == 1: v = np.frombuffer(data, "<f2", w, q).astype(np.float32)
                elif t == 2: v = np.frombuffer(data, "<f4", w, q)
                else: v = np.frombuffer(data, "<u4", w, q).astype(np.float32)
                planes[n][y - y0 + l] = v
                q += bytes_per[t] * w
    img = np.zeros((h, w, 4), np.float32)
    names = [n for n, _t in chans]
    if len(names) == 1:
        img[...] = planes[names[0]][..., None]
    else:
        for k, n in enumerate(("R", "G", "B")):
            if n in planes: img[..., k] = planes[n]
        img[..., 3] = planes["A"] if "A" in planes else 1.0
    return img


def decode_ldr_image(path, raw):
    """.png / .ppm / .bmp -> uint32 [h, w] RGBA8 (r in the low byte), rows in file order - the same decoders as csrc/scene_loader.h
    (8-bit non-interlaced PNG through zlib, binary PPM, uncompressed 24 / 32-bit BMP); JPEG through the library's own reader."""
    import zlib
    low = path.lower()
    if low.endswith(".jpg") or low.endswith(".jpeg"):                        # csrc/jpeg_decode.h, the one JPEG reader of both loaders
        from .api import load_library
        lib = load_library()
        buf = np.frombuffer(raw, np.uint8)
        w, h = C.c_uint32(0), C.c_uint32(0)
        if lib.hpt_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), None, 0) != 0:
            raise NotImplementedError(f"{path}: only 8-bit baseline / progressive Huffman JPEG files (grey or YCbCr) are read")
        out = np.zeros((h.value, w.value, 4), np.uint8)
        if lib.hpt_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), out.ctypes.data, out.size) != 0:
            raise ValueError(f"{path}: JPEG decode failed")
        return np.ascontiguousarray(out).view(np.uint32).reshape(h.value, w.value)
    if low.endswith(".png"):
        if raw[:8] != b"\x89PNG\r\n\x1a\n":
            raise ValueError(f"{path}: not a PNG file")
        p, idat, plte, trns, hdr = 8, b"", b"", b"", None
        while p + 12 <= len(raw):
            ln, typ = struct.unpack_from(">I4s", raw, p)
            d = raw[p + 8:p + 8 + ln]
            if typ == b"IHDR":
                hdr = struct.unpack_from(">IIBBBBB", d)
            elif typ == b"PLTE":
                plte = d
            elif typ == b"tRNS":
                trns = d
            elif typ == b"IDAT":
                idat += d
            elif typ == b"IEND":
                break
            p += 12 + ln
        w, h, depth, ctype, _, _, interlace = hdr
        if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
            raise NotImplementedError(f"{path}: only 8-bit non-interlaced PNG images are read")
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        stride = w * ch
        data = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, stride + 1)
        img = np.zeros((h, stride), np.uint8)
        for y in range(h):                                                   # undo the scanline filters (PNG spec 9)
            ft, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
            up = img[y - 1].astype(np.int32) if y else np.zeros(stride, np.int32)
            if ft == 0:
                out = line
            elif ft == 2:
                out = line + up
            else:
                out = np.zeros(stride, np.int32)
                for x in range(stride):
                    a = out[x - ch] if x >= ch else 0
                    b = up[x]
                    c = up[x - ch] if x >= ch else 0
                    if ft == 1:
                        pred = a
                    elif ft == 3:
                        pred = (a + b) >> 1
                    elif ft == 4:
                        pp = a + b - c
                        pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                        pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    else:
                        raise ValueError(f"{path}: bad PNG filter type {ft}")
                    out[x] = (line[x] + pred) & 255
            img[y] = (out & 255).astype(np.uint8)
        px = img.reshape(h, w, ch).astype(np.uint32)
        if ctype == 0:
            r = g = b = px[..., 0]; a = np.full((h, w), 255, np.uint32)
        elif ctype == 2:
            r, g, b = px[..., 0], px[..., 1], px[..., 2]; a = np.full((h, w), 255, np.uint32)
        elif ctype == 3:
            pal = np.frombuffer(plte, np.uint8).reshape(-1, 3).astype(np.uint32)
            al = np.full(pal.shape[0], 255, np.uint32); al[:len(trns)] = np.frombuffer(trns, np.uint8)[:pal.shape[0]]
            k = px[..., 0]
            r, g, b, a = pal[k, 0], pal[k, 1], pal[k, 2], al[k]
        elif ctype == 4:
            r = g = b = px[..., 0]; a = px[..., 1]
        else:
            r, g, b, a = px[..., 0], px[..., 1], px[..., 2], px[..., 3]
        return (r | (g << 8) | (b << 16) | (a << 24)).astype(np.uint32)
    if low.endswith(".ppm"):
        toks, p = [], 0
        while len(toks) < 4:
            while raw[p:p + 1].isspace():
                p += 1
            if raw[p:p + 1] == b"#":
                p = raw.index(b"\n", p)
                continue
            q = p
            while not raw[q:q + 1].isspace():
                q += 1
            toks.append(raw[p:q]); p = q
        if toks[0] != b"P6" or int(toks[3]) != 255:
            raise NotImplementedError(f"{path}: only binary P6 PPM with 8-bit samples is read")
        w, h = int(toks[1]), int(toks[2])
        px = np.frombuffer(raw, np.uint8, w * h * 3, p + 1).reshape(h, w, 3).astype(np.uint32)
        return (px[..., 0] | (px[..., 1] << 8) | (px[..., 2] << 16) | np.uint32(0xFF000000)).astype(np.uint32)
    if low.endswith(".bmp"):
        off, = struct.unpack_from("<I", raw, 10)
        sw, sh = struct.unpack_from("<ii", raw, 18)
        bpp, = struct.unpack_from("<H", raw, 28)
        comp, = struct.unpack_from("<I", raw, 30)
        if sw <= 0 or sh == 0 or bpp not in (24, 32) or comp not in (0, 3):
            raise NotImplementedError(f"{path}: only uncompressed 24 / 32-bit BMP images are read")
        w, h, bs = sw, abs(sh), bpp // 8
        stride = (w * bs + 3) & ~3
        rows = np.frombuffer(raw, np.uint8, stride * h, off).reshape(h, stride)[:, :w * bs].reshape(h, w, bs).astype(np.uint32)
        a = rows[..., 3] if bs == 4 else np.full((h, w), 255, np.uint32)
        return (rows[..., 2] | (rows[..., 1] << 8) | (rows[..., 0] << 16) | (a << 24)).astype(np.uint32)
    raise NotImplementedError(f"texture file '{path}': only .image4ub / .image4f / .exr / .png / .ppm / .bmp are read here (no JPEG decoder in this image)")


def ies_spherical_texture(path):
    """CreateSphericalTextureFromIES (ies_parser/ies_render.cpp:29-120), axially symmetric photometry only
    (one horizontal angle), which is what scenes/test_228 uses; normalised to max 1
    (integrator_pt_scene_lgt.cpp:171-186)."""
    toks = open(path).read().split("TILT=NONE", 1)[1].split()
    vals = [float(t) for t in toks]
    n_vert, n_horz = int(vals[3]), int(vals[4])
    vals = vals[13:]
    vert = vals[:n_vert]
    horz = vals[n_vert:n_vert + n_horz]
    cand = vals[n_vert + n_horz:]
    if n_horz != 1:
        raise NotImplementedError("only axially symmetric IES files are handled by the fixture loader")
    v0, v1 = vert[0], vert[-1]
    h = n_vert * 2 if (abs(v0) < 1e-5 and abs(v1 - 90) < 1e-5) or (abs(v0 - 90) < 1e-5 and abs(v1 - 180) < 1e-5) else n_vert
    img = np.zeros((h, 1), np.float32)
    step = np.float32(v1 - v0) / np.float32(n_vert)
    theta_grad = np.float32(v0)
    for ti in range(n_vert):
        theta = np.float32(math.pi / 180.0) * theta_grad
        iy = min(int((theta * np.float32(1.0 / math.pi)) * np.float32(h) + np.float32(0.5)), h - 1)
        img[iy, 0] = cand[ti]
        theta_grad = np.float32(theta_grad + step)
    mx = float(img.max()) or 1.0
    return (img * np.float32(1.0 / mx)).astype(np.float32)


def _f(s):
    return [float(v) for v in s.split()]


def load_hydra_xml(xml_path: str, width=None, height=None, spectral=False) -> SceneData:
    folder = os.path.dirname(os.path.abspath(xml_path))
    text = open(xml_path, encoding="utf-8").read()
    text = text.replace('<?xml version="1.0"?>', "")
    root = ET.fromstring("<root>" + text + "</root>")
    sc = SceneData()

    settings = root.find("render_lib/render_settings")
    sc.width = int(settings.findtext("width")) if width is None else width
    sc.height = int(settings.findtext("height")) if height is None else height
    sc.trace_depth = int(settings.findtext("trace_depth") or 0) or 6          # LoadSceneSettings :926-940
    sc.spp = int(settings.findtext("maxRaysPerPixel") or 0) or 1

    cam = root.find("cam_lib/camera")
    sc.fov = float(cam.findtext("fov"))
    sc.near, sc.far = float(cam.findtext("nearClipPlane")), float(cam.findtext("farClipPlane"))
    sc.cam_pos, sc.cam_look_at, sc.cam_up = _f(cam.findtext("position")), _f(cam.findtext("look_at")), _f(cam.findtext("up"))
    optics = cam.find("optical_system")                                       # (the reference's fallback to <optics> never triggers: opticNode != opticNode)
    if optics is not None:
        lines = []
        for k, ln in enumerate(optics.findall("line")):
            ap = ln.get("semi_diameter") if ln.get("semi_diameter") is not None else ln.get("aperture_radius", "0")
            lines.append((int(ln.get("id", k)), float(ln.get("curvature_radius", 0)), float(ln.get("thickness", 0)), float(ln.get("ior", 0)), float(ap)))
        sc.set_optics(lines, float(optics.get("sensor_diagonal", 0.035)), float(optics.get("scale", 1.0)), optics.get("order", ""))

    # textures: LoadSceneTexturesInfo (integrator_pt_scene.cpp:330-355) keeps the nodes with a size, indexed by position; a material's
    # <texture id=..> is loaded on first use, one m_textures entry per distinct (id, address modes, filter) - the HydraSampler
    # equality of integrator_pt.h:75-83 (rows and gamma are not part of the key) - integrator_pt_scene_tex.cpp:105-125
    tex_info = []
    for t in root.findall("textures_lib/texture"):
        w, h = int(t.get("width", 0)), int(t.get("height", 0))
        if w != 0 and h != 0:
            path = t.get("path") if not t.get("loc") else os.path.join(folder, t.get("loc"))
            tex_info.append((path, w, h, int(t.get("bytesize", 0)) // (w * h)))
    tex_cache = {}
    addr_modes = {"clamp": ADDR_CLAMP, "wrap": ADDR_WRAP}

    # LoadSceneSpectrumData (integrator_pt_scene.cpp:358-419): every <spectrum> resampled at 1 nm; one {offset, size} per node, in node order
    sc.spectral_mode = 1 if spectral else 0
    spec_vals = []
    for sn in root.findall("spectra_lib/spectrum"):
        if sn.get("lambda_ref_ids") is not None:                              # a spectrum given by textures (:363-377): "lambda texture lambda texture ..."
            refs = [int(v) for v in _f(sn.get("lambda_ref_ids"))]
            sc.spec_tex_offset_sz.append((len(sc.spec_tex_ids_wavelengths), len(refs) // 2))
            sc.spec_tex_ids_wavelengths += [[refs[2 * k + 1], refs[2 * k]] for k in range(len(refs) // 2)]      # {texture id of the XML, wavelength}
            sc.spec_offset_sz.append((UINT_MAX, 0))
            continue
        sc.spec_tex_offset_sz.append((UINT_MAX, 0))
        if sn.get("value") is not None:                                       # ParseSpectrumStr: "lambda value lambda value ..."
            nums = _f(sn.get("value"))
            wl, vl = nums[0::2], nums[1::2]
        else:
            wl, vl = load_spd(os.path.join(folder, sn.get("loc")))
        u = resample_uniform(wl, vl)
        sc.spec_offset_sz.append((sum(len(v) for v in spec_vals), len(u)))
        spec_vals.append(u)
    if not sc.spec_offset_sz:                                                 # "if no spectra are loaded add uniform 1.0 spectrum" (:406-418)
        u = resample_uniform([200.0, 400.0, 600.0, 800.0], [1.0, 1.0, 1.0, 1.0])
        sc.spec_offset_sz.append((0, len(u))); spec_vals.append(u); sc.spec_tex_offset_sz.append((UINT_MAX, 0))
    sc.spec_values = np.concatenate(spec_vals).astype(np.float32) if spec_vals else np.zeros(0, np.float32)

    def spectrum_id(node):
        """GetSpectrumIdFromNode (integrator_pt_scene_mat.cpp:109-119)."""
        sn = node.find("spectrum") if node is not None else None
        return int(sn.get("id")) & UINT_MAX if sn is not None else UINT_MAX

    def read_sampler(node, from_spectrum=False):
        """ReadSamplerFromColorNode (integrator_pt_scene_mat.cpp:32-91): (key, row0, row1, disable_gamma) or None without a <texture> (a <spectrum>)."""
        tnode = node.find("spectrum" if from_spectrum else "texture") if node is not None else None
        if tnode is None:
            return None
        def addr(name, default):
            v = tnode.get(name)
            if v is None:
                return default
            if v in ("mirror", "border", "mirror_once"):
                raise NotImplementedError(f"texture addressing mode '{v}' is outside the path (wrap and clamp are in)")
            return addr_modes.get(v, ADDR_WRAP)
        au, av = addr("addressing_mode_u", ADDR_WRAP), addr("addressing_mode_v", ADDR_WRAP)
        aw = addr("addressing_mode_w", av)
        filt = FILTER_LINEAR
        if tnode.get("filter") in ("point", "nearest"):
            filt = FILTER_NEAREST
        elif tnode.get("filter") in ("cubic", "bicubic"):
            raise NotImplementedError("bicubic texture filtering is outside the path")
        row0, row1 = [1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]
        mvals = _f(tnode.get("matrix", ""))                                    # rows of the texture matrix; a short list leaves the rest
        for i, v in enumerate(mvals[:8]):
            (row0 if i < 4 else row1)[i % 4] = v
        gamma = tnode.get("input_gamma")
        disable_gamma = gamma is not None and int(float(gamma)) == 1           # attribute.as_int() == 1 (integrator_pt_scene_tex.cpp:118-120)
        return (int(tnode.get("id")), au, av, aw, filt), row0, row1, disable_gamma

    def load_texture_from_node(node):
        """LoadTextureFromNode (integrator_pt_scene_tex.cpp:105-126): (row0, row1, index into the texture table)."""
        sam = read_sampler(node)
        if sam is None:
            return (1, 0, 0, 0), (0, 1, 0, 0), 0
        key, row0, row1, disable_gamma = sam
        return tuple(row0), tuple(row1), load_texture_by_key(key, disable_gamma)

    loaded_spectral = set()

    def load_spectral_textures(spec_id, color_node):
        """LoadSpectralTextures (integrator_pt_scene_mat.cpp:144-173): the textures of a spectrum's bands enter the table with the sampler read
        from the colour node's <spectrum> child (gamma never applied, LoadTextureById :129-144); the bands then name table entries."""
        if spec_id == UINT_MAX or spec_id >= len(sc.spec_tex_offset_sz) or sc.spec_tex_offset_sz[spec_id][1] == 0 or spec_id in loaded_spectral:
            return
        sam = read_sampler(color_node, True)
        off, n = sc.spec_tex_offset_sz[spec_id]
        for k in range(n):
            xml_id = sc.spec_tex_ids_wavelengths[off + k][0]
            key = (xml_id, *sam[0][1:])
            sc.spec_tex_ids_wavelengths[off + k][0] = load_texture_by_key(key, True)
        loaded_spectral.add(spec_id)

    def load_texture_by_key(key, disable_gamma):
        if key not in tex_cache:
            path, w, h, bpp = tex_info[key[0]]
            raw = open(path, "rb").read()
            if ".exr" in path:                                                 # LoadImage4fFromEXR / LoadImage1fFromEXR (imageutils.cpp:317-392): tinyexr's rows, flipped
                img = decode_exr(raw)[::-1]
                if bpp == 16:
                    tex = Texture(np.ascontiguousarray(img), TEX_RGBA32F, False, key[1], key[2], key[4])
                else:                                                          # one float per texel: R, infinities and values beyond the half range clamped
                    r = np.where(np.isinf(img[..., 0]), np.float32(65504.0), np.clip(img[..., 0], 0.0, 65504.0)).astype(np.float32)
                    tex = Texture(np.ascontiguousarray(r), TEX_R32F, False, key[1], key[2], key[4])
                tex_cache[key] = sc.add_texture(tex)
                return tex_cache[key]
            if ".image" not in path:                                           # LDR files through LiteImage::LoadImage<uint32_t> (:24-33)
                tex = Texture(decode_ldr_image(path, raw), TEX_RGBA8, not disable_gamma, key[1], key[2], key[4])
                tex_cache[key] = sc.add_texture(tex)
                return tex_cache[key]
            fw, fh = struct.unpack_from("<II", raw, 0)
            if fw == 0 or fh == 0:                                             # white float dummy (:67-73)
                tex = Texture(np.ones((1, 1, 4), np.float32), TEX_RGBA32F, False, key[1], key[2], key[4])
            elif bpp == 16:
                tex = Texture(np.frombuffer(raw, "<f4", fw * fh * 4, 8).reshape(fh, fw, 4).copy(), TEX_RGBA32F, False, key[1], key[2], key[4])
            else:
                tex = Texture(np.frombuffer(raw, "<u4", fw * fh, 8).reshape(fh, fw).copy(), TEX_RGBA8, not disable_gamma, key[1], key[2], key[4])
            tex_cache[key] = sc.add_texture(tex)
        return tex_cache[key]

    def color4(node):
        """GetColorFromNode (integrator_pt_scene_mat.cpp:124-143): one value splats, three get w = 0, four are taken as they are."""
        v = _f(node.get("val")) if node is not None and node.get("val") is not None else []
        if len(v) == 1:
            return np.array([v[0]] * 4, np.float32)
        if len(v) == 3:
            return np.array([*v, 0.0], np.float32)
        if len(v) == 4:
            return np.array(v, np.float32)
        return np.zeros(4, np.float32)

    def val1f(node, default=0.0):
        """hydra_xml::readval1f (hydraxml.cpp:390-402)."""
        return default if node is None else np.float32(float(node.get("val")) if node.get("val") is not None else float(node.text or 0.0))

    def length(v):          # LiteMath length in float
        v = np.asarray(v, np.float32)
        return np.sqrt(np.float32(np.dot(v, v)), dtype=np.float32)

    # <sensor><response> of the camera (integrator_pt_scene.cpp:688-711): response type, up to three response spectra, m_camRespoceRGB
    resp = cam.find("sensor/response")
    if resp is not None:
        sc.cam_response_type = 0 if resp.get("type", "") in ("xyz", "XYZ") else 1
        ids = [int(sn.get("id")) for sn in resp.findall("spectrum")][:3]
        sc.cam_response_spectrum_id = tuple(ids + [-1] * (3 - len(ids)))
        rgb = color4(resp.find("color"))
        sc.cam_respoce_rgb = (float(rgb[0]), float(rgb[1]), float(rgb[2]), 1.0)

    # lights first: materials with light_id copy intensity from them (integrator_pt_scene.cpp:973-996, 575-599)
    scene_node = root.find("scenes/scene")
    light_nodes = {int(l.get("id")): l for l in root.findall("lights_lib/light")}
    old_to_new = []
    for linst in scene_node.findall("instance_light"):
        lnode = light_nodes[int(linst.get("light_id"))]
        m = np.asarray(_f(linst.get("matrix"))).reshape(4, 4)
        ltype, shape, dist = lnode.get("type"), lnode.get("shape"), lnode.get("distribution")
        inten = lnode.find("intensity")
        color = list(color4(inten.find("color")))                             # GetColorFromNode: one value splats over all four components
        mult_node = inten.find("multiplier")
        power = float(mult_node.get("val")) if mult_node is not None else 1.0
        if ltype == "sky":                  # LIGHT_GEOM_ENV (integrator_pt_scene_lgt.cpp:36-59, integrator_pt_scene.cpp:441-486)
            cnode = inten.find("color")
            env_rows, env_tex, back_tex = ((1, 0, 0, 0), (0, 1, 0, 0)), UINT_MAX, UINT_MAX
            env_sample = None
            if cnode.find("texture") is not None:
                r0, r1, env_tex = load_texture_from_node(cnode)
                env_rows = (r0, r1)
                # "m_textureLoadInfo[lightSource.texId]": the reference indexes the XML table with the loaded-texture index (:460-461)
                info = tex_info[env_tex] if env_tex < len(tex_info) else None
                env_sample = info is not None and (".exr" in info[0] or info[3] > 4)
            if lnode.find("back") is not None:
                back_tex = load_texture_from_node(lnode.find("back"))[2]
            lid = sc.set_environment(color4(cnode), env_tex, power, env_rows[0], env_rows[1], back_tex, env_sample)
            sc.env_spec_id, sc.env_spec_mult = spectrum_id(cnode), power       # m_envSpecId = lightSource.specId, m_envSpecMult = lightSource.mult
            old_to_new.append(lid)          # a plain-colour or LDR environment is not a light to sample
            continue
        size = lnode.find("size")
        if ltype == "directional":
            lt = light_directional(m, color, power)
        elif shape in ("rect", "disk"):
            hw, hl = float(size.get("half_width", 0)), float(size.get("half_length", 0))
            lt = light_rect(m, hl, hw, color, power, float(size.get("radius")) if shape == "disk" else None)
        elif shape == "sphere":
            lt = light_sphere(m, float(size.get("radius")), color, power)
        else:
            lt = light_point(m, color, power, dist)
            if dist == "spot":                                        # integrator_pt_scene_lgt.cpp:125-160
                half_rad = np.float32(0.5) * np.float32(0.017453292519943295769)
                lt["lightCos2"] = np.cos(half_rad * val1f(lnode.find("falloff_angle")), dtype=np.float32)
                lt["lightCos1"] = np.cos(half_rad * val1f(lnode.find("falloff_angle2")), dtype=np.float32)
                proj = lnode.find("projective")
                if proj is not None:
                    set_projective(lt, m, float(val1f(proj.find("fov"))), float(val1f(proj.find("nearClipPlane"))), float(val1f(proj.find("farClipPlane"))),
                                   load_texture_from_node(proj)[2] if proj.find("texture") is not None else UINT_MAX)
        lt["specId"] = spectrum_id(inten.find("color"))                      # LoadLightSourceFromNode (integrator_pt_scene_lgt.cpp:22-25)
        if len(_f(inten.find("color").get("val"))) != 3:
            lt["intensity"] = color                                           # float4(v) / float4 as given (three values keep w = 0)
        ies = lnode.find("ies")
        if ies is not None:
            img = ies_spherical_texture(os.path.join(folder, ies.get("loc")))
            lt["iesId"] = sc.add_texture(Texture(img, TEX_R32F, False, ADDR_CLAMP, ADDR_CLAMP, FILTER_LINEAR))
            if ies.get("matrix") is not None:
                mnode = np.asarray(_f(ies.get("matrix"))).reshape(4, 4)
                inst = m.copy(); inst[:, 3] = (0, 0, 0, 1)
                im = rotate_y(90.0) @ (mnode.T @ inst).T        # mrot*transpose(transpose(matrixFromNode)*instMatrix)
                im[:, 3] = (0, 0, 0, 1)
                lt["iesMatrix"] = colmajor(im)
        old_to_new.append(len(sc.lights))
        sc.lights.append(lt)

    def zero_material():
        """Material mat = {} of the typed loaders; spectra ids stay 'none' (RGB mode)."""
        m = np.zeros((), dtype=MATERIAL_DTYPE)
        m["spdid"] = UINT_MAX
        return m

    def bind_texture(mat, slot, node):
        r0, r1, tid = load_texture_from_node(node)
        mat["row0"][slot] = r0; mat["row1"][slot] = r1; mat["texid"][slot] = tid
        return tid

    def apply_normal_map(mat, mnode):
        """LoadSceneMaterials (integrator_pt_scene.cpp:603-638): texid[1] = none, or the <displacement type="normal_bump"> map."""
        mat["texid"][1] = UINT_MAX
        disp = mnode.find("displacement")
        if disp is None or disp.get("type") != "normal_bump":                 # other bump types: message only in the reference
            return
        nm = disp.find("normal_map")
        bind_texture(mat, 1, nm)
        inv = nm.find("invert") if nm is not None else None
        flag = lambda name: inv is not None and int(float(inv.get(name, "0"))) == 1
        cf = int(mat["cflags"])
        if flag("x"):
            cf |= FLAG_NMAP_INVERT_X
        if flag("y"):
            cf |= FLAG_NMAP_INVERT_Y
        if flag("swap_xy"):
            cf |= FLAG_NMAP_SWAP_XY
        mat["cflags"] = cf

    def convert_gltf(mnode):
        """ConvertGLTFMaterial (integrator_pt_scene_mat.cpp:176-278)."""
        mat = zero_material()
        mat["mtype"] = MAT_TYPE_GLTF
        cflags = GLTF_COMPONENT_LAMBERT | GLTF_COMPONENT_COAT
        mat["data"][GLTF_FLOAT_REFL_COAT] = 1.0
        for i in range(4):
            mat["row0"][i] = (1, 0, 0, 0); mat["row1"][i] = (0, 1, 0, 0)
        ior, gloss, metal = np.float32(1.5), np.float32(1.0), np.float32(0.0)
        base = np.ones(4, np.float32)
        cn = mnode.find("color")
        if cn is not None:
            base = color4(cn)
            if cn.find("texture") is not None:
                bind_texture(mat, 0, cn)
        gn, rn = mnode.find("glossiness"), mnode.find("roughness")
        if gn is not None or rn is not None:
            n = gn if gn is not None else rn
            gloss = val1f(n)
            if gn is None:
                cflags |= FLAG_INVERT_GLOSINESS
            if n.find("texture") is not None:
                bind_texture(mat, 2, n); cflags |= FLAG_FOUR_TEXTURES
        mn = mnode.find("metalness")
        if mn is not None:
            metal = val1f(mn)
            if mn.find("texture") is not None:
                bind_texture(mat, 3, mn); cflags |= FLAG_FOUR_TEXTURES
        if mnode.find("fresnel_ior") is not None:
            ior = val1f(mnode.find("fresnel_ior"))
        if mnode.find("coat") is not None:
            mat["data"][GLTF_FLOAT_REFL_COAT] = val1f(mnode.find("coat"))
        pn = mnode.find("glossiness_metalness_coat")
        if pn is not None:
            metal = gloss = val1f(pn)
            mat["data"][GLTF_FLOAT_REFL_COAT] = gloss
            if pn.find("texture") is not None:
                bind_texture(mat, 2, pn); cflags |= FLAG_FOUR_TEXTURES | FLAG_PACK_FOUR_PARAMS_IN_TEXTURE
        mat["cflags"] = cflags
        mat["colors"][GLTF_COLOR_METAL] = 1.0
        mat["data"][GLTF_FLOAT_ALPHA] = metal
        mat["data"][GLTF_FLOAT_GLOSINESS] = gloss
        set_mi_plastic(mat, float(ior), 1.0, base, (1.0, 1.0, 1.0, 1.0))
        return mat

    def attr_float(node, default=0.0):      # xml_attribute::as_float of a missing node / attribute is 0
        return np.float32(float(node.get("val"))) if node is not None and node.get("val") is not None else np.float32(default)

    def load_rough_conductor(mnode):
        """LoadRoughConductorMaterial (integrator_pt_scene_mat.cpp:452-513), RGB mode."""
        mat = zero_material()
        mat["colors"][0] = 1.0
        mat["mtype"] = MAT_TYPE_CONDUCTOR
        mat["lightId"] = UINT_MAX
        an = mnode.find("alpha")
        if an is not None:
            au = av = attr_float(an)
            if bind_texture(mat, 0, an) != 0:
                au = av = np.float32(1.0)
        else:
            au, av = attr_float(mnode.find("alpha_u")), attr_float(mnode.find("alpha_v"))
        mat["data"][0], mat["data"][1] = au, av                                # CONDUCTOR_ROUGH_U, CONDUCTOR_ROUGH_V
        mat["data"][2], mat["data"][3] = attr_float(mnode.find("eta")), attr_float(mnode.find("k"))
        mat["spdid"][0], mat["spdid"][1] = spectrum_id(mnode.find("eta")), spectrum_id(mnode.find("k"))      # (:493-497)
        if mnode.find("reflectance") is not None and not spectral:           # (the reflectance colour is only read in RGB mode, :507-510)
            mat["colors"][0] = color4(mnode.find("reflectance"))
        return mat

    def load_diffuse(mnode):
        """LoadDiffuseMaterial (integrator_pt_scene_mat.cpp:516-571), RGB mode."""
        mat = zero_material()
        mat["colors"][0] = 1.0
        mat["mtype"] = MAT_TYPE_DIFFUSE
        mat["lightId"] = UINT_MAX
        bsdf = mnode.find("bsdf")
        if bsdf is not None and bsdf.get("type") == "oren-nayar":
            mat["cflags"] = GLTF_COMPONENT_ORENNAYAR
            if mnode.find("roughness") is not None:
                mat["data"][0] = val1f(mnode.find("roughness"))
        rn = mnode.find("reflectance")
        if rn is not None:
            mat["colors"][0] = color4(rn)
            bind_texture(mat, 0, rn)
            mat["spdid"][0] = spectrum_id(rn)                                 # (:559-560)
            if spectral:
                load_spectral_textures(int(mat["spdid"][0]), rn)              # (:562-567)
        return mat

    def load_dielectric(mnode):
        """LoadDielectricMaterial (integrator_pt_scene_mat.cpp:574-616), RGB mode."""
        mat = zero_material()
        mat["colors"][0] = 1.0; mat["colors"][1] = 1.0
        mat["mtype"] = MAT_TYPE_DIELECTRIC
        mat["lightId"] = UINT_MAX
        mat["data"][0], mat["data"][1] = 1.00028, 1.5046                       # DIELECTRIC_ETA_EXT (air), DIELECTRIC_ETA_INT (bk7)
        if mnode.find("int_ior") is not None:
            mat["data"][1] = attr_float(mnode.find("int_ior"))
            mat["spdid"][0] = spectrum_id(mnode.find("int_ior"))               # dispersion: an IOR spectrum (:590-595)
        if mnode.find("ext_ior") is not None:
            mat["data"][0] = attr_float(mnode.find("ext_ior"))
        if mnode.find("reflectance") is not None:
            mat["colors"][0] = color4(mnode.find("reflectance"))
        if mnode.find("transmittance") is not None:
            mat["colors"][1] = color4(mnode.find("transmittance"))
        return mat

    def load_blend(mnode):
        """LoadBlendMaterial (integrator_pt_scene_mat.cpp:619-647)."""
        mat = zero_material()
        mat["mtype"] = MAT_TYPE_BLEND
        mat["data"][0] = 1.0
        for k, name in enumerate(("bsdf_1", "bsdf_2")):
            n = mnode.find(name)
            mat["datai"][k] = int(n.get("id", 0)) if n is not None else 0
        wn = mnode.find("weight")
        if wn is not None:
            mat["data"][0] = val1f(wn)
            bind_texture(mat, 0, wn)
        return mat

    def load_plastic(mnode):
        """LoadPlasticMaterial (integrator_pt_scene_mat.cpp:675-757), RGB mode."""
        rn = mnode.find("reflectance")
        color = color4(rn) if rn is not None else np.zeros(4, np.float32)
        r0, r1, tid = load_texture_from_node(rn) if rn is not None else ((0, 0, 0, 0), (0, 0, 0, 0), 0)
        nl = mnode.find("nonlinear")
        nonlinear = int(float(nl.get("val") if nl.get("val") is not None else (nl.text or 0))) if nl is not None else 0
        mat = sc.material_plastic(color, float(val1f(mnode.find("alpha"), 0.1)), float(val1f(mnode.find("int_ior"), 1.49)),
                                  float(val1f(mnode.find("ext_ior"), 1.000277)), nonlinear, tid, r0, r1)
        mat["spdid"][0] = spectrum_id(rn)                                     # (:704-705)
        if spectral:
            load_spectral_textures(int(mat["spdid"][0]), rn)                  # (:708-713)
            # mi::fresnel_coat_precompute in spectral mode (mi_materials.cpp:383-404): the specular reflectance (1, 1, 1, 1) averages to 1 over FOUR
            # components, the diffuse mean is the mean of the reflectance spectrum (trapezoid rule over 360 .. 830 nm), or 0.5 without one
            sid = int(mat["spdid"][0])
            d_mean = np.float32(0.5)
            if sid != UINT_MAX and sc.spec_offset_sz[sid][0] != UINT_MAX:
                off, sz = sc.spec_offset_sz[sid]
                d_mean = spectrum_mean(sc.spec_values[off:off + sz])
            mat["data"][2] = np.float32(1.0) / (d_mean + np.float32(1.0))      # PLASTIC_SPEC_SAMPLE_WEIGHT = s_mean / (d_mean + s_mean)
        return mat

    def load_thin_film(mnode):
        """LoadThinFilmMaterial (integrator_pt_scene_mat.cpp:1020-1193)."""
        def layer(n):
            en, kn = n.find("eta"), n.find("k")
            d = {"eta": float(val1f(en, 0.0)), "k": float(val1f(kn, 0.0)), "eta_spec": spectrum_id(en), "k_spec": spectrum_id(kn)}
            if n.find("thickness") is not None:
                d["thickness"] = float(val1f(n.find("thickness"), 0.0))
            return d
        ln = mnode.find("layers")
        layers = [layer(c) for c in list(ln)] if ln is not None else []
        substrate = layer(mnode) if mnode.find("eta") is not None else None
        an = mnode.find("alpha")
        alpha_tex = None
        if an is not None:
            alpha = float(val1f(an, 0.0))
            r0, r1, tid = load_texture_from_node(an)
            alpha_tex = (tid, r0, r1)
        else:
            alpha = (float(val1f(mnode.find("alpha_u"), 0.0)), float(val1f(mnode.find("alpha_v"), 0.0)))
        tn = mnode.find("thickness_map")
        tmap = None
        if tn is not None:
            r0, r1, tid = load_texture_from_node(tn)
            tmap = (float(tn.get("min", 0.0)), float(tn.get("max", 0.0)), tid, r0, r1)
        en = mnode.find("ext_ior")
        trn = mnode.find("transparent")
        return sc.material_thin_film(layers, substrate, alpha, float(val1f(en, 1.00028)) if en is not None else 1.00028,
                                     int(float(trn.get("val", 0))) if trn is not None else 0, tmap, alpha_tex)

    typed_loaders = {"thin_film": load_thin_film, "plastic": load_plastic, "gltf": convert_gltf, "rough_conductor": load_rough_conductor, "diffuse": load_diffuse,
                     "dielectric": load_dielectric, "blend": load_blend}

    # ConvertOldHydraMaterial (integrator_pt_scene_mat.cpp:280-450), every branch: emission, diffuse (+ Oren-Nayar), reflectivity with and
    # without Fresnel (coated plastic / Lambert + metal mix / pure metal), transparency (legacy glass)
    for mnode in root.findall("materials_lib/material"):
        mtype_attr = mnode.get("type", "")
        if mtype_attr in typed_loaders:                                       # LoadSceneMaterials dispatch (integrator_pt_scene.cpp:500-570)
            mat = typed_loaders[mtype_attr](mnode)
            for k in range(4):
                if not np.any(mat["row0"][k]) and not np.any(mat["row1"][k]):
                    mat["row0"][k] = (1, 0, 0, 0); mat["row1"][k] = (0, 1, 0, 0)
            apply_normal_map(mat, mnode)
            lid = int(mnode.get("light_id", -1))
            if 0 <= lid < len(sc.lights):
                mat["colors"][EMISSION_COLOR] = sc.lights[lid]["intensity"]
                mat["data"][EMISSION_MULT] = sc.lights[lid]["mult"]
                mat["spdid"][0] = sc.lights[lid]["specId"]
                sc.lights[lid]["matId"] = len(sc.materials)
            sc.materials.append(mat)
            continue
        if mtype_attr != "hydra_material":
            raise NotImplementedError(f"material type '{mtype_attr}' is outside the path (thin_film: SURVEY.md 2a)")
        mat = _blank_material()
        mat["texid"] = (0, 0, 0, 0)                                           # Material mat = {}: no 0xFFFFFFFF sentinels in this converter
        mat["spdid"] = (0, 0, 0, 0)
        mat["row0"][:] = 0.0; mat["row1"][:] = 0.0
        mat["mtype"] = MAT_TYPE_GLTF
        mat["data"][GLTF_FLOAT_ALPHA] = 0.0
        mat["data"][GLTF_FLOAT_REFL_COAT] = 1.0
        mat["colors"][GLTF_COLOR_COAT] = 1.0
        mat["colors"][GLTF_COLOR_METAL] = 0.0
        mat["lightId"] = UINT_MAX
        emis = mnode.find("emission")
        color = np.zeros(4, np.float32)
        is_emission = False
        if mnode.get("light_id") is not None or emis is not None:
            cnode = emis.find("color") if emis is not None else None
            color = color4(cnode)
            is_emission = mnode.get("light_id") is not None or length(color) > 1e-5
            bind_texture(mat, 0, cnode)                                       # rows of the node's sampler (default rows without a texture)
            mat["colors"][EMISSION_COLOR] = color
            mat["lightId"] = int(mnode.get("light_id")) & UINT_MAX if mnode.get("light_id") is not None else UINT_MAX
            mat["spdid"][0] = spectrum_id(cnode)                               # GetSpectrumIdFromNode(nodeEmissColor) (:319-320)
            mat["mtype"] = MAT_TYPE_LIGHT_SOURCE
            mult = cnode.find("multiplier") if cnode is not None else None
            mat["data"][EMISSION_MULT] = val1f(mult) if mult is not None else 1.0
        diff = mnode.find("diffuse")
        dnode = diff.find("color") if diff is not None else None
        if dnode is not None:
            color = color4(dnode)
            if dnode.find("texture") is not None:
                bind_texture(mat, 0, dnode)
        refl_color, refl_gloss, fresnel_ior = np.zeros(4, np.float32), np.float32(1.0), np.float32(1.5)
        refl = mnode.find("reflectivity")
        if refl is not None:
            refl_color = color4(refl.find("color"))
            refl_gloss = val1f(refl.find("glossiness"))
            fresnel_ior = val1f(refl.find("fresnel_ior"))
        transp_color, transp_gloss = np.zeros(4, np.float32), np.float32(1.0)
        transp = mnode.find("transparency")
        if transp is not None:
            transp_color = color4(transp.find("color"))
            transp_gloss = val1f(transp.find("glossiness"))
        fres = refl.find("fresnel") if refl is not None else None
        has_fresnel = fres is not None and int(float(fres.get("val", "0"))) != 0
        if not has_fresnel:
            fresnel_ior = np.float32(0.0)
        if (length(refl_color) > 1e-5 and length(color[:3]) > 1e-5) or has_fresnel:
            mat["mtype"] = MAT_TYPE_GLTF
            mat["lightId"] = UINT_MAX
            mat["colors"][GLTF_COLOR_BASE] = color
            mat["colors"][GLTF_COLOR_COAT] = refl_color
            if has_fresnel:
                mat["data"][GLTF_FLOAT_ALPHA] = 0.0
                mat["data"][GLTF_FLOAT_REFL_COAT] = 1.0
                mat["colors"][GLTF_COLOR_METAL] = 0.0
                mat["cflags"] = GLTF_COMPONENT_LAMBERT | GLTF_COMPONENT_COAT
                set_mi_plastic(mat, float(fresnel_ior), 1.0, color, refl_color)
            else:
                mat["data"][GLTF_FLOAT_ALPHA] = length(refl_color) / (length(refl_color) + length(color[:3]))
                mat["data"][GLTF_FLOAT_REFL_COAT] = 0.0
                mat["colors"][GLTF_COLOR_COAT] = 0.0
                mat["colors"][GLTF_COLOR_METAL] = refl_color
                mat["cflags"] = GLTF_COMPONENT_LAMBERT | GLTF_COMPONENT_METAL
        elif length(refl_color) > 1e-5:
            mat["mtype"] = MAT_TYPE_GLTF
            mat["cflags"] = GLTF_COMPONENT_METAL
            mat["colors"][GLTF_COLOR_BASE] = refl_color
            mat["colors"][GLTF_COLOR_METAL] = 1.0
            mat["colors"][GLTF_COLOR_COAT] = 0.0
            mat["data"][GLTF_FLOAT_ALPHA] = 1.0
        elif length(color[:3]) > 1e-5:
            mat["mtype"] = MAT_TYPE_GLTF
            mat["cflags"] = GLTF_COMPONENT_LAMBERT
            mat["colors"][GLTF_COLOR_BASE] = color
            mat["colors"][GLTF_COLOR_COAT] = 0.0
            mat["colors"][GLTF_COLOR_METAL] = 0.0
            mat["data"][GLTF_FLOAT_ALPHA] = 0.0
            mat["data"][GLTF_FLOAT_REFL_COAT] = 0.0
        if length(transp_color) > 1e-5:                                       # legacy glass
            mat["mtype"] = MAT_TYPE_GLASS
            mat["colors"][0] = refl_color                                     # GLTF_COLOR_BASE == GLASS_COLOR_REFLECT
            mat["colors"][1] = transp_color
            mat["data"][0] = refl_gloss
            mat["data"][1] = transp_gloss
            mat["data"][2] = fresnel_ior
        if is_emission:
            mat["mtype"] = MAT_TYPE_LIGHT_SOURCE
        rough = diff.find("roughness") if diff is not None else None
        if rough is not None:
            mat["data"][GLTF_FLOAT_ROUGH_ORENNAYAR] = val1f(rough)
            mat["cflags"] = int(mat["cflags"]) | GLTF_COMPONENT_ORENNAYAR
        mat["data"][GLTF_FLOAT_GLOSINESS] = refl_gloss
        mat["data"][GLTF_FLOAT_IOR] = fresnel_ior
        # the record the kernels read: texture slots a material does not use still need sane rows / ids (integrator_pt_scene.cpp:600-608)
        for k in range(4):
            if not np.any(mat["row0"][k]) and not np.any(mat["row1"][k]):
                mat["row0"][k] = (1, 0, 0, 0); mat["row1"][k] = (0, 1, 0, 0)
        apply_normal_map(mat, mnode)
        if mat["mtype"] == MAT_TYPE_LIGHT_SOURCE:
            lid = int(mnode.get("light_id", -1))
            if 0 <= lid < len(sc.lights):                                     # LoadScene :973-996: the light's intensity, multiplier and spectrum win
                mat["colors"][EMISSION_COLOR] = sc.lights[lid]["intensity"]
                mat["data"][EMISSION_MULT] = sc.lights[lid]["mult"]
                mat["spdid"][0] = sc.lights[lid]["specId"]
                sc.lights[lid]["matId"] = len(sc.materials)
        sc.materials.append(mat)

    for mesh in root.findall("geometry_lib/mesh"):
        pos, norm, tang, uv, idx, mats = load_vsgf(os.path.join(folder, mesh.get("loc")))
        sc.add_mesh(pos, norm, tang, uv, idx, mats)

    # remap lists (hydraxml.h:267-276, LoadSceneRemapLists integrator_pt_scene.cpp:909-924): `size` ints of (from, to) pairs per list
    rl = scene_node.find("remap_lists")
    if rl is not None:
        lists = []
        for node in rl:
            n = int(node.get("size", 0))
            vals = [int(v) for v in node.get("val", "").split()][:n]
            lists.append(vals + [0] * (n - len(vals)))
        sc.set_remap_lists(lists)

    for inst in scene_node.findall("instance"):
        linst = inst.get("linst_id")
        light_id = old_to_new[int(linst)] if linst is not None and int(linst) >= 0 else -1
        mot = inst.find("motion")
        sc.add_instance(int(inst.get("mesh_id")), np.asarray(_f(inst.get("matrix"))).reshape(4, 4),
                        int(inst.get("rmap_id", -1)), light_id,
                        np.asarray(_f(mot.get("matrix"))).reshape(4, 4) if mot is not None else None)
    return sc
