#!/usr/bin/env python3
"""Writes tests/golden/scenes/legacy_materials: a small Hydra scene (XML + VSGF + image4ub, the formats of the shipped fixtures) whose
materials walk every branch of ConvertOldHydraMaterial (integrator_pt_scene_mat.cpp:280-450): diffuse, diffuse + Oren-Nayar roughness,
textured diffuse, reflectivity + diffuse without Fresnel (Lambert/metal mix), reflectivity with Fresnel (coated plastic), reflectivity only
(metal), transparency (legacy glass), emission with a multiplier and no light, and the light-bound emissive material; one sphere moves
during the exposure (motion blur).
Own data, not the reference's: the two loaders (Python, C++) are checked against each other on it and the GPU against the oracle."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hydracore3_amd import synth                                    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "scenes", "legacy_materials")


def write_vsgf(path, pos4, norm4, tang4, uv2, idx, mats):
    nv, ni = pos4.shape[0], idx.size
    body = pos4.astype("<f4").tobytes() + norm4.astype("<f4").tobytes() + tang4.astype("<f4").tobytes() + uv2.astype("<f4").tobytes() + \
        idx.astype("<u4").tobytes() + mats.astype("<u4").tobytes()
    hdr = struct.pack("<QIIII", 24 + len(body), nv, ni, int(len(set(mats.tolist()))), 1)      # flags bit 0: tangents present
    open(path, "wb").write(hdr + body)


def main():
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    rng = np.random.RandomState(7)
    tex = (rng.randint(40, 255, (8, 8, 3))).astype(np.uint32)
    rgba = tex[..., 0] | (tex[..., 1] << 8) | (tex[..., 2] << 16) | np.uint32(0xFF000000)
    open(os.path.join(OUT, "data", "chunk_00000.image4ub"), "wb").write(struct.pack("<II", 8, 8) + rgba.astype("<u4").tobytes())
    sp = synth._sphere_mesh(2)
    pad4 = lambda a: np.concatenate([np.asarray(a, np.float32).reshape(-1, 3), np.zeros((len(a), 1), np.float32)], 1) if np.asarray(a).shape[-1] == 3 else np.asarray(a, np.float32)
    ntri = sp[4].size // 3
    meshes = []
    for i in range(8):                                                # one sphere mesh per material 1..8
        p = os.path.join("data", f"chunk_{i + 1:05d}.vsgf")
        write_vsgf(os.path.join(OUT, p), pad4(sp[0]), pad4(sp[1]), pad4(sp[2]), np.asarray(sp[3], np.float32), np.asarray(sp[4], np.uint32), np.full(ntri, i + 1, np.uint32))
        meshes.append(p)
    q = synth._quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 3.0)
    write_vsgf(os.path.join(OUT, "data", "chunk_00009.vsgf"), pad4(q[0]), pad4(q[1]), pad4(q[2]), np.asarray(q[3], np.float32), np.asarray(q[4], np.uint32), np.zeros(q[4].size // 3, np.uint32))
    lq = synth._quad((-1, 0, -1), (2, 0, 0), (0, 0, 2))
    write_vsgf(os.path.join(OUT, "data", "chunk_00010.vsgf"), pad4(lq[0]), pad4(lq[1]), pad4(lq[2]), np.asarray(lq[3], np.float32), np.asarray(lq[4], np.uint32), np.full(lq[4].size // 3, 9, np.uint32))
    mats = [
        '<material id="0" name="floor" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.8 0.8 0.8"><texture id="0" type="texref" /></color></diffuse></material>',
        '<material id="1" name="lambert" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.2 0.6 0.3" /></diffuse></material>',
        '<material id="2" name="orennayar" type="hydra_material"><diffuse brdf_type="orennayar"><color val="0.6 0.5 0.2" /><roughness val="0.6" /></diffuse></material>',
        '<material id="3" name="mix" type="hydra_material"><diffuse><color val="0.5 0.1 0.1" /></diffuse><reflectivity brdf_type="ggx"><color val="0.6 0.6 0.6" /><glossiness val="0.7" /><fresnel val="0" /><fresnel_ior val="1.5" /></reflectivity></material>',
        '<material id="4" name="plastic" type="hydra_material"><diffuse><color val="0.1 0.2 0.7" /></diffuse><reflectivity brdf_type="ggx"><color val="0.9 0.9 0.9" /><glossiness val="0.85" /><fresnel val="1" /><fresnel_ior val="1.5" /></reflectivity></material>',
        '<material id="5" name="metal" type="hydra_material"><reflectivity brdf_type="ggx"><color val="0.9 0.7 0.3" /><glossiness val="0.6" /><fresnel val="0" /><fresnel_ior val="8" /></reflectivity></material>',
        '<material id="6" name="glass" type="hydra_material"><reflectivity brdf_type="phong"><color val="1 1 1" /><glossiness val="1" /><fresnel val="1" /><fresnel_ior val="1.5" /></reflectivity><transparency><color val="0.9 1.0 0.95" /><glossiness val="1" /><ior val="1.5" /></transparency></material>',
        '<material id="7" name="glow" type="hydra_material"><emission><color val="0.8 0.4 0.1"><multiplier val="2.5" /></color></emission></material>',
        '<material id="8" name="mirror" type="hydra_material"><reflectivity brdf_type="ggx"><color val="0.95 0.95 0.95" /><glossiness val="1" /><fresnel val="0" /><fresnel_ior val="1.5" /></reflectivity></material>',
        '<material id="9" name="light_material" type="hydra_material" light_id="0" visible="1"><emission><color val="20 20 20" /></emission></material>',
    ]
    geo = [f'<mesh id="{i}" name="s{i}" type="vsgf" loc="{m}" />' for i, m in enumerate(meshes)]
    geo += ['<mesh id="8" name="floor" type="vsgf" loc="data/chunk_00009.vsgf" />', '<mesh id="9" name="lightmesh" type="vsgf" loc="data/chunk_00010.vsgf" light_id="0" />']
    inst = []
    for i in range(8):
        x, z = -3.3 + 0.95 * i, -0.6 * (i % 3)
        s = 0.42 + 0.03 * (i % 2)
        # the third sphere slides and rises during the exposure (<motion matrix=..>, hydraxml.h:170-176: motion blur)
        motion = f'<motion matrix="{s} 0 0 {x + 0.45} 0 {s} 0 {0.85 + 0.2 * (i % 2)} 0 0 {s} {z} 0 0 0 1" />' if i == 2 else ""
        inst.append(f'<instance id="{i}" mesh_id="{i}" rmap_id="-1" matrix="{s} 0 0 {x} 0 {s} 0 {0.45 + 0.2 * (i % 2)} 0 0 {s} {z} 0 0 0 1">{motion}</instance>')
    inst.append('<instance id="8" mesh_id="8" rmap_id="-1" matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" />')
    inst.append('<instance id="9" mesh_id="9" rmap_id="-1" matrix="1 0 0 0 0 1 0 3.5 0 0 1 0.5 0 0 0 1" light_id="0" linst_id="0" />')
    xml = f'''<?xml version="1.0"?>
<textures_lib>
  <texture id="0" name="noise" loc="data/chunk_00000.image4ub" width="8" height="8" channels="4" />
</textures_lib>
<materials_lib>
  {chr(10).join("  " + m for m in mats)}
</materials_lib>
<geometry_lib>
  {chr(10).join("  " + g for g in geo)}
</geometry_lib>
<lights_lib>
  <light id="0" name="area" type="area" shape="rect" distribution="diffuse" visible="1" mat_id="9" mesh_id="9">
    <size half_length="1" half_width="1" />
    <intensity><color val="1 1 1" /><multiplier val="20" /></intensity>
  </light>
  <light id="1" name="sky" type="sky" shape="point" distribution="uniform"><intensity><color val="0.1 0.12 0.16" /><multiplier val="1" /></intensity></light>
</lights_lib>
<cam_lib>
  <camera id="0" name="cam" type="uvn"><fov>42</fov><nearClipPlane>0.01</nearClipPlane><farClipPlane>100.0</farClipPlane><up>0 1 0</up><position>0 2.2 7.5</position><look_at>0 0.6 0</look_at></camera>
</cam_lib>
<render_lib>
  <render_settings type="HydraModern" id="0"><width>96</width><height>64</height><trace_depth>6</trace_depth><maxRaysPerPixel>4</maxRaysPerPixel></render_settings>
</render_lib>
<scenes>
  <scene id="0" name="legacy materials">
    <instance_light id="0" light_id="0" matrix="1 0 0 0 0 1 0 3.5 0 0 1 0.5 0 0 0 1" lgroup_id="-1" />
    <instance_light id="1" light_id="1" matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" lgroup_id="-1" />
    {chr(10).join("    " + i for i in inst)}
  </scene>
</scenes>
'''
    open(os.path.join(OUT, "statex_00001.xml"), "w").write(xml)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
