#!/usr/bin/env python3
"""Writes tests/golden/scenes/typed_materials: a small Hydra scene (XML + VSGF + image4ub / image4f) whose materials use the typed nodes of
LoadSceneMaterials (integrator_pt_scene.cpp:500-570) - gltf (ConvertGLTFMaterial, with colour / glossiness / metalness textures and the
packed glossiness_metalness_coat form), rough_conductor (alpha and alpha_u / alpha_v), diffuse (Lambert, Oren-Nayar, textured),
dielectric, plastic (LoadPlasticMaterial: its transmittance table in m_arrays1f), blend (constant and texture-masked weight, nested) - the sampler attributes of ReadSamplerFromColorNode (addressing modes,
point filter, texture matrix, input_gamma) a remap list, a spot light with falloff angles and a projected texture (<projective>), and normal-map bump (<displacement type="normal_bump">, with and without the invert / swap flags, also on a blend leaf). Own data, not the reference's: the two loaders (Python, C++) are checked
against each other on it and the GPU against the oracle."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hydracore3_amd import synth                                    # noqa: E402
from make_legacy_scene import write_vsgf                            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "scenes", "typed_materials")


def main():
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    rng = np.random.RandomState(11)
    # texture 0: the white 1x1 the exporter always writes first; 1: colour noise (LDR); 2: checker mask (LDR); 3: scalar parameters (HDR, image4f)
    open(os.path.join(OUT, "data", "chunk_00000.image4ub"), "wb").write(struct.pack("<II", 1, 1) + np.array([0xFFFFFFFF], "<u4").tobytes())
    tex = rng.randint(40, 255, (8, 8, 3)).astype(np.uint32)
    rgba = tex[..., 0] | (tex[..., 1] << 8) | (tex[..., 2] << 16) | np.uint32(0xFF000000)
    open(os.path.join(OUT, "data", "chunk_00001.image4ub"), "wb").write(struct.pack("<II", 8, 8) + rgba.astype("<u4").tobytes())
    chk = np.zeros((8, 8), np.uint32)
    for y in range(8):
        for x in range(8):
            chk[y, x] = 0xFFFFFFFF if (x + y) % 2 else 0xFF181818
    open(os.path.join(OUT, "data", "chunk_00002.image4ub"), "wb").write(struct.pack("<II", 8, 8) + chk.astype("<u4").tobytes())
    prm = np.zeros((4, 4, 4), np.float32)
    prm[..., 0] = rng.uniform(0.55, 0.95, (4, 4)); prm[..., 1] = rng.uniform(0.0, 0.8, (4, 4)); prm[..., 2] = rng.uniform(0.3, 1.0, (4, 4)); prm[..., 3] = 1.0
    open(os.path.join(OUT, "data", "chunk_00003.image4f"), "wb").write(struct.pack("<II", 4, 4) + prm.astype("<f4").tobytes())

    # texture 4: a tangent-space normal map (LDR, linear): gentle bumps around +z
    ny, nx = np.mgrid[0:16, 0:16]
    dx, dy = 0.35 * np.sin(nx * np.pi / 4.0), 0.35 * np.cos(ny * np.pi / 4.0)
    nz = np.sqrt(np.maximum(1.0 - dx * dx - dy * dy, 0.0))
    enc = lambda v: np.clip(np.rint((v * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint32)
    nrm = enc(dx) | (enc(dy) << 8) | (np.clip(np.rint(nz * 255.0), 0, 255).astype(np.uint32) << 16) | np.uint32(0xFF000000)
    open(os.path.join(OUT, "data", "chunk_00018.image4ub"), "wb").write(struct.pack("<II", 16, 16) + nrm.astype("<u4").tobytes())

    sp = synth._sphere_mesh(2)
    pad4 = lambda a: np.concatenate([np.asarray(a, np.float32).reshape(-1, 3), np.zeros((len(a), 1), np.float32)], 1) if np.asarray(a).shape[-1] == 3 else np.asarray(a, np.float32)
    ntri = sp[4].size // 3
    nsph = 14
    for i in range(nsph):                                             # one sphere mesh per material 1..12
        write_vsgf(os.path.join(OUT, "data", f"chunk_{i + 4:05d}.vsgf"), pad4(sp[0]), pad4(sp[1]), pad4(sp[2]), np.asarray(sp[3], np.float32), np.asarray(sp[4], np.uint32), np.full(ntri, i + 1, np.uint32))
    q = synth._quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 3.0)
    write_vsgf(os.path.join(OUT, "data", "chunk_00026.vsgf"), pad4(q[0]), pad4(q[1]), pad4(q[2]), np.asarray(q[3], np.float32), np.asarray(q[4], np.uint32), np.zeros(q[4].size // 3, np.uint32))
    lq = synth._quad((-1, 0, -1), (2, 0, 0), (0, 0, 2))
    write_vsgf(os.path.join(OUT, "data", "chunk_00027.vsgf"), pad4(lq[0]), pad4(lq[1]), pad4(lq[2]), np.asarray(lq[3], np.float32), np.asarray(lq[4], np.uint32), np.full(lq[4].size // 3, 15, np.uint32))
    ident = "1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1"
    BUMP = '<displacement type="normal_bump"><normal_map><invert x="0" y="0" swap_xy="0" /><texture id="4" type="texref" matrix="2 0 0 0 0 2 0 0 0 0 1 0 0 0 0 1" input_gamma="1" /></normal_map></displacement>'
    BUMP_INV = BUMP.replace('x="0" y="0" swap_xy="0"', 'x="1" y="1" swap_xy="1"').replace('matrix="2 0 0 0 0 2', 'filter="point" matrix="3 0 0 0 0 3')
    mats = [
        f'<material id="0" name="floor" type="diffuse"><bsdf type="lambert" /><reflectance val="0.8 0.8 0.8"><texture id="1" type="texref" matrix="2 0 0 0.25 0 2 0 0.5 0 0 1 0 0 0 0 1" addressing_mode_u="wrap" addressing_mode_v="wrap" input_gamma="2.2" input_alpha="rgb" /></reflectance></material>',
        f'<material id="1" name="gltf_plain" type="gltf"><color val="0.7 0.25 0.2" /><glossiness val="0.7" /><metalness val="0.1" /><fresnel_ior val="1.45" /><coat val="0.8" />{BUMP}</material>',
        f'<material id="2" name="gltf_textured" type="gltf"><color val="0.9 0.9 0.9"><texture id="1" type="texref" matrix="{ident}" addressing_mode_u="clamp" addressing_mode_v="clamp" input_gamma="2.2" /></color><roughness val="0.4" /><metalness val="0.3" /></material>',
        f'<material id="3" name="gltf_four" type="gltf"><color val="0.3 0.5 0.8" /><glossiness val="1.0"><texture id="3" type="texref" matrix="{ident}" input_gamma="1" /></glossiness><metalness val="1.0"><texture id="3" type="texref" matrix="3 0 0 0 0 3 0 0 0 0 1 0 0 0 0 1" filter="point" input_gamma="1" /></metalness></material>',
        f'<material id="4" name="gltf_packed" type="gltf"><color val="0.8 0.7 0.3" /><glossiness_metalness_coat val="1.0"><texture id="3" type="texref" matrix="{ident}" input_gamma="1" /></glossiness_metalness_coat></material>',
        f'<material id="5" name="gold" type="rough_conductor"><bsdf type="ggx" /><alpha val="0.15" /><eta val="0.2" /><k val="3.6" /><reflectance val="1.0 0.8 0.4" />{BUMP}</material>',
        '<material id="6" name="brushed" type="rough_conductor"><bsdf type="ggx" /><alpha_u val="0.05" /><alpha_v val="0.35" /><eta val="1.1" /><k val="6.8" /></material>',
        '<material id="7" name="mirror" type="rough_conductor"><bsdf type="ggx" /><alpha val="0" /><eta val="0.2" /><k val="3.9" /></material>',
        f'<material id="8" name="orennayar" type="diffuse"><bsdf type="oren-nayar" /><roughness val="0.7" /><reflectance val="0.3 0.6 0.3" />{BUMP_INV}</material>',
        f'<material id="9" name="diffuse_point" type="diffuse"><bsdf type="lambert" /><reflectance val="1 1 1"><texture id="2" type="texref" matrix="{ident}" filter="point" input_gamma="1" /></reflectance></material>',
        '<material id="10" name="glass" type="dielectric"><int_ior val="1.5" /><ext_ior val="1.0" /><reflectance val="1 1 1" /><transmittance val="0.95 1 0.95" /></material>',
        f'<material id="11" name="masked" type="blend"><bsdf_1 id="5" /><bsdf_2 id="8" /><weight val="1.0"><texture id="2" type="texref" matrix="2 0 0 0 0 2 0 0 0 0 1 0 0 0 0 1" input_gamma="1" /></weight></material>',
        '<material id="12" name="nested" type="blend"><bsdf_1 id="11" /><bsdf_2 id="1" /><weight val="0.35" /></material>',
        '<material id="13" name="plastic" type="plastic"><reflectance val="0.2 0.45 0.7" /><alpha val="0.12" /><int_ior val="1.49" /><ext_ior val="1.000277" /></material>',
        f'<material id="14" name="plastic_nl" type="plastic"><reflectance val="0.9 0.9 0.9"><texture id="1" type="texref" matrix="{ident}" input_gamma="2.2" /></reflectance><alpha val="0.3" /><int_ior val="1.6" /><ext_ior val="1.0" /><nonlinear val="1" />{BUMP}</material>',
        '<material id="15" name="light_material" type="hydra_material" light_id="0" visible="1"><emission><color val="20 20 20" /></emission></material>',
    ]
    geo = [f'<mesh id="{i}" name="s{i}" type="vsgf" loc="data/chunk_{i + 4:05d}.vsgf" />' for i in range(nsph)]
    geo += [f'<mesh id="{nsph}" name="floor" type="vsgf" loc="data/chunk_00026.vsgf" />', f'<mesh id="{nsph + 1}" name="lightmesh" type="vsgf" loc="data/chunk_00027.vsgf" light_id="0" />']
    inst = []
    for i in range(nsph):
        x, z = -3.9 + 0.6 * i, -0.7 * (i % 3)
        s = 0.3 + 0.02 * (i % 2)
        rm = 0 if i == 0 else -1                                      # the first sphere is recoloured through remap list 0 (1 -> 6)
        inst.append(f'<instance id="{i}" mesh_id="{i}" rmap_id="{rm}" matrix="{s} 0 0 {x} 0 {s} 0 {0.33 + 0.25 * (i % 2)} 0 0 {s} {z} 0 0 0 1" />')
    inst.append(f'<instance id="{nsph}" mesh_id="{nsph}" rmap_id="-1" matrix="{ident}" />')
    inst.append(f'<instance id="{nsph + 1}" mesh_id="{nsph + 1}" rmap_id="-1" matrix="1 0 0 0 0 1 0 3.5 0 0 1 0.5 0 0 0 1" light_id="0" linst_id="0" />')
    xml = f'''<?xml version="1.0"?>
<textures_lib>
  <texture id="0" name="Map#0" loc="data/chunk_00000.image4ub" offset="8" bytesize="4" width="1" height="1" />
  <texture id="1" name="noise" loc="data/chunk_00001.image4ub" offset="8" bytesize="256" width="8" height="8" />
  <texture id="2" name="checker" loc="data/chunk_00002.image4ub" offset="8" bytesize="256" width="8" height="8" />
  <texture id="3" name="params" loc="data/chunk_00003.image4f" offset="8" bytesize="256" width="4" height="4" />
  <texture id="4" name="normals" loc="data/chunk_00018.image4ub" offset="8" bytesize="1024" width="16" height="16" />
</textures_lib>
<materials_lib>
  {chr(10).join("  " + m for m in mats)}
</materials_lib>
<geometry_lib>
  {chr(10).join("  " + g for g in geo)}
</geometry_lib>
<lights_lib>
  <light id="0" name="area" type="area" shape="rect" distribution="diffuse" visible="1" mat_id="15" mesh_id="{nsph + 1}">
    <size half_length="1" half_width="1" />
    <intensity><color val="1 1 1" /><multiplier val="20" /></intensity>
  </light>
  <light id="1" name="sky" type="sky" shape="point" distribution="uniform"><intensity><color val="0.1 0.12 0.16" /><multiplier val="1" /></intensity></light>
  <light id="2" name="projector" type="point" shape="point" distribution="spot" visible="0">
    <size radius="0" /><falloff_angle val="70" /><falloff_angle2 val="50" />
    <intensity><color val="1 0.9 0.8" /><multiplier val="60" /></intensity>
    <projective><fov val="70" /><nearClipPlane val="0.1" /><farClipPlane val="100" /><texture id="2" type="texref" matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" input_gamma="1" /></projective>
  </light>
</lights_lib>
<cam_lib>
  <camera id="0" name="cam" type="uvn"><fov>42</fov><nearClipPlane>0.01</nearClipPlane><farClipPlane>100.0</farClipPlane><up>0 1 0</up><position>0 2.2 7.5</position><look_at>0 0.6 0</look_at></camera>
</cam_lib>
<render_lib>
  <render_settings type="HydraModern" id="0"><width>96</width><height>64</height><trace_depth>6</trace_depth><maxRaysPerPixel>4</maxRaysPerPixel></render_settings>
</render_lib>
<scenes>
  <scene id="0" name="typed materials">
    <remap_lists>
      <remap_list id="0" size="2" val="1 6 " />
    </remap_lists>
    <instance_light id="0" light_id="0" matrix="1 0 0 0 0 1 0 3.5 0 0 1 0.5 0 0 0 1" lgroup_id="-1" />
    <instance_light id="1" light_id="1" matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" lgroup_id="-1" />
    <instance_light id="2" light_id="2" matrix="1 0 0 -2.5 0 0.8 -0.6 3.0 0 0.6 0.8 2.5 0 0 0 1" lgroup_id="-1" />
    {chr(10).join("    " + i for i in inst)}
  </scene>
</scenes>
'''
    open(os.path.join(OUT, "statex_00001.xml"), "w").write(xml)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
