#!/usr/bin/env python3
"""Writes tests/golden/scenes/env_map: a small Hydra scene lit only by a lat-long HDR environment map (image4f, so LoadSceneLights adds
it to the lights with a pdf table - integrator_pt_scene.cpp:441-478), with a texture matrix on the map, a camera back plate
(<back>, integrator_pt_scene_lgt.cpp:51-58), diffuse / glossy / mirror / glass spheres on a floor, seen through a simulated lens
(<optical_system>, integrator_pt_scene.cpp:1078-1141). Own data, not the reference's: the
two loaders (Python, C++) are checked against each other on it and the GPU against the oracle."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hydracore3_amd import synth                                    # noqa: E402
from make_legacy_scene import write_vsgf                            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "scenes", "env_map")


def main():
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    open(os.path.join(OUT, "data", "chunk_00000.image4ub"), "wb").write(struct.pack("<II", 1, 1) + np.array([0xFFFFFFFF], "<u4").tobytes())
    # texture 1: 32 x 16 lat-long sky: blue-ish gradient, warm horizon band, a small very bright sun
    W, H = 32, 16
    v, u = np.mgrid[0:H, 0:W]
    sky = np.zeros((H, W, 4), np.float32)
    up = 1.0 - (v + 0.5) / H
    sky[..., 0] = 0.15 + 0.35 * (1.0 - up); sky[..., 1] = 0.25 + 0.3 * (1.0 - up); sky[..., 2] = 0.5 + 0.3 * up
    sky[7:9, :, :3] += np.array([0.5, 0.3, 0.1], np.float32)
    sky[4, 21, :3] = (180.0, 160.0, 120.0); sky[4, 22, :3] = (90.0, 80.0, 60.0)
    sky[..., 3] = 1.0
    open(os.path.join(OUT, "data", "chunk_00001.image4f"), "wb").write(struct.pack("<II", W, H) + sky.astype("<f4").tobytes())
    # texture 2: the camera back plate (LDR), texture 3: floor albedo (LDR)
    rng = np.random.RandomState(3)
    def ldr(n):
        t = rng.randint(30, 255, (n, n, 3)).astype(np.uint32)
        return t[..., 0] | (t[..., 1] << 8) | (t[..., 2] << 16) | np.uint32(0xFF000000)
    open(os.path.join(OUT, "data", "chunk_00002.image4ub"), "wb").write(struct.pack("<II", 8, 8) + ldr(8).astype("<u4").tobytes())
    open(os.path.join(OUT, "data", "chunk_00003.image4ub"), "wb").write(struct.pack("<II", 4, 4) + ldr(4).astype("<u4").tobytes())

    sp = synth._sphere_mesh(2)
    pad4 = lambda a: np.concatenate([np.asarray(a, np.float32).reshape(-1, 3), np.zeros((len(a), 1), np.float32)], 1) if np.asarray(a).shape[-1] == 3 else np.asarray(a, np.float32)
    ntri = sp[4].size // 3
    nsph = 5
    for i in range(nsph):
        write_vsgf(os.path.join(OUT, "data", f"chunk_{i + 4:05d}.vsgf"), pad4(sp[0]), pad4(sp[1]), pad4(sp[2]), np.asarray(sp[3], np.float32), np.asarray(sp[4], np.uint32), np.full(ntri, i + 1, np.uint32))
    q = synth._quad((-8, 0, 6), (16, 0, 0), (0, 0, -14), 2, 2, 3.0)
    write_vsgf(os.path.join(OUT, "data", "chunk_00009.vsgf"), pad4(q[0]), pad4(q[1]), pad4(q[2]), np.asarray(q[3], np.float32), np.asarray(q[4], np.uint32), np.zeros(q[4].size // 3, np.uint32))
    ident = "1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1"
    mats = [
        f'<material id="0" name="floor" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.7 0.7 0.7"><texture id="3" type="texref" matrix="{ident}" /></color></diffuse></material>',
        '<material id="1" name="lambert" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.7 0.3 0.2" /></diffuse></material>',
        '<material id="2" name="plastic" type="gltf"><color val="0.2 0.5 0.7" /><glossiness val="0.8" /><metalness val="0.0" /><fresnel_ior val="1.5" /></material>',
        '<material id="3" name="mirror" type="rough_conductor"><bsdf type="ggx" /><alpha val="0" /><eta val="0.2" /><k val="3.9" /></material>',
        '<material id="4" name="glass" type="hydra_material"><reflectivity brdf_type="phong"><color val="1 1 1" /><glossiness val="1" /><fresnel val="1" /><fresnel_ior val="1.5" /></reflectivity><transparency><color val="0.95 1.0 0.95" /><glossiness val="1" /><ior val="1.5" /></transparency></material>',
        '<material id="5" name="rough_metal" type="rough_conductor"><bsdf type="ggx" /><alpha val="0.2" /><eta val="1.1" /><k val="6.8" /><reflectance val="1 0.85 0.6" /></material>',
    ]
    geo = [f'<mesh id="{i}" name="s{i}" type="vsgf" loc="data/chunk_{i + 4:05d}.vsgf" />' for i in range(nsph)]
    geo += [f'<mesh id="{nsph}" name="floor" type="vsgf" loc="data/chunk_00009.vsgf" />']
    inst = []
    for i in range(nsph):
        x, z = -2.6 + 1.3 * i, -0.6 * (i % 2)
        inst.append(f'<instance id="{i}" mesh_id="{i}" rmap_id="-1" matrix="0.5 0 0 {x} 0 0.5 0 0.5 0 0 0.5 {z} 0 0 0 1" />')
    inst.append(f'<instance id="{nsph}" mesh_id="{nsph}" rmap_id="-1" matrix="{ident}" />')
    # a double-Gauss 50 mm f/2 prescription (the widely circulated six-element design; millimetres, front element first) behind scale = 0.001:
    # exercises LoadOpticsFromNode's scale, semi_diameter, the stop (radius 0) and order="scene_to_sensor" (integrator_pt_scene.cpp:1078-1141)
    dgauss = [(29.475, 3.76, 1.67, 12.6), (84.83, 0.12, 1.0, 12.6), (19.275, 4.025, 1.67, 11.5), (40.77, 3.275, 1.699, 11.5), (12.75, 5.705, 1.0, 9.0),
              (0.0, 4.5, 0.0, 8.55), (-14.495, 1.18, 1.603, 8.5), (40.77, 6.065, 1.658, 10.0), (-20.385, 0.19, 1.0, 10.0), (437.065, 3.22, 1.717, 10.0),
              (-39.73, 36.9, 1.0, 10.0)]
    LENS = '<optical_system order="scene_to_sensor" scale="0.001" sensor_diagonal="0.035">' + "".join(
        f'<line id="{i}" curvature_radius="{r}" thickness="{t}" ior="{n}" semi_diameter="{a}" />' for i, (r, t, n, a) in enumerate(dgauss)) + '</optical_system>'
    xml = f'''<?xml version="1.0"?>
<textures_lib>
  <texture id="0" name="Map#0" loc="data/chunk_00000.image4ub" offset="8" bytesize="4" width="1" height="1" />
  <texture id="1" name="sky" loc="data/chunk_00001.image4f" offset="8" bytesize="{W * H * 16}" width="{W}" height="{H}" />
  <texture id="2" name="backplate" loc="data/chunk_00002.image4ub" offset="8" bytesize="256" width="8" height="8" />
  <texture id="3" name="floor" loc="data/chunk_00003.image4ub" offset="8" bytesize="64" width="4" height="4" />
</textures_lib>
<materials_lib>
  {chr(10).join("  " + m for m in mats)}
</materials_lib>
<geometry_lib>
  {chr(10).join("  " + g for g in geo)}
</geometry_lib>
<lights_lib>
  <light id="0" name="sky" type="sky" shape="point" distribution="map">
    <intensity>
      <color val="1 1 1"><texture id="1" type="texref" matrix="1 0 0 0.15 0 1 0 0 0 0 1 0 0 0 0 1" addressing_mode_u="wrap" addressing_mode_v="clamp" input_gamma="1" /></color>
      <multiplier val="1.0" />
    </intensity>
    <back><texture id="2" type="texref" matrix="{ident}" addressing_mode_u="clamp" addressing_mode_v="clamp" input_gamma="2.2" /></back>
  </light>
</lights_lib>
<cam_lib>
  <camera id="0" name="cam" type="uvn"><fov>45</fov><nearClipPlane>0.01</nearClipPlane><farClipPlane>100.0</farClipPlane><up>0 1 0</up><position>0 1.6 6.5</position><look_at>0 0.5 0</look_at>
    {LENS}
  </camera>
</cam_lib>
<render_lib>
  <render_settings type="HydraModern" id="0"><width>96</width><height>64</height><trace_depth>5</trace_depth><maxRaysPerPixel>4</maxRaysPerPixel></render_settings>
</render_lib>
<scenes>
  <scene id="0" name="environment map">
    <instance_light id="0" light_id="0" matrix="{ident}" lgroup_id="-1" />
    {chr(10).join("    " + i for i in inst)}
  </scene>
</scenes>
'''
    open(os.path.join(OUT, "statex_00001.xml"), "w").write(xml)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
