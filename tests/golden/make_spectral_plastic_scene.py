"""Writes tests/golden/scenes/spectral_plastic/statex_00001.xml and tests/golden/scenes/spectral_glass/statex_00001.xml: the reference's spectral Cornell fixture (tests/golden/scenes/test_spectral,
= scenes/test_spectral/spectral_cornell_conductor.xml) with its sphere turned into a rough plastic with a reflectance SPECTRUM (nonlinear
colour shift on) and its white walls into a smooth plastic with a plain grey reflectance - the two forms of the `plastic` node the
reference's spectral test list exercises (testing/run_tests.py:480-511: PlasticRough-025_sphere, Spectral-plastic-sphere). Meshes, spectra and
the texture are shared with the test_spectral fixture (paths point into ../test_spectral/data).  Usage: python tests/golden/make_spectral_plastic_scene.py"""
import os
import re

here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "scenes", "test_spectral", "statex_00001.xml"), encoding="utf-8").read()
out = src.replace('loc="data/', 'loc="../test_spectral/data/')


def material(xml, mat_id, body):
    pat = re.compile(r'<material id="%d".*?</material>' % mat_id, re.S)
    assert pat.search(xml), mat_id
    return pat.sub(lambda _m: body, xml, count=1)


out = material(out, 4, '''<material id="4" name="plastic_sphere" type="plastic">
    <reflectance val="1.0">
      <spectrum id="2" type="ref"/>
    </reflectance>
    <alpha val="0.25" />
    <int_ior val="1.49" />
    <ext_ior val="1.000277" />
    <nonlinear val="1" />
  </material>''')
out = material(out, 3, '''<material id="3" name="white_plastic" type="plastic">
    <reflectance val="0.6" />
    <alpha val="0.05" />
    <int_ior val="1.5" />
    <ext_ior val="1.0" />
    <nonlinear val="0" />
  </material>''')
dst = os.path.join(here, "scenes", "spectral_plastic", "statex_00001.xml")
os.makedirs(os.path.dirname(dst), exist_ok=True)
open(dst, "w", encoding="utf-8").write(out)
print("wrote", dst)

# tests/golden/scenes/spectral_glass: the sphere as a smooth dielectric whose interior IOR is a SPECTRUM given inline (ParseSpectrumStr's
# "lambda value ..." form; a flint-like dispersion curve) - the dispersive case: the first wavelength's IOR bends the ray and the path keeps
# that wavelength only (RAY_FLAG_WAVES_DIVERGED) - and the short box as a plain dielectric without a spectrum (testing/run_tests.py:482-483,
# 502-503: Spectral-ior-sphere, Spectral-ior-model)
glass = src.replace('loc="data/', 'loc="../test_spectral/data/')
glass = glass.replace("</spectra_lib>", '  <spectrum id="7" name="flint_ior" value="360 1.70 400 1.68 450 1.66 500 1.645 550 1.635 600 1.628 650 1.623 700 1.619 760 1.615 830 1.612" />\n</spectra_lib>')
glass = material(glass, 4, '''<material id="4" name="dispersive_glass" type="dielectric">
    <int_ior val="1.63">
      <spectrum id="7" type="ref"/>
    </int_ior>
    <ext_ior val="1.00028" />
  </material>''')
dst = os.path.join(here, "scenes", "spectral_glass", "statex_00001.xml")
os.makedirs(os.path.dirname(dst), exist_ok=True)
open(dst, "w", encoding="utf-8").write(glass)
print("wrote", dst)

# tests/golden/scenes/spectral_sky: the reference's fixture with its area light dimmed and a `sky` light whose colour carries a spectrum (D65) and
# a multiplier - m_envSpecId / m_envSpecMult (integrator_pt_scene.cpp:456-457, integrator_pt_lgt.cpp:181-188): rays that leave the open box see it
sky = src.replace('loc="data/', 'loc="../test_spectral/data/')
sky = sky.replace("</spectra_lib>", '  <spectrum id="7" name="d65" loc="../test_spectral/data/spd/cie.stdillum.D6500.spd" />\n</spectra_lib>')
sky = sky.replace("</lights_lib>", '''  <light id="1" name="sky" type="sky" shape="point" distribution="uniform">
    <intensity>
      <color val="1 1 1">
        <spectrum id="7" type="ref"/>
      </color>
      <multiplier val="0.8" />
    </intensity>
  </light>
</lights_lib>''')
assert sky.count('<instance_light id="0"') == 1
sky = re.sub(r'(<instance_light id="0"[^>]*/>)', r'\1\n    <instance_light id="1" light_id="1" matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" lgroup_id="-1" />', sky, count=1)
dst = os.path.join(here, "scenes", "spectral_sky", "statex_00001.xml")
os.makedirs(os.path.dirname(dst), exist_ok=True)
open(dst, "w", encoding="utf-8").write(sky)
print("wrote", dst)

# tests/golden/scenes/spectral_textures: the white walls' and the sphere's reflectance given by TEXTURES (KSPEC_SPD_TEX; LoadSceneSpectrumData's
# lambda_ref_ids, integrator_pt_scene.cpp:363-377; SampleMatColorSpectrumTexture, integrator_spectrum.cpp:128-180): spectrum 7 = five 8 x 8 maps
# at 400 / 480 / 560 / 640 / 720 nm on the diffuse walls (sampler attributes on the <spectrum> node: clamp, point filter), spectrum 8 = two maps
# on the sphere turned plastic. Wavelengths outside a spectrum's bands read zero; the maps' red channel is the value (gamma never applied).
import struct
import numpy as np
tex = src.replace('loc="data/', 'loc="../test_spectral/data/')
folder = os.path.join(here, "scenes", "spectral_textures")
os.makedirs(os.path.join(folder, "data"), exist_ok=True)
v, u = np.mgrid[0:8, 0:8]
lines = []
for k, lam in enumerate((400, 480, 560, 640, 720)):
    val = np.clip(0.25 + 0.55 * np.exp(-((lam - 400 - 40 * (u + v) / 2.0) / 120.0) ** 2) + 0.1 * ((u // 2 + v // 2 + k) % 2), 0.0, 1.0)
    r = (val * 255.0 + 0.5).astype(np.uint32)
    open(os.path.join(folder, "data", f"band_{lam}.image4ub"), "wb").write(struct.pack("<II", 8, 8) + (r | (r << 8) | (r << 16) | np.uint32(0xFF000000)).astype("<u4").tobytes())
    lines.append(f'  <texture id="{k + 1}" name="band_{lam}" loc="data/band_{lam}.image4ub" offset="8" bytesize="256" width="8" height="8" dl="0" />')
tex = tex.replace("</textures_lib>", "\n".join(lines) + "\n</textures_lib>")
tex = tex.replace("</spectra_lib>", '  <spectrum id="7" name="wall_maps" lambda_ref_ids="400 1 480 2 560 3 640 4 720 5" />\n'
                  '  <spectrum id="8" name="sphere_maps" lambda_ref_ids="450 2 650 4" />\n</spectra_lib>')
tex = material(tex, 3, '''<material id="3" name="white" type="diffuse">
    <reflectance val="0.5">
      <spectrum id="7" type="ref" addressing_mode_u="clamp" addressing_mode_v="clamp" filter="point" matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1"/>
      <texture id="0" type="texref" matrix="2 0 0 0 0 2 0 0 0 0 1 0 0 0 0 1" addressing_mode_u="wrap" addressing_mode_v="wrap" />
    </reflectance>
  </material>''')
tex = material(tex, 4, '''<material id="4" name="mapped_plastic" type="plastic">
    <reflectance val="0.5">
      <spectrum id="8" type="ref"/>
    </reflectance>
    <alpha val="0.2" />
    <int_ior val="1.49" />
    <ext_ior val="1.000277" />
    <nonlinear val="0" />
  </material>''')
dst = os.path.join(folder, "statex_00001.xml")
open(dst, "w", encoding="utf-8").write(tex)
print("wrote", dst)
