"""Writes tests/golden/scenes/thin_film/statex_00001.xml and tests/golden/scenes/thin_film_rough/statex_00001.xml: the reference's spectral Cornell
fixture (tests/golden/scenes/test_spectral = scenes/test_spectral/spectral_cornell_conductor.xml) with `thin_film` materials in the forms
LoadThinFilmMaterial (integrator_pt_scene_mat.cpp:1020-1193) distinguishes. The reference ships no film scene (its test list points at an
external scene library, testing/run_tests.py:534), so these are the only fixtures of the material; both render in RGB and in spectral mode.

thin_film:        sphere   = smooth, transparent film (water-like) on glass with a THICKNESS MAP  (RGB: thickness x angle tables; spectral:
                             one film + a map = no table, the Airy summation per vertex)
                  red wall = rough film on gold (eta / k SPECTRA for the substrate, a constant film), opaque
                  green wall = smooth two-film stack on a constant conductor (three layers: multFrFilm tables in both modes)
thin_film_rough:  sphere   = rough, transparent film on glass (rough reflection AND rough transmission), alpha_u / alpha_v given apart
                  red wall = smooth opaque film on a constant conductor, plain thickness (RGB: the angle-only table)
                  green wall = rough film whose roughness comes through a texture (alpha node with a texture: alpha = 1, min with the texel)
Meshes and spectra are shared with the test_spectral fixture (paths point into ../test_spectral/data); the two small textures are written
here.  Usage: python tests/golden/make_film_scene.py"""
import os
import re
import struct

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "scenes", "test_spectral", "statex_00001.xml"), encoding="utf-8").read()
base = src.replace('loc="data/', 'loc="../test_spectral/data/')


def material(xml, mat_id, body):
    pat = re.compile(r'<material id="%d".*?</material>' % mat_id, re.S)
    assert pat.search(xml), mat_id
    return pat.sub(lambda _m: body, xml, count=1)


def textures(xml, folder):
    """texture 1: 8 x 8 thickness ramp with a ripple (red channel is what the film reads); texture 2: 8 x 8 roughness noise"""
    os.makedirs(os.path.join(folder, "data"), exist_ok=True)
    v, u = np.mgrid[0:8, 0:8]
    ramp = np.clip((u + 0.5) / 8.0 * 0.8 + 0.1 + 0.1 * np.sin(v * 1.7), 0.0, 1.0)
    r = (ramp * 255.0 + 0.5).astype(np.uint32)
    open(os.path.join(folder, "data", "thickness.image4ub"), "wb").write(struct.pack("<II", 8, 8) + (r | (r << 8) | (r << 16) | np.uint32(0xFF000000)).astype("<u4").tobytes())
    rng = np.random.RandomState(11)
    a = rng.randint(10, 90, (8, 8)).astype(np.uint32)
    open(os.path.join(folder, "data", "alpha.image4ub"), "wb").write(struct.pack("<II", 8, 8) + (a | (a << 8) | (a << 16) | np.uint32(0xFF000000)).astype("<u4").tobytes())
    return xml.replace("</textures_lib>", '  <texture id="1" name="thickness" loc="data/thickness.image4ub" offset="8" bytesize="256" width="8" height="8" dl="0" />\n'
                       '  <texture id="2" name="alpha" loc="data/alpha.image4ub" offset="8" bytesize="256" width="8" height="8" dl="0" />\n</textures_lib>')


TEXREF = 'type="texref" matrix="%s" addressing_mode_u="wrap" addressing_mode_v="wrap" input_gamma="1"'

# ---- thin_film ------------------------------------------------------------------------------------------------------------------------------
out = textures(base, os.path.join(here, "scenes", "thin_film"))
out = material(out, 4, '''<material id="4" name="bubble_on_glass" type="thin_film">
    <alpha val="0.0" />
    <transparent val="1" />
    <ext_ior val="1.00028" />
    <thickness_map min="120" max="650"><texture id="1" %s /></thickness_map>
    <layers>
      <layer><eta val="1.33" /><k val="0.0" /><thickness val="300" /></layer>
    </layers>
    <eta val="1.5" /><k val="0.0" />
  </material>''' % (TEXREF % "3 0 0 0 0 2 0 0 0 0 1 0 0 0 0 1"))
out = material(out, 1, '''<material id="1" name="film_on_gold" type="thin_film">
    <alpha_u val="0.18" /><alpha_v val="0.12" />
    <layers>
      <layer><eta val="2.2" /><k val="0.0" /><thickness val="140" /></layer>
    </layers>
    <eta val="0.2"><spectrum id="5" type="ref"/></eta>
    <k val="3.0"><spectrum id="6" type="ref"/></k>
  </material>''')
out = material(out, 2, '''<material id="2" name="two_films_on_metal" type="thin_film">
    <alpha val="0.0" />
    <layers>
      <layer><eta val="1.38" /><k val="0.0" /><thickness val="100" /></layer>
      <layer><eta val="2.3" /><k val="0.01" /><thickness val="60" /></layer>
    </layers>
    <eta val="1.1" /><k val="2.6" />
  </material>''')
dst = os.path.join(here, "scenes", "thin_film", "statex_00001.xml")
open(dst, "w", encoding="utf-8").write(out)
print("wrote", dst)

# ---- thin_film_rough ------------------------------------------------------------------------------------------------------------------------
out = textures(base, os.path.join(here, "scenes", "thin_film_rough"))
out = material(out, 4, '''<material id="4" name="rough_film_on_glass" type="thin_film">
    <alpha_u val="0.12" /><alpha_v val="0.2" />
    <transparent val="1" />
    <layers>
      <layer><eta val="1.7" /><k val="0.0" /><thickness val="420" /></layer>
    </layers>
    <eta val="1.45" /><k val="0.0" />
  </material>''')
out = material(out, 1, '''<material id="1" name="oxide_on_metal" type="thin_film">
    <alpha val="0.0" />
    <layers>
      <layer><eta val="2.4" /><k val="0.02" /><thickness val="230" /></layer>
    </layers>
    <eta val="0.9" /><k val="2.9" />
  </material>''')
out = material(out, 2, '''<material id="2" name="textured_roughness" type="thin_film">
    <alpha val="0.5"><texture id="2" %s /></alpha>
    <layers>
      <layer><eta val="1.9" /><k val="0.0" /><thickness val="310" /></layer>
    </layers>
    <eta val="1.3" /><k val="3.4" />
  </material>''' % (TEXREF % "2 0 0 0 0 2 0 0 0 0 1 0 0 0 0 1"))
dst = os.path.join(here, "scenes", "thin_film_rough", "statex_00001.xml")
open(dst, "w", encoding="utf-8").write(out)
print("wrote", dst)
