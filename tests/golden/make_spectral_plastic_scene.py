"""Writes tests/golden/scenes/spectral_plastic/statex_00001.xml: the reference's spectral Cornell fixture (tests/golden/scenes/test_spectral,
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
