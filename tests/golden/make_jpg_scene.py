#!/usr/bin/env python3
"""Writes tests/golden/scenes/jpg_textures: the test_035 Cornell box with its 256 x 256 texture stored as a JPEG FILE (baseline, 4:2:0 chroma,
restart markers every MCU row) and its 2 x 2 one as a progressive greyscale .jpeg - what LoadTextureAndMakeCombined reads through
LiteImage::LoadImage<uint32_t> (integrator_pt_scene_tex.cpp:24-33). The files are encoded here with PIL (this script is the only user of an
encoder; the loaders carry their own decoder, csrc/jpeg_decode.h); the meshes are referenced where they lie (../test_035/data)."""
import os
import re
import struct

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "tests", "golden", "scenes", "test_035")
OUT = os.path.join(ROOT, "tests", "golden", "scenes", "jpg_textures")


def read_image4ub(path):
    raw = open(path, "rb").read()
    w, h = struct.unpack_from("<II", raw, 0)
    return np.frombuffer(raw, "<u4", w * h, 8).reshape(h, w)


def main():
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    t0, t1 = read_image4ub(os.path.join(SRC, "data", "chunk_00000.image4ub")), read_image4ub(os.path.join(SRC, "data", "chunk_00001.image4ub"))
    rgb = np.stack([(t1 >> s) & 0xFF for s in (0, 8, 16)], -1).astype(np.uint8)
    Image.fromarray(rgb).save(os.path.join(OUT, "data", "texture1.jpg"), "JPEG", quality=92, subsampling=2, restart_marker_rows=1)
    Image.fromarray((t0 & 0xFF).astype(np.uint8)).save(os.path.join(OUT, "data", "texture0.jpeg"), "JPEG", quality=95, progressive=True)
    xml = open(os.path.join(SRC, "statex_00001.xml")).read()
    xml = xml.replace('loc="data/chunk_00000.image4ub"', 'loc="data/texture0.jpeg"').replace('loc="data/chunk_00001.image4ub"', 'loc="data/texture1.jpg"')
    xml = re.sub(r'loc="data/(chunk_0000[234]\.vsgf)"', r'loc="../test_035/data/\1"', xml)
    head, tail = xml.split('<material id="2"', 1)                           # the floor takes the 2 x 2 file, so that both files are on the rendered path
    xml = head + '<material id="2"' + tail.replace('<texture id="1" type="texref" />', '<texture id="0" type="texref" />', 1)
    open(os.path.join(OUT, "statex_00001.xml"), "w").write(xml)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
