#!/usr/bin/env python3
"""Writes tests/golden/scenes/png_textures: the test_035 Cornell box with its two textures stored as ordinary image FILES instead of Hydra's
.image4ub containers - texture 0 as a 24-bit BMP, texture 1 as an 8-bit RGBA PNG whose scanlines cycle through all five PNG filter types -
what LoadTextureAndMakeCombined reads through LiteImage::LoadImage<uint32_t> (integrator_pt_scene_tex.cpp:24-33). Own files made from
the fixture's texels; the meshes are referenced where they lie (../test_035/data). Both loaders must produce the texels of the containers."""
import os
import re
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "tests", "golden", "scenes", "test_035")
OUT = os.path.join(ROOT, "tests", "golden", "scenes", "png_textures")


def read_image4ub(path):
    raw = open(path, "rb").read()
    w, h = struct.unpack_from("<II", raw, 0)
    return np.frombuffer(raw, "<u4", w * h, 8).reshape(h, w)


def write_png(path, texels):
    """8-bit RGBA, non-interlaced; row y uses filter type y % 5 (None, Sub, Up, Average, Paeth)."""
    h, w = texels.shape
    px = np.stack([(texels >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(np.int32).reshape(h, w * 4)
    lines = bytearray()
    for y in range(h):
        ft, cur, up = y % 5, px[y], (px[y - 1] if y else np.zeros(w * 4, np.int32))
        a = np.concatenate([np.zeros(4, np.int32), cur[:-4]]); c = np.concatenate([np.zeros(4, np.int32), up[:-4]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (a + up) >> 1
        else:
            pp = a + up - c
            pa, pb, pc = np.abs(pp - a), np.abs(pp - up), np.abs(pp - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
        lines.append(ft)
        lines += ((cur - pred) & 255).astype(np.uint8).tobytes()

    def chunk(typ, data):
        return struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data) & 0xFFFFFFFF)
    comp = zlib.compress(bytes(lines), 9)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))   # two IDAT chunks: they concatenate


def write_bmp24(path, texels):
    """Uncompressed 24-bit BMP with a NEGATIVE height (top-down rows), so that row 0 of the file order is row 0 of the container."""
    h, w = texels.shape
    stride = (w * 3 + 3) & ~3
    rows = bytearray()
    for y in range(h):
        line = np.stack([(texels[y] >> s) & 0xFF for s in (16, 8, 0)], -1).astype(np.uint8).tobytes()
        rows += line + b"\0" * (stride - len(line))
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 24, 0, len(rows), 2835, 2835, 0, 0)
    open(path, "wb").write(hdr + bytes(rows))


def main():
    os.makedirs(os.path.join(OUT, "data"), exist_ok=True)
    t0, t1 = read_image4ub(os.path.join(SRC, "data", "chunk_00000.image4ub")), read_image4ub(os.path.join(SRC, "data", "chunk_00001.image4ub"))
    write_bmp24(os.path.join(OUT, "data", "texture0.bmp"), t0)
    write_png(os.path.join(OUT, "data", "texture1.png"), t1)
    xml = open(os.path.join(SRC, "statex_00001.xml")).read()
    xml = xml.replace('loc="data/chunk_00000.image4ub"', 'loc="data/texture0.bmp"').replace('loc="data/chunk_00001.image4ub"', 'loc="data/texture1.png"')
    xml = re.sub(r'loc="data/(chunk_0000[234]\.vsgf)"', r'loc="../test_035/data/\1"', xml)
    # the floor plane takes the 2 x 2 BMP (point-sized checker), so that both decoders are on the rendered path
    head, tail = xml.split('<material id="2"', 1)
    xml = head + '<material id="2"' + tail.replace('<texture id="1" type="texref" />', '<texture id="0" type="texref" />', 1)
    open(os.path.join(OUT, "statex_00001.xml"), "w").write(xml)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
