"""Writes tests/golden/scenes/exr_sky: the env_map fixture with its sky map stored as an OpenEXR file (ZIP-compressed FLOAT channels A, B, G, R;
rows flipped, because the reference's LoadImage4fFromEXR flips tinyexr's rows - imageutils.cpp:382-388 - so the loaded texture equals the
.image4f one and the frames of the two fixtures must be identical), plus three small files for the decoders' own test: ZIPS + HALF RGB,
NONE + FLOAT single channel, ZIP + HALF RGBA of 40 rows (three blocks).  Usage: python tests/golden/make_exr_scene.py"""
import os
import struct
import zlib

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))


def write_exr(path, planes, compression, half):
    """planes: dict channel name -> float32 [h, w]; scanline file, one part."""
    names = sorted(planes)
    h, w = planes[names[0]].shape
    ptype = 1 if half else 2

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    hdr = (b"\x76\x2f\x31\x01" + struct.pack("<I", 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression])) +
           attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
           attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) +
           attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lpb = {0: 1, 2: 1, 3: 16}[compression]
    blocks = []
    for y in range(0, h, lpb):
        raw = b""
        for l in range(y, min(y + lpb, h)):
            for n in names:
                raw += planes[n][l].astype("<f2" if half else "<f4").tobytes()
        data = raw
        if compression != 0:
            d = np.frombuffer(raw, np.uint8)
            t = np.concatenate([d[0::2], d[1::2]]).astype(np.int32)           # split into the even and the odd bytes
            t = np.concatenate([t[:1], (t[1:] - t[:-1] + 128 + 256) & 0xFF]).astype(np.uint8)   # predictor
            z = zlib.compress(t.tobytes(), 6)
            data = z if len(z) < len(raw) else raw
        blocks.append((y, data))
    table_at = len(hdr)
    off = table_at + 8 * len(blocks)
    table, body = b"", b""
    for y, data in blocks:
        table += struct.pack("<Q", off)
        chunk = struct.pack("<ii", y, len(data)) + data
        body += chunk; off += len(chunk)
    open(path, "wb").write(hdr + table + body)


src_dir = os.path.join(here, "scenes", "env_map")
dst_dir = os.path.join(here, "scenes", "exr_sky")
os.makedirs(os.path.join(dst_dir, "data"), exist_ok=True)
raw = open(os.path.join(src_dir, "data", "chunk_00001.image4f"), "rb").read()
w, h = struct.unpack_from("<II", raw, 0)
img = np.frombuffer(raw, "<f4", w * h * 4, 8).reshape(h, w, 4)
flipped = img[::-1]
write_exr(os.path.join(dst_dir, "data", "sky.exr"), {"R": flipped[..., 0], "G": flipped[..., 1], "B": flipped[..., 2], "A": flipped[..., 3]}, 3, False)
xml = open(os.path.join(src_dir, "statex_00001.xml"), encoding="utf-8").read()
xml = xml.replace('loc="data/chunk_00001.image4f" offset="8"', 'loc="data/sky.exr" offset="0"')
xml = xml.replace('loc="data/', 'loc="../env_map/data/').replace('loc="../env_map/data/sky.exr"', 'loc="data/sky.exr"')
open(os.path.join(dst_dir, "statex_00001.xml"), "w", encoding="utf-8").write(xml)
rng = np.random.default_rng(7)
t = os.path.join(here, "exr")
os.makedirs(t, exist_ok=True)
a = rng.uniform(0, 4, (5, 7, 3)).astype(np.float32)
write_exr(os.path.join(t, "zips_half_rgb.exr"), {"R": a[..., 0], "G": a[..., 1], "B": a[..., 2]}, 2, True)
y = rng.uniform(0, 1e5, (9, 6)).astype(np.float32); y[2, 3] = np.inf
write_exr(os.path.join(t, "none_float_y.exr"), {"Y": y}, 0, False)
b = rng.uniform(0, 2, (40, 12, 4)).astype(np.float32)
write_exr(os.path.join(t, "zip_half_rgba.exr"), {"R": b[..., 0], "G": b[..., 1], "B": b[..., 2], "A": b[..., 3]}, 3, True)
print("wrote", dst_dir, "and", t)
