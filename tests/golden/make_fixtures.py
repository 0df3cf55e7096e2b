#!/usr/bin/env python3
"""Generates tests/golden/rng_kat.json and tests/golden/packxy_16x16_t8.json.

Plain-Python integer restatements of the formulas in the reference (no reference code is executed: the reference
cannot be built here). They pin the oracle's and the GPU's integer arithmetic:
  * RandomGenInit / NextState / rndFloat4_Pseudo   include/crandom.h:17-55
  * kernel_PackXY                                   integrator_rt.cpp:13-31
Run from the repo root:  python tests/golden/make_fixtures.py
"""
import json
import os
import struct

M = 0xFFFFFFFF
HERE = os.path.dirname(os.path.abspath(__file__))


def next_state(s):
    x = (s[0] * 17 + s[1] * 13123) & M
    s[0] = ((x << 13) & M) ^ x
    s[1] ^= (x << 7) & M
    return x


def gen_init(seed):
    s = [(seed * (seed * seed * 15731 + 74323) + 871483) & M, (seed * (seed * seed * 13734 + 37828) + 234234) & M]
    for _ in range(seed % 7):
        next_state(s)
    return s


def to_f32(u):
    """(float)(uint32) * 2^-32 with round-to-nearest-even on the int->float conversion, returned as raw bits."""
    f = struct.unpack("<f", struct.pack("<f", float(u)))[0]          # float(u) is exact in double; pack rounds to nearest even
    return struct.unpack("<I", struct.pack("<f", f * (1.0 / 4294967296.0)))[0]


def float4(s):
    x = next_state(s)
    polys = ((15731, 74323, 871483), (13734, 37828, 234234), (11687, 26461, 137589), (15707, 789221, 1376312589))
    return [to_f32((x * (x * x * a + b) + c) & M) for a, b, c in polys]


def main():
    kat = {}
    for seed in (0, 1, 7, 12345, 1048575):
        s = gen_init(seed)
        init = list(s)
        draws = [float4(s) for _ in range(8)]
        kat[str(seed)] = {"init": init, "float4_bits": draws, "final": list(s)}
    json.dump(kat, open(os.path.join(HERE, "rng_kat.json"), "w"), indent=1)

    W = H = 16
    ts = 8
    packed = [0] * (W * H)
    for y in range(H):
        for x in range(W):
            off = ((x // ts) + (y // ts) * (W // ts)) * ts * ts + (y % ts) * ts + (x % ts)
            packed[off] = ((y << 16) & 0xFFFF0000) | (x & 0xFFFF)
    json.dump({"width": W, "height": H, "tile": ts, "packedXY": packed}, open(os.path.join(HERE, "packxy_16x16_t8.json"), "w"))


if __name__ == "__main__":
    main()
