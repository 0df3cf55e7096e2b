#!/usr/bin/env python3
"""Builds an A/B variant of the library: python profiles/mkvariant.py <name> [--units pt1,wf_trace] -DFLAG[=v] ...
Only the listed translation units (default: all) are recompiled with the flags; the others are taken from the base build.
The result is hydracore3_amd/libhydra_hip_<name>.so (profiles/ab.sh loads it through HYDRA_HIP_LIB)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

name = sys.argv[1]
units, flags = None, []
for a in sys.argv[2:]:
    if a.startswith("--units="):
        units = a.split("=", 1)[1].split(",")
    else:
        flags.append(a)
g.build_hip_library()                                   # the base objects
todo, objs = [], []
for uname, src, uflags in g.UNITS:
    if units is None or uname in units:
        obj = os.path.join(g.OBJDIR, f"{name}_{uname}.o")
        todo.append([g.HIPCC] + g.HIP_FLAGS + flags + uflags + ["-c", os.path.join(g.CSRC, src), "-o", obj])
    else:
        obj = os.path.join(g.OBJDIR, f"{uname}.o")
    objs.append(obj)
with ThreadPoolExecutor(max(1, min(len(todo), os.cpu_count() or 2))) as ex:
    list(ex.map(g._run, todo))
g._run([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "hydracore3_amd", f"libhydra_hip_{name}.so")] + objs)
