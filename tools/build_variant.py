#!/usr/bin/env python3
"""Developer tool: an A/B build of libmvtools_amd.so -> tools/variants/<name>.so (git-ignored; travels to the GPU box; selected there
with MVX_LIB=tools/variants/<name>.so).  Only the named translation units are recompiled with the extra -D switches; the others are
taken from the default build's objects.

    python tools/build_variant.py <name> "<DEF1> <DEF2=..>" mvx_analyse_spec_u16.hip [more.hip ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vapoursynth-mvtools_amd"))
import build as B  # noqa: E402

name, defs, tus = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
B.build()
vdir = os.path.join(ROOT, "tools", "variants")
odir = os.path.join(vdir, "obj_" + name)
os.makedirs(odir, exist_ok=True)
objs, procs = [], []
for s in B.SOURCES:
    o = os.path.join(B.HERE, "build", s.replace(".hip", ".o"))
    if s in tus:
        o = os.path.join(odir, s.replace(".hip", ".o"))
        procs.append(subprocess.Popen([B._hipcc()] + B.FLAGS + ["-Wno-inline-asm"] + ["-D" + d for d in defs] + ["-c", os.path.join(B.CSRC, s), "-o", o]))
    objs.append(o)
for p in procs:
    if p.wait() != 0:
        raise SystemExit("variant build failed")
out = os.path.join(vdir, name + ".so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
