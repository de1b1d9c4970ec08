"""Builds libmvtools_amd.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python vapoursynth-mvtools_amd/build.py [--force]

The library is git-ignored but travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmvtools_amd.so")
SOURCES = ["mvx_api.hip", "mvx_super.hip", "mvx_analyse.hip", "mvx_analyse_any.hip", "mvx_analyse_u8.hip", "mvx_analyse_u16.hip", "mvx_analyse_spec_u8.hip", "mvx_analyse_spec_u16.hip", "mvx_degrain.hip"]
# -ffp-contract=off: the reference's double arithmetic (lambda scaling, predictor interpolation, degrain weights) must
# not be fused into FMAs; no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-inline-asm",
         "-Wno-unused-function", "-Wno-unused-variable"]
if os.environ.get("MVX_NT_REF"):  # developer-only experiment: non-temporal reference loads
    FLAGS.append("-DMVX_NT_REF")
if os.environ.get("MVX_DEFS"):  # developer-only: extra -D switches for A/B builds, e.g. MVX_DEFS="MVX_NO_EARLY"
    FLAGS += ["-D" + d for d in os.environ["MVX_DEFS"].split()]
if os.environ.get("MVX_PROFILE"):  # developer-only: per-phase cycle counters inside the search kernel
    FLAGS.append("-DMVX_PROFILE")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


VSDIR = os.path.join(HERE, "vsplugin")
VS_PLUGIN = os.path.join(HERE, "libmvtools_vs.so")
VS_HOST = os.path.join(HERE, "mvx_vs_host")


def build_vs(force=False, verbose=False):
    """The VapourSynth API-4 filter shell (plain C, links libmvtools_amd.so) and the mini host used by the tests."""
    srcs = [os.path.join(VSDIR, f) for f in os.listdir(VSDIR)] + [os.path.join(HERE, "..", "include", "mvtools_amd.h")]
    if not force and all(os.path.exists(o) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in srcs) for o in (VS_PLUGIN, VS_HOST)):
        return VS_PLUGIN, VS_HOST
    cc = os.environ.get("CC", "gcc")
    cmds = [[cc, "-std=gnu11", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-fvisibility=hidden", os.path.join(VSDIR, "mvtools_vs.c"), "-o", VS_PLUGIN,
             "-L" + HERE, "-lmvtools_amd", "-Wl,-rpath,$ORIGIN", "-lpthread"],
            [cc, "-std=gnu11", "-O2", "-Wall", "-Wextra", os.path.join(VSDIR, "minihost.c"), "-o", VS_HOST, "-ldl"]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return VS_PLUGIN, VS_HOST


def build(force=False, verbose=False):
    out = _build_lib(force, verbose)
    build_vs(force, verbose)
    return out


def _build_lib(force=False, verbose=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mvtools_amd.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        src = os.path.join(CSRC, s)
        if not force and os.path.exists(o) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps if d.endswith(".h") or d == src):
            continue
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
