#!/usr/bin/env python3
"""Compile one HIP translation unit of the library to gfx950 assembly (no GPU needed) and summarise, per kernel, the
register allocation the compiler reports (-Rpass-analysis=kernel-resource-usage) and the instruction mix of the
assembly (whole kernel and, with --loop, the largest innermost-to-outermost loop body that contains a given label
pattern).  Developer tool: it is how the round-2 register / instruction diet of the search kernel was driven.

    python tools/isa_report.py vapoursynth-mvtools_amd/csrc/mvx_analyse_u16.hip [-D MVX_FAST_PROF ...] [--filter Li16ELi16] [--keep out.s]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wno-unused-function", "-Wno-unused-variable", "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage",
         "-I" + os.path.join(ROOT, "include")]


def classify(op):
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_ld" if not op.startswith("scratch") else "scratch"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vmem_st" if not op.startswith("scratch") else "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(asm):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.strip().startswith(".end_amdhsa_kernel"):
            continue
        if cur and re.match(r"^\s*s_endpgm", line):
            body.append(line)
            yield cur, body
            cur = None
            continue
        if cur is not None:
            body.append(line)


def mix(lines):
    c = collections.Counter()
    for line in lines:
        t = line.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        c[classify(t.split()[0])] += 1
    return c


def main():
    args = sys.argv[1:]
    src = args[0]
    defs, filt, keep = [], None, None
    i = 1
    while i < len(args):
        if args[i] == "-D":
            defs.append("-D" + args[i + 1]); i += 2
        elif args[i] == "--filter":
            filt = args[i + 1]; i += 2
        elif args[i] == "--keep":
            keep = args[i + 1]; i += 2
        else:
            i += 1
    out = keep or tempfile.mktemp(suffix=".s")
    rpt = None
    for a in args:
        if a.startswith("--report="):
            rpt = a.split("=", 1)[1]
    if rpt and os.path.exists(rpt) and os.path.exists(out):  # re-read an earlier compilation
        class P_: pass
        p = P_(); p.stderr = open(rpt).read(); p.returncode = 0
    else:
        p = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + defs + [src, "-o", out], stderr=subprocess.PIPE, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr)
            sys.exit(1)
        if rpt:
            open(rpt, "w").write(p.stderr)
    res = {}
    name = None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); res[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+)", line)
        if m and name:
            res[name][m.group(1).strip()] = int(m.group(2))
    asm = open(out).read()
    for k, body in kernels(asm):
        if filt and filt not in k:
            continue
        r = res.get(k, {})
        c = mix(body)
        print(k)
        print("   VGPRs %s  AGPRs %s  spillV %s  spillS %s  scratch %s  occupancy %s  LDS %s" % (
            r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size")))
        print("   static instr: total %d  %s" % (sum(c.values()), "  ".join("%s %d" % kv for kv in sorted(c.items()))))
    if not keep:
        os.unlink(out)


if __name__ == "__main__":
    main()
