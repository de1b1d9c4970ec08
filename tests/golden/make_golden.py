"""Generates tests/golden/golden.json: FNV-1a hashes of the oracle's outputs (super-frame defined regions, vector blobs,
Degrain / Compensate frames) on seeded synthetic clips.  Run from the repo root:  python tests/golden/make_golden.py
The oracle was first pinned against the reference's recorded blob hashes (tests/test_oracle.py::test_survey_known_answers).
No reference code or data is involved here: inputs come from tests/pipeline.py, expected values from oracle/."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = [
    dict(name="p8_pel2_blk8_ov4_dg1", w=128, h=96, bits=8, radius=1, skw={}, akw=dict(blksize=8, overlap=4)),
    dict(name="p16_pel2_blk16_ov8_dg3", w=192, h=112, bits=16, radius=3, skw={}, akw=dict(blksize=16, overlap=8)),
    dict(name="p8_pel1_blk8_ov0_dg1", w=160, h=96, bits=8, radius=1, skw=dict(pel=1), akw=dict(blksize=8)),
    dict(name="p8_pel4_bicubic_cubic_dg1", w=136, h=80, bits=8, radius=1, skw=dict(pel=4, sharp=1, rfilter=4), akw=dict(blksize=8, overlap=2)),
    dict(name="p10_bilinear_simple_dg2", w=144, h=80, bits=10, radius=2, skw=dict(sharp=0, rfilter=0), akw=dict(blksize=16, overlap=4, search=5, searchparam=4)),
    dict(name="p16_blk32_ov16_dg6", w=256, h=160, bits=16, radius=6, skw={}, akw=dict(blksize=32, overlap=16)),
    dict(name="p8_trymany_nstep_triangle", w=160, h=96, bits=8, radius=1, skw=dict(rfilter=1), akw=dict(blksize=8, overlap=4, trymany=1, search=1, searchparam=2)),
    dict(name="p8_quadratic_exhaustive_badsad", w=160, h=96, bits=8, radius=1, skw=dict(rfilter=3), akw=dict(blksize=8, overlap=4, search=3, badsad=200, badrange=-3)),
]


def run_case(oracle, c):
    import pipeline as pl
    w, h, bits, radius = c["w"], c["h"], c["bits"], c["radius"]
    frames = pl.moving_clip(w, h, bits, 2 * radius + 1, seed=99, noise=3)
    sup = oracle.Super(w, h, bits, **c["skw"])
    sf = [sup.frame(f) for f in frames]
    out = {}
    acc = 2166136261
    regions = sup.defined_regions()
    hs = []
    for (p, lv, k, y0, x0, hh, ww) in regions:
        hs.append(oracle.fnv1a(np.ascontiguousarray(sf[radius][p][y0:y0 + hh, x0:x0 + ww])))
    out["super_regions_fnv"] = "%08x" % oracle.fnv1a(np.array(hs, dtype=np.uint32))
    blobs, refs = [], []
    for d in range(1, radius + 1):
        for isb in (1, 0):
            an = oracle.Analyse(sup, isb=isb, delta=d, **c["akw"])
            nref = radius + (d if isb else -d)
            blobs.append(an.frame(sf[radius], sf[nref]))
            refs.append(sf[nref])
    out["blobs_fnv"] = ["%08x" % oracle.fnv1a(b) for b in blobs]
    dg = oracle.Degrain(radius, sup, an.ad)
    o = dg.frame(frames[radius], refs, blobs)
    out["degrain_fnv"] = ["%08x" % oracle.fnv1a(p) for p in o]
    cp = oracle.Compensate(sup, an.ad)
    o = cp.frame(sf[radius], refs[0], blobs[0])
    out["compensate_fnv"] = ["%08x" % oracle.fnv1a(p) for p in o]
    # round-1 additions: Finest, Recalculate (+ divide), BlockFPS on the first two frames' vector fields
    out["finest_fnv"] = ["%08x" % oracle.fnv1a(np.ascontiguousarray(p)) for p in sup.finest(sf[radius])]
    abw = oracle.Analyse(sup, isb=1, delta=1, **c["akw"])
    afw = oracle.Analyse(sup, isb=0, delta=1, **c["akw"])
    nf = len(frames)
    bbw = [abw.frame(sf[n], sf[n + 1] if n + 1 < nf else None) for n in range(nf)]
    bfw = [afw.frame(sf[n], sf[n - 1] if n >= 1 else None) for n in range(nf)]
    rc = oracle.Recalculate(sup, abw.ad, blksize=8, overlap=4, thsad=100, divide=2)
    out["recalculate_divide_fnv"] = "%08x" % oracle.fnv1a(rc.frame(sf[0], sf[1], bbw[0]))
    for mode in (3, 7):
        bf = oracle.BlockFPS(sup, abw.ad, afw.ad, nf, 24, 1, num=60, den=1, mode=mode, ml=60.0)
        o = bf.frame(1, frames, sf, bbw, bfw)
        out["blockfps_mode%d_fnv" % mode] = ["%08x" % oracle.fnv1a(np.ascontiguousarray(p)) for p in o]
    return out


if __name__ == "__main__":
    import mvoracle
    cases = []
    for c in CASES:
        params = {k: c[k] for k in ("w", "h", "bits", "radius", "skw", "akw")}
        cases.append(dict(name=c["name"], params=params, expect=run_case(mvoracle, params)))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py", cases=cases), f, indent=1)
    print("wrote", len(cases), "cases")
