"""tools/gpu_session.sh traffic: per-kernel HBM bytes per dispatch from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
with the gfx950 correction MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-byte request of 16-byte-per-lane reads)."""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for d, c in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            acc[r["Kernel_Name"]][c] += float(r["Counter_Value"])
            n[(r["Kernel_Name"], c)] += 1
res = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --no-cpu --no-parity --no-traffic --steps 1 --warmup 0 %s (one pass per counter)" % (sys.argv[3] if len(sys.argv) > 3 else ""),
       "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024", "kernels": {}}
for k in acc:
    if not any(t in k for t in ("analyse", "degrain", "super_", "compensate", "blockfps", "bf_")):
        continue
    d = n[(k, "FETCH_SIZE")] or 1
    fs = acc[k]["FETCH_SIZE"] / d
    ws = acc[k]["WRITE_SIZE"] / (n[(k, "WRITE_SIZE")] or 1)
    res["kernels"][k.split("(")[0]] = {"dispatches": d, "FETCH_SIZE_KB_per_dispatch": fs, "WRITE_SIZE_KB_per_dispatch": ws, "hbm_bytes_per_dispatch_corrected": 2 * fs * 1024 + ws * 1024}
print(json.dumps(res, indent=1))
