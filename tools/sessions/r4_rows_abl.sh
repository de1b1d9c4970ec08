#!/bin/bash
# round 4: what a row step is made of -- the instrumented build with the LDS reads / the reference loads / the SADs taken out (results wrong)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r4_rowpass_ablations.txt; : > $O
for v in specprof prof_abl4 prof_abl5 prof_abl6; do echo "== $v" >> $O; MVX_LIB=$PWD/tools/variants/$v.so timeout 200 python tools/specprof.py cfg3 341 2>&1 | grep -E "search launch|inside the row|row passes|live blocks" >> $O; done
cat $O
