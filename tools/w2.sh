# developer experiment: one vs two chains per SIMD
r() { # name lib ldsmin config batch
  MVX_LIB=$PWD/tools/variants/$2.so MVX_LDS_MIN=$3 python bench.py --no-cpu --config $4 --batch $5 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$1\", \"$4\", \"batch\", $5, round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\")"; }
r base base 33792 cfg3 168
r expect expect 33792 cfg3 168
r nb4 nb4 33792 cfg3 168
r w2nb4-5perCU w2nb4 16384 cfg3 210
r w2nb4-5perCU w2nb4 16384 cfg3 168
r base base 33792 cfg3 168
r expect expect 33792 cfg3 168
