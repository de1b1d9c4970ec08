r() { python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\")"; }
for D in 0 1 2 8 32 64 252 -1 -2 -8 -64 -252; do echo -n "order D=$D: "; MVX_JOB_ORDER=$D r; done
