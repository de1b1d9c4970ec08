r() { python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\")"; }
for k in -1 -1 -1 -1 0 0 0 1 1 1 7 7 7; do echo -n "skew $k: "; MVX_ALLOC_SKEW=$k r; done
