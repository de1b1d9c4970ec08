r() { python bench.py --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], \"fps\", d[\"roofline\"][\"avg_launch_ms\"], \"ms/launch\")"; }
timeout 500 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -2
echo B168; r
echo B168 lds33k; MVX_LDS_MIN=33000 r
echo B21; r --batch 21
echo cfg2; r --config cfg2
