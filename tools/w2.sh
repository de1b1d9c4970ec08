# developer experiment runner: r <name> <env assignments...> -- <bench args>   (prints fps, search launch ms, rest of the step)
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")"; }
r cfg5-pflate-w2 MVX_LIB=$PWD/tools/variants/pflate.so MVX_W2_32=1 -- --config cfg5 --batch 168
r cfg3-pflate MVX_LIB=$PWD/tools/variants/pflate.so -- --config cfg3
