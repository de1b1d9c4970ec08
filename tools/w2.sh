r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\")"; }
r x1-tile-1perSIMD X=1 -- --config cfg3
r x1-tile-2perSIMD-b336 X=1 -- --config cfg3 --batch 336
r x1-plain-1perSIMD MVX_TILE=0 -- --config cfg3
