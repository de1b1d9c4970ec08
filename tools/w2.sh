# developer experiment runner (run on the GPU box from the repo root): one bench line per call
#   r <name> <ENV=value ...> -- <bench.py args>      prints fps, the search launch time and the rest of the step
# e.g.  r plain MVX_CPW=1 -- --config cfg3 ; r default X=1 -- --config cfg3      (MVX_LIB=<path to another build> for A/B of builds)
r() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu --steps 2 --warmup 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$name\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\", round(d[\"ms_per_step\"]-d[\"roofline\"][\"avg_launch_ms\"],1), \"ms other\")"; }
r default X=1 -- --config cfg3
r one-chain-per-workgroup MVX_CPW=1 -- --config cfg3 --batch 168
