# usage: tools/ab.sh <variantA> <variantB> [bench args]   -- interleaved A/B runs of bench.py with two library builds
A=$1; B=$2; shift 2
r() { MVX_LIB=$PWD/tools/variants/$1.so python bench.py --no-cpu "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$1\", round(d[\"value\"],1), \"fps\", round(d[\"roofline\"][\"avg_launch_ms\"],1), \"ms/launch\")"; }
for i in 1 2 3; do r $A "$@"; r $B "$@"; done
