#!/bin/bash
# What the round-end driver runs, in one place.  Here (no GPU):   bash tools/verify_round.sh cpu
# On an MI355X box (gpurun -- 'bash tools/verify_round.sh gpu'): GPU tests, smoke, the default bench line, every configuration.
# Every step's exit status counts: the script stops with status 1 at the first failing step.
set -u -o pipefail
cd "$(dirname "$0")/.."
mode=${1:-cpu}
if [ "$mode" = cpu ]; then
  python -c "import __graft_entry__ as g; g.build(); print('build ok')" || exit 1
  python -m pytest tests -x -q -m "not gpu" || exit 1
else
  mkdir -p gpurun_out
  python -m pytest tests -x -q -m gpu 2>&1 | tail -25 || exit 1
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
  python bench.py > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err || { tail -5 gpurun_out/verify_bench.err; exit 1; }
  cut -c1-200 gpurun_out/verify_bench.json
  for c in c3 c4 c5 eval; do
    python bench.py --config $c --no-cpu-baseline --no-kernels 2>gpurun_out/verify_$c.err > gpurun_out/verify_$c.json || { tail -5 gpurun_out/verify_$c.err; exit 1; }
    grep -o '"name": "[a-z0-9]*"\|"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/verify_$c.json | tr '\n' ' '; echo
  done
fi
