#!/bin/bash
# per-kernel durations of the C2 step with and without SEDT_IGEMM_BREG (kernel trace of the same ab_step run); ring depth of the BR kernels 3 / 4 / 5
set -u
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SEDT_DEV=1 SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so
for set in "0 3" "1 3" "1 4" "1 5" "0 3"; do
  set -- $set
  export SEDT_IGEMM_BREG=$1 SEDT_IGEMM_BREG_S=$2
  d=$root/gpurun_out/breg_$1_$2
  rm -rf $d
  rocprofv3 --kernel-trace --stats -d $d -o t --output-format csv -- python3 $root/tools/dev/ab_step.py --config c2 --replays 30 --tag breg=$1,S=$2 2>/dev/null | tail -1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if any(k in n for k in ('igemm3_br', 'igemm3_w8', 'pack_frag')):
        print(f"   {n[:66]:66s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
  rm -rf $d
done
