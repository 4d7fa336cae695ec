#!/bin/bash
# One or more extra PMC passes over bench.py (each ';'-separated counter group is its own run).
# usage: tools/pmc_pass.sh <tag> "<grp1>;<grp2>;..." [bench args...]
set -o pipefail
TAG=$1; GROUPS_=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---steps 3 --warmup 1 --no-cpu-baseline}"
i=0
IFS=';' read -ra GR <<< "$GROUPS_"
for PMC in "${GR[@]}"; do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  timeout -k 10 400 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
python3 $REPO/tools/summarize_prof.py $OUT | tee $OUT/summary.json | head -60
