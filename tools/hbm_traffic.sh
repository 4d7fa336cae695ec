#!/bin/bash
# HBM bytes of the render kernel per launch for a bench.py command, without the full profile: the FETCH_SIZE and WRITE_SIZE passes of
# tools/profile_bench.sh only (each its own rocprofv3 run).  usage: tools/hbm_traffic.sh [bench args...]   (ACGPT_HIP_LIB selects a library)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/hbm_traffic_$$
ARGS="${@:---no-cpu-baseline --warmup 8}"
cd /tmp && export TMPDIR=/tmp
for PMC in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/$PMC -o pmc -- python3 $REPO/bench.py $ARGS > $OUT.$PMC.log 2>&1 || { echo "$PMC pass failed"; tail -3 $OUT.$PMC.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_pw" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
rd = acc["FETCH_SIZE"] / max(1, n["FETCH_SIZE"]) * 1024 * 2      # KB, and the gfx950 correction of the microarch guide
wr = acc["WRITE_SIZE"] / max(1, n["WRITE_SIZE"]) * 1024
print("bench.py $ARGS: per launch of k_render_pw (%d launches): HBM read %.3f GB, written %.3f GB, total %.3f GB" % (n["WRITE_SIZE"], rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
PY
rm -rf $OUT $OUT.*.log
