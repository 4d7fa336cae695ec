#!/bin/bash
# A/B of two builds of the library under the same counters: tools/pmc_ab.sh <tag> <lib.so> [bench args]
# (ACGPT_HIP_LIB selects the library inside acgpathtracing_amd/; each counter group is its own rocprofv3 run)
set -o pipefail
TAG=$1; LIB=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ACGPT_HIP_LIB=$LIB
ARGS="${@:---steps 8 --warmup 8 --no-cpu-baseline}"
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
python3 $REPO/tools/summarize_prof.py $OUT k_render_pw > $OUT/summary.json
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
p=d["pmc_per_launch"]
for k in sorted(p): print("%-32s %.4g" % (k, p[k]))
print(d.get("derived"))
PY
