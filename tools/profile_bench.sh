#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run,
# never combined with tracing domains).  Outputs under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---no-cpu-baseline} --no-ieee-leg"      # one render-kernel instantiation in the trace
echo "== kernel trace" 
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum" \
           "SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64" ; do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  timeout -k 10 400 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; }
done
STEPS_PER_LAUNCH=${STEPS_PER_LAUNCH:-8}
python3 $REPO/tools/summarize_prof.py $OUT ${KERNEL_FILTER:-k_render_pw} "python bench.py $ARGS" $STEPS_PER_LAUNCH > $OUT/summary.json
cp $OUT/trace/trace_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
head -c 1500 $OUT/summary.json
