#!/bin/bash
# A/B of the numbering of the 32-byte nodes on the large scenes — the build's order (0), the two children of a node in
# one 64-byte line (1), depth first (2) — time, L1 miss per access, L2 hit rate, L2 <-> fabric bytes.  Experiments library
# (pt_debug_node_order); same bits in every order.  usage: tools/node_order_ab.sh <out dir> [10m]
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/${1:-gpurun_out/node_order}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ACGPT_EXPERIMENTS=1
SCENES="c5"
[ "$2" = "10m" ] && SCENES="c5 10m"
BIG=""
for S in $SCENES; do
  ARGS="--config 5 --no-cpu-baseline --no-ieee-leg --warmup 2"
  if [ $S = 10m ]; then BIG=$(python3 $REPO/tools/make_big_scene.py --spheres 128 --subdiv 6) || exit 1; ARGS="$ARGS --scene $BIG"; fi
  for ORD in 0 1 2; do
    export ACGPT_NODE_ORDER=$ORD
    T=$OUT/${S}_order$ORD
    timeout -k 10 300 python3 $REPO/bench.py $ARGS > $T.bench.log 2>&1 || { echo "$S order $ORD: bench failed"; tail -3 $T.bench.log; exit 1; }
    grep -q "Memory access fault" $T.bench.log && { echo "GPU fault"; exit 1; }
    i=0
    for PMC in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d $T.pmc$i -o pmc -- python3 $REPO/bench.py $ARGS > $T.pmc$i.log 2>&1 || { echo "$S order $ORD: pmc pass $i failed"; tail -3 $T.pmc$i.log; }
    done
    python3 - $T $S $ORD <<'PY'
import csv, glob, json, sys, collections
t, s, o = sys.argv[1:4]
line = [l for l in open(t + ".bench.log") if l.startswith("{")][-1]
j = json.loads(line)
agg = collections.defaultdict(list)
for f in glob.glob(t + ".pmc*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_pw" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
g = {k: sum(v) / len(v) for k, v in agg.items()}
l1 = g.get("TCP_TCC_READ_REQ_sum", 0) / max(1.0, g.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1))
l2 = g.get("TCC_HIT_sum", 0) / max(1.0, g.get("TCC_HIT_sum", 0) + g.get("TCC_MISS_sum", 0))
fab = g.get("FETCH_SIZE", 0) * 1024 * 2 + g.get("WRITE_SIZE", 0) * 1024
print("%-4s order %s: kernel %.2f ms per launch, %.0f Mray/s | L1 miss / access %.3f, L2 hit %.3f, L2 <-> fabric %.1f GB per launch, TA busy %.2f | scene bytes on the device %.0f MB"
      % (s, o, j["roofline"]["kernel_ms_avg"], j["value"], l1, l2, fab / 1e9, g.get("TA_TA_BUSY_sum", 0) / max(1.0, g.get("GRBM_GUI_ACTIVE", 1) / 8 * 256), j["config"].get("scene_device_bytes", 0) / 1e6), flush=True)
PY
  done
done
