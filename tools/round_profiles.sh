#!/bin/bash
# One GPU call that produces the evidence set of a round under gpurun_out/ (copy what is to be judged into profiles/):
#   TAG_bench_default.json        python bench.py
#   TAG_bench_driver_args.json    python bench.py --steps 20 --warmup 5          (the driver's arguments; optional: FULL=1)
#   TAG_config3_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the headline workload alone
#   TAG_c3_hbm_traffic_pmc.json   FETCH_SIZE / WRITE_SIZE passes of the headline step
#   TAG_tier_a_pmc_traffic.json   the same for the reference-parity kernels (bench.py --tier-a-only)
#   TAG_tiled_*/c2_* pmc + kernel stats: FETCH/WRITE traffic and MFMA-busy passes of bench.py --only tiled / secondary
#   TAG_tiled_phases.txt          kernel-trace phase summary of one localised month
#   TAG_rehearse2.json            2 ranks over gloo on this one GPU (the N > 1 path: broadcast, sharded units, gather)
# usage: tools/round_profiles.sh TAG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:-r03_x}
O=gpurun_out
say() { echo "[$(date +%H:%M:%S)] $*"; }
say bench default
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/${TAG}_bench_default.json"))
print("headline", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_source","")[:60])
print("tiled", d["tiled"]["ms_per_step"], "c4", d["config4_strong"]["seconds"], "c2", d["secondary"]["ms_per_step"])
PY
if [ "${FULL:-0}" = "1" ]; then
  say bench driver args
  python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_args.json 2> $O/${TAG}_bench_driver_args.err || exit 1
fi
say kernel stats
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/${TAG}_ks -o ks --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-config4 --steps 2 > $R/$O/${TAG}_ks.log 2>&1 )
f=$(find $O/${TAG}_ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_config3_kernel_stats.csv; rm -rf $O/${TAG}_ks
head -8 $O/${TAG}_config3_kernel_stats.csv
say pmc c3
tools/pmc_traffic.sh ${TAG}_c3 2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary --no-config4 | tail -3
say pmc tier a
tools/pmc_traffic.sh ${TAG}_tier_a 0 -- python3 bench.py --tier-a-only | tail -8
say pmc tiled / config 2 '(traffic + MFMA busy of their task-graph launch)'
tools/pmc_traffic.sh ${TAG}_tiled 0 -- python3 bench.py --only tiled | tail -3
tools/pmc_busy.sh $O/${TAG}_tiled_mfma_busy_pmc.json potrf_dag_kernel -- python3 bench.py --only tiled | tail -12
tools/pmc_traffic.sh ${TAG}_c2 0 -- python3 bench.py --only secondary | tail -3
tools/pmc_busy.sh $O/${TAG}_c2_mfma_busy_pmc.json potrf_dag_kernel -- python3 bench.py --only secondary | tail -12
tools/pmc_busy.sh $O/${TAG}_c3_mfma_busy_pmc.json potrf_dag_kernel -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary --no-config4 | tail -12
say tiled kernel stats
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/${TAG}_kst -o ks --output-format csv -- python3 $R/bench.py --only tiled > $R/$O/${TAG}_kst.log 2>&1 )
f=$(find $O/${TAG}_kst -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_tiled_kernel_stats.csv; rm -rf $O/${TAG}_kst
head -6 $O/${TAG}_tiled_kernel_stats.csv
say tiled phases
tools/tiled_timeline.sh $O/${TAG}_tiled_phases.txt > /dev/null 2>&1; head -3 $O/${TAG}_tiled_phases.txt
say rehearse 2 ranks gloo, launched by bench.py itself from this plain shell
timeout -k 10 500 python3 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --rehearse-on-device0 --c4-months 3 --c4-passes 1 > $O/${TAG}_rehearse2.json 2> $O/${TAG}_rehearse2.err
echo rc=$?; tail -c 300 $O/${TAG}_rehearse2.json
say done
