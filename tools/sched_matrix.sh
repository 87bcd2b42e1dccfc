#!/bin/bash
# profiling aid: the localised (tiled) leg of bench.py under the scheduling knobs of dense.BatchedFactor
# usage: tools/sched_matrix.sh OUTFILE
out=${1:-gpurun_out/sched_matrix.txt}
: > "$out"
run() {
  echo "== $*" >> "$out"
  env "$@" python bench.py --only tiled 2>>"$out.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['tiled']['ms_per_step'],2), 'ms', round(d['tiled']['solve_tflops_end_to_end'],1),'TFLOP/s')" >> "$out"
}
run OISAT_BATCH_ORDER=smallest
run OISAT_BATCH_ORDER=largest
run OISAT_BATCH_ORDER=largest OISAT_CHAIN_PRIO=3
run OISAT_BATCH_ORDER=largest OISAT_BATCH_SCHEDULE=sequential
run OISAT_BATCH_ORDER=largest OISAT_BATCH_SCHEDULE=sequential OISAT_CHAIN_PRIO=3
run OISAT_BATCH_ORDER=smallest OISAT_BATCH_SCHEDULE=sequential
cat "$out"
