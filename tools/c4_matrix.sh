#!/bin/bash
out=${1:-gpurun_out/c4_matrix.txt}
: > "$out"
run() {
  echo "== $*" >> "$out"
  env "$@" python bench.py --only config4 --c4-passes 3 2>>"$out.err" | python -c "import sys,json; d=json.loads(sys.stdin.read())['config4']; print(round(d['seconds'],4), 's', round(d['solve_tflops_end_to_end'],1),'TFLOP/s')" >> "$out"
}
run OISAT_BATCH_SCHEDULE=overlap
run OISAT_BATCH_SCHEDULE=sequential OISAT_BATCH_ORDER=largest
run OISAT_BATCH_SCHEDULE=sequential OISAT_BATCH_ORDER=smallest
run OISAT_BATCH_SCHEDULE=overlap OISAT_BATCH_RESERVE_CUS=2
cat "$out"
