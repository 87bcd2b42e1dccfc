#!/bin/bash
out=${1:-gpurun_out/mask_matrix.txt}
: > "$out"
run() {
  echo "== $*" >> "$out"
  env "$@" REPS=6 python tools/tiled_once.py 2>>"$out.err" | tail -4 | tr '\n' ' ' >> "$out"; echo >> "$out"
}
run OISAT_BATCH_RESERVE_CUS=0
run OISAT_BATCH_RESERVE_CUS=1
run OISAT_BATCH_RESERVE_CUS=2
run OISAT_BATCH_RESERVE_CUS=4
run OISAT_BATCH_RESERVE_CUS=8
cat "$out"
