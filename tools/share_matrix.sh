#!/bin/bash
out=${1:-gpurun_out/share_matrix.txt}
: > "$out"
run() {
  echo "== $*" >> "$out"
  env "$@" REPS=6 python tools/tiled_once.py 2>>"$out.err" | tail -4 | tr '\n' ' ' >> "$out"; echo >> "$out"
}
run OISAT_BATCH_MAJOR_PRIO=0 OISAT_BATCH_MINOR_WG=2
run OISAT_BATCH_MAJOR_PRIO=3 OISAT_BATCH_MINOR_WG=2
run OISAT_BATCH_MAJOR_PRIO=3 OISAT_BATCH_MINOR_WG=1
run OISAT_BATCH_MAJOR_PRIO=0 OISAT_BATCH_MINOR_WG=1
run OISAT_BATCH_MAJOR_PRIO=1 OISAT_BATCH_MINOR_WG=2
cat "$out"
