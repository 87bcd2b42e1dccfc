#!/bin/bash
# usage: tools/tiled_timeline.sh OUT.txt [ENV=VAL ...]   -- rocprofv3 kernel trace of tools/tiled_once.py + phase summary
R=$GRAFT_REPO_ROOT
out=$R/$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/tl_$$
rocprofv3 --kernel-trace -d $d -o t --output-format csv -- python3 $R/tools/tiled_once.py > $out.run 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 $R/tools/phases.py $f 2 ${TAIL_MS:-0} > $out
rm -rf $d
cat $out.run | tail -4; cat $out
