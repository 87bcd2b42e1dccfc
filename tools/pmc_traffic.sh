#!/bin/bash
# HBM-side traffic per kernel from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (they do not fit one pass), program directly after `--`, FETCH_SIZE doubled (gfx950 tallies the
# 128-byte read requests of wide coalesced loads at 64 bytes), both in KiB.
# usage: tools/pmc_traffic.sh TAG FACTORIZATIONS -- python3 <program> [args]     -> gpurun_out/TAG_hbm_traffic_pmc.json
set -e
tag=$1; nfact=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_${tag}_$c
  rm -rf $d
  (cd $R && timeout -k 10 900 rocprofv3 --pmc $c --output-format csv -d $d -o p -- "$@" > $R/gpurun_out/pmc_${tag}_$c.log 2>&1)
done
cd $R
python3 tools/pmc_traffic.py $tag $nfact "$*"
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
