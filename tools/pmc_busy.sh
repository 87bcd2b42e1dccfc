#!/bin/bash
# MFMA-busy fraction of the factorization kernel from PMC counters (own rocprofv3 --pmc pass, program directly after `--`):
# usage: tools/pmc_busy.sh OUT.json KERNEL_SUBSTRING -- python3 <program> [args]
set -e
out=$1; pat=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_busy_$$
(cd $R && timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $d -o p -- "$@" > $R/gpurun_out/pmc_busy.log 2>&1)
cd $R
python3 - "$d" "$pat" "$out" "$*" <<'PY'
import csv, glob, json, sys, collections, subprocess
d, pat, out, cmd = sys.argv[1:5]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
last = {k: v[-1] for k, v in agg.items()}
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if pat in r["Kernel_Name"]] if kt else []
res = {"command": cmd, "kernel": pat, "dispatches": len(next(iter(agg.values()), [])), "last_dispatch": last, "duration_ns_last": dur[-1] if dur else None}
try:
    res["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    res["commit"] = None
if "SQ_VALU_MFMA_BUSY_CYCLES" in last and "GRBM_GUI_ACTIVE" in last:
    # the counter sums over the 4 SIMDs of all 256 CUs (as in profiles/r02_gemm_pmc.json: 8192^3 -> 0.8986)
    res["mfma_busy_fraction"] = last["SQ_VALU_MFMA_BUSY_CYCLES"] / (last["GRBM_GUI_ACTIVE"] * 128.0)
if "SQ_WAIT_ANY" in last and "SQ_WAVE_CYCLES" in last:
    res["wait_any_fraction_of_wave_cycles"] = last["SQ_WAIT_ANY"] / last["SQ_WAVE_CYCLES"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $d
