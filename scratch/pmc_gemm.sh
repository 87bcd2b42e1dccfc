set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 scratch/gemm_pmc.py"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d gpurun_out/pmc_g1 -o p -- $CMD > gpurun_out/pmc_g1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_g2 -o p -- $CMD > gpurun_out/pmc_g2.log 2>&1
python3 - <<'PY'
import csv, json, glob, collections
out = {}
for tag in ("pmc_g1", "pmc_g2"):
    f = glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_nt_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[tag] = {k: v[-1] for k, v in agg.items()}          # last (warm) dispatch
    kt = glob.glob(f"gpurun_out/{tag}/**/*kernel_trace.csv", recursive=True)
    if kt:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "gemm_nt_kernel" in r["Kernel_Name"]]
        out[tag + "_duration_ns_last"] = d[-1]
json.dump(out, open("gpurun_out/pmc_gemm_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc_g1 gpurun_out/pmc_g2
