cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
  timeout -k 10 100 python3 scratch/gemm_v10_one.py $v 2>/dev/null | tail -1
  timeout -k 10 100 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_swz$v -o p -- python3 scratch/gemm_v10_one.py $v > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pmc_swz$v/**/*counter_collection.csv", recursive=True)[0]
vals = {}
for r in csv.DictReader(open(f)):
    if "gemm_v10" in r["Kernel_Name"]: vals[r["Counter_Name"]] = float(r["Counter_Value"])
print("   swz $v", vals)
PY
  rm -rf gpurun_out/pmc_swz$v
done
