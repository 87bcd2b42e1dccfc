set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_c3_fetch -o f -- $CMD > gpurun_out/pmc_c3_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_c3_write -o w -- $CMD > gpurun_out/pmc_c3_write.log 2>&1
python - <<'PY'
import csv, collections, json
res = {}
for tag, f in (("FETCH_SIZE", "gpurun_out/pmc_c3_fetch/f_counter_collection.csv"), ("WRITE_SIZE", "gpurun_out/pmc_c3_write/w_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == tag:
            import re
            mm = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = mm.group(1) if mm else r["Kernel_Name"][:40]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    res[tag] = {k: {"sum_KiB": v[0], "dispatches": v[1]} for k, v in agg.items()}
json.dump(res, open("gpurun_out/pmc_c3_traffic.json", "w"), indent=1)
for tag in res:
    for k, v in sorted(res[tag].items(), key=lambda kv: -kv[1]["sum_KiB"])[:6]:
        print(tag, k, v)
PY
