# usage: python scratch/trace_gaps.py <kernel_trace.csv>   -> busy time, span and gap statistics of the last analysis
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# find last occurrence of innovation kernel -> start of last analysis
idx = [i for i, e in enumerate(ev) if "innovation" in e[2]]
s = idx[-1]
seg = ev[s:]
# cut at last apply_increment
last = max(i for i, e in enumerate(seg) if "apply_increment" in e[2])
seg = seg[:last + 1]
busy = sum(e[1] - e[0] for e in seg)
span = seg[-1][1] - seg[0][0]
gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
print("kernels", len(seg), "busy %.3f ms span %.3f ms  gaps total %.3f ms mean %.2f us median %.2f us" % (
    busy / 1e6, span / 1e6, sum(gaps) / 1e6, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
by = collections.defaultdict(lambda: [0, 0])
for a, b, n in seg:
    k = n.split("(")[0][-40:]
    by[k][0] += b - a; by[k][1] += 1
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print("  %-42s %8.3f ms  x%-4d avg %.1f us" % (k, t / 1e6, c, t / c / 1e3))
