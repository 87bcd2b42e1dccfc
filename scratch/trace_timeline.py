# usage: python scratch/trace_timeline.py <kernel_trace.csv> [nrows]  -> timeline of the last analysis (relative us, queue, kernel)
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows))
idx = [i for i, e in enumerate(ev) if "innovation" in e[2]]
seg = ev[idx[-1]:]
t0 = seg[0][0]
def short(n):
    m = re.search(r"(\w+)_kernel", n)
    return m.group(1) if m else n[:30]
qs = sorted(set(e[3] for e in seg))
print("queues:", qs)
for a, b, n, q, g in seg[:N]:
    print("%9.1f %9.1f  dur %7.1f  q%-2d grid %-8s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, qs.index(q), g, short(n)))
last = max(i for i, e in enumerate(seg) if "apply_increment" in e[2])
print("span of analysis: %.1f us; busy by queue:" % ((seg[last][1] - t0) / 1e3), {qs.index(q): round(sum(e[1] - e[0] for e in seg[:last + 1] if e[3] == q) / 1e3, 1) for q in qs})
