"""profiling aid: phase summary of the LAST pass in a rocprofv3 kernel trace (csv) of tools/tiled_once.py:
per kernel and group (caps: grid.y == 2 / 512-thread diagonal launches of 2; tiles: the rest) first start, last end, busy, count"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", 0) or 0),
             int(r.get("Grid_Size_Y", 1) or 1), int(r.get("Workgroup_Size_X", 0) or 0)) for r in rows)
def short(n):
    m = re.search(r"(\w+)_kernel", n)
    return m.group(1) if m else n[:30]
passes, cur, end = [], [], None
for e in ev:
    if end is not None and e[0] - end > 20_000_000:
        passes.append(cur); cur = []
    cur.append(e); end = max(end or 0, e[1])
passes.append(cur)
seg = passes[-1]
t0 = seg[0][0]
ncap = int(sys.argv[2]) if len(sys.argv) > 2 else 2
agg = collections.OrderedDict()
for a, b, n, gx, gy, wx in seg:
    k = short(n)
    if k.startswith("gemm"):
        k += " caps" if gy == ncap else " tiles"
    elif k.startswith("potrf"):
        k += " caps" if gx == ncap * wx else " tiles"
    d = agg.setdefault(k, [a, b, 0, 0]); d[0] = min(d[0], a); d[1] = max(d[1], b); d[2] += b - a; d[3] += 1
print("span %.1f ms, %d kernels" % ((max(e[1] for e in seg) - t0) / 1e6, len(seg)))
for k, (a, b, busy, cnt) in sorted(agg.items(), key=lambda kv: kv[1][0]):
    print("%-30s first %8.2f  last end %8.2f  busy %8.2f ms  x%-4d avg %7.1f us" % (k, (a - t0) / 1e6, (b - t0) / 1e6, busy / 1e6, cnt, busy / cnt / 1e3))
if len(sys.argv) > 3 and float(sys.argv[3]) > 0:                    # the last kernels of the pass in time order: what runs after the factorizations are done
    tail_ms = float(sys.argv[3])
    tend = max(e[1] for e in seg)
    print("---- kernels ending in the last %.1f ms" % tail_ms)
    for a, b, n, gx, gy, wx in sorted(seg, key=lambda e: e[0]):
        if (tend - b) / 1e6 <= tail_ms:
            print("%9.3f ms  %8.1f us  %-22s grid %6d x %d" % ((a - t0) / 1e6, (b - a) / 1e3, short(n), gx // max(wx, 1), gy))
