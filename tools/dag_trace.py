"""profiling aid: one task-graph factorization with OISAT_DAG_TRACE -- where the chain's time goes per diagonal block, how long
the bulk tasks compute and wait, how many workgroups compute at a time.
usage: python tools/dag_trace.py M            (one system of M observations)
       python tools/dag_trace.py FILE.bin     (parse an existing trace)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np


def parse(path):
    raw = open(path, "rb").read()
    nt, cr = np.frombuffer(raw, np.int32, 2)
    tasks = np.frombuffer(raw, np.int32, 4 * nt, 8).reshape(nt, 4)
    st = np.frombuffer(raw, np.int64, 4 * nt + 8 * cr, 8 + 16 * nt)
    return tasks, st[:4 * nt].reshape(nt, 4), st[4 * nt:].reshape(cr, 8)


def report(path):
    tasks, tk, ch = parse(path)
    us = 0.01                                            # 100 MHz stamps -> microseconds
    t0 = tk[:, 0][tk[:, 0] > 0].min()
    t1 = tk[:, 1].max()
    print("tasks %d, span %.1f us" % (len(tasks), (t1 - t0) * us))
    names = {0: "chain", 1: "tile", 2: "sub", 3: "pre"}
    for kind in (1, 2, 3):
        sel = tasks[:, 0] == kind
        if not sel.any():
            continue
        dur = (tk[sel, 1] - tk[sel, 0]) * us
        wait = tk[sel, 2] * us
        print("%-5s x%-6d held %.1f us (mean), polling %.1f us, working %.1f us; sum working %.1f ms"
              % (names[kind], sel.sum(), dur.mean(), wait.mean(), (dur - wait).mean(), (dur - wait).sum() * 1e-3))
    sel = tasks[:, 0] == 1
    tail = (tk[sel, 1] - tk[sel, 3])[tk[sel, 3] > 0] * us
    if len(tail):
        print("tile: T_j seen -> L(i,j) published %.1f us (mean), %.1f (median)" % (tail.mean(), np.median(tail)))
    # the chains
    row = 0
    for c in np.where(tasks[:, 0] == 0)[0]:
        sysid, r0 = tasks[c, 1], tasks[c, 2]
        nxt = [tasks[d, 2] for d in np.where(tasks[:, 0] == 0)[0] if tasks[d, 2] > r0]
        r1 = min(nxt) if nxt else len(ch)
        s = ch[r0:r1]
        nb = len(s)
        if nb < 3:
            continue
        full = s[1:-1]                                   # steps with all six stamps
        ph = {"diagonal block + publish": full[:, 1] - full[:, 0], "wait sub(j)": full[:, 2] - full[:, 1],
              "panel product + store + image + publish": full[:, 3] - full[:, 2], "rank-128 update from LDS": full[:, 4] - full[:, 3],
              "wait pre(j+1)": full[:, 5] - full[:, 4], "write-back + barrier": np.concatenate([s[2:-1, 0] - full[:-1, 5], [0]])}
        step = (s[1:, 0] - s[:-1, 0]) * us
        print("chain of system %d: %d blocks, %.1f us per block (median %.1f); total %.1f us"
              % (sysid, nb, step.mean(), np.median(step), (s[-1, 3] - s[0, 0]) * us))
        for k, v in ph.items():
            print("    %-40s mean %6.2f us   median %6.2f   max %7.2f" % (k, v.mean() * us, np.median(v) * us, v.max() * us))
        if c > 3:
            break
    # workgroups computing at a time (bulk tasks; the polling time is taken off the end of a task's interval)
    sel = tasks[:, 0] != 0
    a = tk[sel, 0]
    b = tk[sel, 1] - tk[sel, 2]
    ev = np.concatenate([np.stack([a, np.ones_like(a)], 1), np.stack([b, -np.ones_like(b)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    lvl = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0])
    print("bulk workgroups not polling, time-averaged: %.1f (of the launch's grid)" % ((lvl[:-1] * dt).sum() / max(1, ev[-1, 0] - ev[0, 0])))
    nseg = 10
    edges = np.linspace(t0, t1, nseg + 1)
    line = []
    for q in range(nseg):
        m = (ev[:-1, 0] >= edges[q]) & (ev[:-1, 0] < edges[q + 1])
        line.append("%.0f" % ((lvl[:-1][m] * dt[m]).sum() / max(1, edges[q + 1] - edges[q])))
    print("   by tenth of the span: " + " ".join(line))


if __name__ == "__main__":
    arg = sys.argv[1] if len(sys.argv) > 1 else "10000"
    if os.path.exists(arg):
        report(arg)
        sys.exit(0)
    from oisatgmi import _hip, synthetic as syn, dense
    m = int(arg)
    ctx = _hip.context()
    ny, nx = (360, 720) if m <= 20000 else (720, 1440)
    p = syn.point_obs_case(ny, nx, m, 4000, swaths=m > 20000)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
    os.environ["OISAT_DAG"] = "1"
    for _ in range(3):
        plan.run(500.0 if m <= 20000 else 300.0, refine=2)
    plan.check()
    out = os.path.join(ROOT, "gpurun_out", "dag_trace_%d.bin" % m)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.environ["OISAT_DAG_TRACE"] = out
    plan.run(500.0 if m <= 20000 else 300.0, refine=2)
    plan.check()
    del os.environ["OISAT_DAG_TRACE"]
    report(out)
