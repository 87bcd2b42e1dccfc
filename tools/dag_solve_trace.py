"""profiling aid: one localised month through the one-launch analysis (oisat_batch_analyse) with OISAT_DAG_TRACE -- per task
kind how many, how long a workgroup holds its slot, how much of that is polling; and, by twentieth of the launch, how many
workgroups are inside each kind of task (time-averaged) and how many of them are polling.
usage: python tools/dag_solve_trace.py [FILE.bin]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd"), os.path.join(ROOT, "tools")]
import numpy as np
from dag_trace import parse

NAMES = {0: "chain", 1: "tile", 2: "sub", 3: "pre", 5: "fwd", 6: "bwd", 7: "res", 8: "inc"}


def report(path, nseg=20):
    tasks, tk, ch = parse(path)
    tasks = tasks.copy()
    dyn = tasks[:, 0] < 0                                  # rows of the ready queue: the task is packed into stamp word 3
    w = tk[:, 3]
    tasks[dyn, 0] = (w[dyn] & 15).astype(np.int32)
    tasks[dyn, 3] = ((w[dyn] >> 4) & 15).astype(np.int32)
    tasks[dyn, 1] = ((w[dyn] >> 8) & 0xffff).astype(np.int32)
    tasks[dyn, 2] = ((w[dyn] >> 24) & 0xffffffff).astype(np.int32)
    us = 0.01
    ran = tk[:, 0] > 0
    t0, t1 = tk[ran, 0].min(), tk[ran, 1].max()
    span = (t1 - t0) * us
    print("tasks %d, span %.2f ms" % (len(tasks), span * 1e-3))
    print("%-6s %8s %10s %10s %10s %12s %12s" % ("kind", "count", "held us", "polling", "working", "sum held ms", "sum work ms"))
    for kind, name in NAMES.items():
        sel = (tasks[:, 0] == kind) & ran
        if not sel.any():
            continue
        dur = (tk[sel, 1] - tk[sel, 0]) * us
        wait = tk[sel, 2] * us if kind != 0 else np.zeros(sel.sum())
        print("%-6s %8d %10.1f %10.1f %10.1f %12.2f %12.2f" % (name, sel.sum(), dur.mean(), wait.mean(), (dur - wait).mean(), dur.sum() * 1e-3,
                                                                (dur - wait).sum() * 1e-3))
    # the per-workgroup rows sit behind the chains' rows: those with a plausible task count
    grid_rows = ch[np.where((ch[:, 3] > 0) & (ch[:, 3] < 10000) & (ch[:, 4] > ch[:, 0]) & (ch[:, 5] == 0))[0]]
    if len(grid_rows):
        life = (grid_rows[:, 4] - grid_rows[:, 0]) * us
        print("per workgroup (%d rows): alive %.1f ms, claiming work %.2f ms (%.1f %%), in tasks %.2f ms (%.1f %%), %.0f tasks"
              % (len(grid_rows), life.mean() * 1e-3, grid_rows[:, 1].mean() * us * 1e-3, 100 * grid_rows[:, 1].sum() / max(1, (grid_rows[:, 4] - grid_rows[:, 0]).sum()),
                 grid_rows[:, 2].mean() * us * 1e-3, 100 * grid_rows[:, 2].sum() / max(1, (grid_rows[:, 4] - grid_rows[:, 0]).sum()), grid_rows[:, 3].mean()))
        print("   of the claiming: looking at / claiming from the queue %.2f ms, waiting for the claimed entry %.2f ms, the rest (ticket counter) %.2f ms"
              % (grid_rows[:, 6].mean() * us * 1e-3, grid_rows[:, 7].mean() * us * 1e-3, (grid_rows[:, 1] - grid_rows[:, 6] - grid_rows[:, 7]).mean() * us * 1e-3))
    # last factorization ticket finished / first solve ticket started, per wave marker: when does each system's last solve task end
    fact = (tasks[:, 0] <= 3) & ran
    print("last factorization task ends at %.2f ms; last solve task at %.2f ms" % ((tk[fact, 1].max() - t0) * us * 1e-3, span * 1e-3))
    edges = np.linspace(t0, t1, nseg + 1)
    print("workgroups inside each kind of task, time-averaged by 1/%d of the launch (polling ones in brackets):" % nseg)
    hdr = "%-6s" % "kind" + "".join("%7d" % q for q in range(nseg))
    print(hdr)
    for kind, name in NAMES.items():
        sel = (tasks[:, 0] == kind) & ran
        if not sel.any():
            continue
        a, b = tk[sel, 0].astype(np.float64), tk[sel, 1].astype(np.float64)
        row = "%-6s" % name
        for q in range(nseg):
            lo, hi = edges[q], edges[q + 1]
            ov = np.clip(np.minimum(b, hi) - np.maximum(a, lo), 0, None).sum() / (hi - lo)
            row += "%7.0f" % ov
        print(row)
    sel = (tasks[:, 0] != 0) & ran
    a = tk[sel, 0].astype(np.float64)
    p1 = a + tk[sel, 2]                                   # polling is (mostly) at the start of a solve task, at the end of a tile task: a bound
    row = "%-6s" % "poll"
    for q in range(nseg):
        lo, hi = edges[q], edges[q + 1]
        kinds = tasks[sel, 0]
        b = tk[sel, 1].astype(np.float64)
        w = tk[sel, 2].astype(np.float64)
        start_poll = np.where(kinds >= 5, a, b - w)
        end_poll = np.where(kinds >= 5, a + w, b)
        ov = np.clip(np.minimum(end_poll, hi) - np.maximum(start_poll, lo), 0, None).sum() / (hi - lo)
        row += "%7.0f" % ov
    print(row)


if __name__ == "__main__":
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        report(sys.argv[1])
        sys.exit(0)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import time
    from oisatgmi import _hip, synthetic as syn, dense
    ctx = _hip.context()
    ctx.own_stream()
    p = syn.point_obs_case(720, 1440, 100000, 4000, swaths=True)
    ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=12)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    for _ in range(3):
        t0 = time.perf_counter(); ta.run(300.0, refine=2); print("run %.2f ms" % (1e3 * (time.perf_counter() - t0)))
    out = os.path.join(ROOT, "gpurun_out", "dag_solve_trace.bin")
    os.environ["OISAT_DAG_TRACE"] = out
    ta.run(300.0, refine=2)
    del os.environ["OISAT_DAG_TRACE"]
    report(out)
