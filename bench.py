#!/usr/bin/env python3
"""Headline benchmark of the MI355X optimal-interpolation hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full dense Gaussian-B analysis of one synthetic month on one GPU -- innovation,
covariance build S = H B H^T + R, fp32-MFMA Cholesky, gain solve with float64-residual refinement,
increment B H^T z over every grid cell -- with the gridded background and the observations already
resident in HBM when the timed region starts.

Workload at N = 1: the configuration BASELINE.json's north star quotes its target on, which fits
one MI355X: 0.25 deg grid (720 x 1440 = 1,036,800 cells) with 10^5 OMI-NO2-style swath
observations (configs[2]; S alone is 40 GB of the 288 GB).  configs[1] (360 x 720, 10^4 obs) is
timed as a secondary leg in the same run and is the size the parity tests validate end to end.
With N > 1 every rank analyses its own month (months are independent work units, as in the
reference's one-job-per-month launch, run/job_submitter_sbatch.py:45-68): the shared grid is
broadcast once from rank 0 and the analysis fields are gathered every step over RCCL -- weak scaling.
Every run, at every N, additionally times the FIXED 12-month (month x tile) workload of configs[3]
(`config4_strong`): seconds(1) / seconds(N) is the strong-scaling curve.

Prints ONE JSON line (rank 0).  Extra legs, N = 1 only: per-kernel HIP-event timing on the launch
stream for the roofline object, the element-wise (reference-parity) OI on the same grid, and a
bounded CPU run of the float64 oracle for cpu_baseline.
"""
import argparse
import glob
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before anything initialises HIP: see oisatgmi/_hip.py (lanes of concurrent tiles)

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md, HBM3E spec

WORKLOADS = {
    # name: (ny, nx, nobs, L_km, swaths, refine); refine = 2 is the product default (dense.py) and what every full-size
    # parity test validates (tests/test_gpu_parity.py, test_gpu_round2.py): the timed configuration is the validated one
    "config3_720x1440_1e5obs": (720, 1440, 100000, 300.0, True, 2),
    "config2_360x720_1e4obs": (360, 720, 10000, 500.0, False, 2),
    "config1_72x144_1e3obs": (72, 144, 1000, 500.0, False, 2),
}
DEFAULT = "config3_720x1440_1e5obs"
SECONDARY = "config2_360x720_1e4obs"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT, choices=sorted(WORKLOADS))
    ap.add_argument("--refine", type=int, default=None)
    ap.add_argument("--species", default="NO2", choices=["NO2", "HCHO", "O3"],
                    help="parameter set of the synthetic month (BASELINE configs[4]: control_omino2/omihcho/omio3.yml shapes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--tier-a-only", action="store_true", help="run only the reference-parity (Tier-A) kernel legs (profiling aid)")
    ap.add_argument("--only", default="", help="profiling aid: comma list of legs to run INSTEAD of the whole bench -- tiled, "
                    "secondary (config 2 alone), config4, config5, pcie, regrid; prints {leg: result}")
    ap.add_argument("--no-config4", action="store_true", help="skip the fixed-workload strong-scaling legs (12 months x tiles; 3 species x tiles)")
    ap.add_argument("--c4-months", type=int, default=12)
    ap.add_argument("--c4-passes", type=int, default=2)
    ap.add_argument("--c4-shards", default="", help="e.g. 1,2,4,8: emulate the config-4 strong-scaling curve on ONE GPU by timing "
                    "every rank's shard of each world size alone (prints only that leg)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo only to rehearse N>1 on a 1-GPU box")
    ap.add_argument("--rehearse-on-device0", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    ap.add_argument("--launcher-selftest", type=int, default=None, metavar="RC",
                    help="CPU-only check of the --gpus N launch plumbing: no GPU work, rank RC %% N exits with code RC")
    return ap.parse_args()


def build_case(workload, seed, lat2=None, lon2=None, species="NO2"):
    from oisatgmi import synthetic as syn, dense
    ny, nx, nobs, L, swaths, refine = WORKLOADS[workload]
    p = syn.point_obs_case(ny, nx, nobs, seed, swaths=swaths, species=species)
    if lat2 is None:
        lat2, lon2 = p.lat, p.lon
    cell = dense.regular_grid_cell(lat2, lon2, p.obs_lat, p.obs_lon)
    return p, cell, lat2, lon2


def make_plan(ctx, workload, seed, lat2=None, lon2=None, species="NO2"):
    from oisatgmi import dense
    p, cell, lat2, lon2 = build_case(workload, seed, lat2, lon2, species)
    plan = dense.DenseAnalysis(lat2, lon2, max_obs=int(p.obs_y.size), dtype=np.float32, ctx=ctx)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
    return plan


def time_steps(fn, steps, warmup, sync, barrier=None):
    for _ in range(warmup):
        fn()
    sync()
    if barrier:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if barrier:
        barrier()
    sync()
    return time.perf_counter() - t0


def roofline_leg(ctx, plan, L, refine, psteps):
    """Per-kernel durations from HIP events recorded around every launch on the launch stream."""
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(psteps):
        plan.run(L, refine=refine)
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    m = plan.m
    # the factorization's kernels: ONE potrf_dag launch (task graph: tile GEMMs, diagonal blocks and panels inside it), or --
    # OISAT_DAG=0 -- the recursion's GEMM launches (diagonal blocks apart)
    dag = "potrf_dag" in prof
    names = ("potrf_dag",) if dag else ("syrk_gemm", "trsm_gemm")
    gemm_ms = sum(prof[k]["total_ms"] for k in names if k in prof) / psteps
    gemm_launches = sum(prof[k]["launches"] for k in names if k in prof) / psteps
    chol_flops = m ** 3 / 3.0
    achieved = chol_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    traffic, traffic_src = None, None
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_hbm_traffic_pmc.json")))
    if m > 90000 and tfiles:                      # newest PMC pass of this same workload (tools/pmc_traffic.sh, offline)
        tj = json.load(open(tfiles[-1]))
        if any(k.startswith("potrf_dag_kernel") for k in tj.get("kernels", {})) == dag:      # (a pass of the other schedule says nothing about this one)
            traffic = tj["hbm_bytes_per_factorization_corrected"] / tj["launches_per_factorization"]
            traffic_src = (f"profiles/{os.path.basename(tfiles[-1])} at commit {tj.get('commit')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                           "separate passes, FETCH x2 per the gfx950 note; bytes per launch of the factorization's kernels, mean over "
                           "the launches of one factorization)")
    roof = {
        "bound": "mfma",
        "kernel": ("potrf_dag_kernel: the whole Cholesky factorization as one persistent launch of left-looking tile tasks (128x128 "
                   "tiles, the K-loop of gemm_nt_big_kernel), diagonal blocks and panel products included") if dag else
                  ("gemm_nt_big_kernel (K >= 2048: 93 % of the GEMM time at the headline size) + gemm_nt_kernel (persistent, K < 2048) + "
                   "gemm_nt_small_kernel + gemm_nt_rows64_kernel: the syrk_gemm + trsm_gemm launches of one Cholesky factorization"),
        "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
        "traffic": traffic, "traffic_source": traffic_src,
        "algorithmic_flops_per_step": chol_flops, "kernel_ms_per_step": gemm_ms, "launches_per_step": gemm_launches,
        "avg_launch_ms": gemm_ms / gemm_launches if gemm_launches else None,
    }
    if dag and m <= 90000:                                # config 2: its own PMC passes (tools/pmc_traffic.sh ... bench.py --only secondary)
        tj, src = _newest_profile("r*_c2_hbm_traffic_pmc.json")
        dk = next((k for k in (tj or {}).get("kernels", {}) if k.startswith("potrf_dag_kernel")), None)
        if dk:
            roof["traffic"] = tj["kernels"][dk]["bytes_per_dispatch"]
            roof["traffic_source"] = f"{src} at commit {tj.get('commit')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2; bytes per launch)"
        bj, bsrc = _newest_profile("r*_c2_mfma_busy_pmc.json")
        if bj and bj.get("mfma_busy_fraction") is not None:
            roof["mfma_busy_fraction"] = bj["mfma_busy_fraction"]
            roof["mfma_busy_source"] = f"{bsrc} at commit {bj.get('commit')}"
    per_kernel = {k: round(v["total_ms"] / psteps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
    return roof, per_kernel


def tier_a_leg(ctx, ny, nx, nobs, sync):
    """Element-wise (reference-parity) OI on the same grid, device-resident: sweep + knee + analysis."""
    from oisatgmi import synthetic as syn
    from oisatgmi.optimal_interpolation import DiagOI
    c = syn.diag_case(ny, nx, nobs, 3001)
    d = DiagOI(ny * nx, dtype=np.float32, ctx=ctx)
    d.load(c.Xa, c.Y, c.Sa, c.So)
    el_host = time_steps(lambda: d.run(True), 20, 3, sync)            # sweep -> host knee pick -> analysis
    el = time_steps(lambda: d.run_fused(True), 50, 5, sync)           # everything on the device, no host sync
    idx, _ = d.fused_result()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(10):
        d.run_fused(True)
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    n = ny * nx
    apply_ms = prof["oi_apply"]["total_ms"] / prof["oi_apply"]["launches"]
    curve_ms = prof["oi_curve"]["total_ms"] / prof["oi_curve"]["launches"]
    gbs = DiagOI.algorithmic_bytes(n, 4) / (apply_ms * 1e-3) / 1e9
    # sweep: per cell and scaling t = Sa*s, K = t/(t+So), Sb = (1-K)*t, AK = 1 - Sb/t, sum/count -> 9 flop, two of them
    # IEEE divisions (~10 VALU instructions each with -ffp-contract=off and no fast-math: the reference's arithmetic)
    sweep_flops = 9.0 * 99 * n
    sweep_tf = sweep_flops / (curve_ms * 1e-3) / 1e12
    return {"workload": f"OI(regularization_on=True) {ny}x{nx}, {nobs} observed cells, fp32, device-resident",
            "value": n * 50 / el, "unit": "grid-cells/s", "ms_per_call": 1e3 * el / 50,
            "ms_per_call_with_host_knee_pick": 1e3 * el_host / 20, "knee_index": int(idx),
            "kernel_ms": {k: v["total_ms"] / v["launches"] for k, v in prof.items()},
            "roofline_oi_apply": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": DiagOI.algorithmic_bytes(n, 4)},
            "roofline_oi_curve": {"bound": "valu (fp32 division)", "achieved": sweep_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": sweep_tf / MFMA_F32_PEAK_TFLOPS, "algorithmic_flops": sweep_flops,
                                  "note": "a division counted as one flop; peak = fp32 vector rate (157.3, packed); HBM side: "
                                          "8 B/cell read once = %.1f GB/s" % (8.0 * n / (curve_ms * 1e-3) / 1e9)}}


def _tier_a_pmc():
    """newest PMC pass of the Tier-A leg (tools/pmc_traffic.sh ... -- python3 bench.py --tier-a-only): kernel -> bytes per dispatch"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_tier_a_hbm_traffic_pmc.json")))
    if files:
        tj = json.load(open(files[-1]))
        return ({k: v["bytes_per_dispatch"] for k, v in tj["kernels"].items()},
                f"profiles/{os.path.basename(files[-1])} at commit {tj.get('commit')}")
    old = os.path.join(ROOT, "profiles", "r02_a_tier_a_pmc_traffic.json")
    if os.path.exists(old):
        return {k: v["total"] for k, v in json.load(open(old))["bytes_per_dispatch"].items()}, "profiles/r02_a_tier_a_pmc_traffic.json"
    return {}, None


def _hbm(bytes_, ms, pmc_kernel=None):
    """HBM roofline entry of one kernel launch; `traffic` = bytes the PMC passes of this same leg counted for that kernel
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2, offline), per launch."""
    gbs = bytes_ / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes": int(bytes_), "kernel_ms": ms, "traffic": None}
    if pmc_kernel:
        table, src = _tier_a_pmc()
        if pmc_kernel in table:
            out["traffic"] = table[pmc_kernel]
            out["traffic_source"] = f"{src}: {pmc_kernel}"
    return out


def _prof_mean(ctx, fn, reps):
    """mean duration per launch (ms) of every kernel `fn` enqueues, from HIP events on the launch stream"""
    fn()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(reps):
        fn()
    prof = ctx.prof_collect()
    ctx.prof_enable(False)
    return {k: v["total_ms"] / v["launches"] for k, v in prof.items()}


def tier_a_kernels_leg(ctx, sync):
    """HBM rooflines of the reference-parity (Tier-A) kernels at the sizes SURVEY section 8(d) names, device-resident,
    float32 AND float64 (the reference's arithmetic is float64): monthly averaging of a 30-granule 720x1440 stack
    (stack_reduce_kernel: nanmean and error_averager, (k+1) elements per cell), the element-wise OI analysis (8 per cell),
    _upscaler's box-filter + pick (10x10 window on the 0.25 deg grid -> 2.5 deg model grid) and the nearest-neighbour
    query of a 98,640-pixel granule onto the 0.25 deg global grid."""
    from oisatgmi import _hip, synthetic as syn
    from oisatgmi.optimal_interpolation import DiagOI
    from oisatgmi.interpolator import NNIndex, _UpscalePlan
    ny, nx, k = 720, 1440, 30
    n = ny * nx
    rng = np.random.default_rng(12)
    out = {"sizes": {"grid": [ny, nx], "granules": k}}
    host = rng.uniform(0.1, 5.0, size=(k, n)).astype(np.float32)
    host[rng.uniform(size=host.shape) < 0.3] = np.nan
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        dt = np.dtype(dt)
        code, item = _hip.dtype_code(dt), dt.itemsize
        stack = ctx.upload(host, dtype=dt)
        res = ctx.alloc(n * item)
        t = _prof_mean(ctx, lambda: (ctx.check(ctx.lib.oisat_nanmean_stack(ctx.h, code, stack.ptr, k, n, 1, res.ptr)),
                                     ctx.check(ctx.lib.oisat_error_average(ctx.h, code, stack.ptr, k, n, 1, res.ptr))), 10)
        cname = "float" if item == 4 else "double"
        out[f"nanmean_stack_{tag}"] = _hbm((k + 1) * n * item, t["nanmean_stack"], f"stack_reduce_kernel<{cname}, false>")
        out[f"error_average_{tag}"] = _hbm((k + 1) * n * item, t["error_average"], f"stack_reduce_kernel<{cname}, true>")
        stack.free()
        # element-wise OI, analysis kernel + the whole fused call
        c = syn.diag_case(ny, nx, 100000, 3001)
        d = DiagOI(n, dtype=dt, ctx=ctx)
        d.load(c.Xa, c.Y, c.Sa, c.So)
        t = _prof_mean(ctx, lambda: d.run_fused(True), 10)
        el = time_steps(lambda: d.run_fused(True), 30, 3, sync)
        out[f"oi_apply_{tag}"] = _hbm(8 * n * item, t["oi_apply"], f"oi_apply_kernel<{cname}>")
        out[f"oi_fused_{tag}"] = {"ms_per_call": 1e3 * el / 30, "value": n * 30 / el, "unit": "grid-cells/s",
                                  "oi_curve_ms": t["oi_curve"], "oi_curve_read_GBs": 2 * n * item / (t["oi_curve"] * 1e-3) / 1e9}
        # _upscaler: 10 x 10 box filter evaluated at the fine nodes the 72 x 144 model cells pick, 8 stacked fields
        fine_lat, fine_lon = syn.global_grid(ny, nx)
        ctm = syn.regional_ctm_grid(-88.75, 88.75, -178.75, 178.75, 2.5, 2.5)
        plan = _UpscalePlan(fine_lon, fine_lat, ctm, 0.25, float(np.hypot(2.5, 2.5)))
        nf = 8
        fine = ctx.upload(rng.uniform(0.0, 1.0, size=(nf, n)), dtype=dt)
        t = _prof_mean(ctx, lambda: plan.run(fine, nf, dt, False), 10)
        out[f"boxfilter_pick_{tag}"] = _hbm(nf * plan.T * (plan.kx * plan.ky + 1) * item, t["boxfilter_pick"],
                                            f"boxfilter_pick_rows_kernel<{cname}>")
        out[f"boxfilter_pick_{tag}"]["window"] = [plan.ky, plan.kx]
        out[f"boxfilter_pick_{tag}"]["model_cells"] = plan.T
        fine.free()
    # nearest-neighbour query (coordinates are always double): count / scan / scatter / query passes over the point hash
    g, ctm = _regrid_granule()
    nn = NNIndex(g.longitude_center, g.latitude_center)
    lon2, lat2 = np.meshgrid(np.arange(-179.875, 179.876, 0.25), np.arange(-89.875, 89.876, 0.25))
    T = lon2.size
    tb = ctx.upload(np.concatenate([lon2.ravel(), lat2.ravel()]))
    idx = ctx.alloc(T * 4)
    t = _prof_mean(ctx, lambda: ctx.check(ctx.lib.oisat_nn_query(ctx.h, nn.buf.at(0), nn.buf.at(nn.P * 8), nn.P, tb.at(0),
                                                                 tb.at(T * 8), T, 0.5, idx.ptr, None)), 10)
    total = sum(v for kname, v in t.items() if kname.startswith("nn_"))
    out["nn_query_f64"] = _hbm(16 * nn.P + 16 * T + 4 * T, total)
    out["nn_query_f64"].update(points=nn.P, targets=T, passes_ms=t,
                               note="bytes = point + target coordinates (double lon/lat) read once + int32 index written; the "
                                    "query pass walks the uniform-cell hash (L2-resident), so this is a latency/L2-bound kernel")
    return out


def pcie_leg(sync):
    """The drop-in NumPy surface, host arrays in / host arrays out (PCIe-inclusive; never `value`)."""
    import contextlib, io
    from oisatgmi import synthetic as syn
    from oisatgmi.optimal_interpolation import OI
    from oisatgmi.averaging import error_averager
    c = syn.diag_case(720, 1440, 100000, 3001)
    out = {}
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        a = [x.astype(dt) for x in (c.Xa, c.Y, c.Sa, c.So)]
        with contextlib.redirect_stdout(io.StringIO()):
            OI(a[0], a[1].copy(), a[2], a[3], True)
            t0 = time.perf_counter()
            for _ in range(5):
                OI(a[0], a[1].copy(), a[2], a[3], True)
            out[f"OI_reg_on_720x1440_{tag}_ms"] = 1e3 * (time.perf_counter() - t0) / 5
    e = np.random.default_rng(3).uniform(0.01, 1.0, size=(30, 720, 1440))
    error_averager(e)
    t0 = time.perf_counter()
    error_averager(e)
    out["error_averager_30x720x1440_f64_ms"] = 1e3 * (time.perf_counter() - t0)
    return out


def _newest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return (json.load(open(files[-1])), "profiles/" + os.path.basename(files[-1])) if files else (None, None)


def _dag_roofline(prof_by_handle, flops_per_run, runs, tag, what):
    """MFMA roofline of the task-graph launch(es) of a leg: achieved = algorithmic flops (sum over the systems of m^3/3) /
    the launch's duration, HIP events on the launch stream; traffic / MFMA-busy from the newest PMC passes of this same leg
    under profiles/ (tools/pmc_traffic.sh, tools/pmc_busy.sh: `bench.py --only <leg>`, program directly after `--`)."""
    ms = sum(pr[k]["total_ms"] for pr in prof_by_handle for k in ("potrf_dag", "analyse_dag") if k in pr) / runs
    launches = sum(pr[k]["launches"] for pr in prof_by_handle for k in ("potrf_dag", "analyse_dag") if k in pr) / runs
    if ms <= 0:
        return None
    achieved = flops_per_run / (ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": what, "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / MFMA_F32_PEAK_TFLOPS, "algorithmic_flops_per_step": flops_per_run, "kernel_ms_per_step": ms,
            "launches_per_step": launches, "traffic": None, "mfma_busy_fraction": None}
    tj, src = _newest_profile(f"r*_{tag}_hbm_traffic_pmc.json")
    dk = next((k for k in (tj or {}).get("kernels", {}) if k.startswith("potrf_dag_kernel")), None)
    if dk:
        roof["traffic"] = tj["kernels"][dk]["bytes_per_dispatch"]
        roof["traffic_source"] = f"{src} at commit {tj.get('commit')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2; bytes per launch)"
        roof["flop_per_byte_at_the_fabric"] = flops_per_run / launches / roof["traffic"] if launches else None
    bj, bsrc = _newest_profile(f"r*_{tag}_mfma_busy_pmc.json")
    if bj and bj.get("mfma_busy_fraction") is not None:
        roof["mfma_busy_fraction"] = bj["mfma_busy_fraction"]
        roof["mfma_busy_source"] = f"{bsrc} at commit {bj.get('commit')} (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128), last launch)"
    return roof


def tiled_leg(ctx, workload, sync):
    """BASELINE configs[2] as worded: localised block-B -- 30 deg x 30 deg tiles, halo 3 L -- with its own roofline object:
    the month's ONE task-graph launch (factorization of all 50 systems, and -- round 4 -- their gain solves and increments)."""
    from oisatgmi import dense
    ny, nx, nobs, L, swaths, refine = WORKLOADS[workload]
    p, cell, lat2, lon2 = build_case(workload, 4000)
    ta = dense.TiledAnalysis(lat2, lon2, tile_deg=30.0, halo_km=3.0 * L, dtype=np.float32, ctx=ctx)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    ta.run(L, refine=refine, check_pd=True)
    steps = 5
    el = time_steps(lambda: ta.run(L, refine=refine), steps, 1, sync)
    sizes = [int(t["obs"].size) for t in ta.tiles]
    chol_flops = sum(float(m) ** 3 / 3.0 for m in sizes)
    handles = list(ta.factor.ctxs) if ta.factor is not None else []
    for hc in handles:
        hc.prof_reset()
        hc.prof_enable(True)
    for _ in range(3):
        ta.run(L, refine=refine)
    profs = [hc.prof_collect() for hc in handles]
    for hc in handles:
        hc.prof_enable(False)
    per_kernel = {}
    for pr in profs:
        for k, v in pr.items():
            per_kernel[k] = per_kernel.get(k, 0.0) + v["total_ms"] / 3
    out = {"workload": f"{workload}, localised block-B: {len(ta.tiles)} tiles (30x30 deg; each polar band is one cap tile, its "
                       f"observation set being the same at every longitude), halo {3.0 * L:.0f} km",
           "solve_tflop_executed": ta.flops / 1e12,
           "value": ny * nx * steps / el, "unit": "grid-cells/s", "ms_per_step": 1e3 * el / steps,
           "obs_per_tile_min_median_max": [min(sizes), int(np.median(sizes)), max(sizes)],
           "solve_tflops_end_to_end": ta.flops / (el / steps) / 1e12,
           "roofline": _dag_roofline(profs, chol_flops, 3, "tiled", "potrf_dag_kernel over the month's 50 systems (polar caps + 30 deg tiles) as one "
                                     "persistent launch"),
           "group_stream_kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])}}
    ta.close()
    return out


def regrid_leg(ctx, sync):
    """One OMI-NO2-like granule (1644 x 60 pixels, 35 scattering-weight + 35 pressure levels, vcd, amf,
    uncertainty = 73 fields) regridded onto a 0.25 deg global model grid through the drop-in
    interpolator() (host arrays in, host arrays out): nearest-neighbour type 4, Delaunay type 1, RBF type 3.
    The reference does one k-d tree build + query and one evaluation PER FIELD (interpolator.py:162-209)."""
    from oisatgmi import synthetic as syn
    from oisatgmi.interpolator import interpolator, _plan_cache
    g = syn.swath_granule(7007, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=20.0, width_deg=24.0)
    nz = 35
    rng = np.random.default_rng(5)
    g.scattering_weights = rng.uniform(0.1, 2.0, size=(nz,) + g.vcd.shape).astype(np.float32)
    g.pressure_mid = rng.uniform(50, 1000, size=(nz,) + g.vcd.shape).astype(np.float32)
    ctm = syn.regional_ctm_grid(-89.875, 89.875, -179.875, 179.875, 0.25, 0.25)
    out = {"workload": "interpolator(): 98,640-pixel granule, 73 fields -> 0.25 deg global grid (720x1440), float64, host in/out"}
    import contextlib, io
    for it in (4, 1, 3):
        _plan_cache.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            interpolator(it, 0.25, g, ctm, 0.75)                      # warm-up (allocations, plan cache)
            sync()
            t0 = time.perf_counter()
            r = interpolator(it, 0.25, g, ctm, 0.75)
            sync()
        dt = time.perf_counter() - t0
        out[f"type{it}_s_per_granule"] = dt
        out[f"type{it}_fields_per_s"] = 73 / dt
    # a month's loop over granules (reader.py:1405): type 1 with the triangulations of the granules ahead built by worker
    # processes while the device regrids the current one (interpolator_many) -- amortised over 32 granules of that size
    from oisatgmi.interpolator import interpolator_many
    many = []
    for k in range(32):
        gk = syn.swath_granule(7100 + k, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=-170.0 + 10.0 * k, width_deg=24.0)
        gk.scattering_weights, gk.pressure_mid = g.scattering_weights, g.pressure_mid
        many.append(gk)
    with contextlib.redirect_stdout(io.StringIO()):
        interpolator_many(1, 0.25, many[:2], ctm, 0.75)
        sync()
        t0 = time.perf_counter()
        res = interpolator_many(1, 0.25, many, ctm, 0.75)
        sync()
    dt = time.perf_counter() - t0
    out["type1_many_s_per_granule"] = dt / len(many)
    out["type1_many"] = {"granules": len(many), "seconds": dt, "regridded": sum(r is not None for r in res),
                         "worker_processes": max(1, min(8, len(os.sched_getaffinity(0)) - 1, len(many))),
                         "note": "interpolator_many: qhull for the granules ahead in worker processes, barycentric transforms and regrid of the current one on the device; outputs bit-identical to the serial calls"}
    return out


def _regrid_granule():
    from oisatgmi import synthetic as syn
    g = syn.swath_granule(7007, nscan=1644, npix=60, lat0=-70.0, lat1=70.0, lon_c=20.0, width_deg=24.0)
    ctm = syn.regional_ctm_grid(-89.875, 89.875, -179.875, 179.875, 0.25, 0.25)
    return g, ctm


def cpu_baseline(workload):
    """Time the float64 oracle (oracle/oi_oracle.py: NumPy + SciPy Cholesky, the CPU restatement of the reference's
    algorithm) on the benchmark host, on BOUNDED samples of the workloads (about 10-30 s each):
      * dense analysis: the first `m_s` observations (>= 25,000) of the config-3 month and a random subset of 8,192
        grid cells for the increment; the Cholesky cost grows as m^3, so the full step (m = 1e5) is extrapolated;
      * element-wise OI(regularization_on=True) at 720x1440 (the reference's own function; 16.7 s in BASELINE.md);
      * error_averager on a 30x720x1440 stack (vectorised restatement; the reference's triple Python loop needs ~22 s);
      * _upscaler 10x10 on the 0.25 deg grid; interpolator type 4 for three fields of the 98,640-pixel granule."""
    from oracle import oi_oracle as orc
    import contextlib, io
    ny, nx, nobs, L, swaths, _ = WORKLOADS[workload]
    p, cell, _, _ = build_case(workload, 424242)
    m_full = int(p.obs_y.size)
    m_s = min(m_full, 25000)
    ncell_s = min(p.Xa.size, 8192)
    sel = np.random.default_rng(1).choice(p.Xa.size, ncell_s, replace=False)
    import scipy.linalg as sla
    # the oracle's dense analysis, stage by stage (oracle/oi_oracle.py dense_oi: same functions, same order), so that the
    # O(m^3) stage can be extrapolated to the full step by itself
    t0 = time.perf_counter()
    sb = np.sqrt(p.Sa.ravel())
    po = orc.unit_vectors(p.obs_lat[:m_s], p.obs_lon[:m_s])
    so = sb[cell[:m_s]]
    S = orc.gaussian_corr(po, po, L) * so[:, None] * so[None, :]
    S[np.diag_indices_from(S)] += p.obs_var[:m_s]
    d = np.where(p.obs_y[:m_s] < 0, 0, p.obs_y[:m_s]) - p.Xa.ravel()[cell[:m_s]]
    t_build = time.perf_counter() - t0
    t1 = time.perf_counter()
    cf = sla.cho_factor(S, lower=True, overwrite_a=True, check_finite=False)
    t_chol = time.perf_counter() - t1
    t1 = time.perf_counter()
    z = sla.cho_solve(cf, d, check_finite=False)
    pg = orc.unit_vectors(p.lat.ravel()[sel], p.lon.ravel()[sel])
    inc = sb[sel] * (orc.gaussian_corr(pg, po, L) @ (so * z))
    t_rest = time.perf_counter() - t1
    del S, cf
    dt = t_build + t_chol + t_rest
    assert np.isfinite(inc).all()
    chol_tflops = m_s ** 3 / 3.0 / t_chol / 1e12
    full_chol_s = (m_full / m_s) ** 3 * t_chol
    try:
        import threadpoolctl
        thr = max((i.get("num_threads", 1) for i in threadpoolctl.threadpool_info()), default=1)
    except Exception:
        thr = os.cpu_count() or 1
    out = {"value": sel.size / dt, "unit": "grid-cells/s", "cores": int(thr), "kind": "port",
           "sample": f"oracle dense analysis (float64 NumPy/SciPy: covariance build, cho_factor, cho_solve, increment; BLAS threads = "
                     f"cores) on {sel.size} cells x {m_s} obs drawn from {workload} in {dt:.1f} s",
           "dense_sample_seconds": {"build_S": t_build, "cho_factor": t_chol, "solve_and_increment": t_rest},
           "cholesky_tflops_f64": chol_tflops,
           "cholesky_seconds_extrapolated_to_full": full_chol_s,
           "extrapolation": f"Cholesky cost ~ m^3: ({m_full}/{m_s})^3 x {t_chol:.2f} s = {full_chol_s:.0f} s for the factorization "
                            f"of the full step alone (the GPU step is the whole analysis)",
           "value_extrapolated_full_step": ny * nx / full_chol_s}
    from oisatgmi import synthetic as syn
    c = syn.diag_case(720, 1440, 100000, 3001)
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        orc.OI(c.Xa.copy(), c.Y.copy(), c.Sa, c.So, regularization_on=True)
        out["OI_reg_on_720x1440_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.OI(c.Xa.copy(), c.Y.copy(), c.Sa, c.So, regularization_on=False)
        out["OI_reg_off_720x1440_s"] = time.perf_counter() - t0
    e = np.random.default_rng(3).uniform(0.01, 1.0, size=(30, 720, 1440))
    e[e < 0.3] = np.nan
    t0 = time.perf_counter()
    orc.error_averager(e)
    out["error_averager_30x720x1440_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    np.nanmean(e, axis=0)
    out["nanmean_30x720x1440_s"] = time.perf_counter() - t0
    del e
    fine_lat, fine_lon = syn.global_grid(720, 1440)
    ctm10 = syn.regional_ctm_grid(-88.75, 88.75, -178.75, 178.75, 2.5, 2.5)
    Z = np.random.default_rng(4).uniform(size=(720, 1440))
    t0 = time.perf_counter()
    orc.upscaler(fine_lon, fine_lat, Z, ctm10, 0.25, float(np.hypot(2.5, 2.5)))
    out["upscaler_10x10_720x1440_s_per_field"] = time.perf_counter() - t0
    # the regridding row: the oracle follows the reference's algorithm (one k-d tree build + query per field,
    # interpolator.py:162-209); vcd, amf and uncertainty of the 98,640-pixel granule
    g, ctm = _regrid_granule()
    one = type(g)(g.vcd, g.amf, g.time, g.tropopause, g.latitude_center, g.longitude_center, [], [], g.uncertainty,
                  g.quality_flag, np.empty((1)), np.empty((1)), False, [], [], [], [])
    t1 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        orc.interpolator(4, 0.25, one, ctm, 0.75, record_type=type(g))
    out["regrid_type4_s_for_3_fields"] = time.perf_counter() - t1
    return out

def _month_cases(args):
    """The fixed config-4 workload: `c4_months` seeded synthetic months (every rank derives the same ones, nothing to broadcast)."""
    from oisatgmi import synthetic as syn
    ny, nx, nobs, L, swaths, refine = WORKLOADS[DEFAULT]
    return {mth: syn.point_obs_case(ny, nx, nobs, 4000 + mth, swaths=swaths) for mth in range(args.c4_months)}


def _species_cases():
    """The fixed config-5 workload: one 720x1440 month of each control_*.yml species (run/control_omino2.yml:23,
    control_omihcho.yml, control_omio3.yml: value ranges, ctm_error, observation-error model)."""
    from oisatgmi import synthetic as syn
    ny, nx, nobs, L, swaths, refine = WORKLOADS[DEFAULT]
    return {sp: syn.point_obs_case(ny, nx, nobs, 4000, swaths=swaths, species=sp) for sp in ("NO2", "HCHO", "O3")}


def _c4_workload(cases, lat2, lon2):
    """(analysis x tile) units of a dict of cases with their obs^3 weights and cell counts."""
    from oisatgmi import dense
    halo = 3.0 * WORKLOADS[DEFAULT][3]
    units, weights, cells = [], [], []
    for key, p in cases.items():
        for ti, t in enumerate(dense.tile_partition(lat2, lon2, p.obs_lat, p.obs_lon, 30.0, halo)):
            if t["obs"].size:
                units.append((key, ti))
                weights.append(float(t["obs"].size) ** 3)
                cells.append((t["rows"][1] - t["rows"][0]) * (t["cols"][1] - t["cols"][0]))
    return units, weights, cells


def _c4_shard(ctx, cases, units, part, lat2, lon2, cap):
    """One rank's shard of the (analysis x tile) units as a MonthTileBatch on the lanes of one pool."""
    from oisatgmi import dense
    halo = 3.0 * WORKLOADS[DEFAULT][3]
    batch = dense.MonthTileBatch(lat2, lon2, 30.0, halo, np.float32, ctx=ctx, streams=12)
    for key, p in cases.items():
        only = [units[i][1] for i in part if units[i][0] == key]
        if only:
            batch.add_month(key, p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var, only=only)
    batch.build(min_slab_elems=cap)
    assert sorted((k, ti) for k, ti, _ in batch.units) == sorted(units[i] for i in part)
    return batch


def config4_shards_leg(ctx, args, lat2, lon2, sync, worlds):
    """Single-GPU EMULATION of the strong-scaling curve of config 4: for every world size W in `worlds` the W shards of
    the same obs^3-weighted LPT partition the multi-GPU leg uses are run ONE AFTER THE OTHER on this GPU, each exactly as
    its rank would run it (own MonthTileBatch, own lanes, whole shard enqueued, checked at the end), and timed alone.
    max_r seconds(W, r) is what the slowest rank of a W-GPU job would need before the gather (one dist.gather of a few
    tens of MB per rank over xGMI, not emulated here); seconds(1) / max_r seconds(W, r) is the speed-up that W GPUs of
    this kind would give.  It measures what a load-balance bound cannot: the fixed per-rank costs (polar caps, lock-step
    levels with few members, host enqueue) that do not shrink with the shard."""
    from oisatgmi import parallel
    _, _, _, L, _, refine = WORKLOADS[DEFAULT]
    cases = _month_cases(args)
    units, weights, cells = _c4_workload(cases, lat2, lon2)
    out = {"workload": f"{args.c4_months} months x (720x1440, 1e5 swath obs) as {len(units)} (month x tile) units",
           "note": "single-GPU emulation: the W shards of a W-rank job timed one after the other on one MI355X; no gather"}
    base = None
    for W in worlds:
        parts = parallel.partition_units(len(units), W, weights)
        secs = []
        for r in range(W):
            cap = sum(2 * cells[i] for i in parts[r])
            batch = _c4_shard(ctx, cases, units, parts[r], lat2, lon2, cap)
            batch.run(L, refine=refine, check_pd=True)
            batch.run(L, refine=refine, wait=False)
            batch.check()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.c4_passes):
                batch.run(L, refine=refine, wait=False)
                batch.check()
            sync()
            secs.append((time.perf_counter() - t0) / args.c4_passes)
            batch.close()
        if W == 1:
            base = secs[0]
        loads = [sum(weights[i] for i in part) for part in parts]
        out[f"world_{W}"] = {"rank_seconds": [round(x, 4) for x in secs], "slowest_rank_seconds": max(secs),
                             "speedup_vs_1": (base / max(secs)) if base else None,
                             "speedup_bound_from_load_balance": sum(loads) / max(loads),
                             "units_per_rank": [len(part) for part in parts]}
    return out


def config4_leg(ctx, args, world, rank, local, lat2, lon2, sync, barrier, cases=None, label=None):
    """BASELINE configs[3], STRONG scaling: a FIXED workload -- `c4_months` synthetic 720x1440 months of 10^5 swath
    observations, each cut into 30 deg x 30 deg tiles with a 3 L halo (localised block-B) -- split into (month x tile)
    units, sharded over the ranks by obs^3-weighted LPT (parallel.partition_units), every rank running its whole shard
    on the lanes of one pool with no collective and no host synchronisation inside, and ONE gather of all `xa | inc`
    tiles to rank 0 at the end.  The same leg runs at every --gpus N (N = 1 included), so seconds(1) / seconds(N) is the
    strong-scaling curve (reference: one scheduler job per month, run/job_submitter_sbatch.py:45-68; with months as the
    only unit 8 GPUs cap at 12/2 = 6.0x, hence the finer unit).
    `cases` / `label`: another fixed dict of analyses through the same machinery -- BASELINE configs[4], the three species'
    months as 3 x 50 (species x tile) units (`config5_strong`)."""
    import torch
    import torch.distributed as dist
    from oisatgmi import dense, parallel
    ny, nx, nobs, L, swaths, refine = WORKLOADS[DEFAULT]
    halo = 3.0 * L
    if cases is None:
        cases = _month_cases(args)
        label = f"{args.c4_months} months x (720x1440, 1e5 swath obs)"
    nanalyses = len(cases)
    units, weights, cells = _c4_workload(cases, lat2, lon2)
    parts = parallel.partition_units(len(units), world, weights)
    loads = [sum(weights[i] for i in part) for part in parts]
    cap = max(sum(2 * cells[i] for i in part) for part in parts)      # slab elements every rank sends
    batch = _c4_shard(ctx, cases, units, parts[rank], lat2, lon2, cap)
    del cases
    send = torch.as_tensor(parallel.DeviceView(batch.slab.ptr, cap, "<f4"), device=torch.device("cuda", local))

    def one_pass():
        batch.run(L, refine=refine, wait=False)           # enqueue the whole shard, heaviest unit first
        if world > 1:                                     # wait for the lanes; a failed solve on ANY rank raises on every rank,
            return parallel.checked_gather(send, batch.check)   # then the one collective of the data path
        batch.check()
        return [send]

    batch.run(L, refine=refine, check_pd=True)             # checked pass, untimed
    one_pass()                                             # warm-up (RCCL connections)
    sync()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.c4_passes):
        got = one_pass()
    sync()
    t_own = time.perf_counter() - t0                       # this rank's own time, before waiting for the others
    if barrier:
        barrier()
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el, t_own], device="cuda", dtype=torch.float64)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = tt.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        el, own_max, own_min = float(tmax[0]), float(tmax[1]), float(tmin[1])
    else:
        own_max = own_min = t_own
    out = None
    if rank == 0:
        # every unit's tile arrived: finite, and the increment is not identically zero
        ok = True
        for r, part in enumerate(parts):
            host = got[r].cpu().numpy()
            used = sum(2 * cells[i] for i in part)
            ok = ok and bool(np.isfinite(host[:used]).all()) and bool(np.any(host[:used] != 0))
        sec = el / args.c4_passes
        flops = sum(dense.DenseAnalysis.flops(round(w ** (1.0 / 3.0))) for w in weights)
        out = {"workload": f"{label}, localised block-B: 30 deg tiles (polar bands as single cap tiles), halo {halo:.0f} km",
               "scaling": "strong", "n_gpus": world, "units": len(units), "seconds": sec,
               "value": nanalyses * ny * nx / sec, "unit": "grid-cells/s",
               "solve_tflops_end_to_end": flops / sec / 1e12,
               "max_rank_load_over_mean": max(loads) / (sum(loads) / world),
               "speedup_bound_from_load_balance": sum(loads) / max(loads),
               "rank_seconds_min_max": [own_min / args.c4_passes, own_max / args.c4_passes],
               "units_per_rank": [len(part) for part in parts], "all_tiles_arrived_finite": ok,
               "gather": "one dist.gather of %.1f MB per rank to rank 0 per pass" % (4e-6 * cap)}
    batch.close()
    return out


def config5_leg(ctx, args, world, rank, local, lat2, lon2, sync, barrier):
    """BASELINE configs[4] as a SHARDED workload: one 720x1440 month of each of NO2, HCHO and O3 (the three control_*.yml
    parameter sets) = 3 x 50 (species x tile) units through the same partition / MonthTileBatch / one-gather machinery as
    config 4, at every N (N = 1 included): seconds(1) / seconds(N) is its strong-scaling curve."""
    return config4_leg(ctx, args, world, rank, local, lat2, lon2, sync, barrier, cases=_species_cases(),
                       label="3 species (NO2, HCHO, O3: run/control_omino2.yml, control_omihcho.yml, control_omio3.yml shapes) x (720x1440, 1e5 swath obs)")


def launch_command(ngpus, argv, port):
    """The command line `--gpus N` (N > 1) from a plain shell turns into: one rank per GPU under torch.distributed.run, this
    same script and arguments (reference: one job per month, run/job_submitter_sbatch.py:45-68)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(ngpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes -- before this process has
    imported torch or made any HIP call (a process that has initialised the GPU must not be replaced or forked) -- pass
    the arguments through, print exactly the one JSON line rank 0 prints, hand everything else to stderr, and return the
    children's exit code (non-zero if any rank failed, or if no JSON line came back)."""
    import socket
    import subprocess
    with socket.socket() as s:                             # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env["OISAT_BENCH_SELF_LAUNCHED"] = "1"
    cmd = launch_command(ngpus, argv, port)
    print("bench.py: launching %d ranks: %s" % (ngpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for line in proc.stdout:                               # rank 0's JSON line is the only thing this script prints to stdout
        text = line.strip()
        if text.startswith("{") and text.endswith("}"):
            lines.append(text)
        elif text:
            print(text, file=sys.stderr, flush=True)
    rc = proc.wait()
    if lines:
        print(lines[-1], flush=True)
    if rc == 0 and not lines:
        print("bench.py: the ranks exited cleanly but printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def launcher_selftest(args, world, rank):
    """CPU-only check of the launch plumbing (tests/test_host_logic_cpu.py): every rank joins a gloo group, the ranks are
    counted with an all-reduce, rank 0 prints ONE JSON line, and rank `--launcher-selftest` % world exits with that code."""
    import torch
    import torch.distributed as dist
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "world_size": world, "n_ranks_seen_by_backend": seen, "gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "self_launched": os.environ.get("OISAT_BENCH_SELF_LAUNCHED") == "1"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    code = args.launcher_selftest
    return code if code and rank == code % world else 0


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not launched and args.gpus > 1:                     # plain `python bench.py --gpus N`: this process only launches
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if launched and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} under a launcher that started {world} rank(s): the two must agree")
    if args.launcher_selftest is not None:
        sys.exit(launcher_selftest(args, world, rank))
    if args.rehearse_on_device0:
        local = 0
    os.environ["OISAT_DEVICE"] = str(local)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    from oisatgmi import _hip, synthetic as syn, dense, parallel
    ctx = _hip.context()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    sync = torch.cuda.synchronize
    barrier = dist.barrier if world > 1 else None
    ranks_seen, devices_seen = 1, [local]
    if world > 1:
        torch.zeros(1, device="cuda").add_(1)            # make sure the device context exists before the collectives
        # what the BACKEND sees, not what the launcher was asked for: ranks counted by an all-reduce over the data-path group,
        # and which device every rank sits on (N distinct devices unless --rehearse-on-device0)
        one = torch.ones(1, device="cuda", dtype=torch.float64) if args.backend != "gloo" else torch.ones(1, dtype=torch.float64)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        mine = torch.tensor([local], dtype=torch.int64, device="cuda" if args.backend != "gloo" else "cpu")
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        devices_seen = [int(g.item()) for g in got]

    ny, nx, nobs, L, swaths, refine = WORKLOADS[args.workload]
    if args.refine is not None:
        refine = args.refine
    n = ny * nx
    if args.tier_a_only:
        print(json.dumps({"tier_a": tier_a_leg(ctx, ny, nx, nobs, sync), "tier_a_kernels": tier_a_kernels_leg(ctx, sync)}))
        return
    if args.only:
        if world != 1:
            raise SystemExit("--only is a single-GPU profiling aid")
        res = {}
        for leg in args.only.split(","):
            if leg == "tiled":
                res[leg] = tiled_leg(ctx, args.workload, sync)
            elif leg == "secondary":
                ny2, nx2, _, L2, _, r2 = WORKLOADS[SECONDARY]
                plan2 = make_plan(ctx, SECONDARY, 4000)
                plan2.run(L2, refine=r2, check_pd=True)
                el2 = time_steps(lambda: plan2.run(L2, refine=r2), 20, 3, sync)
                el1 = time_steps(lambda: plan2.run(L2, refine=1), 20, 3, sync)
                roof2, per2 = roofline_leg(ctx, plan2, L2, r2, 5)
                res[leg] = {"ms_per_step": 1e3 * el2 / 20, "ms_per_step_refine1": 1e3 * el1 / 20, "refine": r2, "obs": plan2.m,
                            "roofline": roof2, "kernel_ms_per_step": per2}
                del plan2
            elif leg == "pcie":
                res[leg] = pcie_leg(sync)
            elif leg == "regrid":
                res[leg] = regrid_leg(ctx, sync)
            elif leg == "config4":
                lat2, lon2 = syn.global_grid(ny, nx)
                res[leg] = config4_leg(ctx, args, 1, 0, local, lat2, lon2, sync, None)
            elif leg == "config5":
                lat2, lon2 = syn.global_grid(ny, nx)
                res[leg] = config5_leg(ctx, args, 1, 0, local, lat2, lon2, sync, None)
            else:
                raise SystemExit(f"--only: unknown leg {leg!r}")
        print(json.dumps(res))
        return
    # ---- shared grid: built on rank 0, broadcast once (RCCL) -------------------------------------
    lat2, lon2 = syn.global_grid(ny, nx)
    if args.c4_shards:
        if world != 1:
            raise SystemExit("--c4-shards is a single-GPU emulation")
        worlds = sorted({1} | {int(w) for w in args.c4_shards.split(",")})
        print(json.dumps({"config4_shards_emulated": config4_shards_leg(ctx, args, lat2, lon2, sync, worlds)}))
        return
    if world > 1:
        lat2, lon2 = parallel.broadcast_grid(lat2 if rank == 0 else None, lon2 if rank == 0 else None, (ny, nx), local)
    # ---- this rank's month ------------------------------------------------------------------------
    plan = make_plan(ctx, args.workload, 4000 + rank, lat2, lon2, args.species)
    m = plan.m
    gather = parallel.FieldGather(plan, world, rank, local) if world > 1 else None

    def step():
        plan.run(L, refine=refine)
        if gather is not None:
            gather.run()

    resid = plan.run(L, refine=refine, check_pd=True, want_resid=True)      # checked pass, untimed
    elapsed = time_steps(step, args.steps, args.warmup, sync, barrier)
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        flops = dense.DenseAnalysis.flops(m)
        out = {
            "metric": "analysed grid-cells/s (dense Gaussian-B OI: B build + Kalman-gain solve + increment) "
                      "and Kalman-gain solve TFLOP/s vs MI355X fp32 MFMA roofline",
            "value": world * n * args.steps / elapsed,
            "unit": "grid-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "species": args.species, "grid": [ny, nx], "obs_per_month": m, "corr_length_km": L,
                       "refine": refine, "months_per_step": world,
                       "parallelism": "one month per GPU; RCCL broadcast of the grid, gather of the fields to rank 0"},
            "solve_tflops_end_to_end": world * flops / (elapsed / args.steps) / 1e12,
            "refinement_residuals": resid,
            "n_ranks_seen_by_backend": ranks_seen,
            "backend": {"name": dist.get_backend() if world > 1 else None, "world_size": dist.get_world_size() if world > 1 else 1,
                        "device_of_rank": devices_seen, "self_launched": os.environ.get("OISAT_BENCH_SELF_LAUNCHED") == "1"},
            "parity_anchor": "dense Gaussian-B path: no reference code exists (SURVEY 8 a-ext); oracle = float64 SciPy Cholesky of the "
                             "same definition, reference-anchored only in the L -> 0 limit against OI() golden outputs (DESIGN section 2)",
        }
    if rank == 0 and world > 1 and not args.no_roofline:
        # the dominant kernel's roofline on rank 0 of a multi-GPU run too (the other ranks wait at the next leg's barrier): the
        # launch is per GPU, so the fraction is this GPU's -- the other legs below stay single-GPU extras
        out["roofline"], out["kernel_ms_per_step"] = roofline_leg(ctx, plan, L, refine, 2 if m > 30000 else 5)
    if rank == 0 and world == 1:
        if not args.no_roofline:
            out["roofline"], out["kernel_ms_per_step"] = roofline_leg(ctx, plan, L, refine, 2 if m > 30000 else 5)
        del plan
        if not args.no_secondary and args.workload != SECONDARY:
            ny2, nx2, _, L2, _, r2 = WORKLOADS[SECONDARY]
            plan2 = make_plan(ctx, SECONDARY, 4000)
            plan2.run(L2, refine=r2, check_pd=True)
            el2 = time_steps(lambda: plan2.run(L2, refine=r2), 20, 3, sync)
            roof2, per2 = roofline_leg(ctx, plan2, L2, r2, 5)
            # several independent months in flight on one GPU (one handle + stream each): a 10^4-observation
            # solve is a chain of small launches that leaves most CUs idle, months are independent work units
            lanes = [_hip.Context(ctx.device).own_stream() for _ in range(7)]
            plans = [plan2] + [make_plan(l, SECONDARY, 4001 + i) for i, l in enumerate(lanes)]
            for pl in plans:
                pl.run(L2, refine=r2)

            def months_in_flight():
                for pl in plans:
                    pl.run(L2, refine=r2)
                for l in lanes:
                    l.sync()
            el4 = time_steps(months_in_flight, 10, 2, sync)
            conc = {"months_in_flight": len(plans), "value": len(plans) * ny2 * nx2 * 10 / el4, "unit": "grid-cells/s",
                    "ms_per_month": 1e3 * el4 / (10 * len(plans))}
            del plans
            # the same eight months with their factorizations advanced in LOCK-STEP (oisat_batch_potrf: one launch per
            # recursion node for all eight) between a per-lane build phase and a per-lane solve phase
            blanes = [ctx] + lanes
            bplans = []
            for i, l in enumerate(blanes):
                pb, cellb, _, _ = build_case(SECONDARY, 4000 + i)
                q = dense.DenseAnalysis(pb.lat, pb.lon, max_obs=int(pb.obs_y.size), dtype=np.float32, ctx=l, batched=True)
                q.load_background(pb.Xa, pb.Sa)
                q.load_obs(pb.obs_lat, pb.obs_lon, cellb, np.where(pb.obs_y < 0, 0, pb.obs_y), pb.obs_var)
                bplans.append(q)
            bf = dense.BatchedFactor(ctx.device, bplans)

            bpool = dense.LanePool.from_lanes(blanes)               # the eight handles above as a pool (no new streams)

            def months_batched():
                for q in bplans:
                    q.run_build(L2)
                bf.run(bpool, [[q] for q in bplans], r2)
                for l in blanes + bf.ctxs:               # the lock-step solves run on the group streams
                    l.sync()
            months_batched()
            for q in bplans:
                q.check()
            bf.check()
            el5 = time_steps(months_batched, 10, 2, sync)
            conc["batched"] = {"months": len(bplans), "value": len(bplans) * ny2 * nx2 * 10 / el5, "unit": "grid-cells/s",
                               "ms_per_month": 1e3 * el5 / (10 * len(bplans))}
            bf.close()
            del bplans
            for l in lanes:
                l.close()
            out["secondary"] = {"workload": SECONDARY, "value": ny2 * nx2 * 20 / el2, "unit": "grid-cells/s",
                                "concurrent": conc,
                                "ms_per_step": 1e3 * el2 / 20, "obs_per_month": plan2.m,
                                "solve_tflops_end_to_end": dense.DenseAnalysis.flops(plan2.m) / (el2 / 20) / 1e12,
                                "roofline": roof2, "kernel_ms_per_step": per2}
            del plan2
        if not args.no_secondary and args.workload == DEFAULT and args.species == "NO2":
            # BASELINE configs[4]: the other two species' parameter sets (control_omihcho.yml / control_omio3.yml shapes: value
            # ranges, ctm_error, observation-error model) at the same full size -- one checked and two timed analyses each
            sp_out = {}
            for sp in ("HCHO", "O3"):
                pl = make_plan(ctx, args.workload, 4000, lat2, lon2, sp)
                rs = pl.run(L, refine=refine, check_pd=True, want_resid=True)
                el = time_steps(lambda: pl.run(L, refine=refine), 2, 0, sync)
                pl.check()
                sp_out[sp] = {"ms_per_step": 1e3 * el / 2, "value": n * 2 / el, "unit": "grid-cells/s", "obs": pl.m,
                              "refinement_residuals": rs,
                              "solve_tflops_end_to_end": dense.DenseAnalysis.flops(pl.m) / (el / 2) / 1e12}
                del pl
            out["config5_species"] = sp_out
        if not args.no_secondary:
            out["tiled"] = tiled_leg(ctx, args.workload, sync)
            out["tier_a"] = tier_a_leg(ctx, ny, nx, nobs, sync)
            out["tier_a_kernels"] = tier_a_kernels_leg(ctx, sync)
            out["pcie_inclusive"] = pcie_leg(sync)
            out["regrid"] = regrid_leg(ctx, sync)
    if not args.no_config4:
        del gather
        if world > 1:
            del plan                                       # its 40 GB factor goes back before the tile lanes allocate theirs
        c4 = config4_leg(ctx, args, world, rank, local, lat2, lon2, sync, barrier)
        c5 = config5_leg(ctx, args, world, rank, local, lat2, lon2, sync, barrier)
        if rank == 0:
            out["config4_strong"] = c4
            out["config5_strong"] = c5
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # last: its BLAS threads keep spinning for a while
        out["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
