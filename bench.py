#!/usr/bin/env python3
"""Headline benchmark of the MI355X optimal-interpolation hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full dense Gaussian-B analysis (innovation, covariance build, MFMA Cholesky,
gain solve with float64-residual refinement, increment over every grid cell) of one synthetic
month on one GPU, with the gridded background and the observations already resident in HBM.
N = 1 workload = BASELINE.json configs[1]: 360x720 grid, 10^4 random observations, full B build +
gain solve.  With N > 1 every rank analyses its own month (months are independent work units, as in
the reference's one-job-per-month launch, run/job_submitter_sbatch.py:45-68): the shared grid is
broadcast once from rank 0 and the analysis fields are gathered to rank 0 every step over RCCL.

Prints ONE JSON line (rank 0).  Extra legs, N = 1 only: per-kernel HIP-event timing for the
roofline object and a bounded CPU run of the float64 oracle for cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md, HBM3E spec

WORKLOADS = {
    # name: (ny, nx, nobs, L_km, swaths)
    "config2_360x720_1e4obs": (360, 720, 10000, 500.0, False),
    "config1_72x144_1e3obs": (72, 144, 1000, 500.0, False),
    "config3_720x1440_1e5obs_global": (720, 1440, 100000, 300.0, True),
    "mid_360x720_3e4obs": (360, 720, 30000, 400.0, False),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config2_360x720_1e4obs", choices=sorted(WORKLOADS))
    ap.add_argument("--refine", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(workload):
    """Time the float64 oracle (oracle/oi_oracle.py dense_oi) on a bounded sample of the workload:
    a subset of the observations and of the grid cells, sized for ~10-30 s of CPU work."""
    from oracle import oi_oracle as orc
    from oisatgmi import synthetic as syn, dense
    ny, nx, nobs, L, swaths = WORKLOADS[workload]
    m_s = min(nobs, 4000)
    p = syn.point_obs_case(ny, nx, m_s, 424242, swaths=swaths)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    ncell_s = min(p.Xa.size, 65536)
    sel = np.random.default_rng(1).choice(p.Xa.size, ncell_s, replace=False)
    lat_s, lon_s = p.lat.ravel()[sel], p.lon.ravel()[sel]
    # observations keep pointing at real cells of the sampled grid: remap through a lookup
    lut = -np.ones(p.Xa.size, dtype=np.int64)
    lut[sel] = np.arange(ncell_s)
    keep = lut[cell] >= 0
    if keep.sum() < 16:                      # make sure the sampled cells include the observed ones
        sel[:cell.size] = cell
        lat_s, lon_s = p.lat.ravel()[sel], p.lon.ravel()[sel]
        lut[:] = -1
        lut[sel] = np.arange(ncell_s)
        keep = lut[cell] >= 0
    t0 = time.perf_counter()
    orc.dense_oi(lat_s, lon_s, p.Xa.ravel()[sel], p.Sa.ravel()[sel], p.obs_lat[keep], p.obs_lon[keep], lut[cell[keep]],
                 np.where(p.obs_y[keep] < 0, 0, p.obs_y[keep]), p.obs_var[keep], L)
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        thr = max((i.get("num_threads", 1) for i in threadpoolctl.threadpool_info()), default=1)
    except Exception:
        thr = os.cpu_count() or 1
    return {"value": ncell_s / dt, "unit": "grid-cells/s", "cores": int(thr), "kind": "port",
            "sample": f"oracle dense_oi (float64 NumPy/SciPy) on {ncell_s} cells x {int(keep.sum())} obs of {workload}, "
                      f"{dt:.1f} s; the full workload has {ny*nx} cells x {nobs} obs (Cholesky cost grows as obs^3)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    os.environ.setdefault("OISAT_DEVICE", str(local))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CUDA/HIP device visible); there is no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from oisatgmi import _hip, synthetic as syn, dense, parallel
    ctx = _hip.context()
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    ny, nx, nobs, L, swaths = WORKLOADS[args.workload]
    n = ny * nx
    # ---- shared grid: built on rank 0, broadcast once (RCCL) -------------------------------------
    lat2, lon2 = syn.global_grid(ny, nx)
    if world > 1:
        lat2, lon2 = parallel.broadcast_grid(lat2 if rank == 0 else None, lon2 if rank == 0 else None, (ny, nx), local)
    # ---- this rank's month ------------------------------------------------------------------------
    p = syn.point_obs_case(ny, nx, nobs, 4000 + rank, swaths=swaths)
    cell = dense.regular_grid_cell(lat2, lon2, p.obs_lat, p.obs_lon)
    m = int(p.obs_y.size)
    plan = dense.DenseAnalysis(lat2, lon2, max_obs=m, dtype=np.float32, ctx=ctx)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
    gather = parallel.FieldGather(plan, world, rank, local) if world > 1 else None

    def step():
        plan.run(L, refine=args.refine)
        if gather is not None:
            gather.run()

    # one checked pass: SPD + residual (outside the timed region)
    resid = plan.run(L, refine=args.refine, check_pd=True, want_resid=True)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        flops = dense.DenseAnalysis.flops(m)
        out = {
            "metric": "analysed grid-cells/s (dense Gaussian-B OI: B build + Kalman-gain solve + increment)",
            "value": world * n * args.steps / elapsed,
            "unit": "grid-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "grid": [ny, nx], "obs_per_month": m, "corr_length_km": L,
                       "refine": args.refine, "months_per_step": world,
                       "parallelism": "one month per GPU; RCCL broadcast of the grid, gather of fields"},
            "solve_tflops_end_to_end": world * flops / (elapsed / args.steps) / 1e12,
            "refinement_residuals": resid,
        }
    # ---- roofline leg (N = 1): per-kernel HIP-event timing on the launch stream --------------------
    if rank == 0 and world == 1 and not args.no_roofline:
        ctx.prof_reset()
        ctx.prof_enable(True)
        psteps = max(2, min(args.steps, 5))
        for _ in range(psteps):
            plan.run(L, refine=args.refine)
        prof = ctx.prof_collect()
        ctx.prof_enable(False)
        gemm_ms = sum(prof[k]["total_ms"] for k in ("syrk_gemm", "trsm_gemm") if k in prof) / psteps
        gemm_launches = sum(prof[k]["launches"] for k in ("syrk_gemm", "trsm_gemm") if k in prof) / psteps
        chol_flops = m ** 3 / 3.0
        achieved = chol_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        out["roofline"] = {
            "bound": "mfma", "kernel": "gemm_nt (syrk_gemm + trsm_gemm launches of one factorization)",
            "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
            "traffic": None,
            "algorithmic_flops_per_step": chol_flops, "kernel_ms_per_step": gemm_ms, "launches_per_step": gemm_launches,
        }
        out["kernel_ms_per_step"] = {k: v["total_ms"] / psteps for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
