import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np, torch
from oisatgmi import _hip, synthetic as syn
from oisatgmi.optimal_interpolation import DiagOI
ctx = _hip.context()
for (ny, nx, nobs) in ((72, 144, 1000), (360, 720, 10000), (720, 1440, 100000)):
    c = syn.diag_case(ny, nx, nobs, 3001)
    s = torch.cuda.Stream()
    ctx.set_stream(s.cuda_stream)
    d = DiagOI(ny * nx, dtype=np.float32, ctx=ctx)
    d.load(c.Xa, c.Y, c.Sa, c.So)
    for _ in range(5): d.run_fused(True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(200): d.run_fused(True)
    ctx.sync(); direct = (time.perf_counter() - t0) / 200
    idx0, _ = d.fused_result()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d.run_fused(True)
    ctx.set_stream(s.cuda_stream)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 200
    idx1, _ = d.fused_result()
    print(f"{ny}x{nx}: direct {direct*1e6:.1f} us/call, hipGraph replay {rep*1e6:.1f} us/call, knee index {idx0} / {idx1}")
