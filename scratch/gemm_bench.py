import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np, ctypes as C
from oisatgmi import _hip
ctx = _hip.context(); lib = ctx.lib
rng = np.random.default_rng(0)
# correctness on a small asymmetric case
M, N, K = 256, 384, 160
A = rng.uniform(-1, 1, (M, K)).astype(np.float32); B = rng.uniform(-1, 1, (N, K)).astype(np.float32)
Cm = rng.uniform(-1, 1, (M, N)).astype(np.float32)
a, b, c = ctx.upload(A), ctx.upload(B), ctx.upload(Cm)
ctx.check(lib.oisat_gemm_nt(ctx.h, c.ptr, N, a.ptr, K, b.ptr, K, M, N, K, 0, 0))
out = ctx.download(c.ptr, (M, N), np.float32)
ref = Cm.astype(np.float64) - A.astype(np.float64) @ B.astype(np.float64).T
print('gemm max err', np.abs(out - ref).max())
import torch
for (M, N, K, lower) in [(4096, 4096, 4096, 0), (8192, 8192, 8192, 0), (8192, 8192, 4096, 1), (4992, 4992, 4992, 1), (9984, 128, 128, 0), (5120, 2560, 2560, 1), (16384, 16384, 512, 1), (16384, 16384, 2048, 1)]:
    A = torch.rand(M, K, device='cuda') * 2 - 1; B = torch.rand(N, K, device='cuda') * 2 - 1; Cc = torch.zeros(M, N, device='cuda')
    def run():
        ctx.check(lib.oisat_gemm_nt(ctx.h, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, lower))
    for _ in range(3): run()
    torch.cuda.synchronize(); ctx.sync()
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps): run()
    ctx.sync(); dt = (time.perf_counter() - t0) / reps
    ntiles = (M // 128) * (N // 128) if not lower else (N // 128) * (M // 128) - (N // 128) * (N // 128 - 1) // 2
    fl = 2.0 * ntiles * 128 * 128 * K
    print(f'M={M} N={N} K={K} lower={lower}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF  tiles={ntiles}')
