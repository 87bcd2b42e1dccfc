import sys, time, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import torch
from oisatgmi import _hip
ctx = _hip.context(); plib = ctx.lib
lib = C.CDLL("/root/repo/scratch/libgemm_v10.so")
lib.gemm_v10.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]
def tiles(M, N, lower):
    return (M // 128) * (N // 128) if not lower else (N // 128) * (M // 128) - (N // 128) * (N // 128 - 1) // 2
for (M, N, K, lower) in ((8192, 8192, 8192, 0), (16384, 16384, 2048, 1), (4992, 4992, 4992, 1), (8192, 8192, 4096, 1), (1152, 640, 256, 1)):
    A = torch.rand(M, K, device="cuda") * 2 - 1; B = torch.rand(N, K, device="cuda") * 2 - 1
    outs = {}
    for name in ("prod", "v10"):
        Cc = torch.zeros(M, N, device="cuda")
        def run():
            if name == "prod":
                ctx.check(plib.oisat_gemm_nt(ctx.h, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, lower))
            else:
                rc = lib.gemm_v10(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, lower); assert rc == 0, rc
        run(); torch.cuda.synchronize(); ctx.sync(); lib.gemm_sync10()
        outs[name] = Cc.clone()
        for _ in range(2): run()
        ctx.sync(); lib.gemm_sync10()
        t0 = time.perf_counter()
        for _ in range(6): run()
        ctx.sync(); lib.gemm_sync10()
        dt = (time.perf_counter() - t0) / 6
        print(f"{name} M={M} N={N} K={K} lower={lower}: {dt*1e3:.3f} ms {2.0*tiles(M,N,lower)*128*128*K/dt/1e12:.1f} TF", flush=True)
    print("   max |v10 - prod| = %.3e" % float((outs["v10"] - outs["prod"]).abs().max()))
