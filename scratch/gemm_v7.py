import sys, time, ctypes as C
import torch
lib = C.CDLL("scratch/libgemm_v7.so")
lib.gemm_v6.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]
for (M, N, K) in ((8192, 8192, 8192), (4096, 4096, 4096), (16384, 16384, 2048)):
    A = torch.rand(M, K, device="cuda") * 2 - 1; B = torch.rand(N, K, device="cuda") * 2 - 1; Cc = torch.zeros(M, N, device="cuda")
    torch.cuda.synchronize()
    lib.gemm_v6(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K); lib.gemm_sync6()
    ref = (A[:256].double() @ B[:256].double().T)
    print("max err", float((Cc[:256, :256].double() + ref).abs().max()))
    for _ in range(2): lib.gemm_v6(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
    lib.gemm_sync6()
    t0 = time.perf_counter()
    for _ in range(5): lib.gemm_v6(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
    lib.gemm_sync6()
    dt = (time.perf_counter() - t0) / 5
    print(f"v7 M={M} N={N} K={K}: {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.1f} TF")
