import sys, time, ctypes as C
import torch
lib = C.CDLL("scratch/libgemm_var.so")
lib.gemm_var.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int]
M = N = K = 8192
A = torch.rand(M, K, device="cuda") * 2 - 1; B = torch.rand(N, K, device="cuda") * 2 - 1; Cc = torch.zeros(M, N, device="cuda")
torch.cuda.synchronize()
ref = (A[:256].double() @ B[:256].double().T)
for v in (4, 5):
    Cc.zero_(); lib.gemm_var(v, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K); lib.gemm_sync()
    print("variant", v, "max err", float((Cc[:256, :256].double() + ref).abs().max()))
for v in (4, 5, 4, 5):
    for _ in range(2): lib.gemm_var(v, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
    lib.gemm_sync()
    t0 = time.perf_counter()
    for _ in range(5): lib.gemm_var(v, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K)
    lib.gemm_sync()
    dt = (time.perf_counter() - t0) / 5
    print("variant", v, f"{dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.1f} TF")
