import sys, time, ctypes as C, torch
v = sys.argv[1]
lib = C.CDLL(f"/root/repo/scratch/libgemm_v10_{v}.so")
lib.gemm_v10.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]
M = N = K = 8192
A = torch.rand(M, K, device="cuda") * 2 - 1; B = torch.rand(N, K, device="cuda") * 2 - 1; Cc = torch.zeros(M, N, device="cuda")
torch.cuda.synchronize()
lib.gemm_v10(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0); lib.gemm_sync10()
ref = (A[:128].double() @ B[:128].double().T)
err = float((Cc[:128, :128].double() + ref).abs().max())
for _ in range(2): lib.gemm_v10(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0)
lib.gemm_sync10()
t0 = time.perf_counter()
for _ in range(5): lib.gemm_v10(Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0)
lib.gemm_sync10()
dt = (time.perf_counter() - t0) / 5
print(f"swz {v}: err {err:.2e}  {dt*1e3:.3f} ms {2*M*N*K/dt/1e12:.1f} TF")
