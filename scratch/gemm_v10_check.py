import sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import torch
from oisatgmi import _hip
ctx = _hip.context(); plib = ctx.lib
lib = C.CDLL("/root/repo/scratch/libgemm_v10.so")
lib.gemm_v10.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int]
torch.manual_seed(0)
for (M, N, K, lower) in ((1152, 640, 256, 1), (1152, 640, 256, 0), (640, 640, 256, 1), (256, 128, 32, 0), (256, 256, 64, 1), (1280, 1280, 128, 1)):
    A = torch.rand(M, K, device="cuda") * 2 - 1; B = torch.rand(N, K, device="cuda") * 2 - 1
    ref = -(A.double() @ B.double().T)
    if lower:
        ti = torch.arange(M, device="cuda")[:, None] // 128; tj = torch.arange(N, device="cuda")[None, :] // 128
        ref = torch.where(ti >= tj, ref, torch.zeros_like(ref))
    C1 = torch.zeros(M, N, device="cuda"); C2 = torch.zeros(M, N, device="cuda")
    ctx.check(plib.oisat_gemm_nt(ctx.h, C1.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, lower)); ctx.sync()
    lib.gemm_v10(C2.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, lower); lib.gemm_sync10()
    print((M, N, K, lower), "prod err %.2e" % float((C1.double() - ref).abs().max()), "v10 err %.2e" % float((C2.double() - ref).abs().max()))
