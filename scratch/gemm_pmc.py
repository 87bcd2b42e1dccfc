import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import torch
from oisatgmi import _hip
ctx = _hip.context(); lib = ctx.lib
M = N = K = 8192
A = torch.rand(M, K, device='cuda') * 2 - 1; B = torch.rand(N, K, device='cuda') * 2 - 1; Cc = torch.zeros(M, N, device='cuda')
for _ in range(6):
    ctx.check(lib.oisat_gemm_nt(ctx.h, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, 0))
ctx.sync(); torch.cuda.synchronize()
