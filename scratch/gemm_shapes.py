# per-shape cost of the GEMM launches of one recursive Cholesky (nb diagonal blocks): which shapes lose the time?
import sys, time, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import torch
from oisatgmi import _hip
ctx = _hip.context(); lib = ctx.lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 79
calls = []
def rec(b0, b1):
    if b1 - b0 == 1:
        rows = nb - b0 - 1
        if rows > 0: calls.append(("trsm", rows * 128, 128, 128, 0))
        return
    mid = b0 + (b1 - b0 + 1) // 2
    rec(b0, mid)
    calls.append(("syrk", (nb - mid) * 128, (b1 - mid) * 128, (mid - b0) * 128, 1))
    rec(mid, b1)
rec(0, nb)
shapes = collections.Counter(calls)
S = torch.rand(nb * 128, nb * 128, device='cuda')
tot = 0.0; totf = 0.0
rows = []
for (kind, M, N, K, lower), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3]):
    A = torch.rand(M, K, device='cuda'); B = torch.rand(N, K, device='cuda'); Cc = torch.zeros(M, N, device='cuda')
    def run(): ctx.check(lib.oisat_gemm_nt(ctx.h, Cc.data_ptr(), N, A.data_ptr(), K, B.data_ptr(), K, M, N, K, 0, lower))
    for _ in range(3): run()
    torch.cuda.synchronize(); ctx.sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps): run()
    ctx.sync(); dt = (time.perf_counter() - t0) / reps
    nt = (M // 128) * (N // 128) - (N // 128) * (N // 128 - 1) // 2 if lower else (M // 128) * (N // 128)
    fl = 2.0 * nt * 128 * 128 * K
    tot += dt * cnt; totf += fl * cnt
    rows.append((kind, M // 128, N // 128, K // 128, nt, cnt, dt * 1e6, fl / dt / 1e12, dt * cnt * 1e3))
print("nb", nb, "total GEMM time %.3f ms, %.1f TF/s overall" % (tot * 1e3, totf / tot / 1e12))
print("kind  Mb  Nb  Kb tiles  x   us/launch   TF/s   total ms")
for r in sorted(rows, key=lambda r: -r[-1])[:28]:
    print("%-5s %3d %3d %3d %5d %3d %9.1f %7.1f %8.3f" % r)
by_k = collections.defaultdict(float)
for r in rows: by_k[(r[0], r[3])] += r[-1]
print({k: round(v, 3) for k, v in sorted(by_k.items(), key=lambda kv: -kv[1])})
