"""Time nphm_dense_gemm_nt over shapes (dev tool, GPU box): T(M, N, K) per epilogue - fixed cost per launch / tile against the K loop."""
import sys, torch
sys.path.insert(0, ".")
from nphm_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
def t(M, N, K, epi, reps=20):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); E = torch.randn(1, N, device=dev)
    f = lambda: _lib.check(lib.nphm_dense_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, E.data_ptr() if epi else None, M, 1.0, 100.0, epi, 1, None), "g")
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M in (3200, 32000, 128000):
    for N in (128, 512):
        row = []
        for K in (32, 64, 256, 512, 1024, 2048):
            row.append("%6.1f" % t(M, N, K, 0))
        print(f"M {M:6d} N {N:4d} epi 0: K = 32 64 256 512 1024 2048 ->", " ".join(row), "us;  epi 1 @512: %.1f" % t(M, N, 512, 1))
