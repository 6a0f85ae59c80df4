"""Reference point only (not a runtime backend): what torch.matmul (hipBLASLt / rocBLAS) reaches on the bench GEMM shapes on this
box, next to our kernel, interleaved in one process on the same random data."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from versecrafter_amd import ops

def timeit(fn, rounds=5, inner=2):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    return statistics.median(ts)

g = torch.Generator(device="cuda").manual_seed(0)
M = 65520
for N, K in ((5120, 5120), (13824, 5120), (5120, 13824)):
    a = torch.randn((M + 255) // 256 * 256, K, device="cuda", generator=g).bfloat16()[:M]   # rows readable to the next 256
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t_ours = timeit(lambda: ops.gemm(a, w, bias, out=out, tile=4))   # ping-pong kernel (a is over-allocated below)
    t_vend = timeit(lambda: F.linear(a, w, bias))
    print(f"M={M} N={N} K={K}: ours {fl / t_ours / 1e9:.0f} TF   torch F.linear {fl / t_vend / 1e9:.0f} TF", flush=True)
if "nosdpa" in sys.argv:
    sys.exit(0)
B, H, L = 2, 40, 32760
q = torch.randn(B, H, L, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(B, H, L, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(B, H, L, 128, device="cuda", generator=g).bfloat16()
try:
    t = timeit(lambda: F.scaled_dot_product_attention(q, k, v), rounds=3, inner=1)
    print(f"torch SDPA B={B} H={H} L={L}: {4.0 * B * H * L * L * 128 / t / 1e9:.0f} TF", flush=True)
except Exception as e:
    print("SDPA failed:", str(e)[:200])
