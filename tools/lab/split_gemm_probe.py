"""lab: is ONE bf16 library GEMM over K-concatenated split pieces faster than the fp32 library GEMM?
C = X W^T with X = X1+X2+X3, W = W1+W2+W3 (bf16 pieces): the six products with p+q <= 2 as
[X3|X2|X1|X2|X1|X1] [W1|W2|W3|W1|W2|W1]^T, fp32 accumulate and output."""
import sys, time, torch
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (196608, 500, 1000)
dev = "cuda"
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
def split3(a):
    p1 = a.to(torch.bfloat16); r = a - p1.float()
    p2 = r.to(torch.bfloat16); r = r - p2.float()
    return p1, p2, r.to(torch.bfloat16)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
ref = X.double() @ W.double().T
y32 = X @ W.T
t32 = timeit(lambda: X @ W.T)
x1, x2, x3 = split3(X); w1, w2, w3 = split3(W)
Xc = torch.cat([x3, x2, x1, x2, x1, x1], 1).contiguous(); Wc = torch.cat([w1, w2, w3, w1, w2, w1], 1).contiguous()
try:
    f = lambda: torch.mm(Xc, Wc.T, out_dtype=torch.float32)
    y = f()
except Exception as e:
    print("torch.mm out_dtype not available:", type(e).__name__, str(e)[:200]); sys.exit(0)
tb = timeit(f)
tsplit = timeit(lambda: split3(X))
scale = ref.abs().max()
print(f"M={M} K={K} N={N}: fp32 GEMM {t32:.3f} ms ({2*M*K*N/t32/1e9:.0f} TFLOP/s), bf16 GEMM over 6K {tb:.3f} ms ({12*M*K*N/tb/1e9:.0f} bf16-TFLOP/s), "
      f"torch-op split of X {tsplit:.3f} ms")
print(f"max err / max |ref|: fp32 GEMM {float((y32.double()-ref).abs().max()/scale):.3e}, split GEMM {float((y.double()-ref).abs().max()/scale):.3e}")
