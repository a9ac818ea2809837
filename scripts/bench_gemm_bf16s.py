"""ait_gemm_bf16s on the AIT's linear shapes at cfg5 size (8 x 512 proposals: 262144 token rows) and cfg2 size, f32 and
bf16 outputs, against the f32-storage kernels in their bf16-products mode (ops.set_matmul_dtype('bf16')): TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import _lab_lib  # noqa: F401  (AIT_LAB_LIB=<name>: a lab build of the library)
from ait_amd import ops


def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M in (int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 76800):
    for name, N, K in (("qkv", 1536, 512), ("ffn1", 2048, 512), ("ffn2", 512, 2048), ("emb", 512, 1024), ("trans", 1024, 512),
                       ("q/kv", 512, 512), ("fc", 512, 64)):
        a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        a16, b16 = ops.to_bf16(a), ops.to_bf16(b)
        o32, o16 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t32 = timeit(lambda: ops.gemm_bf16s(a16, b16, bias=bias, out32=o32))
        t16 = timeit(lambda: ops.gemm_bf16s(a16, b16, bias=bias, relu=True, out16=o16, want32=False))
        tb = timeit(lambda: ops.gemm_bf16s(a16, b16, bias=bias, out32=o32, out16=o16))
        tc = timeit(lambda: ops.to_bf16(a, out=a16))
        ops.set_matmul_dtype("bf16")
        told = timeit(lambda: ops.gemm(a, b, bias=bias, out=o32))
        ops.set_matmul_dtype("f32")
        print("%-5s M=%6d N=%4d K=%4d | bf16s f32-out %6.1f  bf16-out %6.1f  both %6.1f TF/s | f32-storage bf16 products %6.1f | "
              "convert A %.3f ms (%.0f GB/s)" % (name, M, N, K, fl / t32 / 1e9, fl / t16 / 1e9, fl / tb / 1e9, fl / told / 1e9, tc,
                                                 M * K * 6 / tc / 1e6), flush=True)
        del a, b, a16, b16, o32, o16

print("weight-gradient products (bf16 operands, reduction outermost, split-K atomics) vs the f32-storage kernels in bf16-products mode")
from ait_amd.system import _wgrad
for R in (262144, 76800):
    for name, Mo, No in (("dW2", 512, 2048), ("dW1", 2048, 512), ("dWqkv", 1536, 512)):
        dy, x = torch.randn(R, Mo, device="cuda"), torch.randn(R, No, device="cuda")
        dy16, x16 = dy.to(torch.bfloat16), x.to(torch.bfloat16)
        out = torch.zeros(Mo, No, device="cuda")
        fl = 2.0 * R * Mo * No
        best = None
        for sk in (8, 16, 32, 64):
            if R % sk or (R // sk) % 32:
                continue
            t = timeit(lambda: ops.gemm_bf16s_tn(dy16, x16, out=out, split_k=sk))
            best = (t, sk) if best is None or t < best[0] else best
        ops.set_matmul_dtype("bf16")
        told = timeit(lambda: _wgrad(dy, x))
        ops.set_matmul_dtype("f32")
        print("%-6s R=%6d %4d x %4d | bf16s-tn %6.1f TF/s (split %d) | f32-storage bf16 products %6.1f" %
              (name, R, Mo, No, fl / best[0] / 1e9, best[1], fl / told / 1e9), flush=True)
