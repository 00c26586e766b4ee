#!/usr/bin/env python3
"""Context for the conv roofline fraction: what the vendor's fp32 GEMM (rocBLAS / hipBLASLt through torch.matmul,
TF32 off) sustains on this box for GEMM shapes like the dominant conv layers."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda:0')
for m, n, k in ((8192, 8192, 8192), (1036800, 128, 1600), (1044480, 128, 1152), (259200, 128, 1152), (65280, 128, 1152)):
    a = torch.randn(m, k, device=dev)
    b = torch.randn(k, n, device=dev)
    for _ in range(2):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('sgemm %8d x %4d x %5d : %8.3f ms  %6.1f TFLOP/s' % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9))
