"""bf16-A NT GEMM at the shape of config 5's projections: weight-stationary kernel vs the tiled kernel (HMP_GEMM_WS=0).
    python tools/exp_gemm_ws.py"""
import sys, torch
sys.path[:0] = ['/root/repo/hydra-gnn_amd']
from hydra_gnn_amd import _lib
lib = _lib.require_device()
dev = 'cuda:0'
import os
def run(M, N, K, c16, reps=20, check=True):
    check = check and not os.environ.get('HMP_WS_DBG')
    torch.manual_seed(0)
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev) * 0.1
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    st = _lib.stream_ptr()
    call = lambda: _lib.check(lib.hmp_gemm_bf16_a16(A.data_ptr(), K, W.data_ptr(), K, Cc.data_ptr(), N, 1 if c16 else 0, M, N, K, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gb = (M * K * 2 + M * N * (2 if c16 else 4)) / 1e9
    msg = f"M={M:8d} N={N:4d} K={K:4d} C={'bf16' if c16 else 'fp32'}: {us:9.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s  {gb / us * 1e6:7.1f} GB/s (A + C once)"
    if check:
        rows = torch.cat([torch.arange(0, min(M, 4096)), torch.arange(max(M - 4096, 0), M)]).to(dev)
        ref = A[rows].double() @ W.to(torch.bfloat16).double().t()
        got = Cc[rows].double()
        err = (got - ref).abs().max().item()
        tol = 2e-2 if c16 else 2e-4
        assert err < tol * max(ref.abs().max().item(), 1.0), err
        msg += f"  max err {err:.2e}"
    print(msg, flush=True)
def run32(M, N, K, reps=10):
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev) * 0.5
    W = torch.randn(N, K, device=dev) * 0.1
    Cc = torch.empty(M, N, device=dev)
    st = _lib.stream_ptr()
    call = lambda: _lib.check(lib.hmp_gemm_bf16(A.data_ptr(), K, 0, W.data_ptr(), K, 1, Cc.data_ptr(), N, M, N, K, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    rows = torch.arange(M - 4096, M, device=dev)
    ref = A[rows].to(torch.bfloat16).double() @ W.to(torch.bfloat16).double().t()
    err = (Cc[rows].double() - ref).abs().max().item()
    print(f"fp32 A: M={M:8d} N={N:4d} K={K:4d} C=fp32: {us:9.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s  max err {err:.2e}", flush=True)
run32(1_000_000, 768, 256)
run(1_000_000, 768, 256, True)
run(1_000_000, 768, 256, False, reps=10)
run(1_000_000, 256, 256, True)
run(100_000, 768, 256, True)
run(1_000_003, 700, 128, True)
