"""fp32 GEMM at the shapes of the GAT configuration (config 3) and of wide SAGE layers: us per launch and TFLOP/s.
    python tools/exp_gemm_big.py        (HMP_GEMM_BIG=0 / HMP_GEMM_BK=64 select kernel variants)"""
import sys, torch
sys.path[:0] = ['/root/repo/hydra-gnn_amd']
from hydra_gnn_amd import _lib
lib = _lib.require_device()
dev = 'cuda:0'
def run(M, N, K, ta, tb, reps=200):
    A = torch.randn((K if ta else M), (M if ta else K), device=dev)
    B = torch.randn((N if tb else K), (K if tb else N), device=dev)
    Cc = torch.empty(M, N, device=dev)
    st = _lib.stream_ptr()
    for _ in range(5):
        lib.hmp_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, Cc.data_ptr(), N, M, N, K, st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hmp_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, Cc.data_ptr(), N, M, N, K, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    form = "NT" if (not ta and tb) else "NN" if (not ta and not tb) else "TN"
    print(f"{form} M={M:6d} N={N:5d} K={K:5d}: {us:8.2f} us/launch  {2 * M * N * K / us / 1e6:7.2f} TFLOP/s", flush=True)
    if M * N * K < 2e9:
        ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
        err = (Cc.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, err
run(5490, 1536, 512, 0, 1)    # GAT projection, layers 1-2 (objects)
run(5490, 1536, 306, 0, 1)    # layer 0
run(5490, 512, 1536, 0, 0)    # input gradient
run(1536, 512, 5490, 1, 0)    # weight gradient (unit op: no split-K)
run(16384, 1024, 1024, 0, 1, reps=50)
run(140000, 768, 256, 0, 1, reps=20)
