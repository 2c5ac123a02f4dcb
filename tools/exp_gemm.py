import sys, time, torch
sys.path[:0] = ['/root/repo/hydra-gnn_amd']
from hydra_gnn_amd import _lib
lib = _lib.require_device()
dev = 'cuda:0'
def run(M,N,K,ta,tb,lda_pad=0,reps=2000):
    A = torch.randn((K if ta else M), (M if ta else K)+lda_pad, device=dev)
    B = torch.randn((N if tb else K), (K if tb else N), device=dev)
    Cc = torch.empty(M, N, device=dev)
    st = _lib.stream_ptr()
    for _ in range(20):
        lib.hmp_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, Cc.data_ptr(), N, M, N, K, st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hmp_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, Cc.data_ptr(), N, M, N, K, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/reps
    print(f"M={M} N={N} K={K} ta={ta} tb={tb} lda={A.stride(0)}: {us:.2f} us/launch  {2*M*N*K/us/1e6:.2f} TFLOP/s", flush=True)
run(2831,192,306,0,1)          # layer-0 projection, x ld=306
run(2831,192,306,0,1,lda_pad=2) # ld=308 (16B aligned rows)
run(2831,192,64,0,1)
run(2831,64,192,0,0)           # dX layer 1
run(192,307,2831,1,0)          # dW layer 0 (no split in unit op)
run(40000,256,256,0,1, reps=200)
run(200000,256,256,0,1, reps=50)
