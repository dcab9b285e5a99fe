"""What does the library F16 GEMM (hipBLASLt through torch.matmul) reach at the prefill shapes?  Y[T][N] = X[T][K] @ W[N][K]^T"""
import torch
torch.manual_seed(0)
dev = "cuda"
for T in (128, 256, 512, 1024, 2048, 4096):
    for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096)):
        x = torch.randn(T, K, device=dev, dtype=torch.float16) * 0.1
        w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
        for _ in range(3):
            y = x @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            y = x @ w.t()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("T=%5d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (T, N, K, us, 2.0 * T * N * K / us / 1e6), flush=True)
