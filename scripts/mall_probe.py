"""Does a producer -> consumer hand-off through the 256 MB Infinity Cache beat HBM?  Copy kernels over buffers of
16 MB .. 4 GB: effective GB/s (read + write) per size.  If the small sizes run far above the ~6.3 TB/s HBM copy rate,
chunked producer/consumer pipelines could hide re-reads in the cache; if not, only removing passes helps."""
import torch

dev = "cuda"
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 4096):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    reps = max(5, min(200, 20000 // mb))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
        a.copy_(b)  # consumer of what was just written
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * reps)
    print(f"{mb:5d} MB  copy {ms*1e3:8.1f} us  {2 * mb / 1024 / (ms * 1e-3):8.1f} GB/s (read+write)")
