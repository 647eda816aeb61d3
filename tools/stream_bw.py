"""What this box streams: a device-to-device copy, a read-only reduction and a write-only fill of 256 / 512 / 1024 MB through torch's
own kernels, HIP-event timed -- the practical roof the HBM-bound kernels (preprocess, geom_bwd, Adam) are read against.  (The first
read-only / write-only lines include those kernels' first-call set-up.)"""
import torch, time
dev = torch.device("cuda", 0)
for mb in (256, 512, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20): b.copy_(a)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    print(f"copy {mb} MB -> {mb} MB: {ms*1000:.1f} us, {2*mb/1024/ms*1000/1000:.2f} TB/s (read + write)")
    ev0.record()
    for _ in range(20): s = a.sum()
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    print(f"read-only sum {mb} MB: {ms*1000:.1f} us, {mb/1024/ms*1000/1000:.2f} TB/s")
    ev0.record()
    for _ in range(20): b.zero_()
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    print(f"write-only fill {mb} MB: {ms*1000:.1f} us, {mb/1024/ms*1000/1000:.2f} TB/s")
