"""HBM bandwidth calibration with plain torch kernels: fill (write), sum (read), copy (read+write) at sizes around the
MLP's saved-activation buffers (536 MB) and below the 256 MB infinity cache."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for mb in (64, 268, 536, 2144):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    tw = timeit(lambda: a.fill_(1.0)); tr = timeit(lambda: a.sum()); tc = timeit(lambda: b.copy_(a))
    print(f"{mb:5d} MB: write {mb/tw/1e3:.2f} TB/s  read {mb/tr/1e3:.2f} TB/s  copy {2*mb/tc/1e3:.2f} TB/s (r+w)", flush=True)
