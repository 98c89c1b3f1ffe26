"""What HBM gives a kernel that reads and writes in equal parts (k_recon's mix), against pure reads / writes:
torch device-to-device copy, fill and a reduction over a 4 GiB tensor."""
import torch
n = 1 << 32
x = torch.empty(n, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
x.fill_(1); torch.cuda.synchronize()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: y.copy_(x)); print("copy   4 GiB -> %.2f ms, %.2f TB/s of traffic (read + write)" % (ms, 2 * n / ms / 1e9))
ms = t(lambda: y.fill_(3)); print("fill   4 GiB -> %.2f ms, %.2f TB/s written" % (ms, n / ms / 1e9))
xi = x.view(torch.int64)
ms = t(lambda: xi.sum()); print("reduce 4 GiB -> %.2f ms, %.2f TB/s read" % (ms, n / ms / 1e9))
