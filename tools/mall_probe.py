import torch, time
dev = torch.device("cuda")
def run(S_mb, chain=12, reps=8):
    S = S_mb << 20
    bufs = [torch.empty(S, dtype=torch.uint8, device=dev) for _ in range(chain + 1)]
    for b in bufs: b.fill_(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        for i in range(chain): bufs[i + 1].copy_(bufs[i])
    e0.record()
    for _ in range(reps):
        for i in range(chain): bufs[i + 1].copy_(bufs[i])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("chain copy  %5d MB per buffer: %.1f GB/s (read+write)  %.1f us per copy" % (S_mb, 2 * S * chain * reps / ms / 1e6, ms * 1e3 / chain / reps), flush=True)
for s in (8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 2048):
    run(s, chain=12 if s <= 512 else 3)
