"""Raw PCIe rates of the box with page-locked memory: H2D, D2H, both at once (what bounds the host-memory API)."""
import time, torch
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
def h2d():
    with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
def both():
    h2d(); d2h()
print(f"H2D 1 GiB: {n / timed(h2d) / 1e9:.1f} GB/s")
print(f"D2H 1 GiB: {n / timed(d2h) / 1e9:.1f} GB/s")
t = timed(both)
print(f"both at once: {t*1e3:.1f} ms = {n / t / 1e9:.1f} GB/s each way")
