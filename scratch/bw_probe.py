"""Plain HBM rates of this box through torch: fill (pure write), copy (read + write), sum (pure read) on 8 GiB."""
import torch, time
n = 8 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print("fill  (write)      %.2f TB/s" % (n / t(lambda: a.fill_(1)) / 1e12))
print("copy  (read+write) %.2f TB/s (2 x bytes)" % (2 * n / t(lambda: b.copy_(a)) / 1e12))
v = a.view(torch.int64)
print("sum   (read)       %.2f TB/s" % (n / t(lambda: v.sum()) / 1e12))
