"""Practical HBM ceiling of this box for a read + write stream: torch device-to-device copy of 0.8 GB (the size of
one keys-only scatter pass at 1e8 records), and a read-only reduction.  Calibrates the roofline fractions."""
import torch
n = 100_000_000
a = torch.empty(n, dtype=torch.int64, device="cuda").random_()
b = torch.empty_like(a)
for name, fn, nbytes in (("copy (8 B read + 8 B write per element)", lambda: b.copy_(a), 16 * n),
                         ("read-only sum (8 B per element)", lambda: a.sum(), 8 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("%-45s %.3f ms  %.0f GB/s  (%.1f %% of 8 TB/s)" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
