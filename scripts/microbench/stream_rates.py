"""HBM stream rates of one MI355X through torch kernels (4 GiB buffers, 20 repetitions, HIP events): write-only (fill), read-only (sum), copy.
The assembly kernels are write-dominated (fused M + K: 1.2 GB read, 4.5 GB written per launch), so the write-only rate — not the 8 TB/s
peak of read + write — is the stream they can be compared with."""
import torch
n = 1 << 29  # doubles: 4 GiB
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: a.fill_(1.5)); print("write-only (fill)   %.2f TB/s" % (8 * n / t / 1e12))
t = timed(lambda: a.zero_());    print("write-only (memset) %.2f TB/s" % (8 * n / t / 1e12))
t = timed(lambda: a.sum());      print("read-only  (sum)    %.2f TB/s" % (8 * n / t / 1e12))
t = timed(lambda: b.copy_(a));   print("copy (read + write) %.2f TB/s" % (16 * n / t / 1e12))
t = timed(lambda: torch.add(a, 1.0, out=b)); print("a + 1 -> b          %.2f TB/s" % (16 * n / t / 1e12))
