import time, numpy as np, torch
n = 1 << 27  # 1 GiB of float64
a = np.ones(n, np.float64)
t = torch.from_numpy(a)
d = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def tm(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
gb = a.nbytes / 1e9
print("pageable H2D  %.1f GB/s" % (gb / tm(lambda: d.copy_(t))))
print("pageable D2H  %.1f GB/s" % (gb / tm(lambda: t.copy_(d))))
p = torch.empty(n, dtype=torch.float64).pin_memory()
print("cpu memcpy -> pinned (1 thread) %.1f GB/s" % (gb / tm(lambda: p.copy_(t))))
print("pinned H2D    %.1f GB/s" % (gb / tm(lambda: d.copy_(p, non_blocking=True))))
print("pinned D2H    %.1f GB/s" % (gb / tm(lambda: p.copy_(d, non_blocking=True))))
rt = torch.cuda.cudart()
b = np.ones(n, np.float64); tb = torch.from_numpy(b)
t0 = time.perf_counter(); r = rt.cudaHostRegister(tb.data_ptr(), tb.numel() * 8, 0); t1 = time.perf_counter()
print("hostRegister 1 GiB: %.1f ms (rc %s)" % ((t1 - t0) * 1e3, r))
print("registered H2D %.1f GB/s" % (gb / tm(lambda: d.copy_(tb, non_blocking=True))))
t0 = time.perf_counter(); rt.cudaHostUnregister(tb.data_ptr()); print("unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
import os; print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
