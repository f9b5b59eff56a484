import time, torch
d = torch.device("cuda")
x = torch.ones(1 << 28, device=d)  # 1 GiB: an elementwise pass takes ~0.4 ms
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def sleep(): torch.cuda._sleep(2_000_000)  # ~1 ms of spinning in one wave
def work():
    for _ in range(3): x.mul_(1.0001)
for _ in range(2):
    a = t(lambda: (sleep()))
    b = t(lambda: (work()))
    def both():
        with torch.cuda.stream(s1): sleep()
        with torch.cuda.stream(s2): work()
    c = t(both)
    print("sleep alone %.2f ms, work alone %.2f ms, on two streams %.2f ms" % (a, b, c))
