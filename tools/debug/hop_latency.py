import time, torch
x = torch.zeros(64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 2000
def single():
    with torch.cuda.stream(s1):
        for _ in range(2 * N): x.add_(1)
def pingpong():
    e1 = [torch.cuda.Event() for _ in range(N)]; e2 = [torch.cuda.Event() for _ in range(N)]
    for i in range(N):
        with torch.cuda.stream(s1): x.add_(1)
        e1[i].record(s1); s2.wait_event(e1[i])
        with torch.cuda.stream(s2): x.add_(1)
        e2[i].record(s2); s1.wait_event(e2[i])
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0)
for _ in range(2):
    a = t(single); b = t(pingpong)
    print("same stream: %.2f us per kernel;  alternating streams with event waits: %.2f us per kernel  -> cross-stream hop ~ %.2f us" % (a / (2 * N) * 1e6, b / (2 * N) * 1e6, (b - a) / (2 * N) * 1e6))
