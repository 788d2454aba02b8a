"""While a long HIP graph replays on one stream: how long do small device-to-host read-backs on ANOTHER stream take?"""
import time, torch
dev = torch.device("cuda", 0)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
z = torch.arange(4, device=dev, dtype=torch.float32)
pin = torch.empty(4, pin_memory=True)
w = torch.randn(2048, 2048, device=dev)
with torch.cuda.stream(a):
    for _ in range(3):
        y = w @ w
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(a):
    with torch.cuda.graph(g, stream=a):
        y = w
        for _ in range(600):
            y = y @ w * 1e-3
torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); print("graph alone: %.1f ms" % ((time.perf_counter() - t0) * 1e3))


def trial(name, fn, graph=True):
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        if graph:
            g.replay()
        else:
            torch.cuda._sleep(int(60e-3 * 2.4e9))
    t0 = time.perf_counter()
    with torch.cuda.stream(b):
        fn()
    dt = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print("%-60s %-14s host waited %6.2f ms" % (name, "(graph)" if graph else "(plain kernel)", dt), flush=True)


def pinned_event():
    pin.copy_(z, non_blocking=True)
    e = torch.cuda.Event(); e.record(); e.synchronize()


def pinned_streamsync():
    pin.copy_(z, non_blocking=True)
    torch.cuda.current_stream().synchronize()


for graph in (False, True):
    trial("z.cpu()", lambda: z.cpu(), graph)
    trial("z.tolist()", lambda: z.tolist(), graph)
    trial("pinned copy_(non_blocking) + event.synchronize()", pinned_event, graph)
    trial("pinned copy_(non_blocking) + stream.synchronize()", pinned_streamsync, graph)
    trial("event.record + event.synchronize (no copy)", lambda: (lambda e: (e.record(), e.synchronize()))(torch.cuda.Event()), graph)
    trial("tiny kernel + stream.synchronize", lambda: (z.add_(1), torch.cuda.current_stream().synchronize()), graph)
