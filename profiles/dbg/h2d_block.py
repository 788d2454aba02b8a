"""Does a small host-to-device copy on one stream wait for a kernel that runs on ANOTHER stream?  (ROCm 7, MI355X.)"""
import time, torch
dev = torch.device("cuda", 0)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(4, device=dev)
torch.cuda.synchronize()


def trial(name, fn):
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(50e-3 * 2.4e9))      # ~50 ms busy kernel on stream a
    t0 = time.perf_counter()
    with torch.cuda.stream(b):
        fn()
    dt = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    print("%-70s host call took %6.2f ms" % (name, dt), flush=True)


pinned = torch.tensor([10, 20, 30]).pin_memory()
idx_dev = torch.tensor([10, 20, 30], device=dev)
q = torch.ones(16, 43, device=dev)
trial("nothing", lambda: None)
trial("torch.tensor([..], device=cuda)", lambda: torch.tensor([10, 20, 30], device=dev))
trial("q[:, [10, 20, 30]] = 0   (list index)", lambda: q.__setitem__((slice(None), [10, 20, 30]), 0))
trial("q[:, idx_dev] = 0   (device index tensor)", lambda: q.__setitem__((slice(None), idx_dev), 0))
trial("q[:, 10] = 0 (int index)", lambda: q.__setitem__((slice(None), 10), 0))
trial("cpu tensor .to(dev)  (pageable)", lambda: torch.tensor([1.0, 2.0]).to(dev))
trial("pinned .to(dev, non_blocking=True)", lambda: pinned.to(dev, non_blocking=True))
trial("pinned .to(dev)  (blocking)", lambda: pinned.to(dev))
trial("torch.full((4,), 3.0, device=dev)", lambda: torch.full((4,), 3.0, device=dev))
trial("x.cpu() of an idle tensor", lambda: x.cpu())
trial("torch.cuda.Event().record()", lambda: torch.cuda.Event().record())
