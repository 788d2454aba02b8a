import importlib, os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
bench.torch = torch
rg = importlib.import_module("rag-gesture_amd")
dev = torch.device("cuda", 0)
wl = bench.Workload(rg, "longform", 10, dev, 0, 32768, clips=10)
m = wl.model
wl.prime()
wl.step()
torch.cuda.synchronize()
orig_submit, orig_flush, orig_gr = m.submit, m.flush, m._graph_run
import traceback
blocks = []
def wrap(owner, name):
    fn = getattr(owner, name)
    def w(*a, **k):
        a0 = time.perf_counter(); r = fn(*a, **k); dt = (time.perf_counter() - a0) * 1e3
        if dt > 1.0:
            fr = [f for f in traceback.extract_stack()[:-1] if "rag-gesture_amd" in f.filename][-3:]
            blocks.append(((a0 - t0[0]) * 1e3, dt, "%s.%s" % (getattr(owner, "__name__", owner), name), " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr))))
        return r
    setattr(owner, name, w)
for owner, names in ((torch.Tensor, ("cpu", "item", "tolist", "numpy", "to", "nonzero", "__bool__", "__int__", "__float__", "copy_", "__setitem__", "__getitem__", "pin_memory", "index_copy_", "clone")),
                     (torch.cuda.Event, ("synchronize",)), (torch.cuda.Stream, ("synchronize",)), (torch.cuda, ("synchronize",)),
                     (torch, ("tensor", "as_tensor", "stack", "cat", "zeros", "empty"))):
    for nm in names:
        wrap(owner, nm)
t0 = [0.0]
def submit(**kw):
    a = time.perf_counter(); r = orig_submit(**kw); print("  submit host %.1f ms (t=%.1f) -> %s" % ((time.perf_counter() - a) * 1e3, (a - t0[0]) * 1e3, "result" if r is not None else "None")); return r
def flush():
    a = time.perf_counter(); r = orig_flush(); print("  flush host %.1f ms (t=%.1f) -> %d results" % ((time.perf_counter() - a) * 1e3, (a - t0[0]) * 1e3, len(r))); return r
def gr(key, inputs, fn, owner=None):
    print("     graph %s" % (str(key)[:60],)); return orig_gr(key, inputs, fn, owner=owner)
m.submit, m.flush, m._graph_run = submit, flush, gr
t0[0] = time.perf_counter()
wl.step()
torch.cuda.synchronize()
print("pass: %.1f ms" % ((time.perf_counter() - t0[0]) * 1e3))
for b in blocks:
    print("  host blocked %6.1f -> %6.1f (%5.1f) %s  %s" % (b[0], b[0] + b[1], b[1], b[2], b[3]))
