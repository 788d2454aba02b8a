import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import transformers
rg = importlib.import_module("rag-gesture_amd")
G = rg.gemm
torch.manual_seed(1)
m = transformers.Wav2Vec2Model(transformers.Wav2Vec2Config()).eval()
wave = torch.randn(160000) * 0.05 + 0.01
xn = (wave - wave.mean()) / torch.sqrt(wave.var(unbiased=False) + 1e-7)
acts = {}
fe = m.feature_extractor
with torch.no_grad():
    h = xn[None, None]
    for i, l in enumerate(fe.conv_layers):
        c = l.conv(h); acts["conv%d" % i] = c[0].T.clone()
        h = l(h); acts["act%d" % i] = h[0].T.clone()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
f = rg.features.Wav2Vec2Features(m.state_dict(), device="cuda", precision="fp32")
hh, dev = f.h, f.dev
x = torch.zeros(160064, device=dev); x[:160000] = xn.cuda()
c0 = f.convs[0]
T = (160000 - 10) // 5 + 1
y = torch.empty(T, 512, device=dev)
G.gemm(hh, M=T, N=512, K=10, W=c0["lin"].w, out=y, segs=[G.Seg(x, ld=5)], bias=None)
torch.cuda.synchronize()
print("conv0", rel(y.cpu(), acts["conv0"]), y.shape, acts["conv0"].shape)
# torch check of the same GEMM
A = x[:160000].unfold(0, 10, 5)
print("conv0 torch-unfold", rel((A @ m.state_dict()["feature_extractor.conv_layers.0.conv.weight"][:, 0].T.cuda()).cpu(), acts["conv0"]))
cur = torch.empty(T, 512, device=dev, dtype=torch.bfloat16); ws = torch.empty(1024, device=dev)
hh.call("time_groupnorm_gelu", y, f.gn_g, f.gn_b, cur, None, T, 512, 1e-5, ws)
torch.cuda.synchronize()
print("act0", rel(cur.float().cpu(), acts["act0"]))
ref_cur = acts["act0"].cuda().contiguous()
for i in range(1, 7):
    cv = f.convs[i]
    Tn = (ref_cur.shape[0] - cv["k"]) // 2 + 1
    out = torch.empty(Tn, 512, device=dev)
    G.gemm(hh, M=Tn, N=512, K=cv["lin"].k, W=cv["lin"].w, out=out, segs=[G.Seg(ref_cur.view(-1), ld=1024)], act=1)
    torch.cuda.synchronize()
    print("act%d (from HF input) fp32-A" % i, rel(out.cpu(), acts["act%d" % i]), Tn, acts["act%d" % i].shape)
    fb = rg.features.Wav2Vec2Features.__new__(rg.features.Wav2Vec2Features)
    ref_cur = acts["act%d" % i].cuda().contiguous()
print("---- chained from own act0")
mine = cur.float().contiguous()
for i in range(1, 7):
    cv = f.convs[i]
    Tn = (mine.shape[0] - cv["k"]) // 2 + 1
    out = torch.empty(Tn, 512, device=dev)
    G.gemm(hh, M=Tn, N=512, K=cv["lin"].k, W=cv["lin"].w, out=out, segs=[G.Seg(mine.view(-1), ld=1024)], act=1)
    torch.cuda.synchronize()
    print("act%d chained" % i, rel(out.cpu(), acts["act%d" % i]), "|ref| rms %.3e mean %.3e" % (acts["act%d" % i].pow(2).mean().sqrt(), acts["act%d" % i].mean()))
    mine = out
print("conv_features()", rel(f.conv_features(xn).cpu(), acts["act6"]))
