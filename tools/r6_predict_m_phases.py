import os, sys, time, torch, torch.nn as nn
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from mural_amd.model import model_predict_m, nn_utils
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(5)
B = 16384
codes = torch.randint(0, 4, (B, 2001), device=dev, generator=g)
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
c = codes[:, 990:1011]
cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
cont = torch.zeros(B, 1, device=dev, dtype=torch.float64)
ys = torch.zeros(B, 1)
loader = [(ys[i:i + 16], cont[i:i + 16].cpu(), cat[i:i + 16].cpu(), x[i:i + 16].cpu()) for i in range(0, B, 16)]
crit = nn.CrossEntropyLoss(reduction="sum")
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
wrap(nn_utils._HostSymbolRoute, "gather", "gather")
wrap(type(model), "forward_symbols", "forward_symbols")
import mural_amd._lib as L
lib = L.lib()
orig = lib.mural_host_dense_to_symbols
def timed_classify(*a):
    t0 = time.perf_counter(); r = orig(*a); T["classify"] = T.get("classify", 0) + time.perf_counter() - t0; return r
class Proxy:
    def __getattr__(self, n):
        return timed_classify if n == "mural_host_dense_to_symbols" else getattr(lib, n)
L_lib = L.lib
L.lib = lambda: Proxy()
with torch.no_grad():
    model_predict_m(model, loader[:64], crit, dev, 4)
    for rep in range(3):
        T.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model_predict_m(model, loader, crit, dev, 4)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("total %.2f ms (returned after %.2f)" % (dt * 1e3, (t1 - t0) * 1e3), {k: round(v * 1e3, 2) for k, v in T.items()})
