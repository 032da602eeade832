"""Diagnostic: per-phase time of the level-0 MFMA ConvBlock kernel (MURAL_CONVBLOCK8_MFMA=1), averaged per tile: mural_debug_cb8_set_stamps gives
every workgroup 8 accumulators; the encoder and the decoder launch of one forward add into the same ones (their tiles are summed)."""
import os, sys
os.environ["MURAL_CONVBLOCK8_MFMA"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd import _lib
from mural_amd.model import model_choice, weights_init
from mural_amd.model import indel_train as IT   # noqa: F401
lib = _lib.lib()
cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
torch.manual_seed(0)
model = model_choice(0, cfg, dict(n_class=8), "indel")
model.apply(weights_init)
model = model.cuda().eval()
B = 2048
codes = torch.randint(0, 4, (B, 8000), device="cuda")
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
with torch.no_grad():
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    stamps = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
    lib.mural_debug_cb8_set_stamps(stamps.data_ptr())
    model(x)
    torch.cuda.synchronize()
    lib.mural_debug_cb8_set_stamps(None)
s = stamps.view(-1, 8)[:2048].cpu().double()
tiles = s[:, 7].clamp(min=1)
names = ["wait/prev", "stage+barrier", "front", "block", "tail/store"]
print("tiles per workgroup (both level-0 launches summed): %.1f" % tiles.mean())
for k, n in enumerate(names):
    print("%-14s %.2f us per tile" % (n, (s[:, k] / tiles).mean() * 0.01))
