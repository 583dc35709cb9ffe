import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swin_v2_weather_amd import _lib as L, ops
from swin_v2_weather_amd.networks import swinv2_global as N
variant = sys.argv[1]
dev = torch.device("cuda:0")
fx = np.load(os.path.join(ROOT, "tests/golden/model_nopos.npz"))
cin, cout, H, W, Cc, depth, h, ratio, relpos, residual, seed = [int(v) for v in fx["meta"]]
m = N.SwinTransformerV2Cr(img_size=(H, W), patch_size=4, depths=(depth,), num_heads=(h,), in_chans=cin, out_chans=cout,
                          embed_dim=Cc, img_window_ratio=ratio, full_pos_embed=True, rel_pos=bool(relpos), residual=bool(residual))
m.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("p:")})
m = m.to(dev).eval()
gy = torch.from_numpy(fx["gy"]).to(dev)
def run(ck):
    m.set_grad_checkpointing(ck)
    m.zero_grad()
    x = torch.from_numpy(fx["x"]).to(dev).requires_grad_(True)
    y = m(x)
    y.backward(gy)
    return y, x.grad
if variant == "A":
    y1, g1 = run(False); y2, g2 = run(True)
elif variant == "B":
    y2, g2 = run(True)
elif variant == "C":
    y1, g1 = run(False); torch.cuda.synchronize(); y2, g2 = run(True)
elif variant == "D":
    y1, g1 = run(False); y2, g2 = run(False)
torch.cuda.synchronize()
print(variant, "done", float(y2.detach().sum()), float(g2.sum()))
