import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from swin_v2_weather_amd.networks import swinv2_global as N
from swin_v2_weather_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = N.SwinTransformerV2Cr(img_size=(72, 144), patch_size=4, depths=(3,), num_heads=(2,), in_chans=3, out_chans=3, embed_dim=32,
                          img_window_ratio=8, drop_path_rate=0.0, full_pos_embed=True, rel_pos=True, residual=True)
with torch.no_grad():
    for n_, p in m.named_parameters():
        if n_.endswith("norm1.weight") or n_.endswith("norm2.weight"): p.uniform_(0.5, 1.0)
        if "meta_mlp" in n_: p.mul_(3.0)
m = m.to(dev).train()
x = torch.randn(2, 3, 72, 144, device=dev)
y = m(x)
st = m.stages[0]._last_cpb
hidden = st.hidden
def decode(bits):
    w = bits.to(torch.int64); dec = (w | (w >> 8) | (w >> 16)) & 0xFF
    j = torch.arange(hidden, device=bits.device); widx = ((j >> 3) & 3) * (hidden // 32) + (j >> 5)
    return ((dec[:, widx] >> (j & 7).view(1, -1)) & 1).bool()
for i, blk in enumerate(m.stages[0].blocks):
    mk = decode(st.keep_bits[i])
    keep = mk.to(torch.bfloat16) * 1.140625
    mm = blk.attn.meta_mlp
    b = torch.empty(2, 162, 162, device=dev)
    ops.cpb_fwd(mm.fc1.weight.detach(), mm.fc1.bias.detach(), mm.fc2.weight.detach(), mm.fc2.bias.detach(), keep.contiguous(), b, 9, 18, 2, hidden, 0.125)
    d = (b - st.bias_all[i]).abs()
    print(i, "max abs diff", float(d.max()), "rel", float(d.norm() / b.norm()), "frac >1e-3", float((d > 1e-3).float().mean()), "keep frac", float(mk.float().mean()))
