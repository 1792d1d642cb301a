import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import torch
from helpers import EB, MODP_ORDER as ORDER
from mpvss_rs_amd import Engine
eng = Engine(0)
fx = lambda v: v.to_bytes(EB, "big")
QH = ORDER // 2
for a0 in (5, QH, QH + 2, QH+3):
    d_pos = torch.tensor([0], dtype=torch.int64, device="cuda:0")
    out = torch.full((EB,), 0xA5, dtype=torch.uint8, device="cuda:0")
    eng.poly_eval_device(fx(a0), d_pos.data_ptr(), 1, out.data_ptr())
    print(list(out.cpu().numpy()[:8]))
