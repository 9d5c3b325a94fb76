"""A/B of two builds of the library on the 1x1 weight-gradient shapes of ResNet-50 (run once per build via SIMHAND_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simhand_amd import ops
from scripts.layer_table import timed
n, dt = 2048, torch.bfloat16
for cin, cout, h in ((256, 64, 56), (64, 256, 56), (64, 64, 56), (512, 128, 28), (128, 512, 28), (128, 128, 28), (1024, 256, 14), (256, 1024, 14), (256, 256, 14), (2048, 512, 7), (512, 2048, 7)):
    d = ops.conv_desc(n, h, h, cin, cout, 1, 1, 1, 0, dt)
    x = torch.randn(n, h, h, cin, device="cuda").to(dt)
    dy = torch.randn(n, h, h, cout, device="cuda").to(dt)
    us = timed(lambda: ops.conv2d_wgrad_oihw(d, x, dy, (cout, cin, 1, 1)))
    by = 2.0 * (x.numel() + dy.numel())
    gram = ""
    if cin == cout:
        st = ops.BNState(cin, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
        us2 = timed(lambda: ops.bn_apply_gram(x, st, True))
        gram = f"  bn_apply_gram {us2:7.0f} us {2 * 2.0 * x.numel() / us2 / 1e3:6.0f} GB/s"
    print(f"wgrad ({cin:4d},{cout:4d})@{h:2d}: {us:7.0f} us {by / us / 1e3:6.0f} GB/s{gram}")
    del x, dy
