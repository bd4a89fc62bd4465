#!/usr/bin/env python3
"""A whole training step on the MI355X path, end to end: a stand-in backbone (two 3x3 convolutions -- NOT the reference's
ResUNet, which stays PyTorch code and is out of scope) -> the embedding heads (pea.OutConv, the HIP 1x1 convolution) at
five scales -> the six-loss section straight from the label images (one autograd node, second HIP stream, class-balance
tables computed ahead) -> backward -> optimizer step; the EMA teacher follows as in scripts_cvppp/main.py:318-323.
Prints the loss of a few steps (it must fall) and the time of the loss section inside the step.

    python examples/train_step_demo.py [steps]
"""
import copy
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pea = ge.load_package()


class TinyNet(nn.Module):
    """stand-in for the reference's encoder-decoder: full-resolution features + four pooled scales, one head each"""

    def __init__(self, emd=16, feat=32):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(3, feat, 3, padding=1), nn.ReLU(), nn.Conv2d(feat, feat, 3, padding=1), nn.ReLU())
        self.outconv_emb = pea.OutConv(feat, emd)                                   # same names as unet2d_residual.py:303-307
        self.outconvs = nn.ModuleList([pea.OutConv(feat, emd) for _ in range(4)])   # emd1..emd4 at 1/2 .. 1/16

    def forward(self, x):
        f = self.body(x)
        emds = [head(F.avg_pool2d(f, 2 ** (j + 1))) for j, head in enumerate(self.outconvs)]
        return emds, self.outconv_emb(f)


def main(steps=6, B=4, H=128, W=160, seed=555):
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    offsets = pea.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half = 2
    model, crit = TinyNet().to(dev), pea.WeightedMSE()
    ema_model = copy.deepcopy(model).requires_grad_(False)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    img = torch.rand(B, 3, H, W, device=dev, generator=g)
    lab = torch.randint(0, 7, (B, H // 16, W // 16), device=dev, generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2).int()
    img = img + 0.5 * lab[:, None].float() / 7.0            # something to learn: the image carries the instance id
    label_downs = [lab[:, ::2 ** j, ::2 ** j].contiguous() for j in range(1, 5)]  # data_provider.py:199-208 (nearest)
    losses, sect_ms = [], []
    for it in range(steps):
        tabs = pea.cvppp_label_weight_tables(lab, label_downs, offsets, nb_half)   # needs the labels only: before the forward
        emds, embedding = model(img)
        with torch.no_grad():
            _, ema_embedding = ema_model(img)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        loss, pred, parts = pea.cvppp_loss_section_from_labels(embedding, emds, ema_embedding, lab, label_downs, crit, offsets, nb_half,
                                                               affs0_weight=1, deep_weight=2, relu_pred=True, weight_tables=tabs)
        t1.record()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        with torch.no_grad():                                                       # EMA teacher, main.py:318-323
            for pe, pm in zip(ema_model.parameters(), model.parameters()):
                pe.mul_(0.99).add_(pm, alpha=0.01)
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
        sect_ms.append(t0.elapsed_time(t1))
        assert pred.shape == (B, len(offsets), H, W) and float(pred.min()) >= 0.0
    print("train_step_demo: loss %s" % " -> ".join("%.4f" % v for v in losses))
    print("train_step_demo: loss section forward %.3f ms (B=%d x %dx%d)" % (min(sect_ms), B, H, W))
    assert losses[-1] < losses[0], "the loss did not fall"
    print("train_step_demo: OK")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
