"""ResidualUNet2D_deep: the 2D backbone of the CVPPP / BBBC039V1 trees, re-declared in plain PyTorch-ROCm.

The backbone is NOT part of the accelerated path (BASELINE.json north_star: "the ResUNet ... encoder-decoder forward/backward
stays in PyTorch-ROCm"); it is here so that the multi-GPU training step (bench.py's train imgs/s leg)
has the real thing to put under DistributedDataParallel / RCCL, and so that the reference's checkpoints load unchanged.
What is mirrored from scripts_cvppp/model/unet2d_residual.py:279-353 is therefore the STATE-DICT LAYOUT (module attribute
names and the order of the layers inside each nn.Sequential: `inconv.conv.conv.0.weight`, `down1.block.project.1.running_mean`,
`up2_emb.block.conv.3.weight`, `outconv_emb.conv.weight`, `binary_seg.3.bias`, ...) and the arithmetic:

    residual unit   two 3x3 conv+BN (ReLU between) plus a 3x3 conv+BN projection of the input, ReLU after the sum   (:5-25)
    encoder         inconv, then four (residual unit -> 2x2 max pool) stages, widths nfeatures = [16,32,64,128,256]      (:29-50)
    decoder         four (bilinear x2 upsample, align_corners -> residual unit) stages; before stages 2-4 the encoder
                    feature of the same resolution is concatenated AFTER the decoder feature (odd sizes: the decoder
                    feature is replicate-padded at the bottom / right first)                                          (:53-64, :320-326)
    heads           five 1x1 convolutions to `emd` channels on x5 (1/16), and on the decoder features at 1/8, 1/4, 1/2, 1/1;
                    a 1x1 conv - BN - ReLU - 1x1 conv mask head on the full-resolution decoder feature               (:300-315)
    forward         -> (emd@1/16, emd@1/8, emd@1/4, emd@1/2, embedding@1, mask_logits)                                (:328-353)

The five heads are this package's OutConv (model/head.py: pea_head_fwd / pea_head_bwd on the GPU), which keeps the
reference's `conv.weight` / `conv.bias` names; `hip_heads=False` builds plain nn.Conv2d heads (CPU tests of the layout)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .head import OutConv as HipOutConv


def _conv_bn(cin, cout):
    return [nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout)]


class ResidualBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Sequential(*_conv_bn(in_ch, out_ch), nn.ReLU(inplace=True), *_conv_bn(out_ch, out_ch))
        self.project = nn.Sequential(*_conv_bn(in_ch, out_ch))

    def forward(self, x):
        return F.relu(self.conv(x) + self.project(x))


class InConv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = ResidualBlock(in_ch, out_ch)

    def forward(self, x):
        return self.conv(x)


class Down(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.block = ResidualBlock(in_ch, out_ch)
        self.pool = nn.MaxPool2d(2)

    def forward(self, x):
        return self.pool(self.block(x))


class Up(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.block = ResidualBlock(in_ch, out_ch)

    def forward(self, x):
        return self.block(self.upsample(x))


class _TorchOutConv(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, 1)

    def forward(self, x):
        return self.conv(x)


def _skip_cat(dec, enc):
    """decoder feature first, encoder feature second; the decoder feature grows to the encoder's size by replication"""
    dh, dw = enc.shape[-2] - dec.shape[-2], enc.shape[-1] - dec.shape[-1]
    if dh or dw:
        dec = F.pad(dec, (0, dw, 0, dh), mode="replicate")
    return torch.cat([dec, enc], dim=1)


class ResidualUNet2D_deep(nn.Module):
    def __init__(self, in_channels=3, out_channels=2, nfeatures=(16, 32, 64, 128, 256), emd=16, if_sigmoid=False,
                 show_feature=False, hip_heads=True):
        super().__init__()
        f = list(nfeatures)
        head = HipOutConv if hip_heads else _TorchOutConv
        self.if_sigmoid, self.show_feature = if_sigmoid, show_feature
        self.inconv = InConv(in_channels, f[0])
        self.down1, self.down2, self.down3, self.down4 = (Down(f[i], f[i + 1]) for i in range(4))
        self.up1_emb = Up(f[4], f[4])
        self.up2_emb = Up(f[4] + f[3], f[3])
        self.up3_emb = Up(f[3] + f[2], f[2])
        self.up4_emb = Up(f[2] + f[1], f[1])
        self.outconv1, self.outconv2 = head(f[4], emd), head(f[4], emd)
        self.outconv3, self.outconv4, self.outconv_emb = head(f[3], emd), head(f[2], emd), head(f[1], emd)
        self.binary_seg = nn.Sequential(nn.Conv2d(f[1], f[1], 1), nn.BatchNorm2d(f[1]), nn.ReLU(), nn.Conv2d(f[1], out_channels, 1))

    def forward(self, x):
        x1 = self.inconv(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        emd16 = self.outconv1(x5)
        d = self.up1_emb(x5)
        emd8 = self.outconv2(d)
        d = self.up2_emb(_skip_cat(d, x4))
        emd4 = self.outconv3(d)
        d = self.up3_emb(_skip_cat(d, x3))
        emd2 = self.outconv4(d)
        d = self.up4_emb(_skip_cat(d, x2))
        return emd16, emd8, emd4, emd2, self.outconv_emb(d), self.binary_seg(d)
