"""A generator-SHAPED network around the hot path.

TEST / BENCH INFRASTRUCTURE (not part of the product package).  The reference generators (model/networks/generator.py)
are consumed unchanged through `install()` where the reference checkout exists; it does not exist on the GPU box.
`WarpGenerator` is a small stand-in network with the same
*shape* of autograd graph around the ops -- what the trainer shell, the gradient reducer and `bench.py --workload
trainer_step` need to exercise on hardware:

    source image ----> source encoder (stride-2 conv / InstanceNorm / LeakyReLU)  --> features at 1/4 and 1/8 scale
    (source, source_B, target_B) --> flow head --> flow field + sigmoid mask at 1/8 and 1/4 scale
    target_B ---------> target encoder --> 1/8 scale
         --> ExtractorAttn layer 3 (k=3) on (source feature, decoder feature, flow) --> mask blend --> up
         --> ExtractorAttn layer 2 (k=5)                                          --> mask blend --> up --> up --> image

with the call convention of PoseGenerator.forward (generator.py:13-36): `(source, source_B, target_B) -> (generated,
flow_fields, masks)`, flow_fields / masks ordered coarse to fine like PoseFlowNet's (generator.py:118-137: attention at
`layers - i in attn_layer` walks the decoder from the coarsest scale), and the blend `out*(1-mask) + attn*mask`
(generator.py:130).  The attention block class is injectable so that the parity tests can build the identical network
on the host with the oracle's op-by-op block (`oracle.cpu_modules.ExtractorAttnCPU`) and compare a whole training step.

`RandomFeaturePyramid` stands in for the frozen VGG19 of PerceptualCorrectness (external_function.py:323-380: relu1_1 ..
relu4_1 = 64/128/256/512 channels at 1, 1/2, 1/4, 1/8 scale): random frozen weights, the same shapes -- the pretrained
weights need torchvision and a download.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from global_flow_local_attention_amd.extractor_attn import ExtractorAttn


def _act():
    return nn.LeakyReLU(0.1)


class _Down(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.Conv2d(cin, cout, 3, 2, 1), nn.InstanceNorm2d(cout), _act())


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm = nn.InstanceNorm2d(cout)
        self.act = _act()

    def forward(self, x):
        return self.act(self.norm(self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))))


class _Res(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.body = nn.Sequential(nn.InstanceNorm2d(c), _act(), nn.Conv2d(c, c, 3, 1, 1),
                                  nn.InstanceNorm2d(c), _act(), nn.Conv2d(c, c, 3, 1, 1))

    def forward(self, x):
        return x + self.body(x)


class WarpGenerator(nn.Module):
    """ngf=64 gives the production widths of the reference (pose_model.py:62-64): 256 channels at 1/8 scale (attention
    layer 3, k=3) and 128 at 1/4 (layer 2, k=5)."""

    def __init__(self, image_nc=3, structure_nc=18, output_nc=3, ngf=64, extractor_kz=None, attn_cls=ExtractorAttn,
                 flow_scale=1.0):
        super().__init__()
        kz = {"2": 5, "3": 3} if extractor_kz is None else extractor_kz
        self.flow_scale = float(flow_scale)
        c1, c2, c3 = ngf, 2 * ngf, 4 * ngf
        self.source = nn.ModuleList([_Down(image_nc, c1), _Down(c1, c2), _Down(c2, c3)])
        self.target = nn.Sequential(_Down(structure_nc, c1), _Down(c1, c2), _Down(c2, c3))
        f1, f2, f3 = ngf // 2, ngf, 2 * ngf
        self.flow_enc = nn.Sequential(_Down(image_nc + 2 * structure_nc, f1), _Down(f1, f2), _Down(f2, f3))
        self.flow3, self.mask3 = nn.Conv2d(f3, 2, 3, 1, 1), nn.Conv2d(f3, 1, 3, 1, 1)
        self.flow_up = _Up(f3, f2)
        self.flow2, self.mask2 = nn.Conv2d(f2, 2, 3, 1, 1), nn.Conv2d(f2, 1, 3, 1, 1)
        self.attn3 = attn_cls(c3, kz["3"], _act(), softmax=True)
        self.dec3 = nn.Sequential(_Res(c3), _Up(c3, c2))
        self.attn2 = attn_cls(c2, kz["2"], _act(), softmax=True)
        self.dec2 = nn.Sequential(_Res(c2), _Up(c2, c1))
        self.dec1 = _Up(c1, c1)
        self.outconv = nn.Sequential(nn.Conv2d(c1, output_nc, 3, 1, 1), nn.Tanh())

    def flow_net(self, source, source_B, target_B):
        h = self.flow_enc(torch.cat((source, source_B, target_B), 1))
        flow3, mask3 = self.flow3(h) * self.flow_scale, torch.sigmoid(self.mask3(h))
        h = self.flow_up(h)
        flow2, mask2 = self.flow2(h) * self.flow_scale, torch.sigmoid(self.mask2(h))
        return [flow3, flow2], [mask3, mask2]

    def forward(self, source, source_B, target_B):
        feats, h = [], source
        for block in self.source:
            h = block(h)
            feats.append(h)
        flow_fields, masks = self.flow_net(source, source_B, target_B)
        out = self.target(target_B)
        out = out * (1 - masks[0]) + self.attn3(feats[2], out, flow_fields[0]) * masks[0]
        out = self.dec3(out)
        out = out * (1 - masks[1]) + self.attn2(feats[1], out, flow_fields[1]) * masks[1]
        out = self.dec1(self.dec2(out))
        return self.outconv(out), flow_fields, masks


class RandomFeaturePyramid(nn.Module):
    """Frozen random stand-in for the reference's VGG19 feature extractor: image -> {'rel1_1' (sic, the reference's
    key), 'relu2_1', 'relu3_1', 'relu4_1'} with 64/128/256/512 channels at 1, 1/2, 1/4, 1/8 scale."""

    def __init__(self, image_nc=3, widths=(64, 128, 256, 512), seed=0):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.convs = nn.ModuleList()
        cin = image_nc
        for i, c in enumerate(widths):
            conv = nn.Conv2d(cin, c, 3, 1 if i == 0 else 2, 1)
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (9 * cin)) ** 0.5)
                conv.bias.copy_(torch.randn(conv.bias.shape, generator=gen) * 0.1)
            self.convs.append(conv)
            cin = c
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        out = {}
        for name, conv in zip(("rel1_1", "relu2_1", "relu3_1", "relu4_1"), self.convs):
            x = F.relu(conv(x))
            out[name] = x
        return out
