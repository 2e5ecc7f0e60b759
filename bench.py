#!/usr/bin/env python3
"""bench.py -- throughput of the GFLA feature-warping hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of B synthetic images
per GPU, forward AND backward, at the feature-map shapes PoseGenerator has for a 256x176 image with
attn_layer=2,3 / kernel_size 2=5,3=3 (SURVEY.md section 8, table of shapes):

    ExtractorAttn layer 3:  source/target (B,256,32,22), flow (B,2,32,22), k=3
    ExtractorAttn layer 2:  source/target (B,128,64,44), flow (B,2,64,44), k=5
       = block_extractor + both FC convolutions + softmax/reshape/aggregate, gradients to
         source, target, flow and the FC parameters (base_function.py:790-810)
    Resample2d(4,1,sigma=2) (PerceptualCorrectness, external_function.py:233,274), fwd + bwd:
       relu4_1-shaped (B,512,32,22) with the layer-3 flow, relu3_1-shaped (B,256,64,44) with the
       layer-2 flow

The stock convolutions/normalisations of the rest of PoseGenerator are out of scope
and the reference network code does not exist on the GPU box, so `value` is images/s THROUGH THE
HOT PATH, not end-to-end generator throughput.  Inputs are resident in HBM before timing starts.

The FC layers of ExtractorAttn (base_function.py:799-807) run on this library's own f32 MFMA kernels
(csrc/fc_*.hip): no vendor GEMM / convolution is called anywhere in the step.  `--fc-mode 4` (default, the headline)
is float32 in the Winograd domain (F(2x2,5x5) / F(4x4,3x3) on v_mfma_f32_16x16x4_f32); mode 0 is the direct f32
convolution, modes 3 / 2 (f16-split operands, f32 accumulation) are labelled experiments, all reported under "variants".

`python bench.py --gpus N` without a torchrun environment spawns its own N ranks (self_spawn); the torchrun form of
the contract works as well.

Rank 0 prints ONE JSON line (contract in the task statement) that additionally carries
  "roofline":     the dominant gfx950 kernel of the step -- an MFMA kernel: executed FLOPs / HIP-event duration
                  against the 157.3 TFLOP/s f32 matrix-core peak
  "kernels":      per C-ABI entry point of the step: HIP-event time, algorithmic bytes and FLOPs; and per internal
                  kernel of the FC path (timed alone through gfla_fc_kernel_f32)
  "oracle_check": forward + input gradients of samples 0 and B-1 of the timed configuration against the CPU oracle;
                  runs AFTER the timed region (its host threads disturb a timed region started behind it) and aborts
                  the run on a mismatch (rank 0, N=1)
  "north_star":   block_extractor + local-attention forward, op by op, against the HBM roofline
  "legs":         the other BASELINE configs, compact (configs[1] ops, config-3 inference, losses, trainer step, face bf16)
  "variants":     the same step with the other FC arithmetic modes / stream arrangements (not the headline)
  "cpu_baseline": the reference composition with the CPU oracle kernels on the host cores,
                  timed on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import re
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))   # warp_generator.py: the stand-in network of --workload trainer_step

import global_flow_local_attention_amd as gfla  # noqa: E402
from global_flow_local_attention_amd import _lib, dist as gdist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 matrix-core peak (MI355X_MICROARCH.md), arithmetic modes 1-3
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input matrix cores = the f32 vector rate (MI355X_MICROARCH.md)

LAYERS = (  # (name, C, H, W, k) for a 256x176 input, layers=3, ngf=64
    ("attn3", 256, 32, 22, 3),
    ("attn2", 128, 64, 44, 5),
)
VGG = (  # Resample2d call sites of PerceptualCorrectness: (name, C, H, W)
    ("relu4_1", 512, 32, 22),
    ("relu3_1", 256, 64, 44),
)


def smooth_flow(B, H, W, device, gen):
    n = torch.randn(B, 2, H, W, device=device, generator=gen) * 12
    n = torch.nn.functional.pad(n, (3, 3, 3, 3), mode="replicate")
    return torch.nn.functional.avg_pool2d(n, 7, 1).contiguous()


class HotPath:
    """Synthetic inputs + modules of one rank."""

    def __init__(self, B, device, seed, modules=None, vgg_grad=True, fc_impl=None, fc_mode=None, with_losses=False):
        gen = torch.Generator(device=device).manual_seed(seed)
        self.B, self.device = B, device
        self.attn, self.inputs, self.vgg = [], [], []
        torch.manual_seed(1234)  # identical FC parameters on every rank
        for i, (name, C, H, W, k) in enumerate(LAYERS):
            mod = modules[i] if modules else gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
            if not modules:
                # LeakyReLU has a kink at 0: among 11.5 M hidden activations per call some land within float rounding of it,
                # and the float32 kernels and the float64-free host oracle then pick different slopes for that unit -- an O(1)
                # difference in its gradient that would fail (and abort) the oracle check for no fault of the kernels.  The
                # first FC layer's bias is +-8 on even / odd hidden channels: every pre-activation is clear of the kink,
                # both slopes stay in use, and no kernel does more or less work.
                with torch.no_grad():
                    bias = mod.fully_connect_layer[0].bias
                    bias.copy_(torch.where(torch.arange(bias.numel()) % 2 == 0, 8.0, -8.0) + 0.1 * bias)
            if fc_impl is not None:
                mod.fc_impl, mod.fc_mode = fc_impl, fc_mode
            self.attn.append(mod.to(device))
            src = torch.randn(B, C, H, W, device=device, generator=gen).requires_grad_()
            tgt = torch.randn(B, C, H, W, device=device, generator=gen).requires_grad_()
            flow = smooth_flow(B, H, W, device, gen).requires_grad_()
            self.inputs.append((src, tgt, flow))
        self.upstream = None
        self.reducer = None
        # the loss-side warps on a second HIP stream (step()): measured 4.54 against 4.66 ms per step
        # (profiles/r4_two_streams.txt); `variants.one_stream` in the line is the same step on one stream
        self.two_streams = True
        for (name, C, H, W) in VGG:
            feat = torch.randn(B, C, H, W, device=device, generator=gen).requires_grad_(vgg_grad)
            self.vgg.append(feat)
        # --with-losses (BASELINE config 4's loss side): the sampling-correctness loss consumes the warps (its
        # best-match cosine similarity is the library's streaming max-GEMM on MFMA) and the affine regulariser reads
        # the flow fields -- external_function.py:223-279, 12-77.  Frozen-VGG features: constants, as in training.
        self.losses = None
        if with_losses:
            self.losses = (gfla.PerceptualCorrectness(), gfla.MultiAffineRegularizationLoss({"2": 5, "3": 3}))
            self.vgg_target = [torch.randn(B, C, H, W, device=device, generator=gen).relu() for (_, C, H, W) in VGG]
            self.vgg = [v.detach().relu() for v in self.vgg]

    def params(self):
        return [p for m in self.attn for p in m.parameters()]

    def step(self, resample, allreduce=True):
        """Forward through both attention layers and both resample sites, then backward from fixed
        upstream gradients (what the rest of the generator / the losses would send back) -- no
        synthetic loss kernels inside the timed region.
        self.two_streams: the warps of the sampling-correctness loss (Resample2d of the VGG features by the flow fields,
        external_function.py:274) are a branch of the training graph that shares only the flow fields with the generator's
        attention layers; they are issued on a side stream (forward here, backward by autograd on the same stream)."""
        if getattr(self, "two_streams", False) and self.losses is None and self.inputs[0][0].is_cuda:
            cur = torch.cuda.current_stream(self.inputs[0][0].device)
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=self.inputs[0][0].device)
            ready = cur.record_event()
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                warps = [resample(feat, flow) for feat, (_, _, flow) in zip(self.vgg, self.inputs)]
                done = self._side.record_event()
            outs = [mod(src, tgt, flow) for mod, (src, tgt, flow) in zip(self.attn, self.inputs)]
            cur.wait_event(done)
            for w in warps:
                w.record_stream(cur)
            outs += warps
            return self._backward(outs, allreduce)
        outs = [mod(src, tgt, flow) for mod, (src, tgt, flow) in zip(self.attn, self.inputs)]
        if self.losses is not None:
            corr, reg = self.losses
            corr.source_vgg = {n: f for (n, _, _, _), f in zip(VGG, self.vgg)}
            corr.target_vgg = {n: f for (n, _, _, _), f in zip(VGG, self.vgg_target)}
            flows = [flow for (_, _, flow) in self.inputs]
            loss = sum(corr.calculate_loss(fl, n) for fl, (n, _, _, _) in zip(flows, VGG)) + 0.0025 * reg(flows)
            outs.append(loss.reshape(1))
        else:
            outs += [resample(feat, flow) for feat, (_, _, flow) in zip(self.vgg, self.inputs)]
        return self._backward(outs, allreduce)

    def _backward(self, outs, allreduce):
        if self.upstream is None:
            gen = torch.Generator(device=outs[0].device).manual_seed(4321)
            self.upstream = [torch.randn(o.shape, device=o.device, generator=gen) / o[0].numel() for o in outs]
        for t in [x for tup in self.inputs for x in tup] + self.vgg + self.params():
            t.grad = None
        if allreduce and self.reducer is None:  # bucketed all-reduce launched from autograd hooks (overlaps backward)
            self.reducer = gdist.GradBucketReducer(self.params())
        torch.autograd.backward(outs, self.upstream)
        if allreduce:
            self.reducer.finish()
        return outs


FACE_LAYERS = (  # (name, C, H, W, k): FaceGenerator at 256x256, layers=3, ngf=64 (generator.py:388-505)
    ("attn3", 256, 32, 32, 3),
    ("attn2", 128, 64, 64, 5),
)


class FacePath:
    """BASELINE config 5 (configs[4]): the hot path as FaceTargetNet runs it (generator.py:468-499) -- TWO ExtractorAttn
    per attention layer (previous frame / reference frame), blended into the decoder features by their masks -- for
    `frames` sequentially generated frames of each clip (generator.py:402-426), bf16 features.  Storage is bf16 end to
    end; the FC layers run in arithmetic mode 1 (one f16 term per operand: exact for bf16 values, f32 accumulation)."""

    def __init__(self, B, device, seed, frames=6, dual_stream=True):
        gen = torch.Generator(device=device).manual_seed(seed)
        self.B, self.device, self.frames, self.dual_stream = B, device, frames, dual_stream
        bf = torch.bfloat16
        torch.manual_seed(1234)
        self.attn = [tuple(gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(device) for _ in range(2))
                     for (_, C, H, W, k) in FACE_LAYERS]
        # attn_p / attn_r of a layer read the same decoder features and meet only in the blend: two HIP streams
        # (global_flow_local_attention_amd/face_step.py; generator.py:490-499)
        self.pairs = [gfla.DualStreamAttn(p, r, enabled=dual_stream) for (p, r) in self.attn]
        self.inputs = []
        for _ in range(frames):
            per = []
            for (_, C, H, W, k) in FACE_LAYERS:
                feat = lambda: torch.randn(B, C, H, W, device=device, generator=gen).to(bf).requires_grad_()
                prev, ref, dec = feat(), feat(), feat()
                flows = [smooth_flow(B, H, W, device, gen).to(bf).requires_grad_() for _ in range(2)]
                masks = [torch.rand(B, 1, H, W, device=device, generator=gen).to(bf) for _ in range(2)]
                per.append((prev, ref, dec, flows, masks))
            self.inputs.append(per)
        self.upstream = None
        self.reducer = None

    images_per_step = property(lambda self: self.B * self.frames)

    def params(self):
        return [p for pair in self.attn for m in pair for p in m.parameters()]

    def leaves(self):
        return [t for per in self.inputs for (prev, ref, dec, flows, _) in per for t in (prev, ref, dec, *flows)]

    def step(self, resample=None, allreduce=True):
        outs = []
        for per in self.inputs:  # frames are generated one after the other
            for pair, (prev, ref, dec, flows, masks) in zip(self.pairs, per):
                outs.append(pair(dec, prev, ref, flows[0], flows[1], masks[0], masks[1]))
        if self.upstream is None:
            gen = torch.Generator(device=outs[0].device).manual_seed(4321)
            self.upstream = [(torch.randn(o.shape, device=o.device, generator=gen) / o[0].numel()).to(o.dtype) for o in outs]
        for t in self.leaves() + self.params():
            t.grad = None
        if allreduce and self.reducer is None:
            self.reducer = gdist.GradBucketReducer(self.params())
        torch.autograd.backward(outs, self.upstream)
        if allreduce:
            self.reducer.finish()
        return outs

    def describe(self, args, world):
        return {
            "metric": "frames/sec (fwd+bwd) FaceGenerator 256x256 attn_layer=2,3, bf16 features -- feature-warping hot path",
            "dtype": "bf16 storage; f16 MFMA operands (exact for bf16 values) with f32 accumulation in the FC layers, f32 "
                     "arithmetic elsewhere",
            "config": {"workload": "GFLA hot path at FaceGenerator 256x256 shapes (BASELINE configs[4]): per generated "
                                   "frame, attn_p + attn_r ExtractorAttn at L3 (C256,32x32,k3) and L2 (C128,64x64,k5) "
                                   "fwd+bwd incl. both FC layers and the mask blend of FaceTargetNet.forward; %d frames "
                                   "generated sequentially per clip" % self.frames,
                       "clips_per_gpu": self.B, "frames_per_clip": self.frames,
                       "batch_per_gpu": self.B * self.frames, "global_batch": self.B * self.frames * world,
                       "parallelism": "dp%d (clips sharded; ExtractorAttn gradients all-reduced in one flat bucket launched "
                                      "from autograd hooks)" % world,
                       "fc_layers": "this library's MFMA kernels, arithmetic mode 1 (one f16 term per operand)"}}


class TrainerPath:
    """`--workload trainer_step` (BASELINE configs[3] on one rank's shard): one
    TrainerShell.optimize_parameters step (pose_model.py:186-196) of the in-repo generator-shaped network
    (tools/warp_generator.py: WarpGenerator at the production widths: ngf 64 -> ExtractorAttn L3 (C256, 1/8 scale, k3) and L2
    (C128, 1/4 scale, k5)) on a 256x176 batch: forward, L1 + sampling-correctness (frozen random VGG-shaped pyramid,
    max-cosine MFMA kernel, Resample2d, fused loss map) + affine-regularisation losses, backward with the bucketed
    reducer's hooks over ALL parameters, Adam step.  GAN / style terms stubbed, as configs[3] prescribes.  The stock
    convolutions / InstanceNorms of the network run through torch (MIOpen), like the reference's own."""

    def __init__(self, B, device, seed, fc_mode=5, ngf=64, size=(256, 176)):
        from global_flow_local_attention_amd.trainer import TrainerShell
        gen = torch.Generator(device=device).manual_seed(seed)
        self.B, self.device = B, device
        torch.manual_seed(1234)  # identical parameters on every rank
        import warp_generator
        self.net = warp_generator.WarpGenerator(3, 18, 3, ngf, flow_scale=8.0).to(device)
        for m in (self.net.attn3, self.net.attn2):
            m.fc_mode = fc_mode
        vgg = warp_generator.RandomFeaturePyramid(seed=11).to(device)
        self.shell = TrainerShell(self.net, lr=1e-4, correctness=gfla.PerceptualCorrectness(vgg=vgg),
                                  regularization=gfla.MultiAffineRegularizationLoss({"2": 5, "3": 3}), attn_layer=(2, 3))
        H, W = size
        rnd = lambda c: torch.rand(B, c, H, W, device=device, generator=gen)
        self.source, self.target = rnd(3) * 2 - 1, rnd(3) * 2 - 1
        self.source_B, self.target_B = rnd(18), rnd(18)
        self.size, self.ngf, self.fc_mode = size, ngf, fc_mode
        self.losses = {}

    images_per_step = property(lambda self: self.B)

    def params(self):
        return list(self.net.parameters())

    def step(self, resample=None, allreduce=True):
        self.losses = self.shell.optimize_parameters((self.source, self.source_B, self.target_B), self.target,
                                                     source=self.source)
        return self.losses

    def describe(self, args, world):
        nparam = sum(p.numel() for p in self.params())
        return {
            "metric": "images/sec, one generator training step (fwd + losses + bwd + Adam) of a generator-shaped network "
                      "around the GFLA hot path, 256x176",
            "dtype": "f32",
            "config": {"workload": "trainer_step: TrainerShell.optimize_parameters on WarpGenerator(ngf=%d) at %dx%d -- conv "
                                   "encoders, flow head, ExtractorAttn L3 (C%d,%dx%d,k3) + L2 (C%d,%dx%d,k5) with mask blend, "
                                   "decoder; losses L1 + PerceptualCorrectness (frozen random VGG-shaped pyramid) + "
                                   "MultiAffineRegularizationLoss; GAN/style terms stubbed (BASELINE configs[3]); Adam step"
                                   % (self.ngf, self.size[0], self.size[1], 4 * self.ngf, self.size[0] // 8, self.size[1] // 8,
                                      2 * self.ngf, self.size[0] // 4, self.size[1] // 4),
                       "batch_per_gpu": self.B, "global_batch": self.B * world, "parameters": nparam,
                       "parallelism": "dp%d (batch shards; ALL %d parameters (%.1f MB) all-reduced in %d bucket(s) launched from "
                                      "autograd hooks, overlapping backward)"
                                      % (world, nparam, nparam * 4 / 1e6, len(self.shell.reducer.buckets)),
                       "fc_layers": "this library's MFMA kernels, arithmetic mode %d; the network's stock convolutions / "
                                    "InstanceNorms run through torch (MIOpen), as the reference's do" % self.fc_mode,
                       "losses": {k: round(v, 5) for k, v in self.losses.items()}}}


# ---- algorithmic bytes per C-ABI call (4 bytes per fp32 element) -------
def algorithmic_bytes(name, a, esz=None):
    base, suffix = name.rsplit("_", 1)
    if esz is None:  # bf16 entries: 2 bytes per element (their float32 reduction outputs are counted at 2 as well)
        esz = {"bf16": 2, "f64": 8}.get(suffix, 4)
    if base == "gfla_block_extractor_unfold_fwd":
        base, a = "gfla_block_extractor_fwd", a[:10]
    if base == "gfla_block_extractor_unfold_bwd":
        base, a = "gfla_block_extractor_bwd", a[:12]
    if base == "gfla_local_attn_source_bwd":  # (src, flow, gunf, attn, gout, gs, gf, B, C, Hs, Ws, H, W, k, layout)
        B, C, Hs, Ws, H, W, k = a[7:14]
        n = B * C * Hs * Ws + 2 * B * H * W
        n += B * C * k * k * H * W if a[2] is not None else 0
        n += (B * k * k * H * W + B * C * H * W) if a[3] is not None else 0
        n += (B * C * Hs * Ws if a[5] is not None else 0) + (2 * B * H * W if a[6] is not None else 0)
        return esz * n
    if base == "gfla_block_extractor_fwd":
        B, C, Hs, Ws, Hf, Wf, k = a[3:10]
        return esz * (B * C * Hs * Ws + 2 * B * Hf * Wf + B * C * k * k * Hf * Wf)
    if base == "gfla_block_extractor_bwd":
        B, C, Hs, Ws, Hf, Wf, k = a[5:12]
        n = B * C * Hs * Ws + 2 * B * Hf * Wf + B * C * k * k * Hf * Wf   # reads
        if a[3] is not None:
            n += B * C * Hs * Ws
        if a[4] is not None:
            n += 2 * B * Hf * Wf
        return esz * n
    if base in ("gfla_local_attn_reshape_fwd", "gfla_local_attn_reshape_bwd"):
        B, H, W, k = a[2:6]
        return 2 * esz * B * k * k * H * W
    if base == "gfla_resample2d_fwd":
        B, C, Hi, Wi, H, W = a[3:9]
        return esz * (B * C * Hi * Wi + B * C * H * W + 3 * B * H * W)
    if base == "gfla_resample2d_bwd":
        B, C, Hi, Wi, H, W = a[5:11]
        n = B * C * H * W + 3 * B * H * W
        if a[3] is not None:
            n += B * C * Hi * Wi
        if a[4] is not None:
            n += B * C * Hi * Wi + 3 * B * H * W
        return esz * n
    if base == "gfla_local_attn_aggregate_fwd":
        B, C, Hs, Ws, H, W, k = a[5:12]
        n = B * C * Hs * Ws + 2 * B * H * W + B * k * k * H * W + B * C * H * W
        if a[4] is not None:
            n += B * k * k * H * W
        return esz * n
    if base == "gfla_local_attn_aggregate_bwd":
        B, C, Hs, Ws, H, W, k = a[7:14]
        n = B * C * Hs * Ws + 2 * B * H * W + B * k * k * H * W + B * C * H * W
        n += (B * C * Hs * Ws if a[4] is not None else 0) + (2 * B * H * W if a[5] is not None else 0)
        n += (B * k * k * H * W if a[6] is not None else 0)
        return esz * n
    if base == "gfla_replicate_pad_bwd":  # (grad_padded, grad_in, planes, H, W, l, r, t, b)
        planes, H, W, l, r, t, b_ = a[2:9]
        return esz * planes * ((H + t + b_) * (W + l + r) + H * W)
    if base == "gfla_fc_tail_fwd":  # (hs, sb, so, ht, b0, w1, b1, logits, B, Hc, HW, KK, slope)
        B, Hc, HW, KK = a[8:12]
        return esz * (2 * B * Hc * HW + B * KK * HW)
    if base == "gfla_fc_tail_bwd":  # (hs, sb, so, ht, b0, w1, gl, g_hs, g_ht, act, partials, B, Hc, HW, KK, slope)
        B, Hc, HW, KK = a[11:15]
        writes = 1 + (a[8] is not None) + (a[9] is not None)
        return esz * ((2 + writes) * B * Hc * HW + B * KK * HW)
    if base == "gfla_local_attn_aggregate_fwd_ws":  # (s, f, l, o, a, ws, B, C, Hs, Ws, H, W, k, sm)
        return algorithmic_bytes("gfla_local_attn_aggregate_fwd_" + suffix, a[:5] + a[6:], esz)
    if base == "gfla_local_attn_aggregate_bwd_ws":  # (s, f, a, go, gs, gf, gl, ws, B, C, Hs, Ws, H, W, k, sm)
        return algorithmic_bytes("gfla_local_attn_aggregate_bwd_f32", a[:7] + a[8:], esz)
    if base == "gfla_resample2d_bwd_ws":  # (in1, in2, go, g1, g2, ws, B, C, Hi, Wi, H, W, k, d, trunc)
        return algorithmic_bytes("gfla_resample2d_bwd_f32", a[:5] + a[6:], esz)
    if base == "gfla_fc_forward":  # (src, tgt, flow, w0, b0, w1, b1, ws, logits, B, C, H, W, k, slope, mode)
        B, C, H, W, k = a[9:14]
        return esz * (2 * B * C * H * W + 2 * B * H * W + B * k * k * H * W + 128 * (2 * C * k * k + k * k))
    if base == "gfla_fc_backward":  # (ws, flow, w1, gl, scratch, gs, gt, gf, gw0, gb0, gw1, gb1, B, C, H, W, k, slope, mode, flags)
        B, C, H, W, k = a[12:17]
        n = 2 * B * C * H * W + 2 * B * H * W + B * k * k * H * W  # saved inputs + grad_logits
        n += (B * C * H * W if a[5] is not None else 0) + (B * C * H * W if a[6] is not None else 0)
        n += (2 * B * H * W if a[7] is not None else 0) + 2 * 128 * (2 * C * k * k + k * k)
        return esz * n
    return 0


def algorithmic_flops(name, a):
    """FLOPs of the reference's formulation of an FC-path call (base_function.py:799-807): conv0 is 2C*k*k -> 128 per
    position, conv1 128 -> k*k; backward = data gradient + weight gradient of both.  0 for the memory-bound entries."""
    base = name.rsplit("_", 1)[0]
    if base == "gfla_fc_forward":
        B, C, H, W, k = a[9:14]
        return 2 * B * H * W * 128 * (2 * C * k * k + k * k)
    if base == "gfla_fc_backward":
        B, C, H, W, k = a[12:17]
        return 2 * 2 * B * H * W * 128 * (2 * C * k * k + k * k)
    return 0


def pmc_traffic(entry, dims, ptrs=""):
    """HBM bytes per launch of the kernels behind one C-ABI call, from the committed rocprofv3 PMC
    passes (profiles/pmc_traffic.json, made by tools/pmc_summary.py from separate --pmc FETCH_SIZE
    / --pmc WRITE_SIZE runs of this bench; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
    correction, so this is an upper bound).  None if no profile matches."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    table = {k: v for k, v in json.load(open(path)).items() if not k.startswith("_")}
    base = entry.rsplit("_", 1)[0]
    B, C = dims[0], dims[1]

    def pick(name, want_write_bytes=None):
        rows = [r for key, rs in table.items() if key.startswith(name) for r in rs]
        if not rows:
            return None
        if want_write_bytes is None or len(rows) == 1:
            return rows[0]
        return min(rows, key=lambda r: abs(r["WRITE_SIZE_KiB"] * 1024 - want_write_bytes))

    names = None
    if base == "gfla_block_extractor_unfold_fwd":
        names = [("be_unfold_fwd_lds_kernel<float, %d>" % dims[6], None)]
    elif base == "gfla_block_extractor_unfold_bwd":
        names = [("be_bwd_lds_kernel<float, %d, true, true, 2" % dims[6], None)]
    elif base == "gfla_block_extractor_bwd":
        names = [("be_bwd_lds_kernel<float, %d, true, true, 0" % dims[6], None)]
    elif base == "gfla_local_attn_source_bwd":
        names = [("be_bwd_lds_kernel<float, %d, true, true, 3" % dims[6], None)]
    elif base == "gfla_local_attn_aggregate_fwd":
        names = [("agg_fwd_lds_kernel<float, %d>" % dims[6], None)]
    elif base == "gfla_local_attn_aggregate_bwd":
        k = dims[6]
        names = [("agg_ga_lds_kernel<float, %d>" % k, None), ("agg_softmax_bwd_kernel<float, %d>" % k, None)]
        if ptrs[4:6] != "00":  # grad_source / grad_flow computed here (not parked for the fused pass)
            names.append(("be_bwd_lds_kernel<float, %d, true, true, 1" % k, None))
    elif base == "gfla_resample2d_fwd":
        names = [("rs_lds_kernel<float, %d, 0" % (dims[6] // 2), 4 * B * C * dims[4] * dims[5])]
    elif base == "gfla_resample2d_bwd":
        kh = dims[6] // 2
        names = []
        if ptrs[3:4] == "1":
            names.append(("rs_lds_kernel<float, %d, 1" % kh, 4 * B * C * dims[2] * dims[3]))
        if ptrs[4:5] == "1":
            names.append(("rs_lds_kernel<float, %d, 2" % kh, 4 * 3 * B * dims[4] * dims[5]))
    if not names:
        return None
    total = 0
    for name, want in names:
        row = pick(name, want)
        if row is None:
            return None
        total += row["traffic_bytes"]
    return total


def pmc_table_provenance():
    """Which counter run profiles/pmc_traffic.json came from (tools/pmc_summary.py stamps it)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return "absent"
    meta = json.load(open(path)).get("_meta")
    if not meta:
        return "unstamped (made before round 5)"
    return "%s, generated %s" % (meta.get("tag") or "untagged", meta.get("generated_utc"))


def pmc_traffic_kernel(kernel_label, in_step=False):
    """HBM bytes per launch of one FC kernel from the committed PMC profile (profiles/pmc_traffic.json), matched by the
    kernel template and the layer's kernel size; when several launches match (source / target half, forward / data
    gradient of the convolution kernel) the LARGEST is reported, i.e. an upper bound for the probed launch.
    in_step: the mean of the two largest grids = the two two-job launches of the step (forward, data gradient)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    table = {k: v for k, v in json.load(open(path)).items() if not k.startswith("_")}
    name = kernel_label.split("<")[0]
    m = re.search(r"k (\d+)>", kernel_label)
    k = m.group(1) if m else None
    rows = []
    for key, rs in table.items():
        if not key.startswith(name + "<"):
            continue
        args = [x.strip() for x in key[len(name) + 1:].rstrip(">").split(",")]
        # fc_conv_kernel<MODE, KS, NMB>, fc_wgrad_f32_kernel<KS>, fc_wgrad_kernel<MODE, KS, NB>,
        # fc_wino_conv_kernel<KS, DBG, DB>, fc_wino_wgrad_kernel<KS>
        ks = args[0] if (len(args) == 1 or name.startswith("fc_wino")) else args[1]
        if name == "fc_wino16_wgrad_kernel":   # <multi-row, DBG>: the k = 5 layer only
            ks = "5"
        if k is None or ks == k:
            rows += rs
    if not rows:
        return None
    if in_step and len(rows) >= 2 and all("grid" in r for r in rows):
        top = sorted(rows, key=lambda r: -int(r["grid"]))[:2]
        return int(sum(r["traffic_bytes"] for r in top) / 2)
    return max(r["traffic_bytes"] for r in rows)


class KernelTimer:
    """Brackets every C-ABI call with HIP events on the stream the kernels are launched on."""

    def __init__(self):
        self.records = []
        self._orig = None

    def __enter__(self):
        self._orig = _lib.call

        def timed(name, ref_tensor, *args):
            stream = torch.cuda.current_stream(ref_tensor.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            self._orig(name, ref_tensor, *args)
            e1.record(stream)
            plain = tuple(x if isinstance(x, (int, float)) else (None if (x is None or x.value is None) else "ptr")
                          for x in args)
            self.records.append((name, plain, e0, e1))

        _lib.call = timed  # the op modules look `_lib.call` up at call time
        return self

    def __exit__(self, *exc):
        _lib.call = self._orig
        return False

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, args, e0, e1 in self.records:
            ptrs = "".join("0" if x is None else "1" for x in args if not isinstance(x, int))
            key = (name, ptrs) + tuple(x for x in args if isinstance(x, int))
            ent = agg.setdefault(key, {"name": name, "ptrs": ptrs, "calls": 0, "ms": 0.0,
                                       "bytes": algorithmic_bytes(name, args), "flops": algorithmic_flops(name, args)})
            ent["calls"] += 1
            ent["ms"] += e0.elapsed_time(e1)
        rows = []
        for key, ent in agg.items():
            avg_ms = ent["ms"] / ent["calls"]
            gbs = ent["bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            row = {"entry": ent["name"], "dims": list(key[2:]), "ptrs": ent["ptrs"], "calls": ent["calls"],
                   "avg_us": round(avg_ms * 1e3, 2), "total_ms": round(ent["ms"], 3),
                   "alg_MB": round(ent["bytes"] / 1e6, 3), "GBps": round(gbs, 1),
                   "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
            if ent["flops"]:
                tf = ent["flops"] / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
                # reference-formulation flops / time of a whole C-ABI call: what the caller gets, NOT a pipe fraction (the
                # Winograd-domain kernels execute fewer multiplies than the reference formulation, so this can pass 1)
                row.update({"effective_GFLOP": round(ent["flops"] / 1e9, 2), "effective_TFLOPs": round(tf, 1),
                            "effective_frac_vs_f32_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4)})
            rows.append(row)
        rows.sort(key=lambda r: -r["total_ms"])
        return rows


def fc_kernel_probes(hp, iters=10):
    """Each internal kernel of the FC path timed alone (HIP events around gfla_fc_kernel_f32 on the launch stream), on
    the state one forward + backward of the bench's own inputs leaves behind.  FLOPs: the reference formulation's
    (2 * B*H*W * C*k*k * 128 per half and pass) -- the kernels multiply somewhat more (the source half's map is
    (H+k-1) x (W+k-1), the data gradient runs on the padded domain), which is NOT counted as achieved work."""
    from global_flow_local_attention_amd import fc_mfma
    names = ("conv fwd source", "conv fwd target", "data-grad source", "data-grad target", "weight-grad source",
             "weight-grad target", "conv fwd source + target (one launch, as in the step)",
             "data-grad source + target (one launch, as in the step)",
             "weight-grad source + target (one launch, as in the step)")
    rows = []
    for mod, (src, tgt, flow) in zip(hp.attn, hp.inputs):
        mode = getattr(mod, "fc_mode", None)
        mode = fc_mfma.DEFAULT_MODE if mode is None else int(mode)
        B, C, H, W = src.shape
        k = mod.kernel_size
        if getattr(mod, "fc_impl", "mfma") != "mfma" or not fc_mfma.supported(C, H, W, k, mode):
            continue
        fc = mod.fully_connect_layer
        with torch.no_grad():
            s, t, f = src.detach().contiguous(), tgt.detach().contiguous(), flow.detach().contiguous()
            w0, w1 = fc[0].weight.detach().contiguous(), fc[2].weight.detach().reshape(k * k, 128).contiguous()
            ws = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 0), dtype=torch.uint8, device=s.device)
            sc = torch.empty(fc_mfma.workspace_bytes(B, C, H, W, k, mode, 1), dtype=torch.uint8, device=s.device)
            logits = s.new_empty(B, k * k, H, W)
            gl = torch.randn_like(logits) * 1e-3
            gs, gt, gf = torch.empty_like(s), torch.empty_like(t), torch.empty_like(f)
            gw0 = torch.empty_like(w0)
            _lib.call("gfla_fc_forward_f32", s, _lib.ptr(s), _lib.ptr(t), _lib.ptr(f), _lib.ptr(w0), _lib.ptr(fc[0].bias),
                      _lib.ptr(w1), _lib.ptr(fc[2].bias), _lib.ptr(ws), _lib.ptr(logits), B, C, H, W, k, 0.1, mode)
            _lib.call("gfla_fc_backward_f32", s, _lib.ptr(ws), _lib.ptr(f), _lib.ptr(w1), _lib.ptr(gl), _lib.ptr(sc),
                      _lib.ptr(gs), _lib.ptr(gt), _lib.ptr(gf), _lib.ptr(gw0), None, None, None, B, C, H, W, k, 0.1, mode, 0)
            stream = torch.cuda.current_stream(s.device)
            flops = 2.0 * B * H * W * C * k * k * 128
            layer_rows = []
            # mode 5 (csrc/fc_block.hip: fc_hyb): the k = 5 convolutions and every data gradient run on the DIRECT kernels with
            # two f16 terms per operand and three cross products, reading the float32 maps in place; tuning key 52 = 1: Winograd
            import global_flow_local_attention_amd as _g
            _old52 = _g.set_tuning(52, 0)
            _g.set_tuning(52, _old52)
            hyb = mode == 5 and _old52 != 1
            geo = {True: fc_mfma.geometry(H, W, k, True), False: fc_mfma.geometry(H, W, k, False)}
            for which, nm in enumerate(names):
                if which > 5 and mode not in (4, 5):
                    continue
                for _ in range(2):
                    _lib.call("gfla_fc_kernel_f32", s, which, _lib.ptr(ws), _lib.ptr(sc), B, C, H, W, k, mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(iters):
                    _lib.call("gfla_fc_kernel_f32", s, which, _lib.ptr(ws), _lib.ptr(sc), B, C, H, W, k, mode)
                e1.record(stream)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / iters * 1e3
                row = {"dims": [B, C, H, W, k], "avg_us": round(us, 1)}
                direct16 = hyb and ((which in (0, 1, 6) and k == 5) or which in (2, 3, 7))
                if direct16 and which < 4:
                    # direct convolution: the kernel executes k*k multiplies per (output row, c, n) over the rows of ITS domain --
                    # the convolved map (Ho x Wo) forward, the padded domain (Hp x Wp) for the data gradient
                    g = geo[which % 2 == 0]
                    rows_out = g["Ho"] * g["Wo"] if which < 2 else g["Hp"] * g["Wp"]
                    done = 2.0 * B * rows_out * C * k * k * 128
                    row.update({"alg_GFLOP": round(done / 1e9, 2), "TFLOPs": round(done / (us * 1e-6) / 1e12, 1),
                                "useful_GFLOP": round(flops / 1e9, 2), "effective_GFLOP": round(flops / 1e9, 2),
                                "effective_TFLOPs": round(flops / (us * 1e-6) / 1e12, 1), "in_step": True, "launches": 1})
                    kern = "fc_conv_kernel<f16x2 from f32>"
                elif direct16:   # which 6 / 7: the two launches above, back to back (the direct kernels take one half each)
                    halves = layer_rows[0:2] if which == 6 else layer_rows[2:4]
                    done = sum(r["alg_GFLOP"] for r in halves) * 1e9
                    eff = sum(r["effective_GFLOP"] for r in halves) * 1e9
                    row.update({"alg_GFLOP": round(done / 1e9, 2), "TFLOPs": round(done / (us * 1e-6) / 1e12, 1),
                                "useful_GFLOP": round(sum(r["useful_GFLOP"] for r in halves), 2),
                                "effective_GFLOP": round(eff / 1e9, 2), "effective_TFLOPs": round(eff / (us * 1e-6) / 1e12, 1),
                                "launches": 2, "note": "the two launches of the rows above, back to back"})
                    kern = "fc_conv_kernel<f16x2 from f32>"
                elif which > 5:   # two jobs in one launch: the sums of the two halves' rows above
                    halves = layer_rows[0:2] if which == 6 else (layer_rows[2:4] if which == 7 else layer_rows[4:6])
                    done = sum(r["alg_GFLOP"] for r in halves) * 1e9
                    eff = sum(r["effective_GFLOP"] for r in halves) * 1e9
                    kern = ("fc_wino16_wgrad_kernel" if mode == 5 and k == 5 else "fc_wino_wgrad_kernel") if which == 8 else \
                        ("fc_wino_conv_kernel" if mode == 4 else "fc_wino16_conv_kernel")
                    row.update({"alg_GFLOP": round(done / 1e9, 2), "TFLOPs": round(done / (us * 1e-6) / 1e12, 1),
                                "useful_GFLOP": round(sum(r["useful_GFLOP"] for r in halves), 2),
                                "effective_GFLOP": round(eff / 1e9, 2), "effective_TFLOPs": round(eff / (us * 1e-6) / 1e12, 1),
                                "in_step": True})
                elif mode in (4, 5):   # (the weight gradients of both layers run in the Winograd domain too: csrc/fc_block.hip)
                    # Winograd domain: the kernel EXECUTES 36 multiplies per (tile, c, n) -- F(2x2,5x5): 2x2 outputs per
                    # tile, F(4x4,3x3): 4x4.  `TFLOPs` / `frac` are these executed MFMA flops against the f32 peak (what the
                    # hardware does); `effective_TFLOPs` = the reference formulation's flops / time (what the caller gets).
                    kern = ("fc_wino_conv_kernel" if mode == 4 else "fc_wino16_conv_kernel") if which < 4 else "fc_wino_wgrad_kernel"
                    m = 2 if k == 5 else 4
                    ext = {0: k - 1, 1: 0, 2: 2 * (k - 1), 3: k - 1, 4: k - 1, 5: 0}[which]
                    tiles = B * (-(-(H + ext) // m)) * (-(-(W + ext) // m))
                    done = 2.0 * 36 * tiles * C * 128
                    # `useful`: the same count over the UN-extended H x W domain -- the tiles whose outputs the caller keeps
                    # (the source half convolves an (H+k-1) x (W+k-1) map, the data gradients the padded domain)
                    useful = 2.0 * 36 * B * (-(-H // m)) * (-(-W // m)) * C * 128
                    row.update({"alg_GFLOP": round(done / 1e9, 2), "TFLOPs": round(done / (us * 1e-6) / 1e12, 1),
                                "useful_GFLOP": round(useful / 1e9, 2),
                                "effective_GFLOP": round(flops / 1e9, 2),
                                "effective_TFLOPs": round(flops / (us * 1e-6) / 1e12, 1)})
                else:
                    kern = "fc_conv_kernel" if which < 4 else ("fc_wgrad_f32_kernel" if mode in (0, 4) else "fc_wgrad_kernel")
                    row.update({"alg_GFLOP": round(flops / 1e9, 2), "TFLOPs": round(flops / (us * 1e-6) / 1e12, 1)})
                # mode 5: which convolutions run on the two-term f16 kernel (csrc/fc_block.hip: fc_w16_dgrad)
                # ... and the weight gradient of the k = 5 layer (both halves in one launch: csrc/fc_block.hip: fc_w16_wgrad)
                w16 = mode == 5 and (which in (0, 1, 6) or (which in (2, 3, 7) and k == 5) or (which == 8 and k == 5)) and not direct16
                if mode == 5 and not w16 and kern == "fc_wino16_conv_kernel":
                    kern = "fc_wino_conv_kernel"
                if mode == 5 and hyb and not direct16 and which < 4:
                    row["in_step"] = False   # (not what the step issues in this configuration: a reference row)
                row["kernel"] = "%s<mode %d, k %d>: %s" % (kern, mode, k, nm)
                row["frac_mfma_f32_peak"] = round(row["TFLOPs"] / MFMA_F32_PEAK_TFLOPS, 4)
                if w16 or direct16:
                    # Winograd-domain kernels: every multiply is formed from two f16 terms per operand, all four cross products --
                    # 4 f16 MACs per f32-equivalent one (two v_mfma_f32_32x32x16_f16 per 8 channels); direct kernels: three
                    row["pipe"] = "f16"
                    row["f16_macs_per_mac"] = 3 if direct16 else 4
                    row["f16_pipe_TFLOPs"] = round(row["f16_macs_per_mac"] * row["alg_GFLOP"] * 1e9 / (us * 1e-6) / 1e12, 1)
                    row["frac_mfma_f16_peak"] = round(row["f16_pipe_TFLOPs"] / MFMA_F16_PEAK_TFLOPS, 4)
                    row["frac_mfma_f32_peak_note"] = "f32-EQUIVALENT Winograd-domain flops / time / f32 peak: not a pipe fraction in this mode"
                rows.append(row)
                layer_rows.append(row)
    return rows


def op_roofline(device, B=32, iters=20, layers=None, flow_kind="smooth"):
    """The north star's own figure (BASELINE.json: ">= 60 % of the HBM3E peak on block_extractor + local-attn forward at
    256x176, attn_layer=2,3"), measured op by op through the C ABI at the attention-layer shapes of the headline:
    `gfla_block_extractor_fwd_f32` in the REFERENCE layout (B,C,kH,kW) (block_extractor_kernel.cu:20-85) and
    `gfla_local_attn_aggregate_fwd_ws_f32` (softmax + reshape + multiply + avg-pool of base_function.py:803-809), each
    HIP-event timed over `iters` back-to-back launches on the launch stream; achieved = SURVEY 8(d)'s algorithmic bytes /
    time, frac = achieved / 8 TB/s, per op and for the pair (sum of bytes / sum of times)."""
    out = {"peak_GBps": HBM_PEAK_GBS, "flow": flow_kind, "batch": B, "layers": {},
           "what": "block_extractor forward (reference layout) + local-attention forward (softmax/aggregate), "
                   "algorithmic bytes (SURVEY 8d) / HIP-event time / 8 TB/s; `pair` = both ops, sum of bytes / sum of times"}
    gen = torch.Generator(device=device).manual_seed(77)
    stream = torch.cuda.current_stream(device)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3  # us

    for (name, C, H, W, k) in (layers or LAYERS):
        src = torch.randn(B, C, H, W, device=device, generator=gen)
        flow = (smooth_flow(B, H, W, device, gen) if flow_kind == "smooth"
                else torch.zeros(B, 2, H, W, device=device))
        logits = torch.randn(B, k * k, H, W, device=device, generator=gen)
        blocks = torch.empty(B, C, k * H, k * W, device=device)
        agg, attn = torch.empty_like(src), torch.empty_like(logits)
        n = _lib.lib().gfla_aggregate_fwd_workspace_bytes(B, H, W, k)
        scratch = torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)
        be_args = (_lib.ptr(src), _lib.ptr(flow), _lib.ptr(blocks), B, C, H, W, H, W, k)
        ag_args = (_lib.ptr(src), _lib.ptr(flow), _lib.ptr(logits), _lib.ptr(agg), _lib.ptr(attn), _lib.ptr(scratch),
                   B, C, H, W, H, W, k, 1)
        t_be = timed(lambda: _lib.call("gfla_block_extractor_fwd_f32", src, *be_args))
        t_ag = timed(lambda: _lib.call("gfla_local_attn_aggregate_fwd_ws_f32", src, *ag_args))
        b_be = algorithmic_bytes("gfla_block_extractor_fwd_f32", (1, 1, 1, B, C, H, W, H, W, k))
        b_ag = algorithmic_bytes("gfla_local_attn_aggregate_fwd_ws_f32", (1, 1, 1, 1, 1, 1, B, C, H, W, H, W, k, 1))

        def row(nbytes, us):
            gbs = nbytes / (us * 1e-6) / 1e9
            return {"alg_MB": round(nbytes / 1e6, 2), "us": round(us, 1), "GBps": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}
        out["layers"][name] = {"dims": [B, C, H, W, k], "block_extractor_fwd": row(b_be, t_be),
                               "local_attn_fwd": row(b_ag, t_ag), "pair": row(b_be + b_ag, t_be + t_ag)}
        del src, flow, logits, blocks, agg, attn, scratch
    return out


def config2_ops(device, iters=20, flows=("smooth", "zero", "wild", "integer", "near_integer", "oob"), with_ref=True, split=False):
    """BASELINE configs[1]: block_extractor (k = 3, 5) and resample2d(4, 1) forward / backward on ONE (1, 64, 256, 176) fp32
    feature map, through the C ABI, HIP-event timed on the launch stream; SURVEY 8(d)'s algorithmic bytes / time / 8 TB/s.
    Where the real reference kernels are present (oracle/_ref: the checker, never the thing measured) the same call is
    timed on them (`ref_us`) and, on the smooth flow, compared in float64 (`max_abs_vs_ref`; the north star's bar is 1e-4)."""
    B, C, H, W = 1, 64, 256, 176
    ref = None
    if with_ref:
        try:
            from oracle import ref_ext
            ref = ref_ext if ref_ext.available() else None
        except Exception:
            ref = None
    stream = torch.cuda.current_stream(device)

    def timed(fn, n):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3  # us

    def flow_of(kind, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        if kind == "zero":
            return torch.zeros(B, 2, H, W, device=device)
        if kind == "smooth":
            return smooth_flow(B, H, W, device, g)
        n = torch.randn(B, 2, H, W, device=device, generator=g)
        if kind == "wild":
            return n * 8
        if kind == "integer":
            return torch.round(n * 3)
        if kind == "near_integer":
            sign = torch.where(torch.randn(B, 2, H, W, device=device, generator=g) > 0, 1.0, -1.0)
            return (torch.round(n * 3) + sign * 2.0 ** -22).contiguous()
        if kind in ("expand", "compress"):   # probes (tools/bench_config2.py): a flow that spreads / squeezes the sample points
            sgn = 0.2 if kind == "expand" else -0.2
            xs = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, 1, W) - W / 2
            ys = torch.arange(H, device=device, dtype=torch.float32).view(1, 1, H, 1) - H / 2
            return torch.cat((sgn * xs.expand(B, 1, H, W), sgn * ys.expand(B, 1, H, W)), 1).contiguous() + 0.37
        f = n * 2                      # oob: every sample far outside the map, through all four sides
        f[:, 0] += 1000.0
        f[:, 1] -= 1000.0
        return f.contiguous()

    def floors_agree(flow, k):
        """flow pixels where (flow + offset) + index floors identically in float32 and float64: elsewhere d/dflow is
        one-sided and a float32 kernel legitimately takes the other side than the float64 reference (a handful of pixels)"""
        ys = torch.arange(H, device=device).view(1, H, 1)
        xs = torch.arange(W, device=device).view(1, 1, W)
        ok = torch.ones(B, H, W, dtype=torch.bool, device=device)
        for t in range(k):
            o = float(t - k // 2)
            for ch, idx in ((0, xs), (1, ys)):
                ok &= torch.floor((flow[:, ch] + o) + idx.float()).double() == torch.floor((flow[:, ch].double() + o) + idx.double())
        return ok.unsqueeze(1)

    rows = []

    def emit(op, kind, entry, plain, us, ref_us, err):
        nbytes = algorithmic_bytes(entry, plain)
        gbs = nbytes / (us * 1e-6) / 1e9
        row = {"op": op, "flow": kind, "alg_MB": round(nbytes / 1e6, 2), "us": round(us, 1), "GBps": round(gbs, 1),
               "frac": round(gbs / HBM_PEAK_GBS, 4)}
        if ref_us is not None:
            row["ref_us"] = round(ref_us, 1)
            row["speedup_vs_ref"] = round(ref_us / us, 2)
        if err is not None:
            # err = (max |got - ref|, that / max |ref|) over the call's outputs: the north star's 1e-4 is a bar on O(1) forward
            # values; gradients here have magnitudes of 40-100, so the RELATIVE figure is the one gated (tests: 2e-5)
            row["max_abs_vs_ref"], row["max_rel_vs_ref"] = float("%.3g" % err[0]), float("%.3g" % err[1])
        rows.append(row)

    def diff(pairs):
        """(max-abs, max-abs / max |reference|), the worse of each over (got, want[, mask]) pairs"""
        worst_abs = worst_rel = 0.0
        for got, want, *mask in pairs:
            d = (got.double() - want).abs()
            if mask:
                d = d * mask[0]
            a = d.max().item()
            worst_abs, worst_rel = max(worst_abs, a), max(worst_rel, a / max(1e-30, want.abs().max().item()))
        return worst_abs, worst_rel

    gen = torch.Generator(device=device).manual_seed(42)
    src = torch.randn(B, C, H, W, device=device, generator=gen)
    for k in (3, 5):
        out = torch.empty(B, C, k * H, k * W, device=device)
        gout = torch.randn(B, C, k * H, k * W, device=device, generator=gen)
        for kind in flows:
            flow = flow_of(kind, 7)
            gs, gf = torch.zeros_like(src), torch.zeros_like(flow)
            fwd = lambda: _lib.call("gfla_block_extractor_fwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(out), B, C, H, W, H, W, k)
            bwd = lambda: _lib.call("gfla_block_extractor_bwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(gout), _lib.ptr(gs),
                                    _lib.ptr(gf), B, C, H, W, H, W, k)
            t_f, t_b = timed(fwd, iters), timed(bwd, max(3, iters // 2))
            r_f = r_b = e_f = e_b = None
            if ref is not None:
                m = ref._mod("block_extractor_cuda")
                ro, rgs, rgf = torch.empty_like(out), torch.zeros_like(src), torch.zeros_like(flow)
                r_f = timed(lambda: m.forward(src, flow, ro, k), 5)
                r_b = timed(lambda: m.backward(src, flow, gout, rgs, rgf, k), 2)
                if kind == "smooth":
                    sd, fd = src.double(), flow.double()
                    e_f = diff([(out, ref.block_extractor_fwd(sd, fd, k))])
                    gs.zero_(), gf.zero_()
                    bwd()
                    ws, wf = ref.block_extractor_bwd(sd, fd, gout.double(), k)
                    keep = floors_agree(flow, k)
                    e_b = diff([(gs, ws), (gf, wf, keep)])
                    del sd, fd, ws, wf
                del ro, rgs, rgf
            emit("block_extractor_fwd k%d" % k, kind, "gfla_block_extractor_fwd_f32", (1, 1, 1, B, C, H, W, H, W, k), t_f, r_f, e_f)
            emit("block_extractor_bwd k%d" % k, kind, "gfla_block_extractor_bwd_f32", (1, 1, 1, 1, 1, B, C, H, W, H, W, k), t_b, r_b, e_b)
            if split:   # one gradient at a time (tools/bench_config2.py --split)
                t_s = timed(lambda: _lib.call("gfla_block_extractor_bwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(gout), _lib.ptr(gs),
                                              None, B, C, H, W, H, W, k), max(3, iters // 2))
                t_w = timed(lambda: _lib.call("gfla_block_extractor_bwd_f32", src, _lib.ptr(src), _lib.ptr(flow), _lib.ptr(gout), None,
                                              _lib.ptr(gf), B, C, H, W, H, W, k), max(3, iters // 2))
                emit("block_extractor_bwd k%d (source only)" % k, kind, "gfla_block_extractor_bwd_f32", (1, 1, 1, 1, None, B, C, H, W, H, W, k), t_s, None, None)
                emit("block_extractor_bwd k%d (flow only)" % k, kind, "gfla_block_extractor_bwd_f32", (1, 1, 1, None, 1, B, C, H, W, H, W, k), t_w, None, None)
        del out, gout
        torch.cuda.empty_cache()
    gout = torch.randn(B, C, H, W, device=device, generator=gen)
    out = torch.empty_like(src)
    for kind in flows:
        i2 = torch.cat((flow_of(kind, 9), torch.full((B, 1, H, W), 2.0, device=device)), 1).contiguous()
        g1, g2 = torch.zeros_like(src), torch.zeros_like(i2)
        fwd = lambda: _lib.call("gfla_resample2d_fwd_f32", src, _lib.ptr(src), _lib.ptr(i2), _lib.ptr(out), B, C, H, W, H, W, 4, 1)
        bwd = lambda: _lib.call("gfla_resample2d_bwd_f32", src, _lib.ptr(src), _lib.ptr(i2), _lib.ptr(gout), _lib.ptr(g1), _lib.ptr(g2),
                                B, C, H, W, H, W, 4, 1, 1)
        t_f, t_b = timed(fwd, iters), timed(bwd, max(3, iters // 2))
        r_f = r_b = e_f = e_b = None
        if ref is not None:
            m = ref._mod("resample2d_cuda")
            ro, r1, r2 = torch.empty_like(out), torch.zeros_like(src), torch.zeros_like(i2)
            r_f = timed(lambda: m.forward(src, i2, ro, 4, 1), 5)
            r_b = timed(lambda: m.backward(src, i2, gout, r1, r2, 4, 1), 2)
            if kind == "smooth":
                e_f = diff([(out, ref.resample2d_fwd(src.double(), i2.double(), 4, 1))])
                g1.zero_(), g2.zero_()
                bwd()
                w1, w2 = ref.resample2d_bwd(src.double(), i2.double(), gout.double(), 4, 1)
                e_b = diff([(g1, w1), (g2, w2)])
                del w1, w2
            del ro, r1, r2
        emit("resample2d_fwd k4", kind, "gfla_resample2d_fwd_f32", (1, 1, 1, B, C, H, W, H, W, 4, 1), t_f, r_f, e_f)
        emit("resample2d_bwd k4", kind, "gfla_resample2d_bwd_f32", (1, 1, 1, 1, 1, B, C, H, W, H, W, 4, 1, 1), t_b, r_b, e_b)
        if split:
            t_1 = timed(lambda: _lib.call("gfla_resample2d_bwd_f32", src, _lib.ptr(src), _lib.ptr(i2), _lib.ptr(gout), _lib.ptr(g1), None,
                                          B, C, H, W, H, W, 4, 1, 1), max(3, iters // 2))
            t_2 = timed(lambda: _lib.call("gfla_resample2d_bwd_f32", src, _lib.ptr(src), _lib.ptr(i2), _lib.ptr(gout), None, _lib.ptr(g2),
                                          B, C, H, W, H, W, 4, 1, 1), max(3, iters // 2))
            emit("resample2d_bwd k4 (input1 only)", kind, "gfla_resample2d_bwd_f32", (1, 1, 1, 1, None, B, C, H, W, H, W, 4, 1, 1), t_1, None, None)
            emit("resample2d_bwd k4 (input2 only)", kind, "gfla_resample2d_bwd_f32", (1, 1, 1, None, 1, B, C, H, W, H, W, 4, 1, 1), t_2, None, None)
    torch.cuda.empty_cache()
    slower = [r for r in rows if "ref_us" in r and r["us"] > r["ref_us"]]
    rels = [r["max_rel_vs_ref"] for r in rows if "max_rel_vs_ref" in r]
    bad = [(r["op"], r["max_abs_vs_ref"], r["max_rel_vs_ref"]) for r in rows
           if r.get("max_rel_vs_ref", 0.0) > 1e-4 or ("fwd" in r["op"] and r.get("max_abs_vs_ref", 0.0) > 1e-4)]
    if bad:   # the leg is gated like oracle_check: forward max-abs and every tensor's relative error <= 1e-4
        raise SystemExit("bench.py config2_ops: differs from the reference's kernels beyond 1e-4: %r" % (bad,))
    return {"gate": "vs the reference's kernels on the smooth flow: forward max_abs <= 1e-4, every output max_abs / max|ref| <= 1e-4",
            "worst_rel_vs_ref": max(rels) if rels else None,
            "what": "BASELINE configs[1]: block_extractor (k 3 / 5) + resample2d(4,1) forward and backward (both gradients) on one "
                    "(1,64,256,176) fp32 map through the C ABI; HIP events; algorithmic bytes (SURVEY 8d) / time / 8 TB/s; ref_us = the "
                    "reference's own kernels (oracle/_ref) on the same inputs in the same process" + ("" if ref is not None else
                                                                                                   " -- not built on this box"),
            "dims": [B, C, H, W], "rows": rows, "slower_than_reference_on": [[r["op"], r["flow"]] for r in slower]}


def oracle_check(hp, resample, tol=1e-4):
    """Samples 0 and B-1 of the timed configuration through the reference composition with the CPU oracle kernels (the
    checker only): forward outputs and the gradients of source, target, flow and the warped VGG features of those samples
    (every op on the path is per-sample, so they do not depend on the rest of the batch).  The backward of this pass is
    driven by O(1) upstream gradients (the timed steps scale theirs by 1/numel, which would hide absolute errors).
    Reported per tensor: max |got - want| (`max_abs`, the north star's measure) and that over max |want| (`max_rel`).
    Raises when a forward output exceeds `tol` in max-abs or any tensor exceeds it relative to its largest entry."""
    from oracle import cpu_modules, cpu_oracle
    cpu_oracle.build()
    timed_upstream = hp.upstream
    gen = torch.Generator(device=timed_upstream[0].device).manual_seed(9876)
    hp.upstream = [torch.randn(u.shape, device=u.device, generator=gen) for u in timed_upstream]
    try:
        outs = hp.step(resample, allreduce=False)
    finally:
        upstream, hp.upstream = hp.upstream, timed_upstream
    max_abs, max_rel = {}, {}

    def cmp(name, got, want, forward):
        want = want.detach().double()
        err = (got.detach().double().cpu() - want).abs().max().item()
        rel = err / max(1e-30, want.abs().max().item())
        max_abs[name] = max(max_abs.get(name, 0.0), float("%.2e" % err))
        max_rel[name] = max(max_rel.get(name, 0.0), float("%.2e" % rel))
        if not rel <= tol or (forward and not err <= tol):
            raise SystemExit("bench.py: %s of the timed configuration differs from the CPU oracle: max abs %.3e, relative "
                             "to the largest entry %.3e" % (name, err, rel))

    rs = cpu_modules.Resample2dCPU(4, 1, 2)
    samples = sorted({0, hp.B - 1})
    for n in samples:
        sl = slice(n, n + 1)
        for i, (mod, (src, tgt, flow)) in enumerate(zip(hp.attn, hp.inputs)):
            name, C, H, W, k = LAYERS[i]
            ref = cpu_modules.ExtractorAttnCPU(C, k, torch.nn.LeakyReLU(0.1), softmax=True)
            ref.load_state_dict({kk: v.detach().cpu() for kk, v in mod.state_dict().items()})
            a = [x[sl].detach().cpu().clone().requires_grad_() for x in (src, tgt, flow)]
            want = ref(*a)
            want.backward(upstream[i][sl].cpu())
            cmp(name + " out", outs[i][sl], want, True)
            # the layer's flow also warps the VGG features of the same resolution (HotPath.step): its gradient is the sum
            feat, j = hp.vgg[i], len(hp.attn) + i
            f1 = feat[sl].detach().cpu().clone().requires_grad_(feat.requires_grad)
            warped = rs(f1, a[2])
            warped.backward(upstream[j][sl].cpu())
            cmp(VGG[i][0] + " warp", outs[j][sl], warped, True)
            for nm, x, w in zip(("grad source", "grad target", "grad flow (attention + warp)"), (src, tgt, flow), a):
                cmp(name + " " + nm, x.grad[sl], w.grad, False)
            if feat.requires_grad:
                cmp(VGG[i][0] + " grad input1", feat.grad[sl], f1.grad, False)
    return {"tolerance": tol, "samples": samples, "upstream": "O(1): randn of the output's shape",
            "gate": "forward outputs: max_abs <= tolerance; every tensor: max_abs / max|reference| <= tolerance",
            "max_abs": max_abs, "max_rel": max_rel}


def cpu_baseline(budget_s=20.0):
    """The reference composition on the host cores with the oracle kernels (kind "port").

    The literal port keeps the reference's atomics (as `omp atomic`); inside ONE image they thrash beyond a few tens of
    threads (measured 0.055 images/s on 256 threads vs 1.3 on 8), so a single process uses 16.  Round 5: the batch is
    what parallelises on a host as it does on the GPU -- independent images -- so the baseline runs one 16-thread worker per
    16 usable CPUs (tools/cpu_baseline_worker.py, pinned to disjoint CPU ranges, same start time, batch 1 each) and reports
    the SUM of their rates with `cores` = the threads actually used."""
    import subprocess
    host = os.cpu_count() or 1
    try:
        usable = sorted(os.sched_getaffinity(0))
    except AttributeError:
        usable = list(range(host))
    per = 16
    workers = max(1, min(len(usable) // per, 16))
    budget = max(5.0, min(budget_s, 20.0))
    start_at = time.time() + 30.0          # imports + the warm-up step of every worker fit in here; late workers start at once
    procs = []
    for w in range(workers):
        cpus = usable[w * per:(w + 1) * per] if len(usable) >= per else usable
        rng = "%d-%d" % (cpus[0], cpus[-1]) if cpus == list(range(cpus[0], cpus[-1] + 1)) else ""
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), "--threads",
                                       str(min(per, len(cpus))), "--cpus", rng, "--budget", str(budget), "--start-at", str(start_at)],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    done = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=budget + 240)
            done.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            pr.kill()
    if not done:
        raise RuntimeError("cpu_baseline: no worker finished")
    rate = sum(d["steps"] / d["seconds"] for d in done)
    threads = min(per, len(usable)) * len(done)
    return {"value": round(rate, 3), "unit": "images/s", "cores": threads, "host_logical_cpus": host,
            "host_cpus_usable_by_this_process": len(usable), "kind": "port", "workers": len(done),
            "per_worker_images_per_s": round(rate / len(done), 3),
            "threads_note": "%d workers x %d threads, one image each at a time: the literal port keeps the reference's atomics "
                            "(omp atomic), which thrash beyond ~16 threads inside one image (0.055 images/s on 256 threads vs "
                            "1.3 on 8); independent images are what parallelises, as on the GPU" % (len(done), min(per, len(usable))),
            "sample": "%d step(s) of batch 1 of the same workload over %d concurrent workers (reference op-by-op composition, "
                      "oracle/gfla_oracle.c kernels with OpenMP + torch CPU convolutions), %.1f s each"
                      % (sum(d["steps"] for d in done), len(done), budget)}


def extra_legs(args, device):
    """The other BASELINE configs, timed by the SAME process right after the headline (rank 0, N = 1): compact results under
    `legs`, each a few steps bracketed by synchronize.  The headline's contract (metric, value, steps) is untouched."""
    legs = {}

    def timeit(step, steps, warmup):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    # configs[1]: the standalone ops on one (1,64,256,176) map, every flow kind, next to the reference's kernels
    try:
        legs["config2_ops"] = config2_ops(device)
    except Exception as exc:   # (the checker's build is optional; the timing itself must not be)
        legs["config2_ops"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    # configs[2]: ExtractorAttn forward only (eval, no_grad) at the attention-layer shapes of a 256x256 image, batch 32
    from global_flow_local_attention_amd import fc_mfma
    for mode in (5, 4, 0):
        mods, ins = [], []
        gen = torch.Generator(device=device).manual_seed(7)
        torch.manual_seed(1234)
        for (_, C, H, W, k) in FACE_LAYERS:
            m = gfla.ExtractorAttn(C, k, torch.nn.LeakyReLU(0.1), softmax=True).to(device).eval()
            m.fc_mode = mode
            mods.append(m)
            ins.append((torch.randn(32, C, H, W, device=device, generator=gen), torch.randn(32, C, H, W, device=device, generator=gen),
                        smooth_flow(32, H, W, device, gen)))

        def fwd():
            with torch.no_grad():
                for m, a in zip(mods, ins):
                    m(*a)
        dt = timeit(fwd, 10, 3)
        flops = sum(2.0 * 32 * H * W * 128 * (2 * C * k * k + k * k) for (_, C, H, W, k) in FACE_LAYERS)
        legs["config3_inference_fc_mode%d" % mode] = {
            "what": "BASELINE configs[2]: ExtractorAttn L3 (C256,32x32,k3) + L2 (C128,64x64,k5) forward only, eval / no_grad, "
                    "batch 32, %s" % fc_mfma.MODE_NAMES[mode],
            "ms": round(dt * 1e3, 3), "images_per_s": round(32 / dt, 1),
            "effective_TFLOPs_of_the_reference_formulation": round(flops / dt / 1e12, 1)}
        del mods, ins
    # configs[3] loss side: the warps inside PerceptualCorrectness + the affine regulariser
    hp = HotPath(args.batch, device, seed=100, vgg_grad=False, fc_impl="mfma", fc_mode=args.fc_mode, with_losses=True)
    dt = timeit(lambda: hp.step(None, allreduce=False), 5, 2)
    legs["config4_with_losses"] = {"what": "the headline's two attention layers + PerceptualCorrectness.calculate_loss (max-cosine "
                                           "MFMA kernel, Resample2d, fused loss map) + MultiAffineRegularizationLoss, fwd+bwd, "
                                           "batch %d" % args.batch,
                                   "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(args.batch / dt, 1)}
    del hp
    # configs[3] as a whole step of a generator-shaped network
    tp = TrainerPath(args.batch, device, seed=100, fc_mode=args.fc_mode)
    dt = timeit(lambda: tp.step(), 3, 2)
    legs["config4_trainer_step"] = {"what": tp.describe(args, 1)["config"]["workload"], "ms_per_step": round(dt * 1e3, 3),
                                    "images_per_s": round(args.batch / dt, 1), "losses": {k: round(v, 5) for k, v in tp.losses.items()}}
    del tp
    # configs[4]: face model shapes, bf16 features, 6 sequential frames, 8 clips
    face = {}
    for dual in (True, False):
        fp = FacePath(8, device, seed=100, frames=6, dual_stream=dual)
        face[dual] = timeit(lambda: fp.step(None, allreduce=False), 3, 2)
        what = fp.describe(args, 1)["config"]["workload"]
        del fp
    dt = face[True]
    legs["config5_face_bf16"] = {"what": what + "; the two blocks of a layer (previous / reference frame) on two HIP streams",
                                 "clips": 8, "frames_per_clip": 6, "ms_per_step": round(dt * 1e3, 3),
                                 "frames_per_s": round(48 / dt, 1),
                                 "ms_per_step_one_stream": round(face[False] * 1e3, 3)}
    # configs[0]: the unmodified reference PoseGenerator forward on the host (custom ops = the CPU oracle).  Needs the
    # reference checkout: run live where it exists (a subprocess: install() rewires sys.modules), otherwise the committed
    # result of the build container is quoted with its provenance
    ref_root = os.environ.get("GFLA_REFERENCE_ROOT", "/root/reference")
    committed = os.path.join(ROOT, "profiles", "r4_config0_cpu_reference_forward.json")
    if os.path.isdir(os.path.join(ref_root, "model", "networks")):
        import subprocess
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_config0_cpu.py"), "--reference", ref_root,
                                  "--iters", "3"], capture_output=True, text=True, timeout=600)
            legs["config1_reference_posegenerator_cpu"] = dict(json.loads(out.stdout.strip().splitlines()[-1]), measured="live")
        except Exception as exc:
            legs["config1_reference_posegenerator_cpu"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
    elif os.path.exists(committed):
        legs["config1_reference_posegenerator_cpu"] = dict(json.load(open(committed)),
                                                           measured="not on this box (no reference checkout): committed "
                                                                    "result of tools/bench_config0_cpu.py, " + os.path.basename(committed))
    torch.cuda.empty_cache()
    return legs


def dist_leg(args, rank, world, device, on_gpu, barrier, tile_shape=None, iters=10):
    """N > 1 only, called by EVERY rank: what the process group is (backend, world size as the group sees it, RCCL
    version) and the one data exchange the north star names -- the all-gather of the generated tiles
    ((B/N, 3, H, W) per rank, face_model.py:81-93 = nn.DataParallel's gather) through dist.all_gather_tiles, timed
    between barrier + synchronize, MAX over ranks."""
    import torch.distributed as td
    info = {"backend": td.get_backend(), "world_size_seen_by_group": td.get_world_size(), "rank0_device": str(device)}
    if on_gpu:
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:  # a torch build without the binding
            info["rccl_version"] = "unavailable (%s)" % type(exc).__name__
        info["gpus_visible"] = torch.cuda.device_count()
    # one all-reduce of rank ids proves every rank is in the collective
    t = torch.tensor([float(rank + 1)], device=device)
    td.all_reduce(t)
    info["allreduce_of_rank_ids"] = t.item()
    info["allreduce_expected"] = world * (world + 1) / 2.0
    shape = tuple(tile_shape or (args.batch, 3, 256, 176))
    tiles = torch.full(shape, float(rank), device=device)
    out = gdist.all_gather_tiles(tiles)
    ok = out.shape[0] == shape[0] * world and all(bool((out[r * shape[0]] == float(r)).all()) for r in range(world))
    for _ in range(2):
        gdist.all_gather_tiles(tiles)
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        gdist.all_gather_tiles(tiles)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = tt.item()
    nbytes = tiles.numel() * tiles.element_size()
    info["all_gather_tiles"] = {"tile_shape_per_rank": list(shape), "dtype": "f32", "correct": bool(ok),
                                "us": round(dt / iters * 1e6, 1), "MB_received_per_rank": round(nbytes * (world - 1) / 1e6, 2),
                                "GBps_received_per_rank": round(nbytes * (world - 1) / (dt / iters) / 1e9, 2)}
    return info


def timed_steps(step, steps, warmup, barrier, world, device):
    """The contract's timing: W untimed steps, then exactly K steps between barrier + synchronize, MAX over ranks."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()
    return elapsed


def run(args, make_hotpath, make_resample, rank, world, device, on_gpu=True, data="synthetic"):
    """Everything after process-group / device setup; `make_hotpath(fc_mode)` builds the per-rank workload.  Split from
    main() so that the N>1 control flow can be exercised on CPU under gloo with a stand-in workload
    (tests/test_dist_cpu.py): a typo must not burn the one multi-GPU hardware run."""
    hp = make_hotpath(args.fc_mode)
    resample = make_resample()
    workload = getattr(args, "workload", "pose")
    face = workload == "face_bf16"
    custom = workload != "pose"  # face_bf16 / trainer_step: the workload object describes itself
    images = getattr(hp, "images_per_step", args.batch)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    hp.step(resample)  # priming step: lazy initialisation, never timed
    # (Capturing the whole step -- forward + backward, ~90 launches -- into one hipGraph was measured twice and dropped:
    # round 1 7.146 ms replayed vs 7.147 ms eager, round 4 4.71 vs 4.69 ms (profiles/r4_hipgraph_step.json).  The gaps
    # between dependent kernels are the same inside a graph; the step is not launch-bound.)
    step = lambda: hp.step(resample)
    elapsed = timed_steps(step, args.steps, args.warmup, barrier, world, device)
    # The oracle check runs AFTER the timed region: it is 16 OpenMP threads of host work, and worker threads still spinning
    # behind their last parallel region compete with the launching thread -- a timed region started right behind the
    # check was measured at 5.4 / 6.2 ms per step instead of 4.7 on two occasions.  It still fails the run on a mismatch.
    check = None
    if rank == 0 and world == 1 and on_gpu and not args.no_cpu_baseline and not custom:
        check = oracle_check(hp, resample)
        time.sleep(1.0)  # let the host threads of the check go idle before the variants / legs are timed

    rows, probes = [], []
    if on_gpu:
        # instrumented pass: the same K steps with every C-ABI call bracketed by HIP events
        with KernelTimer() as kt:
            for _ in range(args.steps):
                hp.step(resample)
        rows = kt.summary()
        if args.fc_impl == "mfma" and not custom:
            probes = fc_kernel_probes(hp)

    variants = {}
    if on_gpu and args.fc_impl == "mfma" and not args.no_variants and not custom:
        notes = {0: "float32, DIRECT convolution kernels (a k-ordered fma chain per output): the same step without the "
                    "Winograd-domain formulation",
                 4: "float32 operands on f32 MFMA, Winograd-domain convolutions and weight gradient (the default of rounds 3-5)",
                 5: "two f16 terms per operand on f16 MFMA, f32 accumulate (the product default)"}
        # the same step with the loss-side warps on a side stream (HotPath.step)
        hv = make_hotpath(args.fc_mode)
        hv.two_streams = not getattr(hp, "two_streams", False)
        hv.step(resample)
        n = max(3, args.steps // 2)
        ev = timed_steps(lambda: hv.step(resample), n, 2, barrier, world, device)
        variants["warps_on_side_stream" if hv.two_streams else "one_stream"] = {
            "value": round(args.batch * world * n / ev, 2), "unit": "images/s", "ms_per_step": round(ev / n * 1e3, 3),
            "note": "the Resample2d sites (a branch of the training graph that shares only the flow fields with the attention "
                    "layers) issued on %s" % ("a second HIP stream" if hv.two_streams else "the same stream as the attention layers")}
        del hv
        for mode, label in ((0, "fc_mode0_f32_direct"), (4, "fc_mode4_f32_winograd"), (5, "fc_mode5_winograd_f16x2_exact"),
                            (3, "fc_mode3_f16x3_split"), (2, "fc_mode2_f16x2_split")):
            if mode == args.fc_mode:
                continue
            hv = make_hotpath(mode)
            hv.step(resample)
            n = max(3, args.steps // 2)
            vstep = lambda: hv.step(resample)
            ev = timed_steps(vstep, n, 2, barrier, world, device)
            variants[label] = {"value": round(args.batch * world * n / ev, 2), "unit": "images/s",
                               "ms_per_step": round(ev / n * 1e3, 3),
                               "note": notes.get(mode, "labelled experiment, not the headline: FC operands split into %d f16 "
                                                  "terms, f32 accumulation in the MFMA; the parity tests hold it to the same "
                                                  "bars as the f32 modes" % mode)}
            del hv

    line = {
        "metric": "images/sec (fwd+bwd) PoseGenerator 256x176 attn_layer=2,3 -- feature-warping hot path",
        "value": round(args.batch * world * args.steps / elapsed, 2),
        "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.fc_mode in (0, 4) else
                 ("f32 (FC products: 2-term f16 splits exact to 2^-24 on f16 MFMA, f32 accumulate)"
                  if args.fc_mode == 5 else "f32 tensors; FC operands split into f16 terms (mode %d)" % args.fc_mode),
        "data": data,
        "config": {"workload": "GFLA hot path at PoseGenerator 256x176 shapes, attn_layer=2,3 kernel_size 2=5,3=3: "
                               "ExtractorAttn L3 (C256,32x22,k3) + L2 (C128,64x44,k5) fwd+bwd incl. both FC layers, "
                               "Resample2d(4,1,2) fwd+bwd at (C512,32x22) and (C256,64x44)"
                               + ("" if not args.no_vgg_grad else " with constant VGG features (no d/d input1)")
                               + ("" if not getattr(args, "with_losses", False) else
                                  "; --with-losses: the warps run inside PerceptualCorrectness.calculate_loss (max-cosine "
                                  "MFMA kernel + fused loss map) and MultiAffineRegularizationLoss reads the flows"),
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "parallelism": "dp%d (batch shards; ExtractorAttn gradients all-reduced in one flat bucket launched "
                                  "from autograd hooks, overlapping backward)" % world,
                   "fc_layers": ("this library's MFMA kernels, arithmetic mode %d (%s); no vendor GEMM / convolution in the step"
                                 % (args.fc_mode, {0: "exact f32, direct convolution", 4: "f32, Winograd-domain convolutions F(2x2,5x5) / "
                                                   "F(4x4,3x3)", 5: "two f16 terms per operand (exact to 2^-24) on f16 MFMA, f32 "
                                                   "accumulate: direct kernels for the k5 convolutions and the data gradients, "
                                                   "Winograd domain for the k3 forward and the k5 weight gradient"}.get(args.fc_mode, "f16-split operands, f32 accumulate")))
                   if args.fc_impl == "mfma" else "round 1's vendor-library path (torch.mm / F.conv2d)"},
        "kernels": rows,
        "fc_kernels": probes,
    }
    if custom:
        line.update(hp.describe(args, world))
        line["value"] = round(images * world * args.steps / elapsed, 2)
        line["unit"] = "frames/s" if face else "images/s"
    if probes:
        # the dominant kernels of the step are the MFMA kernels of the FC path; the roofline object describes the kernel the step
        # spends the most time in (all of its launches in the step: the direct convolutions of the k = 5 layer are four launches,
        # the Winograd-domain ones two, a weight gradient one)
        groups = {}
        for r in probes:
            if r.get("in_step"):
                groups.setdefault((r["kernel"].split(":")[0], tuple(r["dims"])), []).append(r)
        if groups:
            step_rows = max(groups.values(), key=lambda rs: sum(r["avg_us"] for r in rs))
        else:
            step_rows = [max(probes, key=lambda r: r["avg_us"])]
        dom = dict(max(step_rows, key=lambda r: r["avg_us"]))
        nl = sum(r.get("launches", 1) for r in step_rows)
        if len(step_rows) > 1 or nl > 1:
            tus = sum(r["avg_us"] for r in step_rows)
            alg, eff = sum(r["alg_GFLOP"] for r in step_rows), sum(r["effective_GFLOP"] for r in step_rows)
            if all("useful_GFLOP" in r for r in step_rows):
                dom["useful_GFLOP"] = round(sum(r["useful_GFLOP"] for r in step_rows) / nl, 2)
            dom.update(kernel=dom["kernel"].split(":")[0] + ": all %d launches of the step (forward and data gradient, source and target "
                                                          "halves)" % nl,
                       avg_us=round(tus / nl, 1), alg_GFLOP=round(alg / nl, 2),
                       TFLOPs=round(alg * 1e9 / (tus * 1e-6) / 1e12, 1), effective_TFLOPs=round(eff * 1e9 / (tus * 1e-6) / 1e12, 1))
            dom["frac_mfma_f32_peak"] = round(dom["TFLOPs"] / MFMA_F32_PEAK_TFLOPS, 4)
            if dom.get("pipe") == "f16":
                dom["f16_pipe_TFLOPs"] = round(dom.get("f16_macs_per_mac", 4) * alg * 1e9 / (tus * 1e-6) / 1e12, 1)
                dom["frac_mfma_f16_peak"] = round(dom["f16_pipe_TFLOPs"] / MFMA_F16_PEAK_TFLOPS, 4)
        f16pipe = dom.get("pipe") == "f16"
        line["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "dims": dom["dims"],
                            "achieved": dom["f16_pipe_TFLOPs"] if f16pipe else dom["TFLOPs"],
                            "peak": MFMA_F16_PEAK_TFLOPS if f16pipe else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": dom["frac_mfma_f16_peak"] if f16pipe else dom["frac_mfma_f32_peak"], "avg_us": dom["avg_us"],
                            **({"pipe": ("f16 matrix cores, two f16 terms per operand, three cross products: executed f16 MACs = 3 x the "
                                         "direct convolution's multiplies over the kernel's own output domain (DESIGN.md 4)"
                                         if dom.get("f16_macs_per_mac") == 3 else
                                         "f16 matrix cores, two f16 terms per operand: executed f16 MACs = 4 x the Winograd-domain "
                                         "multiplies; the kernel is bound by its LDS / vector work (transforms, f16 split), not by "
                                         "the matrix cores (DESIGN.md 4)"),
                                "f32_equivalent_TFLOPs": dom["TFLOPs"],
                                # (Winograd-domain kernels only) round 5's yardstick for the convolution kernel -- executed
                                # Winograd-domain flops / time / f32 MFMA peak, 0.58 on the f32 pipe then: NOT a pipe fraction
                                **({"f32_equivalent_frac_of_f32_peak": round(dom["TFLOPs"] / MFMA_F32_PEAK_TFLOPS, 4)}
                                   if dom.get("f16_macs_per_mac") == 4 else {})} if f16pipe else {}),
                            "alg_GFLOP_per_launch": dom["alg_GFLOP"],
                            **({"effective_TFLOPs": dom["effective_TFLOPs"]} if "effective_TFLOPs" in dom else {}),
                            **({"useful_frac": round((float(dom.get("f16_macs_per_mac", 4)) if f16pipe else 1.0) * dom["useful_GFLOP"] * 1e9 / (dom["avg_us"] * 1e-6) / 1e12
                                                     / (MFMA_F16_PEAK_TFLOPS if f16pipe else MFMA_F32_PEAK_TFLOPS), 4),
                                "useful_frac_note": "executed Winograd-domain flops over the UN-extended output domain only (the "
                                                    "tiles whose outputs the caller keeps) / time / peak"}
                               if "useful_GFLOP" in dom else {}),
                            "traffic": pmc_traffic_kernel(dom["kernel"], in_step=bool(dom.get("in_step"))),
                            "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
                                              "passes) of this bench, (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; table: "
                                              + pmc_table_provenance(),
                            "flops": ("direct convolution: achieved = 2*B*rows*C*k*k*128 over the rows the kernel computes (the convolved "
                                      "map forward, the padded domain for the data gradient) x 3 f16 MACs each / time; useful = "
                                      "the reference formulation's 2*B*H*W*C*k*k*128" if dom.get("f16_macs_per_mac") == 3 else
                                      "Winograd domain: achieved = the 36 multiplies per (tile, c, n) the kernel executes "
                                      "(2*36*tiles*C*128; tiles = ceil(rows/m)*ceil(cols/m) of the output domain, m = 2 for "
                                      "F(2x2,5x5), 4 for F(4x4,3x3)) / time; effective_TFLOPs = the reference formulation's "
                                      "2*B*H*W*C*k*k*128 / time" if args.fc_mode in (4, 5) else
                                      "reference formulation (2*B*H*W*C*k*k*128 per half and pass); work the kernel adds on "
                                      "top (extended / padded domains) is not counted"),
                            "timing": "HIP events around 10 back-to-back launches of the kernel alone "
                                      "(gfla_fc_kernel_f32) on the launch stream"}
    elif rows:
        dom = rows[0]
        line["roofline"] = {"bound": "hbm", "kernel": dom["entry"], "dims": dom["dims"], "achieved": dom["GBps"],
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac_hbm_peak"], "avg_us": dom["avg_us"],
                            "alg_MB_per_launch": dom["alg_MB"], "traffic": pmc_traffic(dom["entry"], dom["dims"], dom["ptrs"])}
        if "effective_GFLOP" in dom:  # an FC-layer entry: several kernels (packing, MFMA contraction, sampling tails) per call
            peak = MFMA_F16_PEAK_TFLOPS if face else MFMA_F32_PEAK_TFLOPS
            line["roofline"] = {"bound": "mfma", "kernel": dom["entry"], "dims": dom["dims"], "achieved": dom["effective_TFLOPs"],
                                "peak": peak, "unit": "TFLOP/s", "frac": round(dom["effective_TFLOPs"] / peak, 4),
                                "avg_us": dom["avg_us"], "alg_GFLOP_per_call": dom["effective_GFLOP"], "traffic": None,
                                "hbm_GBps_of_call": dom["GBps"],
                                "note": "whole C-ABI call (packing, MFMA contraction, sampling / reduction kernels), HIP "
                                        "events around the call; peak = dense matrix-core rate of the operand type"}
    if variants:
        line["variants"] = variants
    if world > 1:   # every rank takes part (collectives); the fingerprint proves N ranks met on the backend
        line["dist"] = dist_leg(args, rank, world, device, on_gpu, barrier,
                                tile_shape=None if on_gpu else (args.batch, 3, 8, 6))
    if rank == 0 and world == 1 and on_gpu and not custom and not getattr(args, "no_legs", False):
        # the north star's own figure: block_extractor (reference layout) + local-attention forward against the HBM roofline
        line["north_star"] = op_roofline(device, B=args.batch)
    if rank == 0 and world == 1 and on_gpu and not custom and not getattr(args, "no_legs", False) \
            and not getattr(args, "with_losses", False):
        line["legs"] = extra_legs(args, device)
    if check is not None:
        line["oracle_check"] = check
    # calls of this process that left the library's own MFMA kernels for rocBLAS / MIOpen (0 unless --fc-impl library)
    from global_flow_local_attention_amd import extractor_attn as _ea
    line["vendor_fallback_calls"] = _ea.vendor_fallback_calls
    if getattr(args, "one_stream", False):
        line["streams"] = "--one-stream: warps issued on the attention layers' stream (trace run; the headline uses two)"
    if rank == 0:
        if world == 1 and on_gpu and not args.no_cpu_baseline and not custom:
            line["cpu_baseline"] = cpu_baseline(args.cpu_budget)
        print(compact_line(line, write_detail(line))[1], flush=True)
    return line


FINAL_LINE_LIMIT = 4096   # bytes: the driver keeps a bounded tail of stdout; the last line must fit it whole


def compact_line(line, detail_file):
    """The LAST stdout line: the contract's keys + roofline + north_star + oracle_check + cpu_baseline in < 4 KB (one short
    line per run, like the reference's own report, train.py:46-48).  Everything else -- per-call / per-kernel rows, variants,
    the legs of the other BASELINE configs -- is in `detail_file` (and on stderr)."""
    def cut(text, n):
        text = str(text)
        return text if len(text) <= n else text[:n - 3] + "..."

    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in line}
    out["metric"], out["dtype"], out["data"] = cut(out["metric"], 160), cut(out["dtype"], 100), cut(out["data"], 100)
    cfg = line.get("config", {})
    out["config"] = {"workload": cut(cfg.get("workload", ""), 330)}
    for k in ("batch_per_gpu", "global_batch", "clips_per_gpu", "frames_per_clip"):
        if k in cfg:
            out["config"][k] = cfg[k]
    out["config"]["parallelism"] = cut(cfg.get("parallelism", ""), 60).split(" (")[0]
    rf = line.get("roofline")
    if rf:
        out["roofline"] = {k: rf[k] for k in ("bound", "kernel", "dims", "achieved", "peak", "unit", "frac", "useful_frac",
                                              "avg_us", "traffic") if k in rf}
        out["roofline"]["kernel"] = cut(rf["kernel"], 110)
        for k in ("alg_GFLOP_per_launch", "alg_MB_per_launch", "f32_equivalent_TFLOPs", "f32_equivalent_frac_of_f32_peak"):
            if k in rf:
                out["roofline"][k] = rf[k]
        if rf.get("traffic") and "alg_MB_read_write_per_launch" in rf:
            out["roofline"]["alg_MB_read_write_per_launch"] = rf["alg_MB_read_write_per_launch"]
    ns = line.get("north_star")
    if ns:
        out["north_star"] = {"what": "block_extractor fwd + local-attn fwd, alg. bytes / HIP-event us / 8 TB/s",
                             "layers": {name: {"be_fwd": [lay["block_extractor_fwd"]["us"], lay["block_extractor_fwd"]["frac"]],
                                               "attn_fwd": [lay["local_attn_fwd"]["us"], lay["local_attn_fwd"]["frac"]],
                                               "pair": [lay["pair"]["us"], lay["pair"]["frac"]]}
                                        for name, lay in ns["layers"].items()}, "cols": ["us", "frac_of_hbm_peak"]}
    oc = line.get("oracle_check")
    if oc:
        fw = [v for k, v in oc["max_abs"].items() if k.endswith(" out") or k.endswith(" warp")]
        out["oracle_check"] = {"tolerance": oc["tolerance"], "passed": True, "worst_abs_forward": max(fw) if fw else None,
                               "worst_abs": max(oc["max_abs"].values()), "worst_rel": max(oc["max_rel"].values()),
                               "tensors": len(oc["max_abs"])}
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb}
        out["cpu_baseline"]["sample"] = cut(cb.get("sample", ""), 220)
    legs = line.get("legs")
    if legs:   # one number per leg (the rows are in the detail file)
        brief = {}
        for name, leg in legs.items():
            if not isinstance(leg, dict):
                continue
            if "error" in leg:
                brief[name] = "error"
            elif name == "config2_ops":
                smooth = {r["op"]: r["frac"] for r in leg.get("rows", []) if r.get("flow") == "smooth"}
                brief[name] = {"frac_hbm_smooth_flow": smooth, "worst_rel_vs_ref": leg.get("worst_rel_vs_ref"),
                               "slower_than_reference_on": leg.get("slower_than_reference_on")}
            else:
                ms = leg.get("ms_per_step", leg.get("ms"))
                if ms is not None:
                    brief[name] = {"ms": ms}
        out["legs"] = brief
    if "dist" in line:
        d = line["dist"]
        out["dist"] = {k: d[k] for k in ("backend", "world_size_seen_by_group", "rccl_version", "gpus_visible",
                                         "allreduce_of_rank_ids", "allreduce_expected") if k in d}
        ag = d.get("all_gather_tiles")
        if ag:
            out["dist"]["all_gather_tiles"] = {k: ag[k] for k in ("correct", "us", "GBps_received_per_rank") if k in ag}
    for k in ("vendor_fallback_calls", "streams"):
        if k in line:
            out[k] = line[k]
    out["detail_file"] = detail_file
    text = json.dumps(out)
    # belt and braces: shed the optional objects, least important first, until the line fits
    for k in ("legs", "streams", "dist", "north_star"):
        if len(text) < FINAL_LINE_LIMIT:
            break
        out.pop(k, None)
        text = json.dumps(out)
    assert len(text) < FINAL_LINE_LIMIT, len(text)
    return out, text


def write_detail(line):
    """The full record (kernels, fc_kernels, variants, legs, the long notes) -> gpurun_out/bench_detail.json (the directory
    that travels back from a gpurun box; BENCH_DETAIL overrides) and stderr.  Never stdout: the last stdout line is the
    short one, and there is exactly one JSON line on stdout."""
    path = os.environ.get("BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    text = json.dumps(line)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(text + "\n")
    except OSError as exc:
        path = "unwritable (%s); see stderr" % type(exc).__name__
    print("bench_detail " + text, file=sys.stderr, flush=True)
    return os.path.relpath(path, ROOT) if os.path.isabs(path) and path.startswith(ROOT) else path


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (oracle check + baseline)")
    ap.add_argument("--no-variants", action="store_true", help="skip the labelled variants (other FC arithmetic modes)")
    ap.add_argument("--tuning", default="", help="library tuning keys for A/B runs: key=value[,key=value...] (include/gfla_hip.h)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the compact legs for the other BASELINE configs (config-3 inference, with-losses, trainer "
                         "step, face bf16) that the default N=1 run appends under `legs`")
    ap.add_argument("--no-vgg-grad", action="store_true",
                    help="treat the VGG features fed to Resample2d as constants (what the reference's training step "
                         "does: they come from a frozen VGG of the input images), i.e. skip d/d input1")
    ap.add_argument("--fc-impl", choices=("mfma", "library"), default="mfma",
                    help="FC layers of ExtractorAttn: this library's MFMA kernels (default) or round 1's vendor GEMM/conv path")
    ap.add_argument("--workload", choices=("pose", "face_bf16", "trainer_step"), default="pose",
                    help="pose: the headline (BASELINE metric, PoseGenerator 256x176 shapes, f32).  face_bf16: BASELINE "
                         "configs[4] -- FaceGenerator 256x256 shapes, two ExtractorAttn per layer, 6 sequential frames, "
                         "bf16 features; --batch is then clips per GPU.  trainer_step: one TrainerShell.optimize_parameters step "
                         "of the in-repo generator-shaped network (SURVEY 8f row 4 / BASELINE configs[3] per rank)")
    ap.add_argument("--frames", type=int, default=6, help="face_bf16: frames generated per clip")
    ap.add_argument("--one-stream", action="store_true",
                    help="pose workload: the loss-side warps on the same stream as the attention layers (for kernel traces whose "
                         "durations are not inflated by the second stream; the headline is the two-stream step)")
    ap.add_argument("--face-one-stream", action="store_true",
                    help="face_bf16: evaluate attn_p and attn_r of a layer one after the other on one stream (default: two streams)")
    ap.add_argument("--fc-mode", type=int, choices=(0, 1, 2, 3, 4, 5), default=5,
                    help="arithmetic of the FC contraction: 5 = float32 tensors, every operand of a product as two f16 terms "
                         "(exact to 2^-24) on the f16 matrix cores, f32 accumulation -- direct kernels and Winograd-domain kernels, "
                         "chosen per convolution (the product default and the headline, round 6); 4 = Winograd domain on f32 MFMA "
                         "(the default of rounds 3-5); 0 = float32, direct convolution; 3 / 2 = three / two f16 terms per operand in the direct "
                         "convolution (labelled experiments)")
    ap.add_argument("--with-losses", action="store_true",
                    help="replace the bare Resample2d sites by the losses that contain them in training (BASELINE "
                         "config 4): PerceptualCorrectness.calculate_loss on synthetic VGG-shaped features + "
                         "MultiAffineRegularizationLoss on the flow fields (no oracle check / CPU baseline for this leg)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args(argv)
    if args.with_losses:
        args.no_cpu_baseline = True
    return args


def self_spawn(argv):
    """`python bench.py --gpus N` with no torchrun environment: re-launch this script as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1, a free port) and hand back its exit code.
    The torchrun form of the contract keeps working: it sets WORLD_SIZE, so this is never reached from a rank."""
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    n = ap.parse_known_args(argv)[0].gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, GFLA_BENCH_SELF_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print("bench.py: --gpus %d without a torch.distributed.run environment -- spawning %d ranks (port %d)" % (n, n, port),
          file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


class _ControlFlowStub(object):
    """GFLA_BENCH_CPU_STUB=1 (test hook, tests/test_dist_cpu.py): a tiny torch model with the HotPath surface so that
    the launcher, the process group, the barriers, the MAX-over-ranks clock and the rank-0 JSON line of an N>1 run can
    be executed end to end where there is no GPU.  Its line says so in `data`; it is never a measurement."""

    def __init__(self, rank):
        torch.manual_seed(1234)
        self.net = torch.nn.Linear(8, 4)
        self.x = torch.full((3, 8), float(rank + 1))
        self.reducer = None

    def params(self):
        return list(self.net.parameters())

    def step(self, resample, allreduce=True):
        for p in self.params():
            p.grad = None
        if allreduce and self.reducer is None:
            self.reducer = gdist.GradBucketReducer(self.params())
        out = resample(self.net(self.x))
        out.sum().backward()
        if allreduce:
            self.reducer.finish()
        return [out]


def main_cpu_stub(args):
    rank, world, _ = gdist.init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    args.no_cpu_baseline = True
    line = run(args, lambda mode: _ControlFlowStub(rank), lambda: (lambda t: t * 2.0), rank, world, torch.device("cpu"),
               on_gpu=False, data="cpu-stub: control flow of the launcher only (GFLA_BENCH_CPU_STUB), not a measurement")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return line


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1 \
            and not os.environ.get("GFLA_BENCH_SELF_SPAWNED"):
        raise SystemExit(self_spawn(argv))
    if os.environ.get("GFLA_BENCH_CPU_STUB") == "1":
        main_cpu_stub(args)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path); run it through gpurun")
    if args.gpus > torch.cuda.device_count() and "GFLA_DEVICE" not in os.environ:   # (GFLA_DEVICE: ranks sharing a GPU, tests)
        raise SystemExit("--gpus %d but %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    # bind the device BEFORE anything touches the GPU or the process group (RCCL communicators are per device)
    local = int(os.environ.get("GFLA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    # GFLA_DIST_BACKEND / GFLA_DEVICE: test hooks (e.g. two gloo ranks sharing the one GPU of a test box)
    rank, world, _ = gdist.init_from_env(os.environ.get("GFLA_DIST_BACKEND"), device=local)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run, or plain `python bench.py "
                         "--gpus N`, which spawns the ranks itself)" % (args.gpus, world))
    device = torch.device("cuda", local)
    # a benchmark must never time rocBLAS / MIOpen by accident: configurations the library's MFMA kernels do not take raise
    # here (the package default, also after install(), is a warning once per module)
    from global_flow_local_attention_amd import extractor_attn as _ea
    _ea.VENDOR_FALLBACK = "error"
    for kv in filter(None, args.tuning.split(",")):
        gfla.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    # the step forks work onto side streams (bench.HotPath.step, face_step.DualStreamAttn): gradients of shared leaves then
    # arrive from two streams, which autograd handles and announces with a warning per backward
    try:
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    except AttributeError:
        pass
    if args.fc_impl == "library":  # round 1's path (rocBLAS / MIOpen FC layers), kept as a cross-check only
        torch.backends.cudnn.benchmark = True

    def make_hotpath(fc_mode):
        if args.workload == "face_bf16":
            return FacePath(args.batch, device, seed=100 + rank, frames=args.frames, dual_stream=not args.face_one_stream)
        if args.workload == "trainer_step":
            return TrainerPath(args.batch, device, seed=100 + rank, fc_mode=fc_mode)
        hp_ = HotPath(args.batch, device, seed=100 + rank, vgg_grad=not args.no_vgg_grad, fc_impl=args.fc_impl,
                      fc_mode=fc_mode, with_losses=args.with_losses)
        if getattr(args, "one_stream", False):
            hp_.two_streams = False
        return hp_

    run(args, make_hotpath, lambda: gfla.Resample2d(4, 1, 2), rank, world, device)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
