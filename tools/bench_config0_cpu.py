#!/usr/bin/env python3
"""BASELINE configs[0]: the UNMODIFIED reference PoseGenerator (model/networks/generator.py:13-36, 14.05 M parameters,
attn_layer=2,3, kernel_size 2=5,3=3) forward on the HOST for one image pair, with the three custom ops served by the CPU
oracle (the reference has no CPU path of its own: block_extractor.py:23-24 raises).  Needs the reference checkout, so it
runs in the build container only (the GPU box has no /root/reference); the result is committed under profiles/.

256x176 cannot pass through the reference network (PoseFlowNet's five stride-2 stages give 5 -> 10 != 11 columns, SURVEY
0.3); the reference itself feeds 256x256 (data/base_dataset.py:32-35), which is what is timed here, plus 256x192.

    python tools/bench_config0_cpu.py [--reference /root/reference] [--iters 5] [--out profiles/r4_config0_cpu.json]
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.reference, "model", "networks")):
        raise SystemExit("no reference checkout at %s" % args.reference)
    import global_flow_local_attention_amd as gfla
    from oracle import cpu_modules, cpu_oracle
    cpu_oracle.build()
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    cpu_oracle.set_threads(min(threads, 16))
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    bf = gfla.install(args.reference, fuse_extractor_attn=False)   # import paths + stubs; the ops are swapped below

    class BlockExtractor(torch.nn.Module):          # block_extractor.py:45-54 on the oracle's literal kernels
        def __init__(self, kernel_size=3):
            super().__init__()
            self.kernel_size = kernel_size

        def forward(self, source, flow_field):
            return cpu_modules._BlockExtractorCPU.apply(source, flow_field, self.kernel_size)

    class LocalAttnReshape(torch.nn.Module):        # local_attn_reshape.py:40-46
        def forward(self, inputs, kernel_size=3):
            return cpu_modules._LocalAttnReshapeCPU.apply(inputs, kernel_size)

    bf.BlockExtractor, bf.LocalAttnReshape = BlockExtractor, LocalAttnReshape
    import model.networks.generator as gen
    torch.manual_seed(0)
    net = gen.PoseGenerator(image_nc=3, structure_nc=18, ngf=64, img_f=512, layers=3, num_blocks=2, use_spect=False,
                            attn_layer=[2, 3], norm="instance", activation="LeakyReLU", extractor_kz={"2": 5, "3": 3}).eval()
    nparam = sum(p.numel() for p in net.parameters())
    rows = []
    for (H, W) in ((256, 256), (256, 192)):
        g = torch.Generator().manual_seed(1)
        src = torch.rand(1, 3, H, W, generator=g) * 2 - 1
        src_B, tgt_B = torch.rand(1, 18, H, W, generator=g), torch.rand(1, 18, H, W, generator=g)
        with torch.no_grad():
            out = net(src, src_B, tgt_B)            # warm-up
            t0 = time.perf_counter()
            for _ in range(args.iters):
                out = net(src, src_B, tgt_B)
            dt = (time.perf_counter() - t0) / args.iters
        img = out[0]
        rows.append({"input": "%dx%d" % (H, W), "s_per_image": round(dt, 4), "images_per_s": round(1 / dt, 3),
                     "output_shape": list(img.shape), "flow_fields": [list(f.shape) for f in out[1]]})
        print(json.dumps(rows[-1]), flush=True)
    res = {"what": "BASELINE configs[0]: unmodified reference PoseGenerator forward (eval, no_grad, batch 1) on the host; the "
                   "three custom ops = oracle/gfla_oracle.c (OpenMP), everything else torch CPU",
           "parameters": nparam, "torch_threads": threads, "host_logical_cpus": os.cpu_count(), "iters": args.iters,
           "where": "build container (the GPU box has no reference checkout)", "rows": rows}
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
