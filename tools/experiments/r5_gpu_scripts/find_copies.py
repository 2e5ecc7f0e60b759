"""Which torch ops inside the bench step launch copy / fill kernels, with shapes and Python stacks (torch.profiler)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import torch
import bench
import global_flow_local_attention_amd as gfla

dev = torch.device("cuda", 0)
hp = bench.HotPath(32, dev, 0, fc_impl="mfma", fc_mode=4)
resample = gfla.Resample2d(4, 1, sigma=2).to(dev)
for _ in range(3):
    hp.step(resample)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    hp.step(resample)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::to", "aten::_to_copy")
for e in prof.events():
    if e.name in want and e.device_time_total > 3:
        st = [s for s in (e.stack or []) if "torch/" not in s and "<built-in" not in s][:4]
        print("%-18s dev %7.1f us  shapes %s\n      %s" % (e.name, e.device_time_total, e.input_shapes, " <- ".join(st)))
