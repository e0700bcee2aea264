"""Kernel times of the many-rows feed-forward (csrc/ffn32.hip) alone: run under rocprofv3 --kernel-trace --stats.
usage: python3 tools/ffn32_bench.py [rows=65536] [p=0.1] [fused=1]"""
import os
import sys

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
if len(sys.argv) > 3 and sys.argv[3] == "0":
    os.environ["IMMTSF_FFN32"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "imm-tsf_amd"))
import torch  # noqa: E402
from immtsf import config, ops  # noqa: E402

dev = torch.device("cuda:0")
D, F = 32, 2048
torch.manual_seed(0)
conv1, conv2 = torch.nn.Linear(D, F).to(dev), torch.nn.Linear(F, D).to(dev)
n2 = torch.nn.LayerNorm(D).to(dev)
x = torch.randn(1, R, D, device=dev, requires_grad=True)
up = torch.randn(1, R, D, device=dev)
config.precision = "bf16"
for it in range(12):
    y = ops.ffn_block(x, conv1, conv2, n2, "relu", p, True, ops.SITE_LAYER_BASE + 7)
    (y * up).sum().backward()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for it in range(10):
    y = ops.ffn_block(x, conv1, conv2, n2, "relu", p, True, ops.SITE_LAYER_BASE + 7)
    (y * up).sum().backward()
ev1.record()
torch.cuda.synchronize()
print(f"rows {R} p {p}: {ev0.elapsed_time(ev1) / 10 * 1e3:.1f} us per forward+backward (eager, incl. LayerNorm joints)")
