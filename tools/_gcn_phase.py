import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch, bench
from immtsf import _lib, config
lib = _lib.load()
dev = torch.device("cuda", 0)
config.nan_check = "deferred"; config.manual_seed(1234)
W = int(sys.argv[1])
w = bench.Workload("cfg2", dev, W, "bf16")
st = bench.flag_step(w)
for _ in range(10): st()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 24)()
lib.immtsf_debug_gcn_trace.argtypes = [C.c_void_p]
lib.immtsf_debug_gcn_trace(C.cast(buf, C.c_void_p))
names = ["zero/between", "cell_forward", "dZ+sync", "mix grads", "dfeat+sync", "hops bwd", "softmax bwd+sync", "dNV+sync", "gate logits+sync", "gate w", "nd linears", "dx", "end sync", "flush"]
n = max(1, buf[20])
print("windows", W, "calls", buf[20], " ".join(f"{names[i]}={buf[i] / n / 100:.2f}" for i in range(14)))
