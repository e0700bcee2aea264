cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_train.py -x -q -k "fused_tail or cfg2_step or phased or two_rank_flag" 2>&1 | tail -3
python3 - <<'PY'
import sys, os, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'imm-tsf_amd')
import bench
from immtsf import _lib, config
from immtsf.train import FlagStep
_lib.load()
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
config.nan_check = "deferred"; config.manual_seed(1234)
for rep in range(2):
  for tailn in (0, 1, 2, 3):
    w = bench.Workload("cfg2", dev, 64, "bf16")
    st = FlagStep(w.trainer, *bench.flag_fns(w), param_tail=tailn)
    for _ in range(20): st()
    torch.cuda.synchronize()
    best = min(bench.time_steps(st, 100, 0, torch.cuda.synchronize)[0] for _ in range(3)) / 100 * 1e3
    print("param_tail", tailn, round(best, 4), "timed_out", st.timed_out(), flush=True)
    w.close(); del st, w
PY
for form in fold chain; do for w in 64 128 256; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --t2v-form $form --windows-per-gpu $w --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$form', $w, d['ms_per_step'])"
done; done
