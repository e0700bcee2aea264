cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_collate.py -x -q 2>&1 | tail -3
for w in 64 256 1024 4096; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-roofline --windows-per-gpu $w --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($w, d['ms_per_step'], d['engine'])"
done
bash tools/prof_windows.sh 4096 r04b_w4096 > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r04b_w4096_stats.csv")))
for r in rows[:14]:
    print(f'{int(r["Calls"]):5d} {int(r["TotalDurationNs"])/int(r["Calls"])/1e3:9.1f} us  {float(r["Percentage"]):5.2f}%  {r["Name"].replace("(anonymous namespace)::","")[:60]}')
PY
