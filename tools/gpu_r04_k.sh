cd $GRAFT_REPO_ROOT
bash tools/prof_windows.sh 1024 r04c_w1024 > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_r04c_w1024_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
n=max(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])
print("steps", n, "kernel ms/step", tot/n/1e6)
for r in rows[:22]:
    print(f'{int(r["Calls"]):5d} {int(r["TotalDurationNs"])/n/1e3:9.1f} us/step {float(r["Percentage"]):5.2f}%  {r["Name"].replace("(anonymous namespace)::","")[:70]}')
PY
