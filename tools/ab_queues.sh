echo base; python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo hwq8; GPU_MAX_HW_QUEUES=8 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo hwq2; GPU_MAX_HW_QUEUES=2 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo nograph; python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-graph 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
strings /opt/rocm/lib/libamdhip64.so | grep -i "graph" | grep -i "^[A-Z_]*$" | head -20
