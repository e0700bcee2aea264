#!/bin/bash
cd /root/repo
python3 -m pytest tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -3
python3 tools/ttcn_bench.py 1024 32 10 31; python3 tools/ttcn_bench.py 65536 32 10 31; python3 tools/ttcn_bench.py 8192 32 10 31
