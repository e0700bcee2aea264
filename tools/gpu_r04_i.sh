cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py -x -q --tb=short 2>&1 | tail -4
