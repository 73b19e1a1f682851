cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x -s 2>&1 | tail -60 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
