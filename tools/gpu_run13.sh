cd $GRAFT_REPO_ROOT
timeout 900 python tools/probes/train_determinism.py 6 2>&1 | grep "exchange=" 
