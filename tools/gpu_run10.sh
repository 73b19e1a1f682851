cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
TTSC_CHAIN_IL=0 timeout 300 python tools/bench_stage.py --iters 8 --shapes 0,1,10,11,12 --acc 1 2>&1 | grep "chain shape" | tee $O/bench_stage_acc.log
TTSC_CHAIN_IL=0 timeout 300 python tools/bench_stage.py --iters 8 --shapes 0,1,10,11,12 --acc 0 2>&1 | grep "chain shape" | tee $O/bench_stage_noacc.log
timeout 300 python -m pytest tests/test_hifigan_gpu.py -q -x -k "guard or range or below" 2>&1 | tail -3
