cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 1200 python -m pytest tests/test_conv1d_gpu.py -x -q -k "chain" 2>&1 | tail -8 > $O/pytest_chain.log; cat $O/pytest_chain.log
timeout 900 python -m pytest tests/test_hifigan_gpu.py -x -q 2>&1 | tail -8 > $O/pytest_hifigan.log; cat $O/pytest_hifigan.log
TTSC_CHAIN_IL=0 timeout 600 python tools/bench_stage.py --iters 5 --shapes 0,1,2,10,11,12 > $O/bench_stage.log 2>&1; cat $O/bench_stage.log
TTSC_CHAIN_IL=0 BENCH_CHAIN_SHAPES=10,11 timeout 300 python tools/bench_layers.py --stages 3 --only-chain > $O/bench_layers_s3.log 2>&1; cat $O/bench_layers_s3.log
TTSC_CHAIN_IL=0 BENCH_CHAIN_SHAPES=11 timeout 300 python tools/bench_layers.py --stages 2 --ks 3 --only-chain > $O/bench_layers_s2.log 2>&1; cat $O/bench_layers_s2.log
for cfg in "1 0" "1 1" "0 0" "0 1"; do set -- $cfg
TTSC_CHAIN_IL=$1 TTSC_HIFIGAN_STAGE=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('IL=$1 STAGE=$2 ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'rms', d.get('self_check_rms_vs_oracle'))
" | tee -a $O/bench_ab.log
done
