cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
/opt/rocm/bin/hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o /tmp/split2.bin tools/probes/split2_probe.hip 2>/dev/null && /tmp/split2.bin | tee $O/split2_probe.log
timeout 1200 python -m pytest tests/test_conv1d_gpu.py tests/test_hifigan_gpu.py tests/test_conv_train_gpu.py -x -q 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
TTSC_CHAIN_IL=0 timeout 600 python tools/bench_stage.py --iters 5 --shapes 0,1,10,11,12 > $O/bench_stage.log 2>&1; cat $O/bench_stage.log
BENCH_CHAIN_SHAPES=10,11 timeout 300 python tools/bench_layers.py --stages 3 --only-chain > $O/bench_layers_s3.log 2>&1; cat $O/bench_layers_s3.log
timeout 300 python tools/bench_layers.py --stages 1,2 > $O/bench_layers_s12.log 2>&1; cat $O/bench_layers_s12.log
for i in 1 2; do timeout 600 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('run $i ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'rms', d.get('self_check_rms_vs_oracle'))
" | tee -a $O/bench_ab.log; done
