set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_hifigan_gpu.py -x -q -k "stage_launch or both_precisions or config2" 2>&1 | tail -15 > gpurun_out/r4a/pytest_stage.log
cat gpurun_out/r4a/pytest_stage.log
timeout 600 python tools/bench_stage.py --iters 5 > gpurun_out/r4a/bench_stage.log 2>&1
cat gpurun_out/r4a/bench_stage.log
BENCH_CHAIN_SHAPES=2 timeout 300 python tools/bench_layers.py --stages 3 --ks 3 --only-chain > gpurun_out/r4a/bench_layers_s3k3.log 2>&1
cat gpurun_out/r4a/bench_layers_s3k3.log
for st in 1 0; do for sh in 0 1; do
TTSC_HIFIGAN_STAGE=$st TTSC_HIFIGAN_STAGE_SHAPE=$sh timeout 600 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('STAGE=$st SHAPE=$sh ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'rms', d.get('self_check_rms_vs_oracle'))
" | tee -a gpurun_out/r4a/bench_ab.log
[ $st = 0 ] && break
done; done
