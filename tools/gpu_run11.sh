cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 600 python -m pytest tests/test_distributed_gpu.py -q -x 2>&1 | tail -3 | tee $O/pytest_dist.log
for i in 1 2; do timeout 300 python bench.py --mode train --steps 6 --warmup 3 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('train b16 run $i', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), d['exchange']['chunks_launched_during_backward'], 'exposed', d['exchange']['exposed_ms_per_step'])" | tee -a $O/train_ab.log; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_sustained'), d['roofline'].get('traffic'))" | tee -a $O/headline.log
