cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hifigan_gpu.py tests/test_conv1d_gpu.py tests/test_api_gpu.py tests/test_baseline_configs_gpu.py -x -q 2>&1 | tail -6
for fp in 1 0; do TTSC_HIFIGAN_FUSE_POST=$fp timeout 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('FUSE_POST=$fp', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_of_sustained'), d['roofline'].get('traffic'), d['self_check_rms_vs_oracle'])"; done
