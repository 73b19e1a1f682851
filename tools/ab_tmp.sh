python -m pytest tests/test_lstm_gpu.py tests/test_meldecoder_gpu.py tests/test_reference_goldens_gpu.py tests/test_api_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_baseline_configs_gpu.py -x -q -k "e2e or c5 or sentences or cubegan" 2>&1 | tail -3
timeout 200 python tools/bench_lstm.py 2>&1 | tail -10
e2e() { timeout 300 python bench.py --mode e2e --steps 5 --warmup 2 "$@" 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],2), round(d['ms_per_step_sequential_rank0'],2), d['phase_ms_rank0'])"; }
e2e
e2e
for i in 1 2; do timeout 200 python tools/bench_cubegan_step.py --iters 10 --ragged 2>&1 | grep ms/step | cut -c1-80; done
