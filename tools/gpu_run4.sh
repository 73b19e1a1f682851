cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests/test_wavernn_gpu.py tests/test_meldecoder_gpu.py tests/test_lstm_gpu.py tests/test_hifigan_gpu.py tests/test_gemm_gpu.py tests/test_training_gpu.py -q -x -s 2>&1 | tail -40 > $O/pytest_a.log; cat $O/pytest_a.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; tail -c 6000 $O/bench_default.json
timeout 300 python tools/probes/textcoder_time.py > $O/textcoder_time.log 2>&1; tail -5 $O/textcoder_time.log
