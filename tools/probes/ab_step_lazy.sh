mkdir -p gpurun_out/r06x
python -m pytest tests/test_training_gpu.py tests/test_cabi_symbols.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06x/pytest_lazy.log
cat gpurun_out/r06x/pytest_lazy.log
for rep in 1 2 3; do for lazy in 0 1; do echo "STEP_LAZY=$lazy"; TTSC_STEP_LAZY=$lazy timeout 300 python tools/bench_cubegan_step.py --iters 10 2>&1 | grep "ms/step"; done; done > gpurun_out/r06x/step_lazy_ab.log 2>&1
TTSC_STEP_LAZY=1 timeout 300 python tools/bench_cubegan_step.py --iters 5 --batch 128 2>&1 | grep "ms/step" >> gpurun_out/r06x/step_lazy_ab.log
TTSC_STEP_LAZY=0 timeout 300 python tools/bench_cubegan_step.py --iters 5 --batch 128 2>&1 | grep "ms/step" >> gpurun_out/r06x/step_lazy_ab.log
cut -c1-80 gpurun_out/r06x/step_lazy_ab.log
