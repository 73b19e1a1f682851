mkdir -p gpurun_out/r06x
for rep in 1 2; do for at in 0 1 2 3 4; do echo "TEXT_AT=$at"; TTSC_TEXT_AT=$at timeout 300 python tools/bench_cubegan_step.py --iters 10 2>&1 | grep "ms/step"; done; done > gpurun_out/r06x/text_at_ab.log 2>&1
cat gpurun_out/r06x/text_at_ab.log
