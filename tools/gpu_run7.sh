cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_power tools/probes/mfma_power_probe.hip 2>/dev/null && /tmp/mfma_power | tee $O/mfma_power_probe.log
timeout 600 python tools/bench_stage.py --iters 5 --shapes 10,12 > $O/bench_stage_random.log 2>&1; cat $O/bench_stage_random.log
timeout 600 python tools/bench_stage.py --iters 5 --shapes 10,12 --data zeros > $O/bench_stage_zeros.log 2>&1; cat $O/bench_stage_zeros.log
