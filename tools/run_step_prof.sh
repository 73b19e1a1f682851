cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_step
timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_step -- python tools/bench_cubegan_step.py --iters 2 2>&1 | grep "ms/step"
f=$(find gpurun_out/prof_step -name "*kernel_trace.csv" | head -1); wc -l $f
python tools/trace_last_step.py $f 270 gpurun_out/step_last270ms.csv
find gpurun_out/prof_step -name "*kernel_trace.csv" -delete
