cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 1500 gpurun_out/bench_final.json
rm -rf gpurun_out/prof_final; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/bench_prof.json 2>/dev/null
find gpurun_out/prof_final -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/prof_wr; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wr -- python tools/bench_wavernn.py > gpurun_out/wr.log 2>/dev/null; tail -2 gpurun_out/wr.log
find gpurun_out/prof_wr -name "*kernel_trace.csv" -delete
