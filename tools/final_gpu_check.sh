cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest_gpu_full.log 2>&1
grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" gpurun_out/final/pytest_gpu_full.log | tail -4 | tee gpurun_out/final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/final/smoke.log
python bench.py > gpurun_out/final/bench_final.json 2> gpurun_out/final/bench_final.err; tail -c 800 gpurun_out/final/bench_final.json
