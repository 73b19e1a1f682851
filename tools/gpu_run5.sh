cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_hifigan_gpu.py -q -x 2>&1 | tail -5 > $O/pytest_hifigan.log; cat $O/pytest_hifigan.log
timeout 300 python tools/bench_vocoder_step.py --iters 3 > $O/vocoder_step.log 2>&1; tail -2 $O/vocoder_step.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/voc -o v -- python tools/bench_vocoder_step.py --iters 3 > $O/vocoder_trace.log 2>&1
python tools/rocpd_stats.py $O/voc/v_results.db $O/vocoder_step_all.csv -330 $O/vocoder_step_kernel_stats.csv >> $O/vocoder_trace.log 2>&1
tail -3 $O/vocoder_trace.log; head -25 $O/vocoder_step_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o r -- python bench.py --mode train --steps 4 --warmup 2 > $O/train.log 2>&1
python tools/rocpd_stats.py $O/train/r_results.db $O/train_kernel_stats.csv -170 $O/train_last2steps_kernel_stats.csv >> $O/train.log 2>&1
tail -3 $O/train.log; head -40 $O/train_last2steps_kernel_stats.csv
rm -rf $O/voc $O/train
timeout 200 python tools/bench_wavernn.py > $O/wavernn.log 2>&1; tail -3 $O/wavernn.log
