#!/bin/bash
# Round-6 profile passes on the GPU box (run through gpurun): the headline bench under rocprofv3 (kernel trace + PMC passes: the generator's kernels are
# round 5's, the records are re-taken on the final tree), the training step (kernel stats of the last steps, timeline), WaveRNN, the e2e path, and the
# default / train / e2e bench lines.  A trimmed tools/profile_round.sh.      usage: bash tools/profile_r06.sh [r06]
R=${1:-r06}
O=gpurun_out/$R
mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-extra --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $B --steps 5 --warmup 1 > $O/bench_trace.log 2>&1
python tools/rocpd_stats.py $O/trace/t_results.db $O/bench_kernel_stats.csv fwd:ttsc:: $O/bench_last_forward.csv >> $O/bench_trace.log 2>&1
python tools/roofline_table.py $O/bench_last_forward.csv $O/roofline.md >> $O/bench_trace.log 2>&1
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "grbm:GRBM_COUNT GRBM_GUI_ACTIVE"; do
  N=${P%%:*}; C=${P#*:}
  TTSC_HIFIGAN_CALIBRATE=0 timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$N -o p -- $B --steps 1 --warmup 0 > $O/pmc_$N.log 2>&1
done
python tools/pmc_summary.py $O/bench_pmc.csv $O/fetch/p_results.db $O/write/p_results.db $O/sq/p_results.db $O/grbm/p_results.db --note "bench.py --steps 1 --warmup 0 (3 identical forwards, TTSC_HIFIGAN_CALIBRATE=0), one rocprofv3 --pmc pass per counter group; sums over all launches" > $O/pmc_summary.log 2>&1
python tools/hbm_from_pmc.py $O/bench_pmc.csv $O/bench_hbm_pmc.csv 3 >> $O/pmc_summary.log 2>&1
rm -rf $O/trace $O/fetch $O/write $O/sq $O/grbm
# training step through bench.py --mode train (process group + hooked exchange): whole run and the last ~300 ms (warm steps), timeline
timeout 300 rocprofv3 --kernel-trace --stats -d $O/train -o r -- python bench.py --mode train --steps 4 --warmup 2 > $O/train.log 2>&1
python tools/rocpd_stats.py $O/train/r_results.db $O/train_kernel_stats.csv -300 $O/train_last300ms_kernel_stats.csv >> $O/train.log 2>&1
python tools/gpu_timeline.py $O/train/r_results.db 100 2 > $O/train_timeline.txt 2>&1
rm -rf $O/train
# the un-armed step (tools/bench_cubegan_step.py: what bench.py's cubegan_training_step_b16 leg runs): kernel stats of the last ~300 ms, phase timeline
timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o s -- python tools/bench_cubegan_step.py --iters 8 > $O/step.log 2>&1
python tools/rocpd_stats.py $O/step/s_results.db $O/train_step_kernel_stats.csv -300 $O/train_step_last300ms_kernel_stats.csv >> $O/step.log 2>&1
python tools/gpu_timeline.py $O/step/s_results.db 100 2 > $O/train_step_timeline.txt 2>&1
rm -rf $O/step
timeout 200 python tools/probes/train_phase_timeline.py 2>&1 | grep -v Warn | tail -17 > $O/train_phase_timeline.log
(for i in 1 2 3; do timeout 200 python tools/bench_cubegan_step.py --iters 10; done; timeout 300 python tools/bench_cubegan_step.py --iters 5 --batch 128) 2>&1 | grep ms/step | cut -c1-90 > $O/train_step_bench.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/wr -o v -- python tools/bench_wavernn.py > $O/wavernn.log 2>&1
python tools/rocpd_stats.py $O/wr/v_results.db $O/wavernn_kernel_stats.csv >> $O/wavernn.log 2>&1
rm -rf $O/wr
(timeout 200 python tools/bench_wavernn.py --frames 100; timeout 120 python tools/bench_wavernn.py --frames 20 --layers 2) 2>&1 | grep 'net:' > $O/wavernn_bench.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/e2e1 -o e -- python tools/probes/e2e_b1.py > $O/e2e_single_sentence.log 2>&1
python tools/rocpd_stats.py $O/e2e1/e_results.db $O/e2e_single_sentence_kernel_stats.csv >> $O/e2e_single_sentence.log 2>&1
python tools/rocpd_tail.py $O/e2e1/e_results.db 5.4 > $O/e2e_single_sentence_launches.txt 2>&1
rm -rf $O/e2e1
(for i in 1 2 3; do timeout 120 python tools/probes/e2e_b1.py; done) 2>&1 | grep "B=1" > $O/e2e_single_sentence_bench.log
timeout 300 python bench.py --mode e2e --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_e2e.json
timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | grep '^{' > $O/bench_train_b16.json
timeout 400 python bench.py --mode train --train-batch 128 --steps 3 --warmup 1 2>/dev/null | grep '^{' > $O/bench_train_b128.json
timeout 200 python tools/bench_vocoder_step.py --iters 3 2>/dev/null < /dev/null | tail -1 > $O/vocoder_step.log
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err < /dev/null
du -sh $O; ls $O; tail -c 300 $O/bench_final.json
