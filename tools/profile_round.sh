#!/bin/bash
# Round profile passes on the GPU box (run through gpurun): kernel traces and PMC passes of the headline bench, the e2e path,
# the training step and WaveRNN.  The rocpd databases are summarised here and deleted (gpurun returns <= 64 MiB).
#   usage: bash tools/profile_round.sh r02
R=${1:-r04}
O=gpurun_out/$R
mkdir -p $O; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-extra --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $B --steps 5 --warmup 1 > $O/bench_trace.log 2>&1
python tools/rocpd_stats.py $O/trace/t_results.db $O/bench_kernel_stats.csv fwd:ttsc:: $O/bench_last_forward.csv >> $O/bench_trace.log 2>&1
python tools/roofline_table.py $O/bench_last_forward.csv $O/roofline.md >> $O/bench_trace.log 2>&1
# PMC passes: calibration off so that every forward of the run is the same launch sequence (3 forwards each)
for P in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "grbm:GRBM_COUNT GRBM_GUI_ACTIVE"; do
  N=${P%%:*}; C=${P#*:}
  TTSC_HIFIGAN_CALIBRATE=0 timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$N -o p -- $B --steps 1 --warmup 0 > $O/pmc_$N.log 2>&1
done
python tools/pmc_summary.py $O/bench_pmc.csv $O/fetch/p_results.db $O/write/p_results.db $O/sq/p_results.db $O/grbm/p_results.db --note "bench.py --steps 1 --warmup 0 (3 identical forwards, TTSC_HIFIGAN_CALIBRATE=0), one rocprofv3 --pmc pass per counter group; sums over all launches" > $O/pmc_summary.log 2>&1
python tools/hbm_from_pmc.py $O/bench_pmc.csv $O/bench_hbm_pmc.csv 3 >> $O/pmc_summary.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/e2e -o e -- python tools/bench_e2e.py > $O/e2e.log 2>&1
python tools/rocpd_stats.py $O/e2e/e_results.db $O/e2e_kernel_stats.csv >> $O/e2e.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/train -o r -- python bench.py --mode train --steps 4 --warmup 2 > $O/train.log 2>&1
# per-kernel stats of the whole run (includes MIOpen's one-time find pass) and of the last ~2 steps (steady state)
python tools/rocpd_stats.py $O/train/r_results.db $O/train_kernel_stats.csv -300 $O/train_last2steps_kernel_stats.csv >> $O/train.log 2>&1
python tools/gpu_timeline.py $O/train/r_results.db 100 2 > $O/train_timeline.txt 2>&1   # (kernels of different streams run one at a time under the tracer)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/wr -o v -- python tools/bench_wavernn.py > $O/wavernn.log 2>&1
python tools/rocpd_stats.py $O/wr/v_results.db $O/wavernn_kernel_stats.csv >> $O/wavernn.log 2>&1
rm -rf $O/trace $O/fetch $O/write $O/sq $O/grbm $O/e2e $O/train $O/wr
# the same workload on round 1's kernels only (no wide tiles, no fused chain), for the record
(TTSC_CONV_WIDE=0 TTSC_HIFIGAN_CHAIN=0 timeout 200 $B --steps 5 --warmup 2) > $O/bench_r1_kernels.log 2>&1
timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | grep '^{' > $O/bench_train_b16.json
if [ -n "$FULL" ]; then
timeout 400 python bench.py --mode train --train-batch 128 --steps 3 --warmup 1 2>/dev/null | grep '^{' > $O/bench_train_b128.json
timeout 300 python tools/bench_lstm.py > $O/lstm.log 2>&1
fi
timeout 300 python bench.py --mode e2e --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_e2e.json
(timeout 120 python tools/bench_wavernn.py --frames 20 --layers 2; TTSC_WR_TILE2=0 timeout 120 python tools/bench_wavernn.py --frames 4 --layers 2) > $O/wavernn_n2.log 2>&1
# per-layer probes of the training convolutions (exact-fp32 kernels vs the split-precision path) and the vocoder training step
(echo "# tools/probes/prof_disc_layers.py, TTSC_TRAIN_SPLIT=0 (exact-fp32 MFMA kernels), batch 32 x 8192 samples"; TTSC_TRAIN_SPLIT=0 timeout 100 python tools/probes/prof_disc_layers.py 2>/dev/null < /dev/null) > $O/disc_layers_fp32.log
(echo "# tools/probes/prof_disc_layers.py, split-precision training kernels (default), batch 32 x 8192 samples"; timeout 100 python tools/probes/prof_disc_layers.py 2>/dev/null < /dev/null) > $O/disc_layers_split.log
(echo "# tools/probes/prof_train_convs.py: per-layer kernel time of one Cubegan step, b = 16 (synchronising timers)"; timeout 200 python tools/probes/prof_train_convs.py 2>/dev/null < /dev/null | grep -E "ms/step|per step") > $O/train_convs_by_layer.log
timeout 200 python tools/bench_vocoder_step.py --iters 3 2>/dev/null < /dev/null | tail -1 > $O/vocoder_step.log
# CubenetVocoder.training_step under the profiler: whole run (incl. the first, cold step) and the last ~330 ms = two WARM steps
timeout 400 rocprofv3 --kernel-trace --stats -d $O/voc -o v -- python tools/bench_vocoder_step.py --iters 3 > $O/vocoder_trace.log 2>&1
python tools/rocpd_stats.py $O/voc/v_results.db $O/vocoder_step_whole_run_kernel_stats.csv -330 $O/vocoder_step_kernel_stats.csv >> $O/vocoder_trace.log 2>&1
rm -rf $O/voc
# what the matrix pipe sustains on this chip by operand data (power budget), and what LDS reads / VALU beside the MFMAs cost
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_power tools/probes/mfma_power_probe.hip 2>/dev/null && /tmp/mfma_power > $O/mfma_power_probe.log 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_energy tools/probes/mfma_energy_probe.hip 2>/dev/null && /tmp/mfma_energy > $O/mfma_energy_probe.log 2>&1
# the 32-channel stage: chain launches by tile shape, the one-launch stage kernel, random vs all-zero data (the same binaries run 25-45 % faster on zeros)
(TTSC_CHAIN_IL=0 timeout 300 python tools/bench_stage.py --iters 5 --shapes 0,1,2,10,11,12 2>&1 | grep -v amdgpu.ids) > $O/bench_stage_random.log
(TTSC_CHAIN_IL=0 timeout 300 python tools/bench_stage.py --iters 5 --shapes 0,1,10,12 --data zeros 2>&1 | grep -v amdgpu.ids) > $O/bench_stage_zeros.log
(BENCH_CHAIN_SHAPES=10,11 timeout 300 python tools/bench_layers.py --stages 1,2,3 2>&1 | grep -v amdgpu.ids) > $O/layer_bench.log
(TTSC_CHAIN_IL=0 timeout 200 $B --steps 5 --warmup 2) > $O/bench_plain_columns.log 2>&1
timeout 200 python tools/probes/textcoder_time.py > $O/textcoder_time.log 2>&1
# the Cubegan step: host enqueue time vs GPU drain, with and without the text side on its own stream; the LSTM recurrence by utterances per member group
(for t in 1 0; do echo "TTSC_TEXT_STREAM=$t"; TTSC_TEXT_STREAM=$t timeout 200 python tools/probes/train_host_bound.py 2>&1 | tail -1; done) > $O/train_host_bound.log
(timeout 200 python tools/probes/lstm_group_probe.py 2>&1 | grep -v amdgpu.ids) > $O/lstm_group_probe.log
(TTSC_GEMM_SPLIT=0 TTSC_LSTM_NB=1 timeout 300 python bench.py --mode e2e --steps 5 --warmup 2 --no-pipeline 2>/dev/null | grep '^{') > $O/bench_e2e_round3_text_side.json
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err < /dev/null
du -sh $O; ls $O; tail -c 300 $O/bench_final.json
