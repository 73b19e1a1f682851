#!/bin/bash
# Round-5 profile passes on the GPU box (run through gpurun): the headline bench under rocprofv3 (kernel trace, PMC passes), the workgroup phase
# timelines of the wide / chain kernels (ablate build), the default bench line and the e2e / train modes.  A trimmed tools/profile_round.sh.
#   usage: bash tools/profile_r05.sh [r05]
R=${1:-r05}
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
# workgroup phase timelines (measurement build)
for cfg in "--stage 1 --k 3 --L 4032" "--stage 1 --k 11 --L 4032" "--stage 2 --k 7 --L 12032" "--stage 2 --k 3 --chain 1 --acc 0 --L 12032" "--stage 3 --k 3 --acc 0" "--stage 3 --k 7 --acc 1" "--stage 3 --k 11 --acc 1" "--stage 4 --k 3 --acc 0" "--stage 4 --k 7 --acc 1" "--stage 4 --k 11 --acc 1"; do
  timeout 120 python tools/wg_timeline.py $cfg 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
done > $O/wg_timeline.log 2>&1
for cfg in "--stage 1 --k 3 --L 4032" "--stage 2 --k 7 --L 12032"; do TTSC_CONV_ACC_INIT=0 TTSC_CONV_EPI_PREFETCH=0 timeout 120 python tools/wg_timeline.py $cfg 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|census\|mfma :\|other:"; done > $O/wg_timeline_round4_epilogue.log 2>&1
# A/B switches of the round, whole forward
(for env in "TTSC_CONV_ACC_INIT=1" "TTSC_CONV_ACC_INIT=0" "TTSC_CONV_ACC_INIT=0 TTSC_CONV_EPI_PREFETCH=0" "TTSC_HIFIGAN_PITCH=0" "TTSC_TALL_SWIZZLE=0" "TTSC_HIFIGAN_CHAIN_SPLIT=0" "TTSC_CONV_ACC_INIT=1"; do
  echo "== $env"; env $env timeout 300 $B --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('ms_per_step %.3f  frac %.4f  rms_vs_oracle %.3e' % (r['ms_per_step'], r['roofline']['frac'], r['self_check_rms_vs_oracle']))"; done) > $O/bench_switches.log 2>&1
timeout 300 python tools/bench_chain_split.py --stages 3,4 --ks 3,7,11 2>&1 | grep -v amdgpu.ids > $O/chain_split.log
/opt/rocm/bin/hipcc -O3 -w --offload-arch=gfx950 -o /tmp/lobits tools/probes/mfma_lo_bits_probe.hip 2>/dev/null && /tmp/lobits > $O/mfma_lo_bits_probe.log 2>&1
timeout 300 python bench.py --mode e2e --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_e2e.json
timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>/dev/null | grep '^{' > $O/bench_train_b16.json
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err < /dev/null
du -sh $O; ls $O; tail -c 300 $O/bench_final.json
