cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
(for cfg in "4 11 12" "4 11 10" "4 3 10" "4 7 12" "3 11 11" "3 3 11"; do set -- $cfg; timeout 200 python tools/ablate.py --stage $1 --k $2 --shape $3 2>&1 | grep -v amdgpu.ids; done) > $O/ablation_chain.log; cat $O/ablation_chain.log
(for cfg in "1 11" "1 3" "2 7"; do set -- $cfg; timeout 200 python tools/ablate.py --stage $1 --k $2 2>&1 | grep -v amdgpu.ids | head -8; done) > $O/ablation_wide.log; cat $O/ablation_wide.log
