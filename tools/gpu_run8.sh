cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_energy tools/probes/mfma_energy_probe.hip 2>/dev/null && /tmp/mfma_energy | tee $O/mfma_energy_probe.log
