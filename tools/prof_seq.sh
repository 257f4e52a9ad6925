# launch-by-launch durations of the dense factorisation's kernels (fused production schedule)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ps
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ps -- python tools/time_dense.py 4096 1024 4 > gpurun_out/ps.log 2>&1
for k in k_chain_update k_diag_chain k_trsm_ud k_trsm_block k_update_diag; do python tools/trace_seq.py gpurun_out/ps $k; done > gpurun_out/ps.txt
python tools/trace_overlap.py gpurun_out/ps 2>/dev/null | head -12 >> gpurun_out/ps.txt
python tools/trace_step.py gpurun_out/ps > gpurun_out/step.txt
rm -rf gpurun_out/ps
