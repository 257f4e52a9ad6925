cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pso
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pso -- python bench.py --workload sparse_ocp_n100000_m50000 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pso.log 2>&1
python tools/trace_step_any.py gpurun_out/pso k_bcr_extract > gpurun_out/step_ocp.txt 2>&1
python tools/trace_overlap.py gpurun_out/pso 2>/dev/null | head -30 >> gpurun_out/step_ocp.txt
rm -rf gpurun_out/pso
