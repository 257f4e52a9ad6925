# average duration of the dense factorisation's kernels, unfused (PGF_FUSED=0) and fused
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pc_u gpurun_out/pc_f
PGF_FUSED=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc_u -- python tools/time_dense.py 4096 1024 6 > gpurun_out/pc_u.log 2>&1
python tools/trace_overlap.py gpurun_out/pc_u 2>/dev/null | head -7 > gpurun_out/pc.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc_f -- python tools/time_dense.py 4096 1024 6 > gpurun_out/pc_f.log 2>&1
python tools/trace_overlap.py gpurun_out/pc_f 2>/dev/null | head -7 >> gpurun_out/pc.txt
grep "ms/step" gpurun_out/pc_u.log gpurun_out/pc_f.log >> gpurun_out/pc.txt
for i in 1 2 3; do python tools/time_dense.py 4096 1024 20 | tail -1 >> gpurun_out/pc.txt; done
rm -rf gpurun_out/pc_u gpurun_out/pc_f
