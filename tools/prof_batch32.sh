cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pb
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pb -- python tools/time_batch.py 32 > gpurun_out/pb.log 2>&1
python tools/trace_overlap.py gpurun_out/pb 2>/dev/null | head -24 > gpurun_out/pb.txt
python tools/trace_step_any.py gpurun_out/pb kb_assemble >> gpurun_out/pb.txt 2>&1
rm -rf gpurun_out/pb
