cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/time_lu.py 1024 2560 5120 > gpurun_out/lu.log 2>&1
rm -rf gpurun_out/plu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/plu -- python tools/time_lu.py 5120 > /dev/null 2>&1
python tools/trace_overlap.py gpurun_out/plu 2>/dev/null | head -14 >> gpurun_out/lu.log
rm -rf gpurun_out/plu
