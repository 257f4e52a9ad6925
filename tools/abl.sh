cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/abl*
for sk in 0 1 2 4 7; do
  PGF_SKIP=$sk timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl$sk -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  echo "skip=$sk $(grep panel gpurun_out/abl$sk/*/*kernel_stats.csv | cut -d, -f2-4)"
done
