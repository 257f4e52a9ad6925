cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fin && mkdir -p gpurun_out/fin
python bench.py --steps 20 --warmup 3 > gpurun_out/fin/bench_n1.json 2> gpurun_out/fin/bench_n1.err && echo bench1 done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/ks -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 && echo ks done
python bench.py --workload batch256_n1024_m256 --steps 6 --warmup 2 > gpurun_out/fin/bench_batch.json 2> gpurun_out/fin/bench_batch.err && echo benchb done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/ksb -- python bench.py --workload batch256_n1024_m256 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && echo ksb done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fin/pf -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fin/pw -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_summary.py gpurun_out/fin/pf gpurun_out/fin/pw gpurun_out/fin/pmc_batch.json > gpurun_out/fin/pmc_batch.txt 2>&1 && echo pmc done
