# Refresh the measurements kept under profiles/ (run on a GPU box from the repository root;
# results land in gpurun_out/fin/, copy them over afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fin && mkdir -p gpurun_out/fin
python bench.py --steps 20 --warmup 3 > gpurun_out/fin/bench_n1.json 2> gpurun_out/fin/bench_n1.err && echo bench1 done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/ks -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 && echo ks done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fin/pf1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fin/pw1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_summary.py gpurun_out/fin/pf1 gpurun_out/fin/pw1 gpurun_out/fin/pmc_n1.json > gpurun_out/fin/pmc_n1.txt 2>&1 && echo pmc1 done
python bench.py --workload batch256_n1024_m256 --steps 6 --warmup 2 > gpurun_out/fin/bench_batch.json 2> gpurun_out/fin/bench_batch.err && echo benchb done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/ksb -- python bench.py --workload batch256_n1024_m256 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && echo ksb done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fin/pf -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fin/pw -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_summary.py gpurun_out/fin/pf gpurun_out/fin/pw gpurun_out/fin/pmc_batch.json > gpurun_out/fin/pmc_batch.txt 2>&1 && echo pmc done
python bench.py --workload sparse_ocp_n100000_m50000 --steps 100 --warmup 5 > gpurun_out/fin/bench_ocp.json 2>/dev/null && echo ocp done
python bench.py --workload box_qp_n16384 --steps 100 --warmup 5 > gpurun_out/fin/bench_box.json 2>/dev/null && echo box done
python bench.py --workload box_qp_dense_n16384 --steps 6 --warmup 2 > gpurun_out/fin/bench_box_dense.json 2>/dev/null && echo boxd done
