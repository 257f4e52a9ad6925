# Round-2 measurements kept under profiles/ (run on a GPU box from the repository root;
# results land in gpurun_out/r02/, copy the summaries over afterwards).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err && echo bench1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 && cp $(ls $O/ks/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && echo ks done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_summary.py $O/pf1 $O/pw1 $O/pmc_n1.json > $O/pmc_traffic.txt 2>&1 && echo pmc1 done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_mfma_summary.py $O/pm1 > $O/pmc_mfma.txt 2>&1 && echo mfma done
python bench.py --workload batch256_n1024_m256 --steps 6 --warmup 2 > $O/bench_batch256.json 2> $O/bench_batch.err && echo benchb done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksb -- python bench.py --workload batch256_n1024_m256 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && cp $(ls $O/ksb/*/*kernel_stats.csv | head -1) $O/batch256_kernel_stats.csv && echo ksb done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pfb -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pwb -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_summary.py $O/pfb $O/pwb $O/pmc_batch.json > $O/batch256_pmc_traffic.txt 2>&1 && echo pmcb done
python bench.py --workload sparse_ocp_n100000_m50000 --steps 100 --warmup 5 > $O/bench_sparse_ocp.json 2>/dev/null && echo ocp done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kso -- python bench.py --workload sparse_ocp_n100000_m50000 --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 && cp $(ls $O/kso/*/*kernel_stats.csv | head -1) $O/sparse_ocp_kernel_stats.csv && echo kso done
python bench.py --workload box_qp_n16384 --steps 100 --warmup 5 > $O/bench_box_qp.json 2>/dev/null && echo box done
python bench.py --workload box_qp_dense_n16384 --steps 6 --warmup 2 > $O/bench_box_qp_dense.json 2>/dev/null && echo boxd done
bash tools/prof_seq.sh && cp gpurun_out/ps.txt $O/fused_launches.txt && cp gpurun_out/step.txt $O/step_timeline.txt && echo seq done
PGF_CHAIN_TIMING=1 python tools/time_dense.py 4096 1024 3 2>&1 | grep stamps | tail -1 > $O/chain_stamps.txt
python tools/bench_update.py 0 11 43 75 > $O/update_tile_microbench.txt 2>&1 && echo ubench done
python tools/time_lu.py 1024 2560 5120 > $O/lu_timing.txt 2>&1 && echo lu done
rm -rf $O/ks $O/pf1 $O/pw1 $O/pm1 $O/ksb $O/kso $O/pfb $O/pwb
ls -la $O
