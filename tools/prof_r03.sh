# Round-3 measurements kept under profiles/ (run on a GPU box from the repository root;
# results land in gpurun_out/r03/, copy the summaries over afterwards).
#   bash tools/prof_r03.sh dense | batch | sparse | all
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
W=${1:-all}
if [ $W = dense ] || [ $W = all ]; then
python bench.py --steps 100 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err && echo bench1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1 && cp $(ls $O/ks/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv && echo ks done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1 && python tools/pmc_summary.py $O/pf1 $O/pw1 $O/pmc_n1.json > $O/pmc_traffic.txt 2>&1 && echo pmc1 done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm1 -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1 && python tools/pmc_mfma_summary.py $O/pm1 > $O/pmc_mfma.txt 2>&1 && echo mfma done
bash tools/prof_seq.sh && cp gpurun_out/ps.txt $O/fused_launches.txt && cp gpurun_out/step.txt $O/step_timeline.txt && echo seq done
PGF_CHAIN_TIMING=1 python tools/time_dense.py 4096 1024 3 2>&1 | grep stamps | tail -1 > $O/chain_stamps.txt
PGF_CONDENSED=0 python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_n1_natural_order.json 2>/dev/null && echo natural done
rm -rf $O/ks $O/pf1 $O/pw1 $O/pm1
fi
if [ $W = batch ] || [ $W = all ]; then
python bench.py --workload batch256_n1024_m256 --steps 6 --warmup 2 > $O/bench_batch256.json 2> $O/bench_batch.err && echo benchb done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksb -- python bench.py --workload batch256_n1024_m256 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 && cp $(ls $O/ksb/*/*kernel_stats.csv | head -1) $O/batch256_kernel_stats.csv && echo ksb done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pfb -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pwb -- python bench.py --workload batch256_n1024_m256 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 && python tools/pmc_summary.py $O/pfb $O/pwb $O/pmc_batch.json > $O/batch256_pmc_traffic.txt 2>&1 && echo pmcb done
rocprofv3 --kernel-trace --output-format csv -d $O/pb -- python tools/time_batch.py 32 > $O/batch32_time.txt 2>&1 && python tools/trace_step_any.py $O/pb kb_residual > $O/batch32_step_timeline.txt 2>&1 && echo b32 done
rm -rf $O/ksb $O/pfb $O/pwb $O/pb
fi
if [ $W = sparse ] || [ $W = all ]; then
python bench.py --workload sparse_ocp_n100000_m50000 --steps 100 --warmup 5 > $O/bench_sparse_ocp.json 2>/dev/null && echo ocp done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kso -- python bench.py --workload sparse_ocp_n100000_m50000 --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 && cp $(ls $O/kso/*/*kernel_stats.csv | head -1) $O/sparse_ocp_kernel_stats.csv && echo kso done
python bench.py --workload box_qp_n16384 --steps 100 --warmup 5 > $O/bench_box_qp.json 2>/dev/null && echo box done
python bench.py --workload box_qp_dense_n16384 --steps 6 --warmup 2 > $O/bench_box_qp_dense.json 2>/dev/null && echo boxd done
python tools/time_lu.py 1024 2560 5120 > $O/lu_timing.txt 2>&1 && echo lu done
rm -rf $O/kso
fi
ls -la $O
