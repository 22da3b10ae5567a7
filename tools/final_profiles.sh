cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -- python3 bench.py --steps 10 --warmup 2 > $O/r03r_bench_under_rocprof.json 2>$O/p_err1.txt
cp $(ls $O/p_stats/*/*kernel_stats.csv | head -1) $O/r03r_kernel_stats.csv; rm -rf $O/p_stats
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 > /dev/null 2>$O/p_err2.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 > /dev/null 2>$O/p_err3.txt
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/r03r_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 3 --warmup 1 > /dev/null 2>$O/p_err4.txt
python tools/pmc_summarize.py $O/pmc_mfma > $O/r03r_pmc_mfma.txt
rm -rf $O/pmc_mfma
python bench.py > $O/r03r_bench.json 2>/dev/null
tail -c 400 $O/r03r_bench_under_rocprof.json; head -c 600 $O/r03r_pmc_traffic.json; head -20 $O/r03r_pmc_mfma.txt
