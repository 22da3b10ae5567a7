cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for d in 162 322; do
  export CSG_GEMM_CFG=$d
  rm -rf $O/pmc_g
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_g -- python3 tools/gemm_probe.py > $O/pmc_g_out.txt 2> $O/pmc_g_err.txt
  echo "=== cfg $d pass A"; python3 tools/pmc_summarize.py $O/pmc_g
  rm -rf $O/pmc_g
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $O/pmc_g -- python3 tools/gemm_probe.py >> $O/pmc_g_out.txt 2>> $O/pmc_g_err.txt
  echo "=== cfg $d pass B"; python3 tools/pmc_summarize.py $O/pmc_g
  rm -rf $O/pmc_g
done
