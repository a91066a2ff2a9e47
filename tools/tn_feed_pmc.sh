export PYTHONPATH=$PWD
for mode in tn nt; do
  if [ $mode = tn ]; then CMD="python $PWD/tools/gemm_tn_bench.py 1638400 768 768"; PAT=gemm8t; else CMD="python $PWD/tools/gemm_loop_once.py 768 768"; PAT=gemm8; fi
  echo "== $mode"
  bash tools/pmc_pass.sh x$mode "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" -- $CMD | grep $PAT
done
