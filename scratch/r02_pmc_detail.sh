#!/bin/bash
# PMC detail of the tiled pass (separate counter-only passes; no trace domains)
export TMPDIR=/tmp
O=gpurun_out/pmcd; mkdir -p $O
K=${1:-16}
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); d=/tmp/pmcd_$i; rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -o pmc -- python3 scratch/pmc_target.py merged $K > /dev/null 2> $d.err || { echo "pass $i failed: $c"; tail -3 $d.err; continue; }
  echo "== $c" >> $O/pmc_detail_K$K.txt
  python scratch/pmc_summarize.py $d k_sq_tiled >> $O/pmc_detail_K$K.txt
done
cat $O/pmc_detail_K$K.txt
