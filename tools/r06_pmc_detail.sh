#!/bin/bash
# SQ / LDS / TCC counters of the dominant pass on the bench workload (counter-only rocprofv3 passes over tools/pmc_target.py, the program
# directly after `--`; averages per dispatch).  usage: tools/r06_pmc_detail.sh <tag> [K=16] [ENV=VAL ...]
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
tag=$1; K=${2:-16}; shift; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=gpurun_out/${tag}_pmc_detail_K$K.txt; : > "$out"
echo "# $(date -u +%H:%M:%S) K=$K env: $*" >> "$out"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=gpurun_out/${tag}_pmcd
  rm -rf "$d"
  echo "== $set" >> "$out"
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$d" -o p -- python3 tools/pmc_target.py merged $K > "$d.log" 2>&1 || { echo "FAILED: $set" >> "$out"; tail -3 "$d.log" >> "$out"; continue; }
  python3 tools/pmc_summarize.py "$d" k_sq_tiled >> "$out"
done
rm -rf gpurun_out/${tag}_pmcd
cat "$out"
