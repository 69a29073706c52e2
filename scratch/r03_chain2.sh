#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for an in "2,0.5" "4,0.25" "1.4,0.7"; do
echo "== aniso $an"
CWR_ORDER_ANISO=$an timeout -k 10 300 python scratch/r03_chain.py bench 16 40 8 2 3 2>&1 | grep -v Warning | tee -a gpurun_out/r03c_aniso.txt
done
CWR_ORDER_ANISO=2,0.5 timeout -k 10 300 python scratch/r03_chain.py bench 16 400 3 2 4 2>&1 | grep -v Warning | tee -a gpurun_out/r03c_aniso.txt
