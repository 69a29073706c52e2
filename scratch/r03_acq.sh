#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for lib in clearwater-riverine_amd/libcwr_transport.so scratch/libB_acq_workgroup.so scratch/libB_acq_agent.so; do
  echo "== $lib" | tee -a gpurun_out/r03k_acquire.txt
  CWR_TRANSPORT_LIB=$PWD/$lib timeout -k 10 200 python scratch/r03_stiff.py 16 6 40 chains 2 2>&1 | grep -v Warn | tee -a gpurun_out/r03k_acquire.txt
  CWR_TRANSPORT_LIB=$PWD/$lib timeout -k 10 200 python scratch/r03_stiff.py 16 3 400 chains 4 2>&1 | grep -v Warn | tee -a gpurun_out/r03k_acquire.txt
done
