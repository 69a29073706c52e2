#!/bin/bash
out=gpurun_out/r04b_exit_repro.txt; : > $out
gcc -O1 -g -shared -fPIC scratch/segv_trace.c -o gpurun_out/libsegv_trace.so 2>/dev/null
for env in "X=1" "CWR_NO_SMALL_GS=1"; do
  echo "=== reconstructed r03w library, $env" >> $out
  env $env LD_PRELOAD=$PWD/gpurun_out/libsegv_trace.so timeout -k 5 120 python -X faulthandler scratch/_r03w_repro/repro.py >> $out 2>&1
  echo "rc=$?" >> $out
done
echo "=== current tree, tests/models/ohio_like.py under the same tracer" >> $out
LD_PRELOAD=$PWD/gpurun_out/libsegv_trace.so timeout -k 5 300 python -X faulthandler tests/models/ohio_like.py >> $out 2>&1
echo "rc=$?" >> $out
cat $out
