#!/bin/bash
# exit-fault forensics + the product modes, each as a fresh process with a native-stack printer preloaded
out=gpurun_out/r04b_exit.txt; : > $out
gcc -O1 -g -shared -fPIC scratch/segv_trace.c -o gpurun_out/libsegv_trace.so 2>/dev/null
for m in facade_r03 facade_leak engine engine_global facade facade_closed facade_stream; do
  echo "=== $m" >> $out
  LD_PRELOAD=$PWD/gpurun_out/libsegv_trace.so timeout -k 5 120 python -X faulthandler scratch/exit_probe.py $m >> $out 2>&1
  echo "rc=$?" >> $out
done
cat $out
