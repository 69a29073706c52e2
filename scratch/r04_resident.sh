#!/bin/bash
# round 4: resident passes (k_sq_resident) A/B on small and mid-size engines
out=gpurun_out/r04ag_resident.txt
: > $out
for env in "CWR_NO_RESIDENT=1" "CWR_VERBOSE=0"; do
  echo "== $env" >> $out
  env $env timeout -k 10 300 python tests/models/ohio_like.py 2>&1 | grep -v Warn >> $out || { echo "FAILED rc=$?" >> $out; cat $out; exit 1; }
done
cat $out
