#!/bin/bash
# round 4: the pipelined Gauss-Seidel small-mesh solver (scratch/r03_small_gs.patch re-applied on branch small-gs) with the
# stiffness guard (only steps with ||J||inf >= 0.9) and one cross-section per group
out=gpurun_out/r04ab_small_gs.txt
: > $out
for env in "" "CWR_NO_SMALL_GS=1" "CWR_SMALL_GS_ROUND=48" "CWR_SMALL_GS_ROUND=24"; do
  echo "== ${env:-default}" >> $out
  env $env python tests/models/ohio_like.py 2>&1 | grep -v Warn | head -2 >> $out
done
cat $out
