#!/bin/bash
export TMPDIR=/tmp
for g in 4 5; do for r in 16 32 48; do
echo "== group 2^$g, round $r"; CWR_SMALL_GS_GROUP=$g CWR_SMALL_GS_ROUND=$r timeout -k 10 100 python tests/models/ohio_like.py > /tmp/o.txt 2>&1; echo "rc=$?"; grep "^n=2943" /tmp/o.txt
done; done
echo "== jacobi"; CWR_NO_SMALL_GS=1 timeout -k 10 100 python tests/models/ohio_like.py > /tmp/o.txt 2>&1; echo "rc=$?"; grep "^n=2943" /tmp/o.txt; tail -3 /tmp/o.txt
