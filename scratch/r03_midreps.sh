#!/bin/bash
export TMPDIR=/tmp
for r in default 3 4 6 8; do
  if [ $r = default ]; then unset CWR_LOCAL_REPS; else export CWR_LOCAL_REPS=$r; fi
  echo "== CWR_LOCAL_REPS=$r"; timeout -k 10 200 python tests/models/ohio_like.py 2>&1 | grep "^n=8000\|^n=10000" | sed 's/oracle CPU.*iters/iters/'
done
