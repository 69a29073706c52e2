#!/bin/bash
for K in 1 2 4 8 16 32; do
  CWR_VERBOSE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --constituents $K 2>&1 >/dev/null | grep "^\[cwr\]" | cut -c1-230
done
