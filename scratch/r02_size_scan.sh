#!/bin/bash
# pass time per cell against mesh size (does a working set that fits the 256 MB MALL run faster?)
for N in 354 500 708 1000; do
  scratch/r02_quick.sh "K16 ${N}x${N}" -- --constituents 16 --no-pmc --nx $N --ny $N
done
