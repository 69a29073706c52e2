#!/bin/bash
for r in 1 2; do
  scratch/r02_quick.sh "K4 default r$r" -- --constituents 4 --no-pmc
  scratch/r02_quick.sh "K4 four per lane r$r" CWR_TCL_VW=4 -- --constituents 4 --no-pmc
  scratch/r02_quick.sh "K4 nt stream r$r" CWR_NT_STREAM=1 -- --constituents 4 --no-pmc
done
