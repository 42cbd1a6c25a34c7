#!/bin/bash
# the round's GPU calls: tools/call.sh <n>  (each block is one gpurun command line; kept so that profiles/ can name what produced them)
case "$1" in
1) python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/c1_tests.txt
   tools/ab.sh c1_early_ego 3 30 "X=1" "WSMG_EARLY_EGO=0" > gpurun_out/c1_early_ego.txt 2>&1 ;;
esac
