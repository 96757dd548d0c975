#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh PHOTONS "CONFIGS" lib1.so lib2.so ... (two rounds, interleaved)
n=$1; cfgs=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    RTMC_LIBRARY=$PWD/$lib timeout -k 10 100 python tools/rate.py $n $cfgs || exit 1
  done
done
