#!/bin/bash
# A/B of several builds of libfthmc_hip.so on the training gradient, ONE device, ONE call: tools/ab_train.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for i in $(seq 1 $R); do
  for lib in "$@"; do FTHMC_LIB=$PWD/$lib python3 tools/train_ab.py 2>/dev/null; done
done
