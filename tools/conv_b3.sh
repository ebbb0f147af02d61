#!/bin/bash
# the UNet's conv shapes at 1 view per rank (batch 3, the 8-GPU case), split-K off / auto.  -> stdout
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for sk in 0 auto; do
  [ $sk = auto ] && unset DREAMMAT_CONV_SPLITK || export DREAMMAT_CONV_SPLITK=$sk
  for shape in "3 64 64 320 320" "3 32 32 640 640" "3 16 16 1280 1280" "3 8 8 1280 1280" "3 8 8 2560 1280" "3 16 16 2560 1280" \
               "3 32 32 1280 640" "24 8 8 1280 1280" "24 16 16 1280 1280"; do
    echo -n "splitk=$sk "; $R/tools/_abi_pmc conv $shape 20
  done
done
