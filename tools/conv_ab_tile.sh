#!/bin/bash
# conv_ab.sh with one tile variant forced (DREAMMAT_CONV_TILE=$1): tools/conv_ab_tile.sh <tile> <variant>...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/dreammat_amd/csrc/_obj
tile=$1; shift
shapes=("8 512 512 128 128" "8 256 256 128 128" "8 256 256 256 256" "24 64 64 320 320")
for pass in 1 2; do
  for v in main "$@"; do
    for s in "${shapes[@]}"; do
      if [ $v = main ]; then out=$(DREAMMAT_CONV_TILE=$tile $R/tools/_abi_pmc conv $s 10); else out=$(DREAMMAT_CONV_TILE=$tile LD_LIBRARY_PATH=$O/$v:$LD_LIBRARY_PATH $R/tools/_abi_pmc conv $s 10); fi
      echo "$v $pass $s $(echo $out | sed 's|.*"ms":\([0-9.]*\),"TFLOPs":\([0-9.]*\).*|\1 ms \2 TF/s|')"
    done
  done
done
