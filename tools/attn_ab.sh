#!/bin/bash
# attention kernel A/B over library variants (tools/build_variant.sh) on one box: tools/attn_ab.sh <variant> [<variant>...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/dreammat_amd/csrc/_obj
shapes=("24 5 4096 4096 64" "24 10 1024 1024 64")
for pass in 1 2; do
  for v in main "$@"; do
    for s in "${shapes[@]}"; do
      if [ $v = main ]; then out=$($R/tools/_abi_pmc attn $s 10 w128); else out=$(LD_LIBRARY_PATH=$O/$v:$LD_LIBRARY_PATH $R/tools/_abi_pmc attn $s 10 w128); fi
      echo "$v $pass $s $(echo $out | sed 's|.*"ms":\([0-9.]*\),"TFLOPs":\([0-9.]*\).*|\1 ms \2 TF/s|')"
    done
  done
done
