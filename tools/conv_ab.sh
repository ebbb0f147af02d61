#!/bin/bash
# conv kernel A/B over library variants (tools/build_variant.sh) on one box: tools/conv_ab.sh <variant> [<variant>...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/dreammat_amd/csrc/_obj
shapes=("8 512 512 128 128" "8 256 256 256 256" "24 64 64 320 320" "24 32 32 640 640" "24 64 64 640 320" "8 256 256 128 256")
for pass in 1 2; do
  for v in main "$@"; do
    for s in "${shapes[@]}"; do
      if [ $v = main ]; then out=$($R/tools/_abi_pmc conv $s 10); else out=$(LD_LIBRARY_PATH=$O/$v:$LD_LIBRARY_PATH $R/tools/_abi_pmc conv $s 10); fi
      echo "$v $pass $s $(echo $out | sed 's|.*"ms":\([0-9.]*\),"TFLOPs":\([0-9.]*\).*|\1 ms \2 TF/s|')"
    done
  done
done
