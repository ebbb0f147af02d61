#!/bin/bash
# fused GEMM (1-tap) kernel A/B over library variants (tools/build_variant.sh) on one box: tools/gemm_ab.sh <variant> [<variant>...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/dreammat_amd/csrc/_obj
shapes=("98304 320 320 0" "98304 320 2560 1" "98304 1280 320 0" "24576 640 640 0" "24576 2560 640 0" "24576 640 5120 1" "6144 1280 1280 0")
for pass in 1 2; do
  for v in main "$@"; do
    for s in "${shapes[@]}"; do
      if [ $v = main ]; then out=$($R/tools/_abi_pmc gemm $s 10); else out=$(LD_LIBRARY_PATH=$O/$v:$LD_LIBRARY_PATH $R/tools/_abi_pmc gemm $s 10); fi
      echo "$v $pass $s x $(echo $out | sed 's|.*"ms":\([0-9.]*\),"TFLOPs":\([0-9.]*\).*|\1 ms \2 TF/s|')"
    done
  done
done
