#!/bin/bash
# ping-pong form of the two-stage conv / GEMM kernels (DREAMMAT_CONV_PP=1) against the default schedule, same box, alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
shapes=("8 512 512 128 128" "8 256 256 256 256" "8 128 128 512 512" "24 16 16 1280 1280" "8 256 256 128 256")
gemms=("98304 320 2560 1" "24576 640 5120 1" "98304 1280 320 0" "24576 640 640 0")
for pass in 1 2; do
  for pp in 0 1; do
    for s in "${shapes[@]}"; do
      out=$(DREAMMAT_CONV_PP=$pp $R/tools/_abi_pmc conv $s 10)
      echo "pp=$pp $pass conv $s $(echo $out | sed 's|.*"ms":\([0-9.]*\),"TFLOPs":\([0-9.]*\).*|\1 ms \2 TF/s|')"
    done
    for s in "${gemms[@]}"; do
      out=$(DREAMMAT_CONV_PP=$pp $R/tools/_abi_pmc gemm $s 10)
      echo "pp=$pp $pass gemm $s $(echo $out | sed 's|.*"ms":\([0-9.]*\),"TFLOPs":\([0-9.]*\).*|\1 ms \2 TF/s|')"
    done
  done
done
