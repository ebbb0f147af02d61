#!/bin/bash
# the UNet's Linear shapes (16 images) on the fused 1-tap kernel: tile variants side by side.  -> stdout
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for shape in "98304 320 320 0" "98304 1280 320 0" "24576 640 640 0" "24576 2560 640 0" "65536 320 320 0" "65536 320 2560 1" "65536 1280 320 0" "16384 640 640 0" "16384 640 5120 1" "16384 2560 640 0" \
             "4096 1280 1280 0" "4096 1280 10240 1" "4096 5120 1280 0" "1024 1280 1280 0" "1232 1024 320 0"; do
  for tile in 256 320 512; do
    echo -n "tile=$tile "; DREAMMAT_GEMM_TILE=$tile $R/tools/_abi_pmc gemm $shape 10
  done
done
