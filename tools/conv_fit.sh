#!/bin/bash
# fixed (prologue + epilogue) vs per-K cost of the conv kernel on the VAE's 512^2 layers: time at Cin = 128/256/512 for
# the same M and Cout, every tile variant.  -> gpurun_out/conv_fit.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for tile in 640 256 512; do
  for cin in 128 256 512; do
    echo -n "tile=$tile cin=$cin "; DREAMMAT_CONV_TILE=$tile $R/tools/_abi_pmc conv 8 512 512 $cin 128 10
  done
done
for tile in 512 320 256; do
  for cin in 320 640 1280; do
    echo -n "tile=$tile B24 64x64 cin=$cin->320 "; DREAMMAT_CONV_TILE=$tile $R/tools/_abi_pmc conv 24 64 64 $cin 320 10
  done
done
