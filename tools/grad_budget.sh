#!/bin/bash
# cfg2 gradient error budget: default library vs the same library with shade.hip built WITHOUT -ffast-math, each with the
# rgb18e8 and the fp32 atlas.  Run on the GPU box from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
C=dreammat_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -c $C/shade.hip -o /tmp/shade_nofast.o || exit 1
objs=$(ls $C/_obj/*.o | grep -v "/shade.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdm_nofast.so $objs /tmp/shade_nofast.o || exit 1
python tools/grad_budget.py fastmath
DREAMMAT_LIB=/tmp/libdm_nofast.so python tools/grad_budget.py nofastmath
python tools/grad_budget.py fastmath-masked
