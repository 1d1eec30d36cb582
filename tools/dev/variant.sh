#!/bin/bash
# usage: tools/dev/variant.sh NAME "<extra hipcc flags>"  — build tools/dev/variants/NAME.so: the core sources recompiled with the
# extra flags (-DRT_LEAF_VOTE_EXT=48 ...).  tools/dev/ab_lib.sh runs them on the box.
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/rendering-fw_amd/csrc
V=$R/tools/dev/variants
mkdir -p $V/$1.d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$R/include -I$C -Wno-unused-function -Wno-unused-result $2"
for s in rfwhip_api.cpp kernels.hip lbvh.hip; do /opt/rocm/bin/hipcc $F -c $C/$s -o $V/$1.d/$s.o || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $V/$1.d/rfwhip_api.cpp.o $C/rfwhip_group.cpp.o $C/bvh_build.cpp.o $V/$1.d/kernels.hip.o $V/$1.d/lbvh.hip.o \
  -o $V/$1.so -lpthread -ldl && rm -rf $V/$1.d && echo built $1
