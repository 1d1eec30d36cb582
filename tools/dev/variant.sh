#!/bin/bash
# usage: tools/dev/variant.sh NAME "<extra hipcc flags>"  — build tools/dev/variants/NAME.so: kernels.hip recompiled with the extra
# flags (-DRT_LEAF_VOTE_EXT=48 ...) and linked with the objects of the last full build.  tools/dev/ab_lib.sh runs them on the box.
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/rendering-fw_amd/csrc
mkdir -p $R/tools/dev/variants
o=$R/tools/dev/variants/$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I$R/include -I$C -Wno-unused-function -Wno-unused-result $2 \
  -c $C/kernels.hip -o $o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/rfwhip_api.cpp.o $C/rfwhip_group.cpp.o $C/bvh_build.cpp.o $o $C/lbvh.hip.o \
  -o $R/tools/dev/variants/$1.so -lpthread -ldl && rm -f $o && echo built $1
