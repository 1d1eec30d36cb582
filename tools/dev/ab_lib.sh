#!/bin/bash
# usage (GPU box): tools/dev/ab_lib.sh "<bench args>" NAME...  — one tools/ab.sh bench per variant library (tools/dev/variant.sh),
# "base" = the library of the tree; the tree's library is restored at the end
R=${GRAFT_REPO_ROOT:-/root/repo}
args=$1; shift
cp $R/rendering-fw_amd/librfwhip.so /tmp/librfwhip_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so; else cp $R/tools/dev/variants/$v.so $R/rendering-fw_amd/librfwhip.so; fi
  echo -n "$v " | tee -a $R/gpurun_out/ab.log; (cd $R && tools/ab.sh "$args")
done
cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so
