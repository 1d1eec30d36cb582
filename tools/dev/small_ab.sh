#!/bin/bash
# usage (GPU box): tools/dev/small_ab.sh NAME...  — pipelined 1-spp and 8-spp 1080p frames per variant library (tools/dev/variant.sh); "base" = the tree's
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/rendering-fw_amd/librfwhip.so /tmp/librfwhip_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so; else cp $R/tools/dev/variants/$v.so $R/rendering-fw_amd/librfwhip.so; fi
  echo "$v: $(cd $R && python tools/dev/small_trace.py 1 2>&1 | tail -1) | $(cd $R && python tools/dev/small_trace.py 8 2>&1 | tail -1) | $(cd $R && python tools/dev/small_trace.py 32 2>&1 | tail -1)"
done
cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so
