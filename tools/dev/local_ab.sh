#!/bin/bash
# usage (GPU box): tools/dev/local_ab.sh "<scenes>" NAME...  — tools/dev/local_check.py per variant library ("base" = the tree's)
R=${GRAFT_REPO_ROOT:-/root/repo}
scenes=$1; shift
cp $R/rendering-fw_amd/librfwhip.so /tmp/librfwhip_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so; else cp $R/tools/dev/variants/$v.so $R/rendering-fw_amd/librfwhip.so; fi
  echo "== $v"; (cd $R && timeout 600 python tools/dev/local_check.py $scenes 2>&1 | tail -12)
done
cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so
