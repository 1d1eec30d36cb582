#!/bin/bash
# usage (GPU box): tools/dev/ab_rank.sh "RANK WORLD SPP [key=value ...]" NAME...  — tools/dev/rank_one.py once per variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
args=$1; shift
cp $R/rendering-fw_amd/librfwhip.so /tmp/librfwhip_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so; else cp $R/tools/dev/variants/$v.so $R/rendering-fw_amd/librfwhip.so; fi
  echo "$v $(cd $R && python tools/dev/rank_one.py $args 2>/dev/null | tail -1)"
done
cp /tmp/librfwhip_base.so $R/rendering-fw_amd/librfwhip.so
