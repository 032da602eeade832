#!/bin/bash
# A/B runs of the INDEL forward leg on ONE box (development switches of the debug flavour), as tools/r6_train_ab.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export MURAL_HIP_FLAVOR=debug
run() { printf "%-50s " "$1"; env ${1//,/ } timeout 300 python3 $REPO/tools/r6_indel_leg.py 2>&1 | tail -1; }
run X=base
for S in "$@"; do run "$S"; done
run X=base
