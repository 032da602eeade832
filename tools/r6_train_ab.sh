#!/bin/bash
# A/B runs of the SNV training leg on ONE box: every argument is one "VAR=value[,VAR=value...]" setting (development switches of the
# debug flavour); the unmodified build runs first and last.  e.g. tools/r6_train_ab.sh MURAL_CW_FULL_GRID=1 MURAL_SIDE_PRIORITY=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${STEPS:-400}
run() { printf "%-60s " "$1"; env ${1//,/ } timeout 300 python3 $REPO/tools/train_only.py $STEPS 2>&1 | tail -1; }
run X=base
for S in "$@"; do run "$S"; done
run X=base
