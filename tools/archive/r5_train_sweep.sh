#!/bin/bash
export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
# the SNV training step (symbol route, batch 4096) with each of the step's A/B switches flipped -- are the defaults still the best?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-36s " "$*"; env "$@" timeout 200 python3 $REPO/tools/bench_train_sym.py 2>&1 | tail -1; }
run X=1
run MURAL_TRAIN_ORDER=1
run MURAL_TRAIN_NO_POOL_FOLD=1
run MURAL_TRAIN_NO_MID_FOLD=1
run MURAL_TRAIN_NO_FIRST_FOLD=1
run MURAL_TRAIN_NO_FOLD=1
run MURAL_TRAIN_LOCAL_OPS=1
run MURAL_TRAIN_HEAD_OPS=1
run MURAL_TRAIN_FIRST_SEPARATE=1
run X=2
