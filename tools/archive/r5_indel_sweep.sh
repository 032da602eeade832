#!/bin/bash
export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
# the INDEL forward (8192 positions, packed entry) with each of its A/B switches flipped -- are the defaults still the best?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-36s " "$*"; env "$@" timeout 200 python3 $REPO/tools/bench_indel.py 8192 packed 2>&1 | tail -1; }
run X=1
run MURAL_INDEL_ENC0_DOWN=0
run MURAL_INDEL_ENC0=0
run MURAL_INDEL_DEC0=0
run MURAL_INDEL_DEEP=0
run MURAL_INDEL_DEEP_FRONT=0
run MURAL_DEBUG_POLY_NARROW=1
run MURAL_CONVBLOCK_DIRECT=0
run MURAL_CONV1D_DIRECT=0
run MURAL_XCD_SWIZZLE=0
run MURAL_INDEL_ENC0_WGS=6
run MURAL_INDEL_ENC0_WGS=5
run MURAL_INDEL_DEC0_WGS=3
run X=2
