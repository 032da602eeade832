#!/bin/bash
export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
# round 6: the SNV training step with the conv launches asking for fewer workgroup slots (so that the two towers' launches are
# co-resident instead of time-sharing the CUs): MURAL_CW_WGS_{LARGE9,MID9,SHORT}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-70s " "$*"; env "$@" timeout 200 python3 $REPO/tools/bench_train_sym.py 2>&1 | tail -1; }
run X=1
run MURAL_CW_WGS_MID9=256 MURAL_CW_WGS_SHORT=256
run MURAL_CW_WGS_MID9=256 MURAL_CW_WGS_SHORT=256 MURAL_CW_PCAP=2
run MURAL_CW_WGS_MID9=256 MURAL_CW_WGS_SHORT=128 MURAL_CW_PCAP=4
run MURAL_CW_WGS_MID9=256 MURAL_CW_WGS_SHORT=128 MURAL_CW_PCAP=2
run MURAL_CW_WGS_MID9=256 MURAL_CW_WGS_SHORT=64 MURAL_CW_PCAP=4
run MURAL_CW_WGS_MID9=256 MURAL_CW_WGS_SHORT=512 MURAL_CW_PCAP=1
run X=2
