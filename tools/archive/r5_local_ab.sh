#!/bin/bash
export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
# headline with the local branch's fragments in registers (default) / in LDS
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s %.0f %.4f' % ('$*', d['value'], d['roofline']['frac']))"; }
run X=1
run MURAL_LOCAL_REG=0
run X=2
run MURAL_LOCAL_REG=0
