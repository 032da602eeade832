export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.0f %.4f' % ('$*', d['value'], d['roofline']['frac']))"; }
run X=1
run MURAL_SNV_DEFER_SHORT=0
run X=2
run MURAL_SNV_DEFER_SHORT=0
