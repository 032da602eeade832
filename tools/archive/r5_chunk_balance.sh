export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
run() { b=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-train --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-30s batch %d: %.0f %.4f' % ('$*', $b, d['value'], d['roofline']['frac']))"; }
run 524288 X=1
run 491520 MURAL_SNV_CHUNK=122880
run 524288 X=1
run 491520 MURAL_SNV_CHUNK=122880
run 491520 X=1
