for r in 1 2; do
for d in . _ab_head; do (cd $d && python bench.py --no-train --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['roofline']['frac'])"); done
done
