cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
(timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15) > gpurun_out/r4d/tests.txt
cat gpurun_out/r4d/tests.txt
(timeout 1500 python bench.py > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err); tail -3 gpurun_out/r4d/bench.err
python - <<'PY'
import json
l = json.loads(open('gpurun_out/r4d/bench.json').read().strip().splitlines()[-1])
print('value', l['value'], 'roofline', l['roofline']['frac'])
print('train', {k: l['train'].get(k) for k in ('steps_per_s', 'ms_per_step', 'launches_per_step')})
print('indel', l['indel'].get('positions_per_s'), l['indel'].get('roofline', {}).get('frac'))
c = l['config5_e2e']
print('config5', c.get('rows_per_s'), c.get('sink_only'), c.get('error'))
print('long', l['variants'].get('long_window_R4000'))
print('cpu', l['cpu_baseline'].get('value'), l['cpu_baseline'].get('batch_256_repeat_medians'))
PY
