O=gpurun_out/r5c; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --only-workloads configs2_pipes_apd_3iter --only-workloads configs2_pipes_apd_whole_pass --only-workloads configs2_pipes_apd_geometric_pass --only-workloads configs2_pipes_hard_apd_3iter --only-workloads configs2_pipes_hard_whole_pass > $O/line.json 2> $O/err.txt; cp bench_workloads.json $O/full.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c/full.json'))
for k,w in d['workloads'].items():
    if 'kernel_ms_per_pass' in w: print(k, w['ms_per_pass'], {a:b for a,b in w['kernel_ms_per_pass'].items() if b>15}, w.get('quality_within_1pct_depth'))
    else: print(k, w['value'], w['ms_per_step'], w.get('quality_within_1pct_depth'))
PY
