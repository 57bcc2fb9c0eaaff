python -m pytest tests/test_gpu_parity.py -x -q -k "three_pass or fuzz_sweep_easy or randomised or golden" 2>&1 | tail -5
python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline --full-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value',d['value'],'ms/step',d['ms_per_step'])
print(d['kernel_ms_timed_region'])
"
