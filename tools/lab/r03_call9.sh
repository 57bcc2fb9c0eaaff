#!/bin/bash
OUT=gpurun_out/r03x; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -8 | tee $OUT/pytest.txt
echo "== bench apd"
python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); w=d.get('weak_path') or {}
print('value', d['value'], 'k910', w.get('avg_launch_ms'), 'k67', (d.get('strong_path') or {}).get('avg_launch_ms'))" | tee $OUT/bench_apd.txt
bash tools/lab/bench_quick.sh 2>&1 | tee $OUT/bench_quick.txt
APD_EXTRA_FLAGS="-DAPD_LAB_WIN_STATS" python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 600 python tools/weak_stats.py 2>&1 | grep -v "^HIP\|^ROCm" | tee $OUT/weak_stats.txt
