#!/bin/bash
OUT=gpurun_out/r03u; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest split"; timeout 1200 python -m pytest tests/test_gpu_properties.py::test_split_weak_sweep_changes_no_bit tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -15 | tee $OUT/pytest.txt
for o in "" "--opt weak_split=0"; do
  echo "== bench apd $o"
  python bench.py --workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); w=d.get('weak_path') or {}
print('value', d['value'], 'k67', d['roofline']['avg_launch_ms'] if d['roofline'].get('kernel','').startswith('k67') else (d.get('strong_path') or {}).get('avg_launch_ms'), 'k910', w.get('avg_launch_ms'), d['roofline'].get('avg_launch_ms'))"
done 2>&1 | tee $OUT/bench_split.txt
export TUNE_WORKLOAD=eth3d_pipes_fullres_10src_apd TUNE_STEPS=3
tools/tune.sh "-DAPD_K910A_LDS_PAD=3072" "-DAPD_K910A_LDS_PAD=6144" "-DAPD_K910A_LDS_PAD=10240" "-DAPD_K910A_WAVES=3" 2>&1 | tee $OUT/ab_k910a_occupancy.txt
