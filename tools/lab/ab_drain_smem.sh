# s_waitcnt lgkmcnt(0) at the head of the LDS-window body (APD_WIN_DRAIN_SMEM): headline + whole passes, both arms on one box, A B A B
export APD_ALLOW_STALE_LIBRARY=1   # lab builds with ad-hoc flags
O=gpurun_out/lab; mkdir -p $O
for arm in 1 0 1 0; do
  APD_EXTRA_FLAGS="-DAPD_WIN_DRAIN_SMEM=$arm" python apd-mvs_amd/build.py --force > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -3 /tmp/build.log; }
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-workloads configs2_pipes_apd_3iter --only-workloads configs2_pipes_apd_whole_pass --only-workloads configs2_pipes_apd_geometric_pass --only-workloads configs3_tt1080p_20iter 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('drain $arm: headline', d['value'], 'k67 ms', d['roofline']['avg_launch_ms'], {k:v[:2] for k,v in d['workloads'].items()})"
done 2>&1 | tee $O/ab_drain_smem.txt
