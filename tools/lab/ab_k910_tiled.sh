# K9/K10's sub-patch taps from the 7 x 8 pair tiles (lab build) against the row-major pairs, both with the tiled copy resident
# (--opt tiled_copy=2): launch time + L1 tag accesses / L1->L2 requests per launch.  Output: gpurun_out/lab/ab_k910_tiled.txt
export APD_ALLOW_STALE_LIBRARY=1   # lab builds with ad-hoc flags
O=gpurun_out/lab; mkdir -p $O; export TMPDIR=/tmp
WL="--workload eth3d_pipes_fullres_10src_apd --steps 3 --warmup 1 --opt tiled_copy=2"
for arm in rowmajor tiled; do
  if [ $arm = tiled ]; then APD_EXTRA_FLAGS=-DAPD_K910_SUBPATCH_TILED=1 python apd-mvs_amd/build.py --force > $O/build_$arm.log 2>&1; python tools/lab/k910_tiled_parity.py > $O/parity_$arm.txt 2>&1; tail -n 2 $O/parity_$arm.txt; fi
  for rep in 1 2; do python bench.py --full-line --no-cpu-baseline $WL 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); w=d['weak_path']; print('$arm rep $rep: value', d['value'], 'k910 ms/launch', w['avg_launch_ms'], 'k67 ms/launch', (d.get('strong_path') or {}).get('avg_launch_ms'), 'q', d['quality_within_1pct_depth'])"; done
  mkdir -p $O/pmc_$arm; APD_PROFILE_PASSES=tcp,sq timeout 900 python tools/profile_bench.py $O/pmc_$arm $WL > $O/pmc_$arm.log 2>&1; tail -n 3 $O/pmc_$arm.log
done 2>&1 | tee $O/ab_k910_tiled.txt
python - <<'PY' | tee -a gpurun_out/lab/ab_k910_tiled.txt
import json,glob
for arm in ("rowmajor","tiled"):
    for f in glob.glob("gpurun_out/lab/pmc_%s/pmc_extra_*.json" % arm):
        d=json.load(open(f)); k=d["kernels"].get("k910") or {}
        print(arm, {a: (round(b,1) if isinstance(b,float) else b) for a,b in k.items() if not isinstance(b,(dict,list))})
PY
