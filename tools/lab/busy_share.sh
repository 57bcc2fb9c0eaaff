#!/bin/bash
# How much of the passes' wall time has at least one kernel in flight (and how many on average): kernel trace of the 24-view 1080p run.
O=gpurun_out/busy; mkdir -p $O; cd /root/repo; export TMPDIR=/tmp
d=/tmp/tt24; rm -rf $d; mkdir -p $d
python tools/make_synthetic_dense.py $d --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
for mode in "" "--jacobi"; do
rm -rf /tmp/prof_busy $d/APD
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_busy -o t -- apd-mvs_amd/_build/APD $d 0 --seed 7 --clean-exit $mode > $O/run.log 2>&1
f=$(find /tmp/prof_busy -name "*kernel_trace.csv" | head -1)
python - "$f" "$mode" <<'PY' | tee -a $O/busy_share.txt
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
busy = 0; depth = 0; last = t0; weighted = 0
for t, d in ev:
    if depth > 0:
        busy += t - last
    weighted += depth * (t - last)
    last = t; depth += d
big = [(s, e) for s, e, n in rows if any(k in n for k in ("k67", "k910", "k14", "k15"))]
print("APD folder 0 %s: %d launches over %.3f s; >= 1 kernel in flight %.1f %% of it; mean kernels in flight %.2f; the four big kernels hold %.1f %% of the kernel-seconds"
      % (sys.argv[2], len(rows), (t1 - t0) / 1e9, 100.0 * busy / (t1 - t0), weighted / (t1 - t0), 100.0 * sum(e - s for s, e in big) / sum(e - s for s, e, _ in rows)))
PY
grep Stages $O/run.log | tee -a $O/busy_share.txt
done
