# rocprofv3 --kernel-trace --stats of the 24-view 1080p end-to-end run (APD folder 0, six views in flight) on the final build
O=gpurun_out/lab; mkdir -p $O; export TMPDIR=/tmp; cd /tmp || exit 1
python $GRAFT_REPO_ROOT/tools/make_synthetic_dense.py /tmp/tt24 --width 1920 --height 1080 --views 24 --src 10 --textureless 0.2 --jpeg > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/apd_trace_e2e -o trace -- $GRAFT_REPO_ROOT/apd-mvs_amd/_build/APD /tmp/tt24 0 --seed 12345 --clean-exit > $GRAFT_REPO_ROOT/$O/e2e_trace.log 2>&1
find /tmp/apd_trace_e2e -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/kernel_stats_e2e_tt24_final.csv \;
head -12 $GRAFT_REPO_ROOT/$O/kernel_stats_e2e_tt24_final.csv | cut -c1-200; grep Stages $GRAFT_REPO_ROOT/$O/e2e_trace.log
