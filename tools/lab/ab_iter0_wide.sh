# Optimistic bound for 16-byte loads in the window-less first iteration (K5 + iteration 0 of K6/K7): three aligned tile-row loads per
# patch row instead of six dword gathers (results are wrong by construction; timing only).  configs[1]'s shape.
O=gpurun_out/lab; mkdir -p $O
export TUNE_WORKLOAD=eth3d_office_fullres_8src TUNE_STEPS=6
bash tools/tune.sh "-DAPD_LAB_ITER0_WIDE=0" "-DAPD_LAB_ITER0_WIDE=1" "-DAPD_LAB_ITER0_WIDE=0" "-DAPD_LAB_ITER0_WIDE=1" 2>&1 | tee $O/ab_iter0_wide_bound.txt
