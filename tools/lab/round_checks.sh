mkdir -p gpurun_out/r04g; cd /root/repo
# 1. bench.py under torchrun with one rank: the RCCL code path (process group, exchange inside the timed region of the C4-shaped sub-line)
(time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline) > gpurun_out/r04g/bench_torchrun_1rank.json 2> gpurun_out/r04g/bench_torchrun_1rank.err
tail -3 gpurun_out/r04g/bench_torchrun_1rank.err
# 2. e2e on the final build
tools/e2e_timing.sh both > gpurun_out/r04g/e2e_timing.txt 2>&1
tools/lab/tt_like.sh > gpurun_out/r04g/e2e_tt24.txt 2>&1
tools/e2e_c4.sh 152 > gpurun_out/r04g/e2e_c4_152.txt 2>&1; grep -E "Start-up" /tmp/c4.log >> gpurun_out/r04g/e2e_c4_152.txt
tools/e2e_c4.sh 152 --jacobi > gpurun_out/r04g/e2e_c4_152_jacobi.txt 2>&1
cat gpurun_out/r04g/e2e_c4_152.txt gpurun_out/r04g/e2e_c4_152_jacobi.txt
# 3. parity fuzz on the final build (split passes and recycled handles in half of the cases)
python tools/parity_fuzz.py 500 30000 > gpurun_out/r04g/parity_fuzz_500.txt 2>&1; tail -2 gpurun_out/r04g/parity_fuzz_500.txt
python tools/fusion_fuzz.py 40 500 > gpurun_out/r04g/fusion_fuzz_40.txt 2>&1; tail -2 gpurun_out/r04g/fusion_fuzz_40.txt
