#!/bin/bash
# End-to-end part of round_final.sh alone (drop-in tests, e2e_timing, 24 and 152 views, bench.py under torchrun with one nccl rank): gpurun_out/final2/.
mkdir -p gpurun_out/final2; cd /root/repo
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_dropin_binary.py > gpurun_out/final2/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/final2/pytest.log | tail -2
timeout 600 tools/e2e_timing.sh both > gpurun_out/final2/e2e_timing.txt 2>&1
timeout 300 tools/lab/tt_like.sh > gpurun_out/final2/e2e_tt24.txt 2>&1
timeout 600 tools/e2e_c4.sh 152 > gpurun_out/final2/e2e_c4_152.txt 2>&1; grep -E "Start-up" /tmp/c4.log >> gpurun_out/final2/e2e_c4_152.txt
timeout 600 tools/e2e_c4.sh 152 --jacobi > gpurun_out/final2/e2e_c4_152_jacobi.txt 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final2/bench_torchrun_1rank.json 2> gpurun_out/final2/bench_torchrun_1rank.err
grep -E "^==|Stages" gpurun_out/final2/e2e_timing.txt; cat gpurun_out/final2/e2e_tt24.txt gpurun_out/final2/e2e_c4_152.txt gpurun_out/final2/e2e_c4_152_jacobi.txt | grep -E "wall|Stages|views 1920|Fusion \+|Start-up"; head -c 200 gpurun_out/final2/bench_torchrun_1rank.json
