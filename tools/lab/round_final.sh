#!/bin/bash
# The measurement set a round ends with, one gpurun call: full GPU suite, counter profiles + bench lines (tools/profile_round.sh), e2e.
# Usage (from the repo root): gpurun --timeout 3600 -- 'bash tools/lab/round_final.sh'; results under gpurun_out/final/ (copy what is to be judged into profiles/rNN/).
mkdir -p gpurun_out/final; cd /root/repo
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/final/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/final/pytest.log | tail -3
(time timeout 1500 bash tools/profile_round.sh gpurun_out/final/profile_round r04) > gpurun_out/final/profile_round.log 2>&1; tail -32 gpurun_out/final/profile_round.log | cut -c1-220
timeout 600 tools/e2e_timing.sh both > gpurun_out/final/e2e_timing.txt 2>&1
timeout 300 tools/lab/tt_like.sh > gpurun_out/final/e2e_tt24.txt 2>&1
timeout 600 tools/e2e_c4.sh 152 > gpurun_out/final/e2e_c4_152.txt 2>&1; grep -E "Start-up" /tmp/c4.log >> gpurun_out/final/e2e_c4_152.txt
timeout 600 tools/e2e_c4.sh 152 --jacobi > gpurun_out/final/e2e_c4_152_jacobi.txt 2>&1
grep -E "^==|Stages" gpurun_out/final/e2e_timing.txt; cat gpurun_out/final/e2e_tt24.txt gpurun_out/final/e2e_c4_152.txt gpurun_out/final/e2e_c4_152_jacobi.txt | grep -E "wall|Stages|views 1920"
# bench.py under torchrun with one rank: the RCCL code path (process group, exchange inside the timed region of the C4-shaped sub-line)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_torchrun_1rank.json 2> gpurun_out/final/bench_torchrun_1rank.err; head -c 300 gpurun_out/final/bench_torchrun_1rank.json
