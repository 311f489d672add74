#!/bin/bash
timeout 1200 bash tools/gpu_evidence.sh r05_w
timeout 400 bash tools/gpu_cfg5.sh r05_w
timeout 120 python tools/drqn_bench.py --profile > gpurun_out/r05_w_drqn_config4.txt 2>&1; tail -4 gpurun_out/r05_w_drqn_config4.txt
timeout 120 python tools/cfg1_bench.py --profile > gpurun_out/r05_w_config1.txt 2>&1; head -3 gpurun_out/r05_w_config1.txt
timeout 120 bash tools/gpu_pmc_cfg5.sh r05_w_cfg5_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 2>&1 | tail -2
