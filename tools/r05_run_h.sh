#!/bin/bash
tag=r05_h
bash tools/gpu_profile.sh $tag --replay 10000 --env-steps 0
bash tools/gpu_pmc.sh ${tag}_fetch "FETCH_SIZE" --env-steps 0
bash tools/gpu_pmc.sh ${tag}_write "WRITE_SIZE" --env-steps 0
bash tools/gpu_pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --env-steps 0
cat gpurun_out/${tag}_step.txt
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2; do for w in 0 256 128 64; do
  $B --dw-wgs $w 2>/dev/null > /tmp/x.json; python tools/bench_summary.py /tmp/x.json | grep -E "steps/s|adam|dw_conv" | tr '\n' ' ' | sed "s|^|dw_wgs=$w |"; echo
done; done 2>&1 | tee gpurun_out/${tag}_dw_wgs.txt
