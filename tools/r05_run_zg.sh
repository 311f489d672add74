R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/zg -o r -- python $R/tools/adam_mode_probe.py --child > $R/gpurun_out/zg.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/zg/r_results.db | head -8
tail -3 $R/gpurun_out/zg.log
rm -rf $R/gpurun_out/zg
