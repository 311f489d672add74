for v in 0 1; do echo "== DQN_PRIO_FORK=$v"; if [ $v = 1 ]; then export DQN_PRIO_FORK=1; fi
python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-steps 0 > gpurun_out/ab_b.json 2>/dev/null; python tools/bench_summary.py gpurun_out/ab_b.json | head -12; done
