set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/s10_tests.txt
for i in 1 2 3; do python bench.py --workload transducer 2>/dev/null | tail -1 >> gpurun_out/s10_cfg4.jsonl; done
python scripts/host_overhead.py > gpurun_out/s10_host.txt 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/s10_cfg2.json
