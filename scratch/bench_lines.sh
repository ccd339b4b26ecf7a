rm -f gpurun_out/bench_lines.jsonl
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
python bench.py --workload asg 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
python bench.py --workload transducer 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
python bench.py --workload ctc --T 2000 --C 512 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
