set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "ctc or CTC" 2>&1 | tail -3 > gpurun_out/s11_tests.txt
for i in 1 2 3; do
python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1 >> gpurun_out/s11_cfg2.jsonl
done
WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_dbg.so python scratch/timeline3.py > gpurun_out/tl3.txt 2>&1
python bench.py --workload ctc --T 2000 --C 512 2>/dev/null | tail -1 > gpurun_out/s11_cfg5.json
