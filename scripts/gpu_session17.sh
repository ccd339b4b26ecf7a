for i in 1 2 3; do
for p in 0 1; do
WFL_CTC_THIN=$p python bench.py --mode abi --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('thin=$p', round(d['ms_per_step']*1e3,2), {k: round(v*1e3,2) for k,v in d['roofline']['kernel_ms'].items()})" >> gpurun_out/s17.txt
done; done
