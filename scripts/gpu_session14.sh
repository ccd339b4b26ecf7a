mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "transducer or Transducer or cfg4 or lattice or stc or STC" 2>&1 | tail -3 > gpurun_out/s14_tests.txt
for i in 1 2; do python bench.py --workload transducer --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'fresh', round(d.get('fresh_targets',{}).get('ms_per_step',0),4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items()})" >> gpurun_out/s14.txt; done
