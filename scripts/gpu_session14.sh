mkdir -p gpurun_out
run() { lbl=$1; shift
for i in 1 2; do env "$@" python bench.py --workload transducer --steps 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items()})" >> gpurun_out/s14.txt; done; }
run base A=1
run tight WFL_CHAIN_TIGHT=1
WFL_CHAIN_TIGHT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "transducer or Transducer or cfg4 or stc or STC" 2>&1 | tail -3 >> gpurun_out/s14.txt
