mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "transducer or Transducer or cfg4 or stc or STC or lattice" 2>&1 | tail -3 > gpurun_out/s14_tests.txt
run() { lbl=$1; shift
env "$@" python bench.py --workload transducer --steps 40 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items()})" >> gpurun_out/s14.txt; }
for i in 1 2 3; do run new A=1; run old WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_old.so; done
