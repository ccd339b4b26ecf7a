mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "asg or ASG or dense" 2>&1 | tail -3 > gpurun_out/s12_tests.txt
run() { # label, env...
  lbl=$1; shift
  env "$@" python bench.py --workload asg --steps 40 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items() if 'dense_fast_chain' in k})" >> gpurun_out/s12.txt
}
for i in 1 2 3; do run new A=1; run old WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_old.so; done
