mkdir -p gpurun_out
run() { # label, env...
  lbl=$1; shift
  env "$@" python bench.py --workload asg --steps 40 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items() if 'dense_fast_chain' in k})" >> gpurun_out/s12.txt
}
for i in 1 2 3; do run base A=1; run hp3 WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_hp3.so; done
