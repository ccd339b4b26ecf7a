mkdir -p gpurun_out
run() { # label, env...
  lbl=$1; shift
  for i in 1 2; do env "$@" python bench.py --workload asg --targets same --steps 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items()})" >> gpurun_out/s12.txt; done
}
run base A=1
for v in p0 v2 v3 v4; do run $v WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_$v.so; done
run base A=1
