mkdir -p gpurun_out
run() { # label, env...
  lbl=$1; shift
  for i in 1 2; do env "$@" python bench.py --workload asg --targets same --steps 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items() if 'dense_fast_grad' in k})" >> gpurun_out/s12.txt; done
}
run base A=1
run wgs768 WFL_DENSE_GRAD_WGS=768
run wgs1024 WFL_DENSE_GRAD_WGS=1024
run ts16 WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_ts16.so
run ts16_768 WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_ts16.so WFL_DENSE_GRAD_WGS=768
run wgs384 WFL_DENSE_GRAD_WGS=384
