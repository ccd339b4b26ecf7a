mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "asg or ASG or lattice or dense" 2>&1 | tail -3 > gpurun_out/s12_tests.txt
run() { # label, env...
  lbl=$1; shift
  for i in 1 2; do env "$@" python bench.py --workload asg --targets same --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lbl', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms'].items()})" >> gpurun_out/s12.txt; done
}
run band A=1
run d24 WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_d24.so
run d31 WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_d31.so
bash scratch/run_fal_alone.sh > gpurun_out/fal_alone.txt 2>&1
