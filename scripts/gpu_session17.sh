timeout 1500 python -m pytest tests -m gpu -x -q -k "ctc or CTC" 2>&1 | tail -3 > gpurun_out/s17_tests.txt
for i in 1 2 3; do
for p in new old; do
L=""; [ $p = new ] || L=$PWD/gtn_applications_amd/libwfl_$p.so
WFL_LIB_PATH=$L python bench.py --mode abi --steps 400 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$p', round(d['ms_per_step']*1e3,2), {k: round(v*1e3,2) for k,v in d['roofline']['kernel_ms'].items()}, d['config'].get('utterances_repaired_in_log_domain'))" >> gpurun_out/s17.txt
done; done
for p in new old; do
L=""; [ $p = new ] || L=$PWD/gtn_applications_amd/libwfl_$p.so
WFL_LIB_PATH=$L python bench.py --mode abi --T 2000 --C 512 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5 $p', round(d['ms_per_step']*1e3,2))" >> gpurun_out/s17.txt
done
