for kb in 24 32 48 64 96; do
  echo "== WFL_GRAD_LDS_KB=$kb"
  WFL_GRAD_LDS_KB=$kb python bench.py --workload transducer --targets same --steps 50 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
done
for kb in 24 48; do
  echo "== asg WFL_GRAD_LDS_KB=$kb"
  WFL_GRAD_LDS_KB=$kb python bench.py --workload asg --targets same --steps 50 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
done
