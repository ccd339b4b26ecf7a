python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for kb in 24 32 40 64; do
  echo "== WFL_GRAD_LDS_KB=$kb"
  for w in transducer asg; do
  WFL_GRAD_LDS_KB=$kb python bench.py --workload $w --targets same --steps 40 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
  done
done
