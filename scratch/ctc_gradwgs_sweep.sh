for g in 0 256 384 448 512 640 768; do
  echo "== WFL_CTC_GRAD_WGS=$g"
  WFL_CTC_GRAD_WGS=$g python bench.py --mode abi --steps 300 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
done
echo "== default"; python bench.py --mode abi --steps 300 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k ctc 2>&1 | tail -2
