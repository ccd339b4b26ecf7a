for i in 1 2; do for v in 0 1; do
echo "== WFL_CTC_XCD=$v"; WFL_CTC_XCD=$v bash scratch/kstats.sh 2>&1 | grep "ctc_fast_pipelined"
WFL_CTC_XCD=$v python bench.py --mode abi --steps 300 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('abi', j['ms_per_step'])"
done; done
for v in 0 1; do echo "== cfg5 WFL_CTC_XCD=$v"; WFL_CTC_XCD=$v python bench.py --workload ctc --T 2000 --C 512 --mode abi --steps 100 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('abi', j['ms_per_step'])"; done
