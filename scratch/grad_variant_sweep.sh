for w in 2 3 4; do for kb in 24 32 48; do
  echo "== minwaves=$w WFL_GRAD_LDS_KB=$kb"
  WFL_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libwfl_w$w.so WFL_GRAD_LDS_KB=$kb python bench.py --workload transducer --targets same --steps 40 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
done; done
for w in 3 4; do for kb in 24 48; do
  echo "== asg minwaves=$w WFL_GRAD_LDS_KB=$kb"
  WFL_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libwfl_w$w.so WFL_GRAD_LDS_KB=$kb python bench.py --workload asg --targets same --steps 40 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['ms_per_step'], json.dumps(j['roofline']['kernel_ms']))"
done; done
WFL_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libwfl_w4.so python -m pytest tests -x -q -m gpu 2>&1 | tail -2
