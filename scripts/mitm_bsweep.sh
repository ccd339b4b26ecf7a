#!/bin/bash
# GPU box: CTC tests with the 8-wave shape forced, then the ABI step at B = 128 .. 1024 for both shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WFL_CTC_MITM_WAVES=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ctc" 2>&1 | tail -3
for w in 16 8; do for b in 128 256 512 1024; do
  echo -n "waves $w B=$b: "
  WFL_CTC_MITM_WAVES=$w python bench.py --mode abi --B $b --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms  frac %.3f' % (r['ms_per_step'], r['roofline']['frac']))"
done; done
