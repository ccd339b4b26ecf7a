cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ctc" 2>&1 | tail -3
for w in 1 0; do echo -n "WIDE=$w cfg5: "; WFL_CTC_MITM_WIDE=$w python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],4), round(r['roofline']['frac'],3), r['roofline']['kernel_ms'], 'abi', round(r['abi_kernels_only']['ms_per_step'],4))"; done
