cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -- python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/k4.log 2>&1
python scripts/kstats.py /tmp/k4 | head -12
for v in 1 0; do WFL_LATTICE_OCC_GRAD=$v python bench.py --config cfg4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('occ=$v', r['ms_per_step'], r['roofline']['kernel_ms'])"; done
