cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for round in 1 2; do
for v in nofx fx; do
  for fx in 0 1; do
  [ $v = nofx ] && [ $fx = 1 ] && continue
  echo "== build $v FX=$fx"
  WFL_CTC_MITM_FX=$fx WFL_LIB_PATH=$PWD/scripts/_build/libwfl_mm_${v}t.so python bench.py --mode abi --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['roofline']['kernel_ms'])"
  [ $round = 1 ] && WFL_CTC_MITM_FX=$fx WFL_LIB_PATH=$PWD/scripts/_build/libwfl_mm_${v}s.so timeout 60 python scripts/mitm_stats.py 2>&1 | grep -E "THREE|chain    simd|stager0 |Error|error" | head -12
  done
done
done
