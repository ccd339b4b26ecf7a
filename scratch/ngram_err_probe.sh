# which arithmetic the n-gram parity error comes from: the same tests with variant builds of the library
for lib in "" scripts/_build/libwfl_acc1.so scripts/_build/libwfl_acc2.so; do
  echo "== ${lib:-default}"
  WFL_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python -m pytest tests/test_gpu_ngram.py -q -x -k "ngram_transitions and 2 or backoff" 2>&1 | tail -2
  python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_r04_ngram.json'))
for k,v in d.items():
    if 'dx' in k or 'dparams' in k: print('  ',k.ljust(20), 'err/scale %.3g'%v['max_abs_err_over_scale'], 'err/counts %.3g'%v.get('max_err_over_counts',0))
PY
done
