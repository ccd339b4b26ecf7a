for n in 1 16 32; do WFL_PACK_TRACE=1 python scratch/pool_sweep.py $n 2>&1 | tail -4; done
