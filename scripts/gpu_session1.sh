#!/bin/bash
# round 2, first GPU session: full GPU suite (incl. the new BASELINE-size parity tests), PMC calibration
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
cd /tmp && export TMPDIR=/tmp
cd "$R"; rm -rf "$O"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_configs.log 2>&1; echo "configs rc=$?" >> $O/pytest_configs.log
tail -30 $O/pytest_configs.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; echo "parity rc=$?" >> $O/pytest_parity.log
tail -5 $O/pytest_parity.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/pmc_calib.hip -o $O/pmc_calib
$O/pmc_calib > $O/pmc_calib.out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- $O/pmc_calib > $O/calib_$c.log 2>&1
  cp $(find $O/calib_$c -name "*counter_collection.csv" | head -1) $O/calib_$c.csv
done
python scripts/pmc_calib.py $O/pmc_calib.out $O/calib_FETCH_SIZE.csv $O/calib_WRITE_SIZE.csv > $O/pmc_calibration.json; cat $O/pmc_calibration.json
rm -rf $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/pmc_calib
