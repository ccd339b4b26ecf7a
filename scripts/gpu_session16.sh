bash scripts/collect_round.sh r02f > gpurun_out/r02f_collect.log 2>&1
python scripts/host_overhead.py > gpurun_out/r02f/host_overhead.txt 2>&1
bash scratch/timeline.sh asg > gpurun_out/r02f/timeline_cfg3.txt 2>&1
bash scratch/run_fal_alone.sh > gpurun_out/r02f/fal_alone.txt 2>&1
WFL_LIB_PATH=$PWD/gtn_applications_amd/libwfl_dbg.so python scratch/timeline3.py > gpurun_out/r02f/ctc_timeline.txt 2>&1
rm -rf gpurun_out/tl gpurun_out/fal_alone gpurun_out/ks
