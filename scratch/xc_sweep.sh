for g in 5 10 20 40 63; do for ga in 0 1; do
echo "== gx=$g gather=$ga"; WFL_XC_GX=$g WFL_XC_GATHER=$ga bash scratch/kstats.sh --workload ctc --T 2000 --C 512 2>&1 | grep compact_x
done; done
