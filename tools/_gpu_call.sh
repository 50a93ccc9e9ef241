set -x
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/ab_taper_cat.txt
for i in 1 2; do
for t in 1 3; do
  echo "taper $t" >> gpurun_out/r4/ab_taper_cat.txt
  DLAGP_OPTIONS=plan_taper=$t python3 tools/measure_catalogue.py 2048 10000 32 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/r4/ab_taper_cat.txt
done; done
cat gpurun_out/r4/ab_taper_cat.txt
