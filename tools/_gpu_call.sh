set -x
mkdir -p gpurun_out/r4
for i in 1 2; do
for lib in base new; do
  DLAGP_LIB=$PWD/ab/$lib.so python3 tools/measure_catalogue.py 2048 10000 32 2>/dev/null | tail -1 | cut -c1-140 >> gpurun_out/r4/ab_catalogue3.txt
done; done
cat gpurun_out/r4/ab_catalogue3.txt
