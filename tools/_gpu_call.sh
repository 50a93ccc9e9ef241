set -x
mkdir -p gpurun_out/r4
bash tools/ab_catalogue_kernels.sh "1024 10000 32" ab/base.so ab/new.so ab/base.so ab/new.so > gpurun_out/r4/ab_catalogue2.txt 2>&1
cut -c1-400 gpurun_out/r4/ab_catalogue2.txt
