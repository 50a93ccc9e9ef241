set -x
mkdir -p gpurun_out/r4
bash tools/ab_kernels.sh "--config 5 --steps 3" ab/new.so ab/exp1.so ab/exp2.so ab/exp3.so > gpurun_out/r4/ab_exp.txt 2>&1
bash tools/ab_kernels.sh "--config 5 --steps 3" ab/new.so ab/exp1.so ab/exp2.so ab/exp3.so >> gpurun_out/r4/ab_exp.txt 2>&1
cut -c1-60 gpurun_out/r4/ab_exp.txt
