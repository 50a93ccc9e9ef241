set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/soak_tail; mkdir -p $out
for seed in 1 2 3; do timeout -k 10 300 python3 tools/soak_evidence_tail.py $seed 250 2>$out/err_$seed.txt | tail -1; done > $out/soak_evidence_tail.json
cat $out/soak_evidence_tail.json
tail -3 $out/err_1.txt
