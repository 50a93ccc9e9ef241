set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6; mkdir -p $out
DLAGP_PROFILE=1 timeout -k 10 300 python3 tools/measure_route_a.py 256 > $out/route_a_profile.txt 2>&1
head -60 $out/route_a_profile.txt | cut -c1-170
