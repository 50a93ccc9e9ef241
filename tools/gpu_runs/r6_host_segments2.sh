set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/host_segments; mkdir -p $out
python3 tools/route_a_host_segments.py 256 > $out/segments.json 2> $out/segments.err
python3 tools/measure_route_a.py 256 2>/dev/null | tail -1 > $out/plain.json
tail -3 $out/segments.err
