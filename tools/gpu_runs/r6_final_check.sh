set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final_check; mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
python3 -m pytest tests -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python3 bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
