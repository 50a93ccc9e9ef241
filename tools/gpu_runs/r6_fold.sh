set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fold; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "finalisation_kernels_tail" > $out/pytest.log 2>&1
echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
