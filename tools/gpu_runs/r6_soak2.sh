set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 tools/soak_pipeline.py 1024 10000 3 2>/dev/null | tail -1 | cut -c1-900
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or native_batch or process_qso" 2>&1 | tail -3
DLAGP_RUNS=3 DLAGP_POLL_IDLE=1 timeout -k 10 600 python3 tools/measure_catalogue.py 2048 10000 32 dr12q 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('catalogue', [round(x) for x in d['spectra_per_s_runs']], round(d['kernel_stream_idle']['share'],3))"
