cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -12 ) > gpurun_out/exp.txt 2>&1
python tools/quick_bench.py 65536:256 48000:256 16400:256 32768:512 >> gpurun_out/exp.txt 2>&1
