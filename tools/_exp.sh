cd $GRAFT_REPO_ROOT
python -m pytest tests/test_golden.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/exp.txt
for rep in 1 2; do for B in 0 1; do WF_HIP_CURVE_BOTH=$B python tools/quick_case.py plugin_defaults 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('both=$B', d['ms'], d['frac'])"; done; done >> gpurun_out/exp.txt
