cd $GRAFT_REPO_ROOT
q() { WF_HIP_LIB=variants/lib_$1.so python tools/quick_bench.py $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['kernel'][-30:], d['ms'], d['frac'])"; }
for rep in 1 2; do q g1024 1024:16384; q g1024s4 1024:16384; WF_HIP_TLDS=0 q g1024 1024:16384; done > gpurun_out/exp.txt 2>&1
WF_HIP_LIB=variants/lib_g1024s4.so python -m pytest tests/test_golden.py -m gpu -q -k "test_hip_reproduces and 1024" 2>&1 | tail -2 >> gpurun_out/exp.txt
