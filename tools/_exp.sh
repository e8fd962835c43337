cd $GRAFT_REPO_ROOT
q() { WF_HIP_LIB=variants/lib_$1.so python tools/quick_bench.py $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms'], d['frac'])"; }
for rep in 1 2; do
q g2048 2048:8192; q g2048ps3 2048:8192
q g4096 4096:4096; q g4096ps3 4096:4096
q g8192 8192:2048; q g8192ps3 8192:2048
q g16384 16384:1024; q g16384ps3 16384:1024
done > gpurun_out/exp.txt 2>&1
