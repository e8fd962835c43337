cd $GRAFT_REPO_ROOT
q() { WF_HIP_LIB=variants/lib_$1.so python tools/quick_bench.py $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms'], d['frac'])"; }
for rep in 1 2; do
q g1024nn 1024:16384; q g1024 1024:16384
q g2048nn 2048:8192; q g2048 2048:8192
q g8192nn 8192:2048; q g8192 8192:2048
q g16384nn 16384:1024; q g16384 16384:1024
for V in g16384nn g16384; do WF_HIP_LIB=variants/lib_$V.so python tools/shape_bench.py 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V cfg4', d['ms_per_step'], d['roofline']['frac'])"; done
for V in g4096nn g4096; do WF_HIP_LIB=variants/lib_$V.so python tools/shape_bench.py 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V cfg5', d['ms_per_step'], d['roofline']['frac'])"; done
for V in g4096nn g4096; do WF_HIP_LIB=variants/lib_$V.so python tools/shape_bench.py 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V cfg3x2', d['ms_per_step'], d['roofline']['frac'])"; done
done > gpurun_out/exp.txt 2>&1
