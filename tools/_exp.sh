cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for V in g1024 g1024w4 g1024t0; do WF_HIP_LIB=variants/lib_$V.so python tools/quick_bench.py 1024:16384 2>/dev/null | cut -c1-30,90-200; done
done > gpurun_out/exp.txt 2>&1
for V in g1024 g1024w4; do WF_HIP_LIB=variants/lib_$V.so python tools/quick_bench.py 512:16384 256:16384 128:16384 2>/dev/null | cut -c1-30,90-200; done >> gpurun_out/exp.txt 2>&1
