cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_fuzz.py -m gpu -x -q -n 4 -k "any or random_size or huge" 2>&1 | tail -6 > gpurun_out/exp.txt
for V in variants/lib_before_packed.so waveform_amd/libwaveform_hip.so; do WF_HIP_LIB=$V python tools/quick_bench.py 800:8192 1600:4096 4160:2048 8000:1024 10912:1024 2>/dev/null | cut -c1-30,60-110,150-250; done >> gpurun_out/exp.txt
