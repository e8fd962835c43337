cd $GRAFT_REPO_ROOT
for V in variants/lib_g4096.so variants/lib_g4096d.so variants/lib_g4096.so variants/lib_g4096d.so; do
WF_HIP_LIB=$V python - 2>&1 <<'PY'
import sys
sys.path.insert(0,'.')
from tools import quick_bench as q
q.run(4096, 4096, stereo=0, curve=1, interp_mode=2)
q.run(4096, 4096, stereo=1, curve=1, interp_mode=1)
q.run(4096, 4096, stereo=1, bars=1, interp_mode=1)
PY
done > gpurun_out/exp.txt
WF_HIP_LIB=variants/lib_g4096d.so python -m pytest tests/test_golden.py -m gpu -x -q -k "4096 or 800" 2>&1 | tail -3 >> gpurun_out/exp.txt
