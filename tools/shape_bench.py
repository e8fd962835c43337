"""One of bench.py's other_configs shapes on its own (profiling runs: rocprofv3 wraps this command).
usage: python tools/shape_bench.py INDEX [steps]      INDEX into bench.shape_list()"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import waveform_amd as wf

if __name__ == "__main__":
    i = int(sys.argv[1])
    name, cfg, streams, steps, flags, shape = bench.shape_list(wf)[i]
    if len(sys.argv) > 2:
        steps = int(sys.argv[2])
    print(json.dumps(bench.measure_shape(wf, name, cfg, streams, steps, 8, 0, flags, shape)), flush=True)
