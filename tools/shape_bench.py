"""One of bench.py's other_configs shapes on its own (profiling runs: rocprofv3 wraps this command).
usage: python tools/shape_bench.py SHAPE [steps]      SHAPE: the profile key of bench.shape_list() (cfg4_n16384_bars, cfg5shape_8192streams_barsonly, ...) or an index into it"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import waveform_amd as wf

if __name__ == "__main__":
    shapes = bench.shape_list(wf)
    i = int(sys.argv[1]) if sys.argv[1].isdigit() else [s[5] for s in shapes].index(sys.argv[1])
    name, cfg, streams, steps, flags, shape = shapes[i]
    if len(sys.argv) > 2:
        steps = int(sys.argv[2])
    print(json.dumps(bench.measure_shape(wf, name, cfg, streams, steps, 8, 0, flags, shape)), flush=True)
