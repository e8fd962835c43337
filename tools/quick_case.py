"""named shapes for tools/quick_bench.run (profiling runs wrap this command)
usage: python tools/quick_case.py plugin_defaults | lanczos_curve"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import quick_bench as q

CASES = {
    # the plugin's default configuration (src/source.cpp:119-174): mono mixdown of two channels, 800-point Catmull-Rom curve
    "plugin_defaults": dict(n=4096, streams=4096, stereo=0, curve=1, interp_mode=2),
    "lanczos_curve": dict(n=4096, streams=4096, stereo=1, curve=1, interp_mode=1),
}
if __name__ == "__main__":
    c = dict(CASES[sys.argv[1]])
    q.run(c.pop("n"), c.pop("streams"), **c)
