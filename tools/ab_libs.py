"""Interleaved A/B of library builds (variants/lib_*.so, tools/variant.sh) on one shape: every build in a process of its own,
round-robin, `reps` rounds.  usage: python tools/ab_libs.py SHAPE reps lib_a.so lib_b.so[@ENV=VALUE...] ...   (SHAPE: an index into
tools/ab_bars.py's SHAPES or N:streams[:bars[:flags]])"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import waveform_amd as wf
from tools import ab_bars
spec = sys.argv[1]
if ":" in spec:
    p = spec.split(":")
    kw = dict(fft_size=int(p[0]), stereo=1, slope=1.0)
    if len(p) > 2 and int(p[2]):
        kw.update(bars=1, interp_mode=int(p[2]))
    name, streams, flags = spec, int(p[1]), int(p[3]) if len(p) > 3 else 0
else:
    name, kw, streams, flags = ab_bars.SHAPES[int(spec)]
ms, byt = ab_bars.measure(kw, streams, flags)
print(json.dumps([name, round(byt / ms / 1e6 / 8000, 4)]))
''' % ROOT

if __name__ == "__main__":
    shape, reps, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    res = {os.path.basename(l): [] for l in libs}
    name = shape
    for _ in range(reps):
        for l in libs:
            path, *sets = l.split("@")  # lib.so@WF_HIP_LANES=1@...: environment of that build's runs
            env = dict(os.environ, WF_HIP_LIB=os.path.abspath(path))
            env.update(dict(kv.split("=", 1) for kv in sets))
            r = subprocess.run([sys.executable, "-c", CHILD, shape], capture_output=True, text=True, env=env)
            if r.returncode != 0:
                res[os.path.basename(l)].append("error: " + r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "error")
                continue
            name, frac = json.loads(r.stdout.strip().splitlines()[-1])
            res[os.path.basename(l)].append(frac)
    print(json.dumps({"shape": name, "frac_of_8TBps": res}), flush=True)
