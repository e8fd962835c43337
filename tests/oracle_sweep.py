"""extended CPU sweep (development aid): the restatement (oracle/wf_oracle*.c) against the reference itself (oracle/_ref)
on the fuzz scripts of seeds [LO, HI) -- what tests/test_oracle_vs_ref.py does for its committed seeds, in parallel.
usage: python tests/oracle_sweep.py LO HI [pow2,any,huge,meter,wave] [processes]"""
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(job):
    kind, seed = job
    import test_oracle_vs_ref as t
    fn = {"pow2": t.test_restatement_matches_reference_pow2, "any": t.test_restatement_matches_reference_any_size,
          "huge": t.test_restatement_matches_reference_huge_size, "wide": t.test_restatement_matches_reference_wide_ranges, "meter": t.test_restatement_matches_reference_meter,
          "wave": t.test_restatement_matches_reference_waveform}[kind]
    try:
        fn(seed)
        return None
    except BaseException as e:  # pytest.skip included
        return (kind, seed, type(e).__name__, str(e).replace("\n", " ")[-300:])


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["pow2", "any", "huge", "meter", "wave"]
    procs = int(sys.argv[4]) if len(sys.argv) > 4 else (os.cpu_count() or 4)
    jobs = [(k, s) for k in kinds for s in range(lo, hi)]
    with Pool(procs) as p:
        bad = [r for r in p.imap_unordered(one, jobs, chunksize=8) if r]
    for r in bad:
        print("FAIL", *r)
    print("done", lo, hi, kinds, "failures", len(bad), "of", len(jobs))
