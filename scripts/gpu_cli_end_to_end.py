#!/usr/bin/env python3
"""GPU box: ssw_test_gpu END TO END (process start -> stdout flushed to a file) on config 2's own reads: 100 000 and 1 000 000 reads of 150 bp
vs the 1 Mb target, score only and with -c; next to the batch-ABI rate of the same work (bench.py's line).  Checks:
  * 100 000 reads, score only: the stdout is rebuilt in Python from the unmodified reference's records (tests/golden/full/config2_block0.npz)
    and must be byte-identical;
  * 1 000 000 reads (read blocks 0..9 of the config-2 generator): the blocks 0..7 samples of config2_blocks_sample.npz (2 000 reads each) are
    looked up in the output;
  * -c: the 10 000-read byte-for-byte comparison with the reference's own main.c stays scripts/gpu_dropin_cli.py.
usage: gpu_cli_end_to_end.py [--reads 100000,1000000] [--trace]      -> one JSON object on stdout (profiles/round5_cli_end_to_end.json)"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import workloads as W   # noqa: E402

sizes = [100_000, 1_000_000]
if "--reads" in sys.argv:
    sizes = [int(x) for x in sys.argv[sys.argv.index("--reads") + 1].split(",")]
trace = "--trace" in sys.argv
work = "/tmp/cli_e2e"
os.makedirs(work, exist_ok=True)
exe = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd", "ssw_test_gpu")
LUT = np.frombuffer(b"ACGTN", dtype=np.uint8)
FULL = os.path.join(ROOT, "tests", "golden", "full")


def write_inputs(nreads):
    ref, _, p = W.dna_config(2, 0, reads=1)
    with open(os.path.join(work, "ref.fa"), "wb") as f:
        f.write(b">ref\n" + LUT[ref].tobytes() + b"\n")
    path = os.path.join(work, "reads_%d.fq" % nreads)
    with open(path, "wb") as f:
        k = 0
        for b in range(-(-nreads // 100_000)):
            _, reads, _ = W.dna_config(2, b)
            for r in reads[:nreads - k]:
                f.write(b"@r%d\n" % k + LUT[r].tobytes() + b"\n+\n" + b"I" * len(r) + b"\n")
                k += 1
    return path


def expected_score_only(fields, first=0):
    """ssw_write's BLAST-like record without -c (reference src/main.c:126-137) from reference records"""
    out = []
    for i, f in enumerate(fields):
        s1, s2, rb, re, qb, qe = int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5])
        if s1 <= 0:
            continue
        line = "target_name: ref\nquery_name: r%d\noptimal_alignment_score: %d\t" % (first + i, s1)
        if s2 > 0:
            line += "suboptimal_alignment_score: %d\t" % s2
        line += "strand: +\ttarget_end: %d\tquery_end: %d\n\n" % (re + 1, qe + 1)
        out.append(line)
    return out


res = {"binary": "ssw_test_gpu (three stages: parse | device | format + write; stdout to a file)", "protocol": "best of 2 runs, 3 s pause before each run",
       "batch_abi_rate_gcups": "bench.py's default line of the same work: see profiles/round5_bench_default*.json", "runs": []}
for n in sizes:
    fq = write_inputs(n)
    cells = float(n) * 150 * 1e6
    for opts in ([], ["-c"]):
        outp = os.path.join(work, "out_%d%s.txt" % (n, "_c" if opts else ""))
        best = None
        for rep in range(2):
            time.sleep(3.0)      # (the driver scrubs what the previous process released; a run that allocates right behind it waits for that)
            env = dict(os.environ)
            if trace and rep == 1:
                env["SSW_CLI_TRACE"] = "1"
            t0 = time.perf_counter()
            with open(outp, "wb") as fo:
                r = subprocess.run([exe] + opts + [os.path.join(work, "ref.fa"), fq], stdout=fo, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        run = {"reads": n, "options": " ".join(opts) or "(scores only)", "rc": r.returncode, "wall_s": round(best, 3), "gcups_end_to_end": round(cells / best / 1e9, 1),
               "stdout_bytes": os.path.getsize(outp), "input_bytes": os.path.getsize(fq)}
        if trace:
            run["trace"] = [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("[ssw_test_gpu")][:60]
        if not opts:
            got = open(outp, "rb").read()
            if n == 100_000:
                z = np.load(os.path.join(FULL, "config2_block0.npz"))
                exp = "".join(expected_score_only(z["fields"])).encode()
                run["stdout_equals_reference_records"] = got == exp
                run["stdout_md5"] = hashlib.md5(got).hexdigest(); run["expected_md5"] = hashlib.md5(exp).hexdigest()
            else:
                zs = np.load(os.path.join(FULL, "config2_blocks_sample.npz"))
                recs = got.decode().split("\n\n")
                byname = {}
                for rec in recs:
                    if rec.startswith("target_name"):
                        byname[rec.split("\n")[1]] = rec + "\n\n"
                bad = tot = 0
                for b in range(min(len(zs["idx"]), n // 100_000)):
                    for i, f in zip(zs["idx"][b], zs["fields"][b]):
                        e = expected_score_only([f], first=b * 100_000 + int(i))
                        tot += 1
                        if e and byname.get("query_name: r%d" % (b * 100_000 + int(i))) != e[0]:
                            bad += 1
                run["sampled_records_checked"] = tot; run["sampled_records_wrong"] = bad
        res["runs"].append(run)
print(json.dumps(res))
