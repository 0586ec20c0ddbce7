#!/usr/bin/env python3
"""Generate the committed golden vectors from the UNMODIFIED reference (oracle/_ref/libssw_ref.so,
built by oracle/Makefile from /root/reference/src/ssw.c) and from the reference's demo data.

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py

Outputs (committed; the GPU box has no /root/reference):
  tests/golden/golden_small.json   demo pairs + seeded random/adversarial cases: inputs, parameters and the
                                   reference's s_align fields + CIGAR words
  tests/golden/chr3_1M.npz         the demo target demo/1M.fa (codes, 1 MB -> ~250 KB compressed), the 100 demo
                                   54-mers and the numbers of demo/new.txt (the reference's own golden stdout)
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from sswutil import (RES_FIELDS, blosum50, dna_matrix, encode_aa, encode_dna, mutate, random_ref,  # noqa: E402
                     ref_align)

DEMO = "/root/reference/demo"


def read_fx(path):
    """minimal FASTA/FASTQ reader -> [(name, seq)]"""
    out, name, seq, mode = [], None, [], None
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f]
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith(">"):
            if name is not None:
                out.append((name, "".join(seq)))
            name, seq, mode = l[1:].split()[0] if len(l) > 1 else "", [], "fa"
        elif l.startswith("@") and mode != "fa":
            if name is not None:
                out.append((name, "".join(seq)))
            name, seq = l[1:].split()[0], [lines[i + 1]]
            i += 3
            mode = "fq"
        elif mode == "fa" and l:
            seq.append(l)
        i += 1
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def case(name, kind, read, ref, mat, n, gapO, gapE, flag, filters, filterd, maskLen, score_size=2):
    d, cig = ref_align(read, mat, n, ref, gapO, gapE, flag, filters, filterd, maskLen, score_size)
    return {"name": name, "kind": kind, "read": [int(x) for x in read], "ref": [int(x) for x in ref],
            "mat": [int(x) for x in mat], "n": n, "gapO": gapO, "gapE": gapE, "flag": flag, "filters": filters,
            "filterd": filterd, "maskLen": maskLen, "score_size": score_size,
            "expect": None if d is None else {k: d[k] for k in RES_FIELDS}, "cigar": cig}


def main():
    cases = []
    dna = dna_matrix(2, 2)
    b50 = blosum50()
    # 1. the known-answer pair of reference src/example.c:105-156 (2/-2/3/1, maskLen 15, flag 1)
    cases.append(case("example_c", "dna", encode_dna("CTGAGCCGGTAAATC"),
                      encode_dna("CAGCCTTTCTGACCCGGAAATCAAAATAGGCACAACAAA"), dna_matrix(2, 2), 5, 3, 1, 1, 0, 0, 15))
    # 2. BASELINE config 1: demo/target.fastq x demo/query.fastq, as ssw_test -c does (flag 2, maskLen readLen/2)
    tg = read_fx(os.path.join(DEMO, "target.fastq"))
    qs = read_fx(os.path.join(DEMO, "query.fastq"))
    for qn, q in qs:
        for tn, t in tg:
            for flag in (0, 2):
                cases.append(case("config1:%s:%s:flag%d" % (qn, tn, flag), "dna", encode_dna(q), encode_dna(t), dna, 5, 3, 1,
                                  flag, 0, 0, len(q) // 2))
    # 3. protein demo (ssw_test -p -c): BLOSUM50, 3/1
    p1 = read_fx(os.path.join(DEMO, "protein1.fa"))[0][1]
    p2 = read_fx(os.path.join(DEMO, "protein2.fa"))[0][1]
    cases.append(case("protein1x2", "aa", encode_aa(p2), encode_aa(p1), b50, 24, 3, 1, 2, 0, 0, len(p2) // 2))
    cases.append(case("protein2x1", "aa", encode_aa(p1), encode_aa(p2), b50, 24, 3, 1, 2, 0, 0, len(p1) // 2))
    # 4. pRef / pRead, r1 (a 56-op CIGAR on the + strand), 1k.fa x query.fastq (README sample)
    pr = read_fx(os.path.join(DEMO, "pRef.fa"))[0][1]
    pq = read_fx(os.path.join(DEMO, "pRead.fa"))[0][1]
    cases.append(case("pRef_pRead", "dna", encode_dna(pq), encode_dna(pr), dna, 5, 3, 1, 2, 0, 0, len(pq) // 2))
    r1 = read_fx(os.path.join(DEMO, "r1.fa"))[0][1]
    r1q = read_fx(os.path.join(DEMO, "r1_query.fq"))[0][1]
    cases.append(case("r1_plus", "dna", encode_dna(r1q), encode_dna(r1), dna, 5, 3, 1, 2, 0, 0, len(r1q) // 2))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    r1rc = "".join(comp.get(ch, "N") for ch in reversed(r1q.upper()))
    cases.append(case("r1_minus", "dna", encode_dna(r1rc), encode_dna(r1), dna, 5, 3, 1, 2, 0, 0, len(r1q) // 2))
    k1 = read_fx(os.path.join(DEMO, "1k.fa"))[0][1]
    for qn, q in qs:
        cases.append(case("1k:%s" % qn, "dna", encode_dna(q), encode_dna(k1), dna, 5, 3, 1, 2, 0, 0, len(q) // 2))

    # 5. seeded random / adversarial cases (gapO > gapE: the GPU path's domain), all flags and score sizes
    rng = np.random.default_rng(20250925)
    for it in range(260):
        if it % 4 != 3:
            n, kind, nc = 5, "dna", 4
            mat = dna_matrix(int(rng.integers(1, 4)), int(rng.integers(1, 6)))
            refLen = int(rng.integers(30, 700))
            ref = random_ref(refLen, int(rng.integers(1 << 30)), 4, 0.02)
            if it % 16 == 0:   # low complexity: many ties
                ref = np.tile(rng.integers(0, 4, size=int(rng.integers(1, 5)), dtype=np.int8), refLen)[:refLen]
        else:
            n, kind, nc = 24, "aa", 20
            mat = b50
            refLen = int(rng.integers(30, 400))
            ref = rng.integers(0, 20, size=refLen, dtype=np.int8)
        rl = int(rng.integers(4, 260))
        if rng.random() < 0.75 and refLen > rl + 24:
            off = int(rng.integers(0, refLen - rl - 16))
            read = mutate(ref[off:off + rl + 8], rng, 0.06 if kind == "dna" else 0.2, 0.03, 0.03, nc)[:rl]
            if len(read) < 4:
                read = rng.integers(0, nc, size=rl, dtype=np.int8)
        else:
            read = rng.integers(0, nc, size=rl, dtype=np.int8)
        gapE = int(rng.integers(1, 4))
        gapO = gapE + int(rng.integers(1, 6))
        flag = int(rng.choice([0, 1, 2, 8, 9, 15, 4, 6, 3]))
        filters = int(rng.choice([0, 0, 25, 70]))
        filterd = int(rng.choice([0, 25, 1000]))
        maskLen = int(rng.choice([len(read) // 2, len(read) // 2, 15, 10, 40]))
        ss = int(rng.choice([2, 2, 2, 0, 1]))
        cases.append(case("rand%03d" % it, kind, read, ref, mat, n, gapO, gapE, flag, filters, filterd, maskLen, ss))
    with open(os.path.join(HERE, "golden_small.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "source": "oracle/_ref/libssw_ref.so (reference ssw.c v1.2.6)",
                   "cases": cases}, f, separators=(",", ":"))
    print("golden_small.json:", len(cases), "cases")

    # 6. the reference's own golden stdout: ssw_test demo/1M.fa demo/54mer_hap1_1.100.fastq == demo/new.txt
    t = read_fx(os.path.join(DEMO, "1M.fa"))[0][1]
    reads = read_fx(os.path.join(DEMO, "54mer_hap1_1.100.fastq"))
    txt = open(os.path.join(DEMO, "new.txt")).read()
    rows = re.findall(r"optimal_alignment_score: (\d+)\tsuboptimal_alignment_score: (\d+)\tstrand: (.)\ttarget_end: (\d+)\tquery_end: (\d+)", txt)
    names = re.findall(r"query_name: (\S+)", txt)
    assert len(rows) == len(reads) == 100 and names == [r[0] for r in reads]
    expect = np.array([[int(a), int(b), int(d), int(e)] for a, b, c, d, e in rows], dtype=np.int32)   # 1-based ends as printed
    # cross-check new.txt against the compiled reference through the C API before committing it
    tcodes = encode_dna(t)
    for i in (0, 1, 57, 99):
        d, _ = ref_align(encode_dna(reads[i][1]), dna, 5, tcodes, 3, 1, 0, 0, 0, len(reads[i][1]) // 2)
        assert [d["score1"], d["score2"], d["ref_end1"] + 1, d["read_end1"] + 1] == list(expect[i]), (i, d, expect[i])
    np.savez_compressed(os.path.join(HERE, "chr3_1M.npz"), target=tcodes,
                        reads=np.stack([encode_dna(r[1]) for r in reads]), expect=expect)
    print("chr3_1M.npz: target", len(tcodes), "reads", len(reads))


def cli_goldens():
    """stdout of the reference's own ssw_test (built out-of-tree from /root/reference/src) on BASELINE config 1 and on
    the protein demo: committed so that the batched CLI can be diffed against it on the GPU box."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp()
    exe = os.path.join(tmp, "ssw_test_ref")
    subprocess.run(["gcc", "-O2", "-I/root/reference/src", "/root/reference/src/main.c", "/root/reference/src/ssw.c", "-o", exe, "-lz", "-lm"],
                   check=True, stderr=subprocess.DEVNULL)
    out = os.path.join(HERE, "cli")
    os.makedirs(out, exist_ok=True)
    for f in ("target.fastq", "query.fastq", "protein1.fa", "protein2.fa", "r1.fa", "r1_query.fq", "1k.fa"):
        shutil.copy(os.path.join(DEMO, f), os.path.join(out, f))
    shutil.copy(os.path.join(DEMO, "r1.fa"), os.path.join(out, "ref.fa")); shutil.copy(os.path.join(DEMO, "r1_query.fq"), os.path.join(out, "query_reads.fq"))
    runs = {"config1_c": ["-c", "target.fastq", "query.fastq"], "config1_csh": ["-c", "-s", "-h", "target.fastq", "query.fastq"],
            "config1_plain": ["target.fastq", "query.fastq"], "config1_cr": ["-c", "-r", "target.fastq", "query.fastq"],
            "protein_pc": ["-p", "-c", "protein1.fa", "protein2.fa"], "r1_cr": ["-c", "-r", "r1.fa", "r1_query.fq"],
            "r1_csr": ["-c", "-s", "-r", "r1.fa", "r1_query.fq"], "readme_1k_cs": ["-c", "-s", "1k.fa", "query.fastq"],
            "config1_f15": ["-c", "-f", "15", "target.fastq", "query.fastq"],
            # option values are given with two characters and followed by another option: the reference's hand-written parser
            # (main.c:253-300) keeps scanning the characters after a consumed value, past the end of a one-character string
            "scoring_m1x3o5e2": ["-m", "01", "-x", "03", "-o", "05", "-e", "02", "-c", "target.fastq", "query.fastq"],
            "matrix_file_csh": ["-a", "wt.tbl", "-c", "-s", "-h", "target.fastq", "query.fastq"],   # (a file name without option letters)
            "gap_o10_e10": ["-m", "05", "-x", "04", "-o", "10", "-e", "10", "-c", "target.fastq", "query.fastq"],
            "protein_o11e1": ["-o", "11", "-e", "01", "-p", "-c", "protein1.fa", "protein2.fa"],
            # an empty record between two reads (common after adapter trimming): the reference reports it on stderr and goes on
            "empty_read_c": ["-c", "target.fastq", "empty_read.fq"], "empty_read_cs": ["-c", "-s", "target.fastq", "empty_read.fq"],
            # -r together with a matrix file that has neither 5 nor 24 rows: the reference ignores -r (src/main.c:457-490)
            # -r with a 5-row matrix file: reverse complements are aligned, coded through the FILE's table (src/main.c:457-490);
            # with -p and a matrix of neither 5 nor 24 rows -r is silently ignored
            "matrix_file_r5": ["-a", "wt.tbl", "-c", "-r", "target.fastq", "query.fastq"],
            "matrix_file_p_r_ignored": ["-p", "-a", "wt4.tbl", "-c", "-r", "target.fastq", "query.fastq"],
            # alignments whose traceback fails (s_align.flag 1, no CIGAR; SURVEY 8.0 "Quirk"): SAM output still carries the soft
            # clips that mark_mismatch adds (tbfail.fa / tbfail.fq are fixtures of this repo, found by random search)
            "tbfail_cs": ["-m", "09", "-x", "04", "-c", "-s", "tbfail.fa", "tbfail.fq"], "tbfail_c": ["-m", "09", "-x", "04", "-c", "tbfail.fa", "tbfail.fq"],
            # round 6: what the reference's scanner (main.c:247-330) really does with command lines that are not written for it -- the CLI reproduces it:
            # a one-character value directly in front of the files: the scan runs on into "ref.fa" / "query_reads.fq" (-r, -e ref.fa = gap extension 0, -f query_reads.fq,
            # -r, -s) and ends on the terminator of the LAST argument -- file names chosen so: a scan that leaves the arguments reads the environment, and the
            # output would depend on the caller's first environment string (ref.fa / query_reads.fq are copies of r1.fa / r1_query.fq)
            "scan_overrun_e2": ["-c", "-e", "2", "ref.fa", "query_reads.fq"],
            "scan_combined_cs": ["-cs", "target.fastq", "query.fastq"],                                  # every character of a '-' argument is an option letter
            "scan_m_without_value": ["-m", "-c", "target.fastq", "query.fastq"],                         # a value may not start with '-': -m keeps its default
            "scan_attached_value_ignored": ["-c", "-x5", "-s", "target.fastq", "query.fastq"],           # "-x5": x takes the NEXT argument unless it starts with '-'; here it does
            "scan_options_behind_files": ["target.fastq", "query.fastq", "-f", "15", "-c"],
            "scan_missing_target_csh": ["-c", "-s", "-h", "no_such_file.fa", "query.fastq"]}              # a target file that cannot be opened: header line only, exit code 0
    # fixture of this repo (not a reference file): the first two demo reads with an empty FASTQ record between them
    q = open(os.path.join(DEMO, "query.fastq")).read().split("\n")
    with open(os.path.join(out, "empty_read.fq"), "w") as f:
        f.write("\n".join(q[:4]) + "\n@empty_after_trimming\n\n+\n\n" + "\n".join(q[4:8]) + "\n")
    for name, args in runs.items():
        r = subprocess.run([exe] + args, cwd=out, capture_output=True, text=True)
        with open(os.path.join(out, name + ".stdout"), "w") as f:
            f.write(r.stdout)
        with open(os.path.join(out, name + ".args"), "w") as f:
            f.write(" ".join(args))
    print("cli goldens:", len(runs))
    # the reference's example programs (C API and C++ wrapper), all-reference builds
    ex_c = os.path.join(tmp, "example_c_ref"); ex_cpp = os.path.join(tmp, "example_cpp_ref")
    subprocess.run(["gcc", "-O2", "-I/root/reference/src", "/root/reference/src/example.c", "/root/reference/src/ssw.c", "-o", ex_c, "-lm"],
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["g++", "-O2", "-I/root/reference/src", "/root/reference/src/example.cpp", "/root/reference/src/ssw_cpp.cpp",
                    "-x", "c", "/root/reference/src/ssw.c", "-o", ex_cpp, "-lm"], check=True, stderr=subprocess.DEVNULL)
    for exe, name in ((ex_c, "example_c"), (ex_cpp, "example_cpp")):
        r = subprocess.run([exe], capture_output=True, text=True)
        with open(os.path.join(out, name + ".example_stdout"), "w") as f:
            f.write(r.stdout)
    print("example goldens: 2")


if __name__ == "__main__":
    cli_goldens()
    main()
