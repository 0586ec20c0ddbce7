#!/usr/bin/env python3
"""Differential fuzz of the batched CLI (csrc/ssw_cli.c: tests/emu/ssw_test_emu on the SIMT emulator in the build container, ssw_test_gpu itself with --gpu on the GPU
box) against the reference's own ssw_test (oracle/_ref/ssw_test_ref: main.c + ssw.c compiled from where they lie by oracle/Makefile `refcli`): random FASTA / FASTQ inputs (several targets, multi-line records,
lower case, N and other letters, reads of 1..300 residues, DNA and protein), random options (-m -x -o -e -f -c -r -s -h -p).  stdout must be byte-identical, the exit code
equal.  (The run time line "CPU time: ..." goes to stderr in both.)
usage: cli_fuzz.py <seconds> <seed> [--gpu]        -> one JSON line   (--gpu: on the GPU box, ssw_test_gpu itself against the prebuilt oracle/_ref/ssw_test_ref)"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
emu_dir = os.path.join(ROOT, "tests", "emu")
on_gpu = "--gpu" in sys.argv      # GPU box: the product binary ssw_test_gpu against the prebuilt oracle/_ref/ssw_test_ref (oracle/Makefile `refcli`; /root/reference does not exist there)
if on_gpu:
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "ssw_test_ref")
    our_exe = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd", "ssw_test_gpu")
    assert os.path.exists(ref_exe) and os.path.exists(our_exe), "build oracle/_ref/ssw_test_ref (make -C oracle refcli) and ssw_test_gpu first"
else:
    subprocess.run(["make", "-C", emu_dir, "-s", "libssw_emu.so", "ssw_test_emu"], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "refcli"], check=True, stdout=subprocess.DEVNULL)
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "ssw_test_ref")
    our_exe = os.environ.get("SSW_CLI_EXE") or os.path.join(emu_dir, "ssw_test_emu")      # (SSW_CLI_EXE: another build of the emulated CLI, e.g. an AddressSanitizer one)
DNA, AA = "ACGT", "ARNDCQEGHILKMFPSTWYV"


def seq(letters, n, weird):
    s = "".join(letters[i] for i in rng.integers(0, len(letters), size=n))
    if weird and n > 0:
        s = list(s)
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, n))] = str(rng.choice(list("NnXxacgtRYBZ*-")))
        s = "".join(s)
    return s


def finish(path, text, noise):
    """file-format noise kseq.h has to swallow the same way in both programs: CRLF line ends, no newline at the end, blank lines between
    records, gzip (both programs read through zlib)"""
    if noise == "crlf": text = text.replace("\n", "\r\n")
    elif noise == "nonl": text = text.rstrip("\n")
    elif noise == "blank": text = text.replace("\n>", "\n\n>").replace("\n@r", "\n\n@r")
    data = text.encode()
    if noise == "gz":
        import gzip
        data = gzip.compress(data)
    with open(path, "wb") as f: f.write(data)


def write_fasta(path, recs, width, noise=None):
    out = []
    for name, s in recs:
        out.append(">%s some comment\n" % name if rng.random() < 0.8 else ">%s\n" % name)
        if width and len(s) > 0:
            for i in range(0, len(s), width): out.append(s[i:i + width] + "\n")
        else:
            out.append(s + "\n")
    finish(path, "".join(out), noise)


def write_fastq(path, recs, noise=None):
    out = []
    for name, s in recs:
        q = "".join(chr(int(c)) for c in rng.integers(35, 74, size=len(s)))      # (no '@' or '+' at a line start can confuse kseq: 35..73 = '#'..'I')
        if len(s) > 20 and rng.random() < 0.2:      # sequence and quality over several lines
            h = len(s) // 2
            out.append("@%s\n%s\n%s\n+%s\n%s\n%s\n" % (name, s[:h], s[h:], name if rng.random() < 0.5 else "", q[:h], q[h:]))
        else:
            out.append("@%s\n%s\n+\n%s\n" % (name, s, q))
    finish(path, "".join(out), noise)


def write_matrix(path, letters):
    with open(path, "w") as f:
        f.write("# random matrix\n   " + "  ".join(letters) + "\n")
        for i, a in enumerate(letters):
            f.write(a + " " + " ".join("%2d" % (int(rng.integers(1, 9)) if i == j else -int(rng.integers(0, 6))) for j in range(len(letters))) + "\n")


t_end = time.time() + secs
runs = wrong = skipped = 0
first = []
tmp = tempfile.mkdtemp(prefix="cli_fuzz_")
while time.time() < t_end:
    protein = rng.random() < 0.3
    letters = AA if protein else DNA
    # -a FILE: only letters of the file's matrix (lower case included) -- any other letter indexes the reference's matrix / profile out of
    # bounds (src/main.c:342-392 keeps the protein table's codes for letters the file does not name), i.e. there is no reference answer
    use_matrix = rng.random() < 0.15
    mletters = ("ACGT" if rng.random() < 0.3 else "ACGTN") if not protein else AA
    weird = rng.random() < 0.4 and not use_matrix
    if use_matrix: letters = mletters + mletters.lower()
    nt = int(rng.integers(1, 4))
    long_run = on_gpu and rng.random() < 0.15      # GPU box: now and then targets of some kb and reads of several hundred residues (strip kernel, team traceback)
    targets = [("t%d" % i, seq(letters, int(rng.integers(1, 500 if not long_run else 6000)), weird)) for i in range(nt)]
    reads = []
    for i in range(int(rng.integers(1, 8))):
        tname, ts = targets[int(rng.integers(0, nt))]
        L = int(rng.integers(1, (300 if not protein else 200) if not long_run else 1500))
        if rng.random() < 0.7 and len(ts) > L:
            o = int(rng.integers(0, len(ts) - L)); s = list(ts[o:o + L])
            for _ in range(int(rng.integers(0, 1 + L // 10))):
                s[int(rng.integers(0, L))] = letters[int(rng.integers(0, len(letters)))]
            s = "".join(s)
            if not protein and rng.random() < 0.4:
                s = s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
        else:
            s = seq(letters, L, weird)
        reads.append(("r%d" % i, s))
    # file names WITH option letters on purpose: the reference's scanner runs on behind a short option value into the following arguments
    # (src/main.c:253-300) and the CLI reproduces what it then does; the files live in the current directory so that the names are all it sees
    tf, qf = str(rng.choice(["t.fa", "target.fasta", "ref.fa", "TT.NN", "db"])), str(rng.choice(["q.fx", "reads.fq", "query.fastq", "QQ", "chr"]))
    noise = [None, None, None, "crlf", "nonl", "blank", "gz"]
    write_fasta(os.path.join(tmp, tf), targets, int(rng.choice([0, 0, 60, 7])), noise[int(rng.integers(0, len(noise)))])
    if rng.random() < 0.5: write_fastq(os.path.join(tmp, qf), reads, noise[int(rng.integers(0, len(noise)))])
    else: write_fasta(os.path.join(tmp, qf), reads, int(rng.choice([0, 50])), noise[int(rng.integers(0, len(noise)))])
    num = lambda v: ("%02d" % v) if rng.random() < 0.5 else str(v)      # two characters keep the scanner inside the value's string
    opts = []
    if protein: opts.append(["-p"])
    if rng.random() < 0.5: opts.append(["-m", num(int(rng.integers(1, 6)))])
    if rng.random() < 0.5: opts.append(["-x", num(int(rng.integers(1, 7)))])
    if rng.random() < 0.6:
        e = int(rng.integers(1, 4)); opts += [["-o", num(e + int(rng.integers(1, 9)))], ["-e", num(e)]]
    elif rng.random() < 0.15:      # the gapO <= gapE regime (lane-model kernel)
        e = int(rng.integers(1, 5)); opts += [["-o", num(int(rng.integers(1, e + 1)))], ["-e", num(e)]]
    if rng.random() < 0.3: opts.append(["-f", num(int(rng.integers(1, 80)))])
    if rng.random() < 0.6: opts.append(["-c"])
    if rng.random() < 0.4 and not protein: opts.append(["-r"])
    if rng.random() < 0.5:
        opts.append(["-s"])
        if rng.random() < 0.5: opts.append(["-h"])
    if use_matrix:
        write_matrix(os.path.join(tmp, "mat.tbl"), mletters); opts.append(["-a", "mat.tbl"])
    if rng.random() < 0.1: opts.append([str(rng.choice(["-m", "-x", "-f", "-q", "-cs", "-sc", "-csh", "-"]))])      # a value option without a value, unknown letters, combined flags
    order = rng.permutation(len(opts))
    args = [a for k in order for a in opts[int(k)]]
    if rng.random() < 0.15 and args:      # options behind the files
        cut = int(rng.integers(0, len(opts)))
        args = [a for k in order[:cut] for a in opts[int(k)]] + [tf, qf] + [a for k in order[cut:] for a in opts[int(k)]]
    else:
        args = args + [tf, qf]
    runs += 1
    try:
        a = subprocess.run([ref_exe] + args, capture_output=True, timeout=60, cwd=tmp)
    except subprocess.TimeoutExpired:
        skipped += 1; continue      # (the reference can spin on what its scanner made of the arguments)
    if a.returncode < 0: skipped += 1; continue      # the reference itself crashed (its matrix parser, a NULL file): nothing to compare with
    b = subprocess.run([our_exe] + args, capture_output=True, timeout=600, cwd=tmp)
    if b" -a " in b.stderr.split(b"took from them:")[-1] and b"took from them:" in b.stderr:
        skipped += 1; continue      # the overrun made one of the sequence files the MATRIX file: the reference's matrix parser overflows its 4-byte token buffer on it
    if a.returncode != b.returncode or a.stdout != b.stdout:
        wrong += 1
        if len(first) < 4:
            la, lb = a.stdout.decode(errors="replace").splitlines(), b.stdout.decode(errors="replace").splitlines()
            k = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), min(len(la), len(lb)))
            first.append({"args": args, "rc": [a.returncode, b.returncode], "first_differing_line": k, "reference": la[k][:160] if k < len(la) else None, "ours": lb[k][:160] if k < len(lb) else None,
                          "targets": [len(s) for _, s in targets], "reads": [len(s) for _, s in reads], "stderr_ours": b.stderr.decode(errors="replace")[-200:]})
            for fn in (tf, qf) + (("mat.tbl",) if use_matrix else ()): shutil.copy(os.path.join(tmp, fn), os.path.join(tmp, "bad%d_%s" % (wrong, fn)))
print(json.dumps({"fuzz": "ssw_test_gpu (%s) vs the reference's ssw_test: stdout bytes and exit code" % ("on the GPU" if on_gpu else "emulated"), "seconds": secs, "seed": seed, "runs": runs, "reference_crashed_or_hung": skipped, "runs_with_a_difference": wrong, "kept_inputs_in": tmp if wrong else None, "first": first}))
sys.exit(1 if wrong else 0)
