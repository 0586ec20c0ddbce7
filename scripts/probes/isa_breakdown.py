#!/usr/bin/env python3
"""Build container: instruction classes of a kernel's hot loop from the compiler's own assembly (hipcc -S --cuda-device-only of csrc/ssw_kernels.hip).
usage: isa_breakdown.py <mangled-kernel-substring> [<loop-label>]       e.g. isa_breakdown.py k_filldbILi20ELi16ELb1E
Prints, for the innermost loop with the most instructions (or the given label), how many instructions of each class one trip issues; the
numbers in profiles/round5_filldb_breakdown.txt come from here."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
asm = "/tmp/ssw_kernels_gfx950.s"
src = os.path.join(ROOT, "complete-striped-smith-waterman-library_amd", "csrc", "ssw_kernels.hip")
if not os.path.exists(asm) or os.path.getmtime(asm) < os.path.getmtime(src):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src),
                    "--cuda-device-only", "-S", src, "-o", asm], check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
key = sys.argv[1]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
fn = lines[start:end]
meta = [l.strip() for l in lines[end:end + 60] if re.search(r"; (NumVgprs|Occupancy|ScratchSize|codeLenInByte)", l)]
print(lines[start].split(":")[0], " ".join(meta))
labels = {m.group(1): i for i, l in enumerate(fn) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(fn):
    m = re.search(r"s_(?:c)?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
if len(sys.argv) > 2:
    loops = [x for x in loops if x[2] == sys.argv[2]]
else:      # innermost = contains no other loop; take the longest of those
    inner = [x for x in loops if not any(y != x and y[0] >= x[0] and y[1] <= x[1] for y in loops)]
    loops = [max(inner, key=lambda x: x[1] - x[0])]
s0, e0, lab = loops[0]


def klass(l):
    op = l.split()[0]
    if op.startswith("v_pk_maximum3"): return "VALU: v_pk_maximum3_f16 (recurrence maxima)"
    if op.startswith("v_pk_"): return "VALU: other packed 16-bit (" + op.rsplit("_e", 1)[0] + ")"
    if "dpp" in l: return "VALU: DPP (" + op + ")"
    if op.startswith("v_bfi"): return "VALU: v_bfi_b32 (best-cell snapshots / selects)"
    if op.startswith("v_add") or op.startswith("v_sub"): return "VALU: 32-bit add / sub"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"): return "VALU: compare / select"
    if op.startswith("v_max") or op.startswith("v_min"): return "VALU: 32-bit max / min"
    if op.startswith("v_mov"): return "VALU: v_mov"
    if op.startswith("v_"): return "VALU: other (" + op.rsplit("_e", 1)[0] + ")"
    if op.startswith("ds_"): return "LDS: " + op
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "VMEM: " + op
    if op.startswith("s_nop"): return "scalar: s_nop"
    if op.startswith("s_waitcnt"): return "scalar: s_waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "scalar: branch"
    return "scalar: other"


c = collections.Counter()
for l in fn[s0:e0 + 1]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    c[klass(l)] += 1
tot = sum(c.values()); valu = sum(v for k, v in c.items() if k.startswith("VALU"))
print("loop %s: %d instructions per trip (all paths of the body), %d of them VALU" % (lab, tot, valu))
for k, v in sorted(c.items(), key=lambda x: (-x[1], x[0])):
    print("%6d  %5.1f %%  %s" % (v, 100.0 * v / tot, k))
