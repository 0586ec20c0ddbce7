"""The oracle against the committed golden vectors (generated from the reference build by
tests/golden/make_golden.py).  CPU only; works without /root/reference."""
import json
import os

import numpy as np
import pytest

from sswutil import RES_FIELDS, dna_matrix, oracle_align

HERE = os.path.dirname(os.path.abspath(__file__))


def load_small():
    with open(os.path.join(HERE, "golden", "golden_small.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("model", [0, 1])
def test_oracle_reproduces_golden_small(model):
    cases = load_small()
    assert len(cases) >= 250
    for c in cases:
        d, cig = oracle_align(np.array(c["read"], dtype=np.int8), np.array(c["mat"], dtype=np.int8), c["n"],
                              np.array(c["ref"], dtype=np.int8), c["gapO"], c["gapE"], c["flag"], c["filters"], c["filterd"],
                              c["maskLen"], c["score_size"], model)
        if c["expect"] is None:
            assert d is None, c["name"]
        else:
            assert {k: d[k] for k in RES_FIELDS} == c["expect"], c["name"]
            assert cig == c["cigar"], c["name"]


def test_config1_table():
    """BASELINE config 1 (demo/target.fastq x demo/query.fastq, ssw_test -c): the table of SURVEY section 4."""
    cases = {c["name"]: c for c in load_small()}
    tbl = [("6:163296599", "20:8823533", 16, 12, (2, 9), (38, 45), "8M"),
           ("6:163296599", "5:106802036", 19, 8, (15, 42), (1, 29), "4M2D11M3I11M"),
           ("3:153409880", "20:8823533", 13, 10, (30, 51), (1, 21), "7M1D1M2I1M1D7M1D3M"),
           ("3:153409880", "5:106802036", 12, 10, (1, 10), (45, 54), "10M")]
    from sswutil import cigar_str
    for q, t, s1, s2, tb, qb, cg in tbl:
        hit = [c for k, c in cases.items() if k.startswith("config1:" + q) and (":" + t) in k and k.endswith("flag2")]
        assert len(hit) == 1
        e = hit[0]["expect"]
        assert (e["score1"], e["score2"]) == (s1, s2)
        assert (e["ref_begin1"] + 1, e["ref_end1"] + 1) == tb and (e["read_begin1"] + 1, e["read_end1"] + 1) == qb
        assert cigar_str(hit[0]["cigar"]) == cg


@pytest.mark.parametrize("model", [0, 1])
def test_oracle_reproduces_demo_new_txt_sample(model):
    """demo/new.txt (the reference's golden stdout for demo/1M.fa x 54mer_hap1_1.100.fastq): a sample here,
    all 100 reads on the GPU (tests/test_gpu_parity.py)."""
    z = np.load(os.path.join(HERE, "golden", "chr3_1M.npz"))
    tgt = z["target"]
    mat = dna_matrix(2, 2)
    for i in (0, 13, 57):
        rd = z["reads"][i]
        d, _ = oracle_align(rd, mat, 5, tgt, 3, 1, 0, 0, 0, len(rd) // 2, 2, model)
        assert [d["score1"], d["score2"], d["ref_end1"] + 1, d["read_end1"] + 1] == [int(x) for x in z["expect"][i]]
