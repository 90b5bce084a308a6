"""Generates tests/golden/*.json: small seeded texts plus, for a query set, the count and the sorted
occurrence positions computed by PURE-PYTHON brute force on the raw text (no oracle, no product code).
This is the definition the reference's own integration test pins (src/fm_index.rs:612-664:
count == number of occurrences in the text, sorted locate == occurrence positions), with both sides
passed through the alphabet's ascii->index map (src/alphabet.rs:169-248).  The reference itself cannot
run here (Rust), so these vectors are produced by that definition, not by the reference binary.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def norm_table(alphabet):
    t = bytearray(256)
    for c in range(256):
        ch = chr(c).upper()
        if alphabet == 0:
            t[c] = ord({"A": "A", "C": "C", "G": "G", "T": "T", "U": "T", "$": "$", "#": "$"}.get(ch, "N"))
        else:
            t[c] = ord(ch) if ch in "ACDEFGHIKLMNPQRSTVWY" else ord("$" if ch in "$#" else "X")
    return bytes(t)


def occurrences(text_n, q_n):
    out, i = [], text_n.find(q_n)
    while i >= 0:
        out.append(i)
        i = text_n.find(q_n, i + 1)
    return out


def make(name, alphabet, n, seed, qlen, n_records=1, n_frac=0.0, extra_queries=()):
    text, starts, headers = synth.make_text(n, alphabet, seed, n_records, n_frac)
    tb = bytes(text)
    tab = norm_table(alphabet)
    tn = tb.translate(tab)
    rng = np.random.default_rng(seed + 1000)
    queries = set()
    for p in range(0, n - qlen, max(1, (n - qlen) // 150)):  # present k-mers (reference: every k-mer)
        queries.add(tb[p:p + qlen].decode())
    for q in synth.random_queries(60, qlen, alphabet, seed + 7):  # mostly absent
        queries.add(bytes(q).decode())
    for L in (1, 2, 3, 5, 9, 10, 11, 13):  # short queries: path A / path B boundary (kmer_len 10 / 4)
        for _ in range(6):
            p = int(rng.integers(0, n - L))
            queries.add(tb[p:p + L].decode())
    queries.update(extra_queries)
    recs = []
    for q in sorted(queries):
        occ = occurrences(tn, q.encode().translate(tab))
        recs.append({"q": q, "count": len(occ), "pos": occ})
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump({"alphabet": alphabet, "text": tb.decode(), "seq_starts": starts, "headers": headers,
                   "queries": recs}, f, separators=(",", ":"))
    print(name, "text", len(tb), "queries", len(recs))


if __name__ == "__main__":
    # sizes follow the reference's own integration tests: 1847 nt / 24-mers (src/fm_index.rs:666-700),
    # 300 aa / 8-mers (:702-743), 24 short records (:994-1032)
    make("nt_1847", 0, 1847, 0, 24, extra_queries=("acgt", "ACGU", "N", "NN", "ACGTR", "a", "T", "GATTACA"))
    make("aa_300", 1, 300, 999, 8, extra_queries=("mk", "X", "B", "acd", "W", "Y"))
    make("nt_multi", 0, 900, 5, 12, n_records=24, n_frac=0.05, extra_queries=("N", "NN", "NNN", "AN", "NA", "ANA", "n"))
    make("aa_multi", 1, 700, 6, 6, n_records=9, extra_queries=("X", "AX", "XA", "x"))
