"""Seeded synthetic texts and query sets shared by tests, smoke() and bench.py.

Texts follow the reference's text model (src/fm_index.rs:148-153,220-223): records joined by one
delimiter byte ('N' nucleotide / 'X' amino) and terminated by a single '$'; upper-case canonical
letters only (the input contract under which the reference's behaviour is pinned).
"""
import numpy as np

NT = np.frombuffer(b"ACGT", dtype=np.uint8)
AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
# Swiss-Prot background frequencies (approx., release-notes order A C D E F G H I K L M N P Q R S T V W Y)
AA_FREQ = np.array([8.25, 1.38, 5.46, 6.72, 3.86, 7.07, 2.27, 5.91, 5.80, 9.65, 2.41, 4.06, 4.74, 3.93,
                    5.53, 6.65, 5.36, 6.86, 1.10, 2.92])
AA_FREQ = AA_FREQ / AA_FREQ.sum()


def make_text(n, alphabet=0, seed=0, n_records=1, n_frac=0.0, n_runs=3):
    """-> (text uint8[n+1] incl. '$', seq_starts list, headers list).  `n` counts everything but '$'."""
    rng = np.random.default_rng(seed)
    if alphabet == 0:
        body = NT[rng.integers(0, 4, size=n, dtype=np.uint8)]
        amb = ord("N")
    else:
        body = AA[rng.choice(20, size=n, p=AA_FREQ).astype(np.uint8)]
        amb = ord("X")
    if n_frac > 0 and n > 100:
        tot = int(n * n_frac)
        for r in range(n_runs):
            ln = max(1, tot // n_runs)
            st = int(rng.integers(0, max(1, n - ln)))
            body[st:st + ln] = amb
    starts = [0]
    if n_records > 1:
        cuts = np.sort(rng.choice(np.arange(1, n - 1), size=n_records - 1, replace=False))
        for c in cuts:
            body[c] = amb  # the delimiter byte between records
            starts.append(int(c) + 1)
    text = np.empty(n + 1, dtype=np.uint8)
    text[:n] = body
    text[n] = ord("$")
    return text, starts, ["seq%d" % i for i in range(len(starts))]


def genome_like_text(n, seed=7):
    """a text shaped like an assembled chromosome rather than i.i.d. letters: megabase runs of N (centromere / telomere
    gaps), a 171-bp satellite array with 2 % divergence per copy, an exact tandem array of a 37-bp unit and four exact
    segmental duplications -- the inputs on which prefix doubling needs many rounds and ranges stay wide.
    Returns (text with a trailing '$', dict of the region starts / sizes)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    text = acgt[rng.integers(0, 4, n, dtype=np.uint8)].copy()
    gap, tel = n // 14, min(10_000, n // 300)
    text[n // 2: n // 2 + gap] = ord("N")
    text[:tel] = ord("N")
    text[n - 1 - tel: n - 1] = ord("N")
    sat0, copies = n // 2 + gap, min(20_000, (n // 8) // 171)
    arr = np.tile(text[n // 250: n // 250 + 171], copies)
    mut = rng.random(arr.size) < 0.02
    arr[mut] = acgt[rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)]
    text[sat0: sat0 + arr.size] = arr
    ex0, ecopies = sat0 + arr.size, min(50_000, (n // 16) // 37)
    text[ex0: ex0 + 37 * ecopies] = np.tile(text[n // 125: n // 125 + 37], ecopies)
    seg = min(200_000, n // 100)
    for k in range(4):
        src, dst = n // 80 + k * 3 * seg, n // 4 + k * 2 * seg
        text[dst: dst + seg] = text[src: src + seg]
    text[n - 1] = ord("$")
    return text, dict(gap=gap, tel=tel, sat0=sat0, copies=copies, ex0=ex0, ecopies=ecopies, seg=seg, dup0=n // 4)


def random_queries(nq, qlen, alphabet=0, seed=1):
    """uniform-random fixed-length queries -> uint8[nq, qlen]"""
    rng = np.random.default_rng(seed)
    if alphabet == 0:
        return NT[rng.integers(0, 4, size=(nq, qlen), dtype=np.uint8)]
    return AA[rng.choice(20, size=(nq, qlen), p=AA_FREQ).astype(np.uint8)]


def sampled_queries(text, nq, qlen, seed=2, skip_amb=True, alphabet=0):
    """fixed-length queries drawn from the text at uniform positions (present => all L-1 steps)"""
    rng = np.random.default_rng(seed)
    n = len(text) - 1
    amb = ord("N") if alphabet == 0 else ord("X")
    out = np.empty((nq, qlen), dtype=np.uint8)
    filled = 0
    while filled < nq:
        pos = rng.integers(0, n - qlen, size=(nq - filled) * 2 + 16)
        win = text[pos[:, None] + np.arange(qlen)[None, :]]
        if skip_amb:
            win = win[~(win == amb).any(axis=1)]
        k = min(len(win), nq - filled)
        out[filled:filled + k] = win[:k]
        filled += k
    return out


def fixed_to_csr(q2d):
    """uint8[nq, L] -> (bytes uint8[nq*L], offsets uint64[nq+1])"""
    nq, L = q2d.shape
    return np.ascontiguousarray(q2d).reshape(-1), (np.arange(nq + 1, dtype=np.uint64) * np.uint64(L))


def write_fasta(path, text, starts, headers, width=80):
    n = len(text) - 1
    ends = [s - 1 for s in starts[1:]] + [n]
    with open(path, "wb") as f:
        for s, e, h in zip(starts, ends, headers):
            f.write(b">" + h.encode() + b"\n")
            rec = bytes(text[s:e])
            for i in range(0, len(rec), width):
                f.write(rec[i:i + width] + b"\n")
