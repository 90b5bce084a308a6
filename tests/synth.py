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


# ------------------------------------------------------------------------------------------------ repeat-rich genome
# A GRCh38-shaped text: the composition a real mammalian genome confronts an FM-index with and an i.i.d. text does not --
# interspersed repeat families (300 bp .. 6 kb units, 10^3 .. 10^6 copies at full scale, 2 .. 15 % divergence from the
# family consensus), satellite and simple tandem arrays, segmental duplications, assembly gaps (runs of N) and 25 records.
# Fractions are of the text length, so the same recipe scales from test sizes (a few Mbp) to 3.1 Gbp.
#   (name, unit length, fraction of the text, per-base divergence from the consensus, truncated 5' ends)
REPEAT_FAMILIES = (
    ("alu_old", 300, 0.100, 0.12, False),     # 1.0 M copies at 3.1 Gbp
    ("alu_young", 300, 0.019, 0.04, False),   # 0.2 M
    ("l1", 6000, 0.180, 0.10, True),          # 0.18 M copies of 300 .. 6000 bp (the 3' end of the unit)
    ("ltr", 1000, 0.085, 0.15, False),        # 0.26 M
    ("dna", 2000, 0.032, 0.06, False),        # 50 k
    ("l1_young", 6000, 0.010, 0.02, False),   # 5 k full-length copies at 2 %
    ("mir", 260, 0.035, 0.15, False),         # 0.4 M
)
SATELLITE_FRACTION, SATELLITE_ARRAYS, SATELLITE_UNIT = 0.020, 120, 171  # per-copy divergence 2 %, array-specific unit variants 5 %
# exact tandem arrays, each with a unit of its own: (unit lengths, array lengths, fraction) -- microsatellites and minisatellites
SIMPLE_TANDEM = (((2, 6), (20, 100), 0.003), ((20, 60), (500, 5000), 0.0005))  # (microsatellites stay below read length, as in real genomes)
SEGDUP_FRACTION, SEGDUP_MIN, SEGDUP_MAX = 0.048, 10_000, 200_000        # copies of existing stretches at 1 .. 5 % divergence
GAP_FRACTION_BIG, GAP_BIG, GAP_FRACTION_SMALL, GAP_SMALL = 0.043, 24, 0.007, 500


def repeat_rich_text(n, seed=11, n_records=25, device="cpu", scale=1.0):
    """-> (text uint8[n+1] numpy incl. '$', seq_starts, headers, info dict).  Generated with torch on `device` (a 3.1 Gbp
    text takes ~2 s on the GPU, minutes in numpy).  The text depends on (n, seed, device type): CPU and GPU generators draw
    different streams.  `scale` multiplies every repeat fraction (1.0: ~43 % of the text in repeats, counted after overlaps)."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)

    def rnd_letters(shape):
        return torch.randint(0, 4, shape, dtype=torch.uint8, device=dev, generator=gen)

    code = rnd_letters((n,))                      # 0..3, mapped to ACGT at the end; 4 = N
    covered = torch.zeros(n, dtype=torch.bool, device=dev)
    info = {"families": {}}

    def mutate(x, div):
        m = torch.rand(x.shape, device=dev, generator=gen) < div
        return torch.where(m, rnd_letters(x.shape), x)   # a quarter of the "mutations" redraw the same letter

    def spaced_starts(copies, span, hi):
        """`copies` sorted start positions in [0, hi) at least `span` apart (copies of one family never overlap, so the
        result does not depend on the order in which a scatter writes them)"""
        if copies <= 0 or hi <= span * copies:
            copies = max(0, min(copies, int(hi // (2 * span))))
        if copies == 0:
            return torch.zeros(0, dtype=torch.int64, device=dev)
        p = torch.sort(torch.randint(0, int(hi - span * copies), (copies,), device=dev, generator=gen)).values
        return p + torch.arange(copies, device=dev) * span

    chunk_elems = 48_000_000
    for name, unit, frac, div, trunc in REPEAT_FAMILIES:
        frac *= scale
        mean_len = (unit + 300) / 2 if trunc else unit
        copies = int(frac * n / mean_len)
        cons = rnd_letters((unit,))
        starts = spaced_starts(copies, unit, n - unit - 1)
        per = max(1, chunk_elems // unit)
        ar = torch.arange(unit, device=dev)
        for a in range(0, len(starts), per):
            st = starts[a:a + per]
            vals = mutate(cons[None, :].expand(len(st), unit), div)
            idx = st[:, None] + ar[None, :]
            if trunc:  # the copy keeps the last `len` letters of the unit
                ln = torch.randint(min(300, unit), unit + 1, (len(st),), device=dev, generator=gen)
                keep = ar[None, :] >= (unit - ln)[:, None]
                idx, vals = idx[keep], vals[keep]
            code[idx.reshape(-1)] = vals.reshape(-1)
            covered[idx.reshape(-1)] = True
            del vals, idx
        info["families"][name] = {"unit": unit, "copies": int(len(starts)), "divergence": div}
    # satellite arrays: tandem copies of an array-specific variant of one of three base units
    bases = [rnd_letters((SATELLITE_UNIT,)) for _ in range(3)]
    per_array = int(SATELLITE_FRACTION * scale * n / SATELLITE_ARRAYS / SATELLITE_UNIT)
    if per_array >= 2:
        a_starts = spaced_starts(SATELLITE_ARRAYS, per_array * SATELLITE_UNIT, n - per_array * SATELLITE_UNIT - 1)
        for j, s in enumerate(a_starts.tolist()):
            unit_a = mutate(bases[j % 3], 0.05)
            arr = mutate(unit_a.repeat(per_array), 0.02)
            code[s:s + arr.numel()] = arr
            covered[s:s + arr.numel()] = True
        info["satellite"] = {"arrays": int(len(a_starts)), "copies_per_array": per_array, "unit": SATELLITE_UNIT}
    for (u_lo, u_hi), (a_lo, a_hi), frac in SIMPLE_TANDEM:
        a_hi = max(a_lo // 10 + 1, min(a_hi, n // 1000))
        a_lo = min(a_lo, a_hi)
        narr = int(frac * scale * n / ((a_lo + a_hi) / 2))
        a_starts = spaced_starts(narr, a_hi, n - a_hi - 1) if narr >= 1 else []
        if len(a_starts) == 0:
            continue
        na = len(a_starts)
        units = torch.randint(u_lo, u_hi + 1, (na,), device=dev, generator=gen)
        lens = torch.randint(a_lo, a_hi + 1, (na,), device=dev, generator=gen)
        pool = rnd_letters((na * u_hi,))                     # array j repeats pool[j * u_hi : j * u_hi + units[j]]
        arr_id = torch.repeat_interleave(torch.arange(na, device=dev), lens)
        within = torch.arange(int(lens.sum().item()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
        idx = a_starts[arr_id] + within
        code[idx] = pool[arr_id * u_hi + within % units[arr_id]]
        covered[idx] = True
        info.setdefault("tandem_arrays", []).append({"arrays": na, "unit": [u_lo, u_hi], "length": [a_lo, a_hi]})
        del arr_id, within, idx
    # segmental duplications: stretches of the text as it now stands (repeats included), copied elsewhere at 1 .. 5 %
    seg_hi = max(SEGDUP_MIN // 10, min(SEGDUP_MAX, n // 100))
    seg_lo = max(100, min(SEGDUP_MIN, seg_hi // 2))
    ndup = int(SEGDUP_FRACTION * scale * n / ((seg_lo + seg_hi) / 2))
    if ndup:
        lens = torch.randint(seg_lo, seg_hi + 1, (ndup,), device=dev, generator=gen).tolist()
        srcs = torch.randint(0, n - seg_hi - 1, (ndup,), device=dev, generator=gen).tolist()
        dsts = spaced_starts(ndup, seg_hi, n - seg_hi - 1).tolist()
        divs = (0.01 + 0.04 * torch.rand(ndup, device=dev, generator=gen)).tolist()
        for ln, s, d, dv in zip(lens, srcs, dsts, divs):
            code[d:d + ln] = mutate(code[s:s + ln].clone(), dv)
            covered[d:d + ln] = True
        info["segdups"] = {"count": ndup, "len": [seg_lo, seg_hi]}
    info["repeat_fraction"] = float(covered.float().mean().item())
    del covered
    # assembly gaps: a few megabase runs and many small ones
    big = int(GAP_FRACTION_BIG * n / GAP_BIG)
    if big >= 1:
        for s in spaced_starts(GAP_BIG, big, n - big - 1).tolist():
            code[s:s + big] = 4
    nsmall = min(GAP_SMALL, n // 20_000)
    if nsmall:
        small = max(1, int(GAP_FRACTION_SMALL * n / nsmall))
        for s in spaced_starts(nsmall, small, n - small - 1).tolist():
            code[s:s + small] = 4
    starts = [0]
    if n_records > 1:
        cuts = torch.sort(torch.randperm(n - 2, device=dev, generator=gen)[:n_records - 1] + 1).values.tolist() if n < (1 << 24) else \
            sorted(set(torch.randint(1, n - 1, (n_records - 1,), device=dev, generator=gen).tolist()))
        for c in cuts:
            code[c] = 4  # the delimiter byte between records
            starts.append(int(c) + 1)
    info["n_fraction"] = float((code == 4).float().mean().item())
    lut5 = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    body = lut5[code.long()] if n < (1 << 28) else torch.cat([lut5[code[a:a + (1 << 28)].long()] for a in range(0, n, 1 << 28)])
    del code
    text = np.empty(n + 1, dtype=np.uint8)
    text[:n] = body.cpu().numpy()
    text[n] = ord("$")
    return text, starts, ["seq%d" % i for i in range(len(starts))], info
