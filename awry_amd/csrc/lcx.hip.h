// lcx.hip.h -- the left-context index (LCX, layout.h): device helpers for searching it with a wavefront quad, and the
// kernels that build it.  Included by kernels.hip.h (needs quad_sum, Text20).
//
// Why: backward search (/root/reference src/fm_index.rs:402-438) consumes a query right to left, one LF step per letter,
// and every step of a range wider than a block costs two random 128-B lines.  The seed table replaces the first k steps
// by one probe, but a seed k-mer inside a repeat family still has 10^3..10^5 rows, and the copies of a family differ in
// one letter out of 8..50: the range stays wide for dozens of letters (GRCh38-scale text with 43 % repeats: 31-mers drawn
// from the text ran 4.3x, 101-bp reads 8x slower than on an i.i.d. text of the same size).  The LCX keeps every seed
// bucket's suffixes sorted by the 32 letters in FRONT of them, so the letters a query has left of its seed window are
// matched by a 16-ary search -- one line per level, log16(rows) levels -- instead of letter by letter.  The answer is the
// same set of suffixes (a suffix of the bucket extends to the whole query iff the query's left letters precede it), so
// counts are unchanged; the locate pass emits them in ascending BWT row as the reference does (src/fm_index.rs:521).
#pragma once

namespace awry {

// 16 nibbles (4-bit text codes, all < 8) -> 16 two-bit letters in the low 32 bits
__device__ __forceinline__ uint64_t lcx_letters16(uint64_t x) {
  x &= 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  return (x | (x >> 16)) & 0x00000000FFFFFFFFull;
}

// the m <= 32 text symbols text[t0 .. t0 + m) as 2-bit letters (symbol j in bits [2j, 2j + 2)); *ok = all of them are ACGT
__device__ __forceinline__ uint64_t lcx_text_letters(const uint32_t* __restrict__ text4, uint64_t t0, int m, bool* ok) {
  const Text20 t = *reinterpret_cast<const Text20*>(text4 + (t0 >> 3));  // (the text buffer has 8 words of slack)
  const int sh = 4 * (int)(t0 & 7);
  const uint64_t a0 = (uint64_t)t.w[0] | ((uint64_t)t.w[1] << 32), a1 = (uint64_t)t.w[2] | ((uint64_t)t.w[3] << 32), a2 = t.w[4];
  uint64_t lo = sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;  // nibbles 0..15
  uint64_t hi = sh ? (a1 >> sh) | (a2 << (64 - sh)) : a1;  // nibbles 16..31
  if (m < 16) { lo &= (1ull << (4 * m)) - 1; hi = 0; }
  else if (m < 32) hi &= m == 16 ? 0ull : (1ull << (4 * (m - 16))) - 1;
  *ok = ((lo | hi) & 0x8888888888888888ull) == 0;
  return lcx_letters16(lo) | (lcx_letters16(hi) << 32);
}

// ---- search -------------------------------------------------------------------------------------------------------
// Keys of a bucket's complete entries are ascending over rows [sp, sp + nc).  Level t of the index holds the key of every
// row that is a multiple of 16^t (level 0 = lcx_key itself), in aligned nodes of 16 keys = one 128-B line; a node of level
// t spans 16^(t+1) rows.  A search keeps, for each of its two bounds, the half-open row range [a, b) whose keys it has
// not compared yet (bound = a + #keys in [a, b) below the threshold) and consults the levels top-down.  The top level is
// the highest one with a sampled row inside the bucket: all its samples inside the bucket then lie in ONE node (two of
// them in different nodes would have a multiple of 16^(t+1) between them), and every range a level leaves behind lies
// between two neighbouring samples, i.e. inside one node of the level below.  So a search costs one line per level per
// bound, and the bounds share their lines until they part.
struct LcxQ {
  uint32_t a0, b0;  // lower bound: first row whose key is >= qlo
  uint32_t a1, b1;  // upper bound: first row whose key is >  qhi
  int t;            // next level to consult; < 0: done (lb = a0, ub = a1)
};

__device__ __forceinline__ int lcx_top_level(uint32_t sp, uint32_t nc) {  // nc >= 1
  int t = 0;
  while (t < 7) {
    const int s = 4 * (t + 1);
    const uint64_t f = (((uint64_t)sp + (1ull << s) - 1) >> s) << s;
    if (f >= (uint64_t)sp + nc) break;
    t++;
  }
  return t;
}

__device__ __forceinline__ void lcx_begin(LcxQ& q, uint32_t sp, uint32_t nc) {
  q.a0 = q.a1 = sp;
  q.b0 = q.b1 = sp + nc;
  q.t = nc ? lcx_top_level(sp, nc) : -1;
}

// first multiple of 2^s that is >= a
__device__ __forceinline__ uint64_t lcx_first_sample(uint32_t a, int s) { return (((uint64_t)a + (1ull << s) - 1) >> s) << s; }

// One level for both bounds (quad-cooperative: lane l of the quad holds 4 of a node's 16 keys).  Levels at which
// neither range holds a sampled row are passed without a load.  Returns the number of lines it asked for (0, 1 or 2).
struct LcxRefs { uint32_t &a0, &b0, &a1, &b1; int& t; };  // the same state held in the caller's own variables
__device__ __forceinline__ int lcx_quad_step(const DevIndex& ix, LcxRefs q, uint64_t qlo, uint64_t qhi, int l) {
  while (q.t >= 0) {
    const int s = 4 * q.t;
    const uint64_t f0 = lcx_first_sample(q.a0, s), f1 = lcx_first_sample(q.a1, s);
    const bool has0 = f0 < q.b0, has1 = f1 < q.b1;
    if (!has0 && !has1) { q.t--; continue; }
    const uint64_t* __restrict__ lev = q.t == 0 ? ix.lcx_key : ix.lcx_inner + ix.lcx_off[q.t];
    const uint64_t n0 = (f0 >> s) & ~15ull, n1 = (f1 >> s) & ~15ull;  // node = 16 aligned entries
    const bool same = has0 && has1 && n0 == n1;
    ulonglong2 k0a{0, 0}, k0b{0, 0}, k1a{0, 0}, k1b{0, 0};
    if (has0) { const ulonglong2* p = reinterpret_cast<const ulonglong2*>(lev + n0 + 4 * l); k0a = p[0]; k0b = p[1]; }
    if (has1 && !same) { const ulonglong2* p = reinterpret_cast<const ulonglong2*>(lev + n1 + 4 * l); k1a = p[0]; k1b = p[1]; }
    if (same) { k1a = k0a; k1b = k0b; }
    uint32_t c = 0;
    const uint64_t kk0[4] = {k0a.x, k0a.y, k0b.x, k0b.y}, kk1[4] = {k1a.x, k1a.y, k1b.x, k1b.y};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t r0 = (n0 + 4 * l + u) << s, r1 = (n1 + 4 * l + u) << s;
      c += (has0 && r0 >= q.a0 && r0 < q.b0 && kk0[u] < qlo) ? 1u : 0u;
      c += (has1 && r1 >= q.a1 && r1 < q.b1 && kk1[u] <= qhi) ? 0x100u : 0u;
    }
    c = quad_sum(c);
    const uint32_t c0 = c & 0xffu, c1 = c >> 8;
    if (has0) {
      const uint64_t nb = f0 + ((uint64_t)c0 << s);
      q.b0 = nb < q.b0 ? (uint32_t)nb : q.b0;
      if (c0) q.a0 = (uint32_t)(f0 + ((uint64_t)(c0 - 1) << s) + 1);
    }
    if (has1) {
      const uint64_t nb = f1 + ((uint64_t)c1 << s);
      q.b1 = nb < q.b1 ? (uint32_t)nb : q.b1;
      if (c1) q.a1 = (uint32_t)(f1 + ((uint64_t)(c1 - 1) << s) + 1);
    }
    q.t--;
    return same || !(has0 && has1) ? 1 : 2;
  }
  return 0;
}
__device__ __forceinline__ int lcx_quad_step(const DevIndex& ix, LcxQ& q, uint64_t qlo, uint64_t qhi, int l) {
  return lcx_quad_step(ix, LcxRefs{q.a0, q.b0, q.a1, q.b1, q.t}, qlo, qhi, l);
}

// the thresholds of a search for the m = min(i, 32) letters nearest the seed window: ctx = those letters packed like a
// query word (the letter next to the window in bits [2m - 2, 2m))
__device__ __forceinline__ void lcx_thresholds(uint64_t ctx, int m, uint64_t* qlo, uint64_t* qhi) {
  const uint64_t lo = m >= 32 ? ctx : (ctx & ((1ull << (2 * m)) - 1)) << (64 - 2 * m);
  *qlo = lo;
  *qhi = m >= 32 ? lo : lo | ((1ull << (64 - 2 * m)) - 1);
}

// letters [i - m, i) of a packed read (W words, letter j in word j / 32, bits 2 (j % 32)), m = min(i, 32), as a word
__device__ __forceinline__ uint64_t lcx_read_ctx(const uint64_t* __restrict__ qw, int W, int i) {
  const int m = i < 32 ? i : 32, f = i - m, a = f >> 5, sh = 2 * (f & 31);
  uint64_t x = qw[a] >> sh;
  if (sh && a + 1 < W) x |= qw[a + 1] << (64 - sh);
  return m >= 32 ? x : x & ((1ull << (2 * m)) - 1);
}

// ---- construction -------------------------------------------------------------------------------------------------
// Per row of a chunk [r0, r0 + n): does its suffix belong to a bucket the index covers (its first k letters are ACGT, its
// seed entry has 2 .. max_bucket rows), and if so its bucket key -- the k-mer with the FIRST letter most significant, so
// that keys ascend with the rows -- with the completeness of its left context as the lowest bit, and its context key.
__global__ __launch_bounds__(256) void lcx_rowinfo_kernel(DevIndex ix, uint32_t r0, uint32_t n, uint32_t max_bucket, uint64_t* __restrict__ bucket_key,
                                                          uint64_t* __restrict__ ctx_key, uint8_t* __restrict__ valid) {
  const int k = ix.seed_k;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const uint32_t p = ix.dense_sa[r0 + j];
    bool ok = false;
    const uint64_t kmer = lcx_text_letters(ix.text4, p, k, &ok);  // letter t of the suffix in bits [2t, 2t + 2): the seed-table index
    uint8_t v = 0;
    uint64_t bk = 0, ck = 0;
    if (ok) {
      const SeedEntry e = ix.seed[kmer];
      const uint32_t cnt = seed_cnt(e);
      if (cnt >= 2u && cnt <= max_bucket && cnt != SEED_CNT_SAT) {
        uint64_t lex = 0;  // first letter most significant
        for (int t = 0; t < k; t++) lex = (lex << 2) | ((kmer >> (2 * t)) & 3ull);
        bool complete = false;
        if (p >= (uint32_t)LCX_CTX) ck = lcx_text_letters(ix.text4, (uint64_t)p - LCX_CTX, LCX_CTX, &complete);
        if (!complete) ck = ~0ull;  // (sorts behind the complete entries of its bucket either way: the bucket key's low bit)
        bk = (lex << 1) | (complete ? 0ull : 1ull);
        v = 1;
      }
    }
    valid[j] = v;
    bucket_key[j] = bk;
    ctx_key[j] = ck;
  }
}

// marks the entries the index does not cover (more rows than max_bucket, saturated counts)
__global__ __launch_bounds__(256) void lcx_flag_big_kernel(SeedEntry* __restrict__ table, uint64_t nentries, uint32_t max_bucket) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nentries; o += stride) {
    const SeedEntry e = table[o];
    if (seed_has_ctx(e)) continue;
    const uint32_t cnt = e.cnt & SEED_CNT_SAT;
    if (cnt >= 2u && (cnt > max_bucket || cnt == SEED_CNT_SAT)) table[o].cnt = e.cnt | SEED_LCX_NONE;
  }
}

// chunk boundary: the first row of the bucket that holds row r (r itself when it is in no covered bucket)
__global__ void lcx_bucket_start_kernel(DevIndex ix, uint32_t r, uint32_t max_bucket, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t res = r;
  if ((uint64_t)r < ix.bwt_len) {
    const uint32_t p = ix.dense_sa[r];
    bool ok = false;
    const uint64_t kmer = lcx_text_letters(ix.text4, p, ix.seed_k, &ok);
    if (ok) {
      const SeedEntry e = ix.seed[kmer];
      const uint32_t cnt = seed_cnt(e);
      if (cnt >= 2u && cnt <= max_bucket && cnt != SEED_CNT_SAT) res = e.sp;
    }
  }
  *out = res;
}

__global__ __launch_bounds__(256) void lcx_gather_u64_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint64_t* __restrict__ dst) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = src[idx[j]];
}
__global__ __launch_bounds__(256) void lcx_gather2_u64_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ outer, const uint32_t* __restrict__ inner,
                                                              uint64_t n, uint64_t* __restrict__ dst) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = src[inner[outer[j]]];
}
__global__ __launch_bounds__(256) void lcx_iota_kernel(uint32_t* __restrict__ dst, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dst[j] = (uint32_t)j;
}

// Sorted entry t of a chunk (nv of them, ascending by bucket key, then context key, then row -- both sorts are stable)
// goes to the t-th covered row of the chunk: the covered rows of a bucket are its rows, and the buckets ascend with the
// rows, so every bucket's entries land on its own rows.  k1 / p1: context keys sorted by themselves and the covered-row
// index each came from; q2: the order of those by bucket key; slot: the covered rows of the chunk, ascending (local).
__global__ __launch_bounds__(256) void lcx_place_kernel(DevIndex ix, uint32_t r0, uint64_t nv, const uint64_t* __restrict__ k1, const uint32_t* __restrict__ p1,
                                                        const uint32_t* __restrict__ q2, const uint32_t* __restrict__ slot,
                                                        uint64_t* __restrict__ key, uint64_t* __restrict__ rowpos) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += stride) {
    const uint32_t j = q2[t];
    const uint32_t from = r0 + slot[p1[j]], to = r0 + slot[t];
    key[to] = k1[j];
    rowpos[to] = (uint64_t)ix.dense_sa[from] | ((uint64_t)from << 32);
  }
}

// The incomplete entries of a bucket are its last ones (bucket key bit 0).  The thread of a bucket's LAST incomplete entry
// counts them, leaves the count in that entry's key slot -- the bucket's last row -- and flags the seed entry.
// b2: the chunk's bucket keys in sorted order.
__global__ __launch_bounds__(256) void lcx_tail_kernel(DevIndex ix, uint32_t r0, uint64_t nv, const uint64_t* __restrict__ b2, const uint32_t* __restrict__ slot,
                                                       uint64_t* __restrict__ key, SeedEntry* __restrict__ table) {
  const int k = ix.seed_k;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += stride) {
    const uint64_t b = b2[t];
    if (!(b & 1ull) || (t + 1 < nv && b2[t + 1] == b)) continue;
    uint64_t c = 1;
    while (c <= t && b2[t - c] == b) c++;
    key[r0 + slot[t]] = c;
    uint64_t lex = b >> 1, kmer = 0;
    for (int j = 0; j < k; j++) { kmer = (kmer << 2) | (lex & 3ull); lex >>= 2; }  // back to the table's letter order
    atomicOr(&table[kmer].cnt, SEED_LCX_TAIL);
  }
}

// level t of the index: the key of every row that is a multiple of 16^t
__global__ __launch_bounds__(256) void lcx_sample_kernel(const uint64_t* __restrict__ key, uint64_t bwt_len, int t, uint64_t* __restrict__ level, uint64_t nlevel) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nlevel; j += stride) {
    const uint64_t r = j << (4 * t);
    level[j] = r < bwt_len ? key[r] : 0ull;
  }
}

}  // namespace awry
