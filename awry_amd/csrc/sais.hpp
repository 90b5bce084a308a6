// sais.hpp -- host suffix-array construction by induced sorting (SA-IS, Nong/Zhang/Chan 2009), O(n).
// Replaces the third-party libsufr SufrBuilder the reference calls at build time
// (/root/reference src/fm_index.rs:156-181); what matters for parity is only its result: the plain
// lexicographic suffix array of the byte text (which ends in a unique, smallest '$').
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace awry {

template <class Sym, class Idx>
void sais_rec(const Sym* s, Idx* sa, Idx n, Idx sigma);

namespace detail {

template <class Sym, class Idx>
struct Sais {
  const Sym* s;
  Idx* sa;
  Idx n, sigma;
  std::vector<bool> stype;       // true = S-type
  std::vector<Idx> bkt_l, bkt_s;  // bucket heads for L-type / heads of the S-region, per symbol

  bool is_lms(Idx i) const { return i > 0 && stype[i] && !stype[i - 1]; }

  void classify() {
    stype.assign((size_t)n, false);
    for (Idx i = n - 2; i >= 0; i--) stype[i] = s[i] == s[i + 1] ? stype[i + 1] : s[i] < s[i + 1];
    bkt_l.assign((size_t)sigma + 1, 0);
    bkt_s.assign((size_t)sigma + 1, 0);
    for (Idx i = 0; i < n; i++) {
      if (!stype[i]) bkt_s[s[i]]++;
      else bkt_l[(size_t)s[i] + 1]++;
    }
    // bkt_s[c] = first slot of the S-region of bucket c, bkt_l[c] = first slot of bucket c
    for (Idx c = 0; c <= sigma; c++) {
      bkt_s[c] += bkt_l[c];
      if (c < sigma) bkt_l[(size_t)c + 1] += bkt_s[c];
    }
  }

  // place the given LMS suffixes (in order) at the heads of their S-regions, then induce L and S
  template <class It>
  void induce(It lms_begin, It lms_end) {
    std::fill(sa, sa + n, (Idx)-1);
    std::vector<Idx> buf(bkt_s.begin(), bkt_s.end());
    for (It it = lms_begin; it != lms_end; ++it) {
      Idx d = *it;
      if (d == n) continue;
      sa[buf[s[d]]++] = d;
    }
    std::copy(bkt_l.begin(), bkt_l.end(), buf.begin());
    sa[buf[s[n - 1]]++] = n - 1;
    for (Idx i = 0; i < n; i++) {
      Idx v = sa[i];
      if (v >= 1 && !stype[v - 1]) sa[buf[s[v - 1]]++] = v - 1;
    }
    std::copy(bkt_l.begin(), bkt_l.end(), buf.begin());
    for (Idx i = n - 1; i >= 0; i--) {
      Idx v = sa[i];
      if (v >= 1 && stype[v - 1]) sa[--buf[(size_t)s[v - 1] + 1]] = v - 1;
    }
  }

  void run() {
    classify();
    std::vector<Idx> lms;
    for (Idx i = 1; i < n; i++)
      if (is_lms(i)) lms.push_back(i);
    const Idx m = (Idx)lms.size();
    induce(lms.begin(), lms.end());
    if (m == 0) return;
    // LMS substrings are now in sorted order inside sa; name them
    std::vector<Idx> sorted;
    sorted.reserve((size_t)m);
    for (Idx i = 0; i < n; i++)
      if (is_lms(sa[i])) sorted.push_back(sa[i]);
    // lms_rank[i/2]: position of LMS suffix i among LMS suffixes in text order (LMS are >= 2 apart)
    std::vector<Idx> lms_rank((size_t)n / 2 + 1, (Idx)-1);
    for (Idx j = 0; j < m; j++) lms_rank[(size_t)lms[j] / 2] = j;
    auto end_of = [&](Idx p) {
      Idx j = lms_rank[(size_t)p / 2];
      return j + 1 < m ? lms[j + 1] : n;
    };
    std::vector<Idx> reduced((size_t)m);
    Idx names = 0;
    reduced[lms_rank[(size_t)sorted[0] / 2]] = 0;
    for (Idx i = 1; i < m; i++) {
      Idx l = sorted[i - 1], r = sorted[i];
      Idx el = end_of(l), er = end_of(r);
      bool same = (el - l) == (er - r);
      if (same) {
        while (l < el && s[l] == s[r]) { l++; r++; }
        same = !(l == n || s[l] != s[r]);
      }
      if (!same) names++;
      reduced[lms_rank[(size_t)sorted[i] / 2]] = names;
    }
    if (names + 1 < m) {
      std::vector<Idx> sub((size_t)m);
      sais_rec<Idx, Idx>(reduced.data(), sub.data(), m, names);
      for (Idx i = 0; i < m; i++) sorted[i] = lms[sub[i]];
    } else {
      for (Idx i = 0; i < m; i++) sorted[reduced[i]] = lms[i];
    }
    std::vector<Idx>().swap(reduced);
    std::vector<Idx>().swap(lms_rank);
    induce(sorted.begin(), sorted.end());
  }
};

}  // namespace detail

// s[0..n) with symbols in [0, sigma]; sa receives the suffix array
template <class Sym, class Idx>
void sais_rec(const Sym* s, Idx* sa, Idx n, Idx sigma) {
  if (n == 0) return;
  if (n == 1) { sa[0] = 0; return; }
  if (n == 2) {
    bool lt = s[0] < s[1];
    sa[0] = lt ? 0 : 1;
    sa[1] = lt ? 1 : 0;
    return;
  }
  detail::Sais<Sym, Idx> w{s, sa, n, sigma, {}, {}, {}};
  w.run();
}

// Suffix array of a byte text; u64 output.  32-bit workspace when n < 2^31, else 64-bit.
inline void suffix_array_bytes(const uint8_t* text, uint64_t n, uint64_t* sa_out) {
  if (n < (1ull << 31)) {
    std::vector<int32_t> sa((size_t)n);
    sais_rec<uint8_t, int32_t>(text, sa.data(), (int32_t)n, 255);
    for (uint64_t i = 0; i < n; i++) sa_out[i] = (uint64_t)sa[i];
  } else {
    sais_rec<uint8_t, int64_t>(text, reinterpret_cast<int64_t*>(sa_out), (int64_t)n, 255);
  }
}

}  // namespace awry
