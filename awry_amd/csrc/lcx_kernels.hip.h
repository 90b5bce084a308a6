// lcx_kernels.hip.h -- phase 2 of the two-phase count schedules when the left-context index (layout.h, lcx.hip.h) is
// resident: one survivor per LANE.  Included at the end of kernels.hip.h.
//
// The quad kernels (count_nt2_resume_kernel, count_nt2_reads_kernel) search a bucket with four lanes per query: 16
// searches per wave, each a chain of 3..6 dependent line reads -- ~100 K line requests in flight on the chip, a third of
// what the memory system needs to reach its random-line rate (measured: 14 G lines/s on k-mers drawn from a repeat-rich
// GRCh38-scale text).  Here every lane carries its own search, 64 per wave, and the 128-B nodes the lanes of a wave need in
// an iteration are fetched COOPERATIVELY -- eight lanes per node, so one load instruction covers eight whole lines, as in
// locate_walk_nt_lane_kernel -- and handed to their lanes through a wave-private LDS tile.  A lane is a small state
// machine (FETCH the survivor record, TAIL count, SEARCH levels, candidate / tail entries against the text); every
// iteration first issues the one load group each lane's state needs and only then consumes the results.
// The block-private survivor lists of the probe pass are consumed as ONE pool: a block keeps the prefix sums of the list
// lengths in LDS and its waves draw batches of 64 items from a device-wide counter, so every wave has work until the pool is
// empty whatever the lists' lengths (a random batch leaves a few survivors per list, a batch from repeats thousands).
// What this pass cannot settle -- saturated or uncovered buckets, more candidates than are worth comparing one by one,
// ranges the locate pass needs as rows -- goes to one device-wide list, which count_nt2_resume_pool_kernel /
// count_nt2_reads_pool_kernel then work through with the quad code (LF steps).
#pragma once

namespace awry {

enum : int { LL_IDLE = 0, LL_FETCH, LL_FETCH2, LL_TAILCNT, LL_SEARCH, LL_POS, LL_TXT, LL_DONE };

// READS: survivors of count_nt2_reads_probe_kernel (in.w = the <= 32 letters left of the seed window, in.range = the probed
// entry, in.q = the read); else survivors of count_nt2_probe_kernel (in.w = the k-mer).  out: the lists of what is left for
// the quad code, same geometry.  nlists block-private lists, worked through by gridDim.x blocks.
template <bool READS, bool RAGGED, bool TALLY>
__global__ __launch_bounds__(256) void lcx_lane_kernel(DevIndex ix, const uint64_t* __restrict__ queries, int L, uint64_t* __restrict__ counts,
                                                       uint64_t* __restrict__ range_start, Nt2Survivors in, Nt2Survivors out, uint32_t nlists,
                                                       const uint32_t* __restrict__ lens, unsigned long long* __restrict__ ctr,
                                                       unsigned long long* __restrict__ tally) {
  constexpr int ROW = 9;  // 16-B pieces per tile row: 8 + 1 of padding against bank conflicts
  __shared__ ulonglong2 s_blk[4][64 * ROW];
  __shared__ uint32_t s_pref[LIST_MAX_LISTS + 1];  // exclusive prefix sums of the lists' lengths
  const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
  const uint64_t lane_lt = (1ull << lane) - 1;
  const int k = ix.seed_k, W = (L + 31) / 32;
  const uint32_t* __restrict__ text4 = ix.text4;
  unsigned long long t_nodes = 0, t_rp = 0, t_txt = 0;
  uint32_t ns = 0;  // items in the pool
  {
    const uint32_t per = (nlists + blockDim.x - 1) / blockDim.x, l0 = threadIdx.x * per;
    uint64_t mine = 0;
    for (uint32_t j = 0; j < per; j++) mine += l0 + j < nlists ? in.count[l0 + j] : 0u;
    uint64_t tot;
    uint64_t run = block_excl_scan(mine, &tot);
    for (uint32_t j = 0; j < per; j++)
      if (l0 + j < nlists) { s_pref[l0 + j] = (uint32_t)run; run += in.count[l0 + j]; }
    if (threadIdx.x == 0) s_pref[nlists] = (uint32_t)tot;
    ns = (uint32_t)tot;
  }
  __syncthreads();
  {
    uint32_t cur = 0, end = 0;  // wave-uniform: the unassigned rest of this wave's batch
    bool exhausted = false;
    int state = LL_IDLE;
    uint64_t w = 0, qlo = 0, qhi = 0;
    uint32_t q = 0, sp = 0, cnt = 0, inc = 0, flags = 0;
    uint64_t item = 0;  // where this lane's survivor record lies
    uint32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
    int t0 = -1, t1 = -1, i = 0;
    uint32_t hits = 0, lb = 0, vmask = 0, vp = 0;
    int vj = 0, vn = 0, vc = 0;   // candidate / tail entry being checked, how many there are, text chunk
    bool tail_pass = false;
    for (;;) {
      const uint64_t nm = __ballot(state == LL_IDLE);
      if (nm) {
        if (cur == end && !exhausted) {
          unsigned long long base = 0;
          if (lane == 0) base = atomicAdd(&ctr[0], 64ull);
          base = __shfl(base, 0, 64);
          cur = base < ns ? (uint32_t)base : ns;
          end = base + 64ull < ns ? (uint32_t)base + 64u : ns;
          exhausted = cur == end;
        }
        if (exhausted && nm == ~0ull) break;
        if (state == LL_IDLE) {
          const uint32_t idx = cur + (uint32_t)__popcll(nm & lane_lt);
          if (idx < end) {  // item idx of the pool: list l with s_pref[l] <= idx < s_pref[l + 1]
            uint32_t lo = 0, hi = nlists;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pref[mid] <= idx) lo = mid; else hi = mid; }
            item = (uint64_t)lo * in.cap + (idx - s_pref[lo]);
            state = LL_FETCH;
          }
        }
        const uint32_t adv = cur + (uint32_t)__popcll(nm);
        cur = adv < end ? adv : end;
      }
      // ---- SEARCH: which node does this lane need next?  (levels without a sampled row inside a bound's range are passed)
      bool want_node = false, shared = false;
      int bound = 0, tl = 0;
      uint64_t nb = 0;
      const uint64_t* node = nullptr;
      if (state == LL_SEARCH) {
        while (t0 >= 0 && lcx_first_sample(a0, 4 * t0) >= b0) t0--;
        while (t1 >= 0 && lcx_first_sample(a1, 4 * t1) >= b1) t1--;
        if (t0 < 0 && t1 < 0) state = LL_DONE;
        else {
          shared = t0 == t1 && a0 == a1 && b0 == b1;
          bound = shared ? 2 : (t0 >= t1 ? 0 : 1);
          tl = bound == 1 ? t1 : t0;
          nb = (lcx_first_sample(bound == 1 ? a1 : a0, 4 * tl) >> (4 * tl)) & ~15ull;
          node = (tl == 0 ? ix.lcx_key : ix.lcx_inner + ix.lcx_off[tl]) + nb;
          want_node = true;
        }
      }
      // ---- DONE: what the search found decides what comes next (no load in this iteration)
      bool settle = false, fallback = false;
      uint64_t out_count = 0, out_rs = 0;
      if (state == LL_DONE) {
        if (!tail_pass && vn == 0) {  // straight from the search
          lb = a0;
          hits = a1 - a0;
          if (!READS || i <= LCX_CTX) {
            if (inc && i < LCX_CTX) { tail_pass = true; vj = 0; vn = (int)inc; vmask = 0; state = LL_POS; }
            else if (!READS || !range_start || hits <= 8u) {
              settle = true;
              out_count = hits;
              out_rs = hits && hits <= 8u ? ((RS_LCX << RS_MODE_SHIFT) | (uint64_t)lb | ((uint64_t)i << 32) | ((uint64_t)((1u << hits) - 1u) << 48)) : ((RS_PLAIN << RS_MODE_SHIFT) | 1ull);
            } else fallback = true;  // the locate pass wants the rows of a larger range
          } else if (hits == 0u) { settle = true; out_rs = (RS_PLAIN << RS_MODE_SHIFT) | 1ull; }
          else if (hits <= 8u) { vj = 0; vn = (int)hits; vmask = 0; state = LL_POS; }  // compare each with the text
          else fallback = true;
        } else if (tail_pass) {  // the bucket's incomplete entries have been compared with the text
          const uint32_t th = (uint32_t)__popc(vmask);
          if (!READS || !range_start || (th == 0u && hits <= 8u)) {
            settle = true;
            out_count = (uint64_t)hits + th;
            out_rs = hits && hits <= 8u ? ((RS_LCX << RS_MODE_SHIFT) | (uint64_t)lb | ((uint64_t)i << 32) | ((uint64_t)((1u << hits) - 1u) << 48)) : ((RS_PLAIN << RS_MODE_SHIFT) | 1ull);
          } else fallback = true;
        } else {  // the candidates have been compared with the text
          settle = true;
          out_count = (uint64_t)__popc(vmask);
          if (vn == 1 && vmask) out_rs = (RS_SINGLE << RS_MODE_SHIFT) | ((uint64_t)vp - (uint64_t)i);
          else out_rs = (RS_LCX << RS_MODE_SHIFT) | (uint64_t)lb | ((uint64_t)i << 32) | ((uint64_t)vmask << 48);
        }
      }
      // ---- issue.  (The load groups of the states share their registers: a lane is in one state.)
      uint64_t f0 = 0, f1 = 0, f2 = 0;  // FETCH: word, entry, query index; FETCH2: length; TAILCNT: key; POS: (position, row); TXT: query words
      ulonglong2 B[8];                  // the node pieces this lane fetches for the wave, then (SEARCH) this lane's own node
      Text20 tx[2];                     // TXT: two text windows of 32 letters
#pragma unroll
      for (int j = 0; j < 8; j++) B[j] = ulonglong2{0, 0};
      tx[0] = tx[1] = Text20{{0u, 0u, 0u, 0u, 0u}};
      if (state == LL_FETCH) {
        f0 = in.w[item];
        f1 = in.range[item];
        f2 = in.q[item];
      }
      if (READS && RAGGED && state == LL_FETCH2) f2 = lens[q];
      if (state == LL_TAILCNT) f0 = ix.lcx_key[sp + cnt - 1u];
      if (state == LL_POS) f0 = ix.lcx_rowpos[(tail_pass ? sp + (cnt - inc) : lb) + (uint32_t)vj];
      if (state == LL_TXT) {
        const uint64_t g = (uint64_t)vp - (uint64_t)i;
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const int ch = vc + c;
          if (32 * ch < i) {
            tx[c] = *reinterpret_cast<const Text20*>(text4 + ((g + 32ull * ch) >> 3));
            const uint64_t qword = READS ? queries[(uint64_t)q * W + ch] : w;
            if (c == 0) f0 = qword; else f1 = qword;
          }
        }
      }
      const uint64_t wm = __ballot(want_node);
      if (wm) {  // round r: the eight lanes 8j..8j+7 fetch the node of lane 8r + j, one 16-B piece each
        const unsigned long long mine = reinterpret_cast<unsigned long long>(node);
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const int src = 8 * r + (lane >> 3);
          const unsigned long long p = __shfl(mine, src, 64);
          if ((wm >> src) & 1ull) B[r] = reinterpret_cast<const ulonglong2*>(p)[lane & 7];
        }
      }
      asm volatile("" : "+v"(tx[0].w[0]), "+v"(tx[0].w[1]), "+v"(tx[0].w[2]), "+v"(tx[0].w[3]), "+v"(tx[0].w[4]),
                        "+v"(tx[1].w[0]), "+v"(tx[1].w[1]), "+v"(tx[1].w[2]), "+v"(tx[1].w[3]), "+v"(tx[1].w[4]));
      asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(B[0].x), "+v"(B[0].y), "+v"(B[1].x), "+v"(B[1].y), "+v"(B[2].x), "+v"(B[2].y),
                   "+v"(B[3].x), "+v"(B[3].y), "+v"(B[4].x), "+v"(B[4].y), "+v"(B[5].x), "+v"(B[5].y), "+v"(B[6].x), "+v"(B[6].y),
                   "+v"(B[7].x), "+v"(B[7].y));
      if (wm) {  // pieces -> tile, then every searching lane picks up its own node
#pragma unroll
        for (int r = 0; r < 8; r++) s_blk[wv_id][(8 * r + (lane >> 3)) * ROW + (lane & 7)] = B[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (want_node) {
#pragma unroll
          for (int j = 0; j < 8; j++) B[j] = s_blk[wv_id][lane * ROW + j];
        }
        __builtin_amdgcn_wave_barrier();
      }
      // ---- consume
      bool to_lf = false;  // this lane's survivor goes to the LF list
      if (settle) {
        counts[q] = out_count;
        if (READS && range_start) range_start[q] = out_rs;
        state = LL_IDLE;
      } else if (fallback) {
        to_lf = true;
      } else if (state == LL_FETCH) {
        w = f0;
        q = (uint32_t)f2;
        const uint64_t f_rg = f1;
        sp = (uint32_t)f_rg;
        const uint32_t cf = (uint32_t)(f_rg >> 32);
        cnt = cf & SEED_CNT_SAT;
        flags = cf & (SEED_LCX_TAIL | SEED_LCX_NONE);
        inc = 0;
        vn = 0;
        tail_pass = false;
        i = L - k;
        // what only LF steps can do: saturated / uncovered buckets, entries the probe pass did not read, nothing left of the window
        const bool lf = f_rg == ~0ull || cnt == SEED_CNT_SAT || cnt < 2u || (flags & SEED_LCX_NONE) || i <= 0 || i >= 65536;
        if (lf && !(READS && RAGGED)) to_lf = true;
        else if (READS && RAGGED) state = LL_FETCH2;
        else {
          lcx_thresholds(w, i < LCX_CTX ? i : LCX_CTX, &qlo, &qhi);
          if (flags & SEED_LCX_TAIL) state = LL_TAILCNT;
          else { a0 = a1 = sp; b0 = b1 = sp + cnt; t0 = t1 = lcx_top_level(sp, cnt); state = LL_SEARCH; }
        }
      } else if (READS && RAGGED && state == LL_FETCH2) {
        i = (int)(uint32_t)f2 - k;
        const bool lf = cnt == SEED_CNT_SAT || cnt < 2u || (flags & SEED_LCX_NONE) || i <= 0 || i >= 65536 || sp == 0xFFFFFFFFu;
        if (lf) to_lf = true;
        else {
          lcx_thresholds(w, i < LCX_CTX ? i : LCX_CTX, &qlo, &qhi);
          if (flags & SEED_LCX_TAIL) state = LL_TAILCNT;
          else { a0 = a1 = sp; b0 = b1 = sp + cnt; t0 = t1 = lcx_top_level(sp, cnt); state = LL_SEARCH; }
        }
      } else if (state == LL_TAILCNT) {
        inc = (uint32_t)f0;
        if (TALLY) t_nodes++;
        if (i < LCX_CTX && inc > (uint32_t)LCX_TAIL_MAX) to_lf = true;  // too many incomplete entries to check one by one
        else {
          const uint32_t nc = cnt - inc;
          a0 = a1 = sp; b0 = b1 = sp + nc;
          t0 = t1 = nc ? lcx_top_level(sp, nc) : -1;
          state = LL_SEARCH;
        }
      } else if (want_node) {  // one node of level tl: the keys of rows (nb + u) << 4 tl
        const int s = 4 * tl;
        uint32_t c0 = 0, c1 = 0;
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const uint64_t key = (u & 1) ? B[u >> 1].y : B[u >> 1].x;
          const uint64_t row = (nb + (uint64_t)u) << s;
          if (bound != 1) c0 += (row >= a0 && row < b0 && key < qlo) ? 1u : 0u;
          if (bound != 0) c1 += (row >= a1 && row < b1 && key <= qhi) ? 1u : 0u;
        }
        if (TALLY) t_nodes++;
        if (bound != 1) {
          const uint64_t f = lcx_first_sample(a0, s), nbnd = f + ((uint64_t)c0 << s);
          b0 = nbnd < b0 ? (uint32_t)nbnd : b0;
          if (c0) a0 = (uint32_t)(f + ((uint64_t)(c0 - 1) << s) + 1);
          t0--;
        }
        if (bound != 0) {
          const uint64_t f = lcx_first_sample(a1, s), nbnd = f + ((uint64_t)c1 << s);
          b1 = nbnd < b1 ? (uint32_t)nbnd : b1;
          if (c1) a1 = (uint32_t)(f + ((uint64_t)(c1 - 1) << s) + 1);
          t1--;
        }
      } else if (state == LL_POS) {
        vp = (uint32_t)f0;
        if (TALLY) t_rp++;
        if (vp >= (uint32_t)i) { state = LL_TXT; vc = 0; }
        else { vj++; if (vj >= vn) state = LL_DONE; }  // the suffix starts too close to the text's beginning
      } else if (state == LL_TXT) {
        uint32_t bad = 0;
        const uint64_t g = (uint64_t)vp - (uint64_t)i;
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const int ch = vc + c, m = i - 32 * ch;
          if (m > 0) {
            TextWin tw;
            tw.t = tx[c];
            tw.m = m > 32 ? 32 : m;
            tw.sh = 4 * (int)((g + 32ull * ch) & 7);
            bad |= text_window_differs(tw, c == 0 ? f0 : f1);
          }
        }
        if (TALLY) t_txt++;
        if (!bad && 32 * (vc + 2) < i) vc += 2;  // more of the read to compare
        else {
          if (!bad) vmask |= 1u << vj;
          vj++;
          state = vj >= vn ? LL_DONE : LL_POS;
        }
      }
      const uint64_t fm = __ballot(to_lf);
      if (fm) {  // one device-wide list (out.count[0] its length), one atomic per wave
        unsigned int slot0 = 0;
        if (lane == 0) slot0 = atomicAdd(out.count, (unsigned int)__popcll(fm));
        slot0 = __shfl(slot0, 0, 64);
        if (to_lf) {
          const uint64_t slot = slot0 + (uint64_t)__popcll(fm & lane_lt);
          out.q[slot] = q;
          if (!READS) { out.w[slot] = w; out.range[slot] = (uint64_t)sp | ((uint64_t)(cnt | flags | SEED_LCX_NONE) << 32); }
          state = LL_IDLE;
        }
      }
    }
  }
  if (TALLY && tally && (t_nodes | t_rp | t_txt)) {
    atomicAdd(&tally[6], t_nodes);
    atomicAdd(&tally[7], t_rp);
    atomicAdd(&tally[4], t_txt);
  }
}

}  // namespace awry
