// lcx_kernels.hip.h -- phase 2 of the two-phase count schedules when the left-context index (layout.h, lcx.hip.h) is
// resident.  Included at the end of kernels.hip.h.
//
// count_nt2_resume_kernel / count_nt2_reads_kernel<.., LIST> search the index too, but they are built around LF steps: one
// survivor per quad at a time, block b working through the list block b wrote.  A search is a chain of 3..6 dependent
// line reads, so those kernels keep ~100 K requests in flight -- a third of what the memory system needs -- and a batch
// from repeats leaves lists of very different lengths (measured on k-mers drawn from a repeat-rich GRCh38-scale text:
// 14 G lines/s, 960 us for 4.3 M survivors).  This kernel does nothing but search:
//   * the block-private survivor lists of the probe pass are consumed as ONE pool (prefix sums of the list lengths in
//     LDS; every wave takes an equal, contiguous share), so every wave has work until the pool is empty;
//   * per iteration a quad first issues the load group its search needs -- both bounds' nodes once they have parted -- and
//     only then consumes it (SLOTS searches per quad: one, see the end of this file);
//   * a search loads one 128-B node per level as two 16-B loads per lane (a 64-B half line per instruction), counts with
//     four 64-bit compares per lane and two quad_perm DPP adds;
//   * what it cannot settle -- saturated or uncovered buckets, more candidates than are worth comparing with the text one
//     by one, ranges the locate pass needs as rows -- goes to ONE device-wide list that count_nt2_reads_pool_kernel works through
//     with LF steps.
// (A one-search-per-LANE variant with cooperative node fetches through LDS was measured first: 64 searches per wave, but
//  ~600 instructions per wave iteration -- every lane compares all 16 keys of its node, and the lanes of a wave are in
//  different states -- made it instruction-bound: 800 us for the same 4.3 M survivors, 9.2 ms against 14.5 for reads.)
#pragma once

namespace awry {

enum : int { LQ_IDLE = 0, LQ_FETCH, LQ_FETCH2, LQ_TAILCNT, LQ_SEARCH, LQ_POS, LQ_TXT, LQ_DONE };
constexpr int LCX_LF_CHUNK = 256;  // slots of the LF list a wave reserves at a time (16 quads x 2 slots append at most 32 per step)

// READS: survivors of count_nt2_reads_probe_kernel (in.w = the <= 32 letters left of the seed window, in.range = the probed
// entry or ~0, in.q = the read); else survivors of count_nt2_probe_kernel (in.w = the k-mer).  out: the device-wide list of
// what is left for the LF kernels (out.count[0] slots, empty ones marked q = ~0; k-mers: records as the probe pass writes
// them, with SEED_LCX_NONE set); the probe pass clears out.count[0] (Nt2Survivors::lf_count).
template <bool READS, bool RAGGED, bool TALLY, int SLOTS>
__device__ __forceinline__ void lcx_quad_body(const DevIndex& ix, const uint64_t* __restrict__ queries, int L, uint64_t* __restrict__ counts,
                                              uint64_t* __restrict__ range_start, const Nt2Survivors& in, const Nt2Survivors& out, uint32_t nlists,
                                              const uint32_t* __restrict__ lens, unsigned long long* __restrict__ tally) {
  __shared__ uint32_t s_pref[LIST_MAX_LISTS + 1];  // exclusive prefix sums of the lists' lengths
  const int lane = threadIdx.x & 63, l = lane & 3;
  const uint64_t leader_lt = ((1ull << (lane & ~3)) - 1) & 0x1111111111111111ull;  // the leader lanes of the quads before this one
  const int k = ix.seed_k, W = (L + 31) / 32;
  const uint32_t* __restrict__ text4 = ix.text4;
  uint32_t t_nodes = 0, t_rp = 0, t_txt = 0;
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
  // Every wave owns an equal, contiguous share of the pool.  (Batches drawn from a device-wide counter were measured first:
  // one atomic per 64 items on one address is ~12 ns each, 0.8 ms for 4.3 M survivors -- and 40 us for a batch with next to
  // no survivors, every wave finding the counter exhausted.)  The items are alike -- a search is 2..6 dependent lines -- so
  // equal shares end together; what is not alike (LF chains) is not done here.
  const uint32_t nwaves = gridDim.x * 4u, wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t share = (ns + nwaves - 1) / nwaves;
  uint32_t cur = (uint64_t)wave_id * share < ns ? wave_id * share : ns;
  const uint32_t end = (uint64_t)cur + share < ns ? cur + share : ns;
  uint32_t lf_base = 0, lf_used = LCX_LF_CHUNK;  // wave-uniform: this wave's current chunk of the LF list
  uint32_t li = 0;  // wave-uniform: the list that holds item `cur` (s_pref[li] <= cur < s_pref[li + 1]); the share is walked in order
  if (cur < end) {
    uint32_t lo = 0, hi = nlists;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pref[mid] <= cur) lo = mid; else hi = mid; }
    li = lo;
  }
  int state[SLOTS];
  uint64_t w[SLOTS], qlo[SLOTS];  // (the upper threshold is qlo with the letters the query does not have set to ones)
  uint32_t q[SLOTS], sp[SLOTS], cnt[SLOTS], inc[SLOTS], flags[SLOTS];
  uint32_t a0[SLOTS], b0[SLOTS], a1[SLOTS], b1[SLOTS], vmask[SLOTS], vp[SLOTS];
  int t0[SLOTS], t1[SLOTS], ii[SLOTS], vj[SLOTS], vn[SLOTS], vc[SLOTS];
  bool tail_pass[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    state[s] = LQ_IDLE;
    w[s] = qlo[s] = 0;
    q[s] = sp[s] = cnt[s] = inc[s] = flags[s] = a0[s] = b0[s] = a1[s] = b1[s] = vmask[s] = vp[s] = 0;
    t0[s] = t1[s] = -1;
    ii[s] = vj[s] = vn[s] = vc[s] = 0;
    tail_pass[s] = false;
  }
  for (;;) {
    // K[s][0..3], the registers a slot's loads land in: FETCH word / entry / index; TAILCNT key; POS entry; SEARCH the lane's
    // four keys of the node; TXT the lane's text window (5 words) and query word
    uint64_t K[SLOTS][8];  // ([4..7]: the upper bound's node once the two bounds have parted)
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
#pragma unroll
      for (int u = 0; u < 8; u++) K[s][u] = 0;
    // ---- hand out items to idle slots, in item order (quad-uniform decisions throughout); their records are asked for at once
    bool any_busy = false;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      const uint64_t nm = cur < end ? __ballot(state[s] == LQ_IDLE && l == 0) : 0ull;
      if (nm) {
        if (state[s] == LQ_IDLE) {
          const uint32_t idx = cur + (uint32_t)__popcll(nm & leader_lt);
          if (idx < end) {  // item idx of the pool: list lo with s_pref[lo] <= idx < s_pref[lo + 1], at or a little after li
            uint32_t lo = li;
            while (s_pref[lo + 1] <= idx) lo++;
            const uint64_t item = (uint64_t)lo * in.cap + (idx - s_pref[lo]);
            K[s][0] = in.w[item];
            K[s][1] = in.range[item];
            K[s][2] = in.q[item];
            state[s] = LQ_FETCH;
          }
        }
        const uint32_t adv = cur + (uint32_t)__popcll(nm);
        cur = adv < end ? adv : end;
        while (cur < end && s_pref[li + 1] <= cur) li++;
      }
      any_busy = any_busy || state[s] != LQ_IDLE;
    }
    if (!__any(any_busy)) break;  // (idle slots were offered the rest of the share above: nothing is left)
    // ---- per slot: what to load
    bool want_node[SLOTS], to_lf[SLOTS], shared[SLOTS];
    uint32_t nb0[SLOTS], nb1[SLOTS];  // the nodes the two bounds consult (levels t0 / t1; < 0: that bound is done)
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      want_node[s] = to_lf[s] = shared[s] = false;
      nb0[s] = nb1[s] = 0;
      bool settle = false;
      uint64_t out_count = 0, out_rs = 0;
      if (state[s] == LQ_SEARCH) {  // levels without a sampled row inside a bound's range are passed without a load
        while (t0[s] >= 0 && lcx_first_sample(a0[s], 4 * t0[s]) >= b0[s]) t0[s]--;
        while (t1[s] >= 0 && lcx_first_sample(a1[s], 4 * t1[s]) >= b1[s]) t1[s]--;
        if (t0[s] < 0 && t1[s] < 0) state[s] = LQ_DONE;
        else {
          shared[s] = t0[s] == t1[s] && a0[s] == a1[s] && b0[s] == b1[s];
          if (t0[s] >= 0) nb0[s] = (uint32_t)((lcx_first_sample(a0[s], 4 * t0[s]) >> (4 * t0[s])) & ~15ull);
          if (t1[s] >= 0) nb1[s] = (uint32_t)((lcx_first_sample(a1[s], 4 * t1[s]) >> (4 * t1[s])) & ~15ull);
          want_node[s] = true;
        }
      }
      if (state[s] == LQ_DONE) {  // what the search found decides what comes next
        const uint32_t lb = a0[s], hits = a1[s] - a0[s];
        const int i = ii[s];
        const uint64_t rs_run = hits && hits <= 8u ? ((RS_LCX << RS_MODE_SHIFT) | (uint64_t)lb | ((uint64_t)i << 32) | ((uint64_t)((1u << hits) - 1u) << 48))
                                                   : ((RS_PLAIN << RS_MODE_SHIFT) | 1ull);
        if (!tail_pass[s] && vn[s] == 0) {  // straight from the search
          if (!READS || i <= LCX_CTX) {
            if (inc[s] && i < LCX_CTX) { tail_pass[s] = true; vj[s] = 0; vn[s] = (int)inc[s]; vmask[s] = 0; state[s] = LQ_POS; }
            else if (!READS || !range_start || hits <= 8u) { settle = true; out_count = hits; out_rs = rs_run; }
            else to_lf[s] = true;  // the locate pass wants the rows of a larger range
          } else if (hits == 0u) { settle = true; out_rs = (RS_PLAIN << RS_MODE_SHIFT) | 1ull; }
          else if (hits <= 8u) { vj[s] = 0; vn[s] = (int)hits; vmask[s] = 0; state[s] = LQ_POS; }  // compare each with the text
          else to_lf[s] = true;
        } else if (tail_pass[s]) {  // the bucket's incomplete entries have been compared with the text
          const uint32_t th = (uint32_t)__popc(vmask[s]);
          if (!READS || !range_start || (th == 0u && hits <= 8u)) { settle = true; out_count = (uint64_t)hits + th; out_rs = rs_run; }
          else to_lf[s] = true;
        } else {  // the candidates have been compared with the text
          settle = true;
          out_count = (uint64_t)__popc(vmask[s]);
          if (vn[s] == 1 && vmask[s]) out_rs = (RS_SINGLE << RS_MODE_SHIFT) | ((uint64_t)vp[s] - (uint64_t)i);
          else out_rs = (RS_LCX << RS_MODE_SHIFT) | (uint64_t)lb | ((uint64_t)i << 32) | ((uint64_t)vmask[s] << 48);
        }
        if (settle) {
          if (l == 0) {
            counts[q[s]] = out_count;
            if (READS && range_start) range_start[q[s]] = out_rs;
          }
          state[s] = LQ_IDLE;  // (takes its next item in the next iteration)
        }
      }
    }
    // ---- issue
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      if (READS && RAGGED && state[s] == LQ_FETCH2) K[s][0] = lens[q[s]];
      if (state[s] == LQ_TAILCNT) K[s][0] = ix.lcx_key[sp[s] + cnt[s] - 1u];
      if (state[s] == LQ_POS) K[s][0] = ix.lcx_rowpos[(tail_pass[s] ? sp[s] + (cnt[s] - inc[s]) : a0[s]) + (uint32_t)vj[s]];
      if (want_node[s]) {
        if (t0[s] >= 0) {
          const ulonglong2* p = reinterpret_cast<const ulonglong2*>((t0[s] == 0 ? ix.lcx_key : ix.lcx_inner + ix.lcx_off[t0[s]]) + (uint64_t)nb0[s] + 4 * l);
          const ulonglong2 x = p[0], y = p[1];
          K[s][0] = x.x; K[s][1] = x.y; K[s][2] = y.x; K[s][3] = y.y;
        }
        if (t1[s] >= 0 && !shared[s]) {
          const ulonglong2* p = reinterpret_cast<const ulonglong2*>((t1[s] == 0 ? ix.lcx_key : ix.lcx_inner + ix.lcx_off[t1[s]]) + (uint64_t)nb1[s] + 4 * l);
          const ulonglong2 x = p[0], y = p[1];
          K[s][4] = x.x; K[s][5] = x.y; K[s][6] = y.x; K[s][7] = y.y;
        }
      }
      if (state[s] == LQ_TXT) {  // lane l: letters [128 vc + 32 l, + 32) of the i letters in front of the candidate
        const int j0 = 128 * vc[s] + 32 * l;
        if (j0 < ii[s]) {
          const uint64_t g = (uint64_t)vp[s] - (uint64_t)ii[s] + (uint64_t)j0;
          const Text20 t = *reinterpret_cast<const Text20*>(text4 + (g >> 3));
          K[s][0] = (uint64_t)t.w[0] | ((uint64_t)t.w[1] << 32);
          K[s][1] = (uint64_t)t.w[2] | ((uint64_t)t.w[3] << 32);
          K[s][2] = t.w[4];
          const int wi = 4 * vc[s] + l;
          K[s][3] = READS ? (wi < W ? queries[(uint64_t)q[s] * W + wi] : 0ull) : w[s];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
      asm volatile("" : "+v"(K[s][0]), "+v"(K[s][1]), "+v"(K[s][2]), "+v"(K[s][3]), "+v"(K[s][4]), "+v"(K[s][5]), "+v"(K[s][6]), "+v"(K[s][7]));
    // ---- consume
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      const int i = ii[s];
      const int m = i < LCX_CTX ? i : LCX_CTX;
      const uint64_t qhi = m >= 32 ? qlo[s] : qlo[s] | ((1ull << (64 - 2 * (m > 0 ? m : 1))) - 1);
      if (to_lf[s]) {
        // (appended below)
      } else if (state[s] == LQ_FETCH) {
        if (!READS || RAGGED) w[s] = K[s][0];  // (reads: the context word is used up by the thresholds, but for the ragged pass, which still waits for its length)
        q[s] = (uint32_t)K[s][2];
        const uint64_t rg = K[s][1];
        sp[s] = (uint32_t)rg;
        const uint32_t cf = (uint32_t)(rg >> 32);
        cnt[s] = cf & SEED_CNT_SAT;
        flags[s] = cf & (SEED_LCX_TAIL | SEED_LCX_NONE);
        inc[s] = 0;
        vn[s] = 0;
        tail_pass[s] = false;
        ii[s] = L - k;
        // what only LF steps can do: saturated / uncovered buckets, entries the probe pass did not read, nothing left of the window
        const bool lf = rg == ~0ull || cnt[s] == SEED_CNT_SAT || cnt[s] < 2u || (flags[s] & SEED_LCX_NONE) || ii[s] <= 0 || ii[s] >= 65536;
        if (lf && !(READS && RAGGED)) to_lf[s] = true;
        else if (READS && RAGGED) state[s] = LQ_FETCH2;
        else {
          { uint64_t hi_unused; lcx_thresholds(READS ? K[s][0] : w[s], ii[s] < LCX_CTX ? ii[s] : LCX_CTX, &qlo[s], &hi_unused); }
          if (flags[s] & SEED_LCX_TAIL) state[s] = LQ_TAILCNT;
          else { a0[s] = a1[s] = sp[s]; b0[s] = b1[s] = sp[s] + cnt[s]; t0[s] = t1[s] = lcx_top_level(sp[s], cnt[s]); state[s] = LQ_SEARCH; }
        }
      } else if (READS && RAGGED && state[s] == LQ_FETCH2) {
        ii[s] = (int)(uint32_t)K[s][0] - k;
        const bool lf = cnt[s] == SEED_CNT_SAT || cnt[s] < 2u || (flags[s] & SEED_LCX_NONE) || ii[s] <= 0 || ii[s] >= 65536;
        if (lf) to_lf[s] = true;
        else {
          { uint64_t hi_unused; lcx_thresholds(w[s], ii[s] < LCX_CTX ? ii[s] : LCX_CTX, &qlo[s], &hi_unused); }
          if (flags[s] & SEED_LCX_TAIL) state[s] = LQ_TAILCNT;
          else { a0[s] = a1[s] = sp[s]; b0[s] = b1[s] = sp[s] + cnt[s]; t0[s] = t1[s] = lcx_top_level(sp[s], cnt[s]); state[s] = LQ_SEARCH; }
        }
      } else if (state[s] == LQ_TAILCNT) {
        inc[s] = (uint32_t)K[s][0];
        if (TALLY) t_nodes++;
        if (i < LCX_CTX && inc[s] > (uint32_t)LCX_TAIL_MAX) to_lf[s] = true;  // too many incomplete entries to check one by one
        else {
          const uint32_t nc = cnt[s] - inc[s];
          a0[s] = a1[s] = sp[s]; b0[s] = b1[s] = sp[s] + nc;
          t0[s] = t1[s] = nc ? lcx_top_level(sp[s], nc) : -1;
          state[s] = LQ_SEARCH;
        }
      } else if (want_node[s]) {  // lane l holds keys 4 l .. 4 l + 3 of each node: rows (nb + 4 l + u) << 4 t
        const bool do0 = t0[s] >= 0, do1 = t1[s] >= 0;
        const int sh0 = 4 * (do0 ? t0[s] : 0), sh1 = 4 * (do1 ? t1[s] : 0);
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint64_t row0 = ((uint64_t)nb0[s] + (uint64_t)(4 * l + u)) << sh0, row1 = ((uint64_t)nb1[s] + (uint64_t)(4 * l + u)) << sh1;
          const uint64_t k1 = shared[s] ? K[s][u] : K[s][4 + u];
          if (do0) c += (row0 >= a0[s] && row0 < b0[s] && K[s][u] < qlo[s]) ? 1u : 0u;
          if (do1) c += (row1 >= a1[s] && row1 < b1[s] && k1 <= qhi) ? 0x100u : 0u;
        }
        c = quad_sum(c);
        if (TALLY) t_nodes += do0 && do1 && !shared[s] ? 2u : 1u;
        if (do0) {
          const uint32_t c0 = c & 0xffu;
          const uint64_t f = lcx_first_sample(a0[s], sh0), nbnd = f + ((uint64_t)c0 << sh0);
          b0[s] = nbnd < b0[s] ? (uint32_t)nbnd : b0[s];
          if (c0) a0[s] = (uint32_t)(f + ((uint64_t)(c0 - 1) << sh0) + 1);
          t0[s]--;
        }
        if (do1) {
          const uint32_t c1 = c >> 8;
          const uint64_t f = lcx_first_sample(a1[s], sh1), nbnd = f + ((uint64_t)c1 << sh1);
          b1[s] = nbnd < b1[s] ? (uint32_t)nbnd : b1[s];
          if (c1) a1[s] = (uint32_t)(f + ((uint64_t)(c1 - 1) << sh1) + 1);
          t1[s]--;
        }
      } else if (state[s] == LQ_POS) {
        vp[s] = (uint32_t)K[s][0];
        if (TALLY) t_rp++;
        if (vp[s] >= (uint32_t)i) { state[s] = LQ_TXT; vc[s] = 0; }
        else { vj[s]++; if (vj[s] >= vn[s]) state[s] = LQ_DONE; }  // the suffix starts too close to the text's beginning
      } else if (state[s] == LQ_TXT) {
        uint32_t bad = 0;
        const int j0 = 128 * vc[s] + 32 * l;
        if (j0 < i) {
          TextWin tw;
          tw.t = Text20{{(uint32_t)K[s][0], (uint32_t)(K[s][0] >> 32), (uint32_t)K[s][1], (uint32_t)(K[s][1] >> 32), (uint32_t)K[s][2]}};
          tw.m = i - j0 > 32 ? 32 : i - j0;
          tw.sh = 4 * (int)(((uint64_t)vp[s] - (uint64_t)i + (uint64_t)j0) & 7);
          bad = text_window_differs(tw, K[s][3]);
        }
        bad = quad_sum(bad);
        if (TALLY) t_txt++;
        if (!bad && 128 * (vc[s] + 1) < i) vc[s]++;  // more of the read to compare
        else {
          if (!bad) vmask[s] |= 1u << vj[s];
          vj[s]++;
          state[s] = vj[s] >= vn[s] ? LQ_DONE : LQ_POS;
        }
      }
      const uint64_t fm = __ballot(to_lf[s] && l == 0);
      if (fm) {  // the device-wide LF list: the wave appends to a chunk of its own and reserves the next one with ONE atomic
        const uint32_t nf = (uint32_t)__popcll(fm);
        if (lf_used + nf > (uint32_t)LCX_LF_CHUNK) {  // (what is left of the old chunk stays empty: q = ~0, skipped by the LF kernels)
          for (uint32_t h = lf_used + (uint32_t)lane; h < (uint32_t)LCX_LF_CHUNK; h += 64) out.q[lf_base + h] = 0xFFFFFFFFu;
          unsigned int b = 0;
          if (lane == 0) b = atomicAdd(out.count, (unsigned int)LCX_LF_CHUNK);
          lf_base = __shfl(b, 0, 64);
          lf_used = 0;
        }
        const uint32_t slot0 = lf_base + lf_used;
        lf_used += nf;
        if (to_lf[s]) {
          const uint64_t slot = slot0 + (uint64_t)__popcll(fm & leader_lt);
          if (l == 0) {
            out.q[slot] = q[s];
            if (!READS) { out.w[slot] = w[s]; out.range[slot] = (uint64_t)sp[s] | ((uint64_t)(cnt[s] | flags[s] | SEED_LCX_NONE) << 32); }
          }
          state[s] = LQ_IDLE;
        }
      }
    }
  }
  if (lf_used < (uint32_t)LCX_LF_CHUNK)  // the rest of the wave's last chunk
    for (uint32_t h = lf_used + (uint32_t)lane; h < (uint32_t)LCX_LF_CHUNK; h += 64) out.q[lf_base + h] = 0xFFFFFFFFu;
  if (TALLY && tally && l == 0 && (t_nodes | t_rp | t_txt)) {
    atomicAdd(&tally[6], (unsigned long long)t_nodes);
    atomicAdd(&tally[7], (unsigned long long)t_rp);
    atomicAdd(&tally[4], (unsigned long long)t_txt);
  }
}

// One search per quad (SLOTS = 1: 84 registers, 5 waves per SIMD); two per quad were measured as well -- 146 registers and 3 waves
// per SIMD, or 128 with spills: 12.4 / 13.3 ms against 9.9 ms for the count phase of 20 M 101-bp reads.  Packed k-mers keep
// count_nt2_resume_kernel (it searches the index with the same quad code): their survivors are numerous and need nothing but
// the search, and there this kernel's fewer resident quads cost more than its pooling gains (1.9 against 1.34 ms per 10 M
// k-mers from the repeat-rich text).
template <bool RAGGED>
__global__ __launch_bounds__(256) void lcx_quad_reads_kernel(DevIndex ix, const uint64_t* __restrict__ queries, int L, uint64_t* __restrict__ counts,
                                                             uint64_t* __restrict__ range_start, Nt2Survivors in, Nt2Survivors out, uint32_t nlists,
                                                             const uint32_t* __restrict__ lens) {
  lcx_quad_body<true, RAGGED, false, 1>(ix, queries, L, counts, range_start, in, out, nlists, lens, nullptr);
}

}  // namespace awry
