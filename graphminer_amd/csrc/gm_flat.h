// gm_flat.h -- the flattened wave64 set-intersection passes shared by the mining kernels (gm_mine.hip) and the
// big-LDS kernels (gm_wide.hip): per-wave LDS scratch, the hashed membership filter, flat_pass (owner marks + DPP scan),
// flat_pass_filtered (LDS filter -> ballot queue -> batched bisection) and drain_candidates. See DESIGN.md section 4.1.
#pragma once
#include "gm_mine.h"
#include "gm_setops.h"

namespace gm {

// per-wave scratch of the flattened passes
struct alignas(16) WaveLds {
  int4 desc[GM_WAVE];                // per-edge descriptors of the current pass
  unsigned char marks[kMarkWindow];  // owner marks of the flattened positions
  unsigned cnt[GM_WAVE];             // per-edge match counts (diamond)
  int qkey[kQueueCap];               // candidate queue of the filtered pass: keys that passed the bit filter
  unsigned char qown[kQueueCap];     // ... and the batch lane (edge) they belong to
};


// the same scratch without the per-edge counters (kernels whose match handler does not use L.cnt): 2496 B per wave
struct alignas(16) WaveLdsLean {
  int4 desc[GM_WAVE];
  unsigned char marks[kMarkWindow];
  int qkey[kQueueCap];
  unsigned char qown[kQueueCap];
};

// ... with the STREAM INDEX of every queued candidate beside its key (the re-hosted k-clique build, gm_cbuild.hip: a match found while
// the owner's list N+(u) is streamed against a staged N+(v) is bit `stream index` of u's row) and one word of per-edge metadata
struct alignas(16) WaveLdsIdx {
  int4 desc[GM_WAVE];
  unsigned char marks[kMarkWindow];
  int qkey[kQueueCap];
  unsigned char qown[kQueueCap];
  unsigned short qidx[kQueueCap];
  int meta[GM_WAVE];
};

struct Acc {
  unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
};

__device__ __forceinline__ int bitlen(int x) { return 32 - __clz(x); }

// Filter hash of (neighbour id, row salt) -> bit index. 24-bit multiplicative hash (v_mul_u32_u24 is full rate).
template <int FL2 = kFilterLog2>
__device__ __forceinline__ unsigned filter_hash(int x, unsigned salt) {
  // (v_mul_u32_u24 takes bits 23..0 of the id: ids that differ only above bit 23 -- graphs beyond 16 M vertices -- share a filter
  // bit, a false positive the exact search behind the filter rejects; folding the top byte in cost one VALU per key)
  return ((__umul24((unsigned)x, 0x9E3779u) >> (32 - FL2)) ^ salt) & (unsigned)((1u << FL2) - 1u);
}
// SPLIT chunks of the symmetric-graph patterns: the 12 KB stage is idle (the row does not fit), so 8 KB of it hold a 2^16-bit
// hashed filter of the WHOLE hub row; only the keys that pass it are verified against the row's dense bitmap in HBM.
#ifndef GM_SPLIT_FL2
#define GM_SPLIT_FL2 16
#endif
constexpr int kSplitFilterLog2 = GM_SPLIT_FL2;
// (a second, independent 2^15-bit filter in the stage's last 4 KB was measured: diamond R-MAT-22 28.2 -> 26.2 ms, 3-motif 460 -> 449 ms,
// but diamond R-MAT-20 10.3 -> 11.3 ms -- not kept)
constexpr int kSplitFilterMaxRow = 1 << 16;  // longer rows: the filter would pass > 60 % -- probe the bitmap directly
__device__ __forceinline__ unsigned filter_salt(int local_row) { return ((unsigned)local_row * 0x2545u) & (unsigned)(kFilterBits - 1); }

// One flattened pass over the 64 edges of a batch.
//   llen       lookup-list length of this lane's edge (0 = edge not in this pass)
//   key_base   index into col[] of the lookup list
//   s_base     search list: index into L.stage (SLDS) or col[] (!SLDS)
//   s_len_flag search-list length | flag << 30
// act(found, owner_lane, key_index, pos_in_search_list, flag)
//
// Latency hiding: kTiles tiles (64 positions each) are resolved together, so kTiles independent
// key loads and kTiles independent bisection chains are in flight per wave. The bisection is the
// branch-free "binary lifting" form with a wave-uniform trip count (bit length of the longest
// search list of the pass): no exec-mask juggling, only v_cmp/v_cndmask and one load per step.
constexpr int kTiles = GM_TILES;  // filtered LDS pass (X): tiles resolved together per wave
static_assert(GM_TILES == 2 || GM_TILES == 4, "the mark window / queue arithmetic assumes a power-of-two tile group (3 miscounts)");
constexpr int kTilesG = 4;        // passes that bisect / probe in HBM (Y, SPLIT chunks): more loads in flight pay off

// MODE: how membership of a key in the search list is decided
//   SEARCH_HBM   bisect the sorted list in global memory (col[s_base ..))
//   SEARCH_LDS   bisect the LDS-staged copy (stage[s_base ..))
//   SEARCH_BITMAP one probe of the row's dense bitmap over vertex ids (hub rows of SPLIT chunks); s_base then
//                 carries the exclusive upper bound of the admissible keys (prefix bound of 3-motif, else INT_MAX)
//   SEARCH_BITMAP_ROW one probe of the dense bitmap of the SEARCHED row, which differs per edge: the slot travels in the
//                 length field of the descriptor, bm = base of all bitmaps, bm_words = words per bitmap
//   SEARCH_NONE  no search: act(in_range, owner, key_index, 0, flag, key) for every key of every lookup list
enum : int { SEARCH_HBM = 0, SEARCH_LDS = 1, SEARCH_BITMAP = 2, SEARCH_NONE = 3, SEARCH_BITMAP_ROW = 4 };

template <int MODE, class LT, class Act>
__device__ __forceinline__ void flat_pass(LT &L, const int *__restrict__ stage, const int *__restrict__ col,
                                          const unsigned *__restrict__ bm, const int lane, const int llen,
                                          const int key_base, const int s_base, const int s_len_flag, Act act,
                                          const unsigned long long bm_words = 0) {
  constexpr bool SLDS = (MODE == SEARCH_LDS);
  const int incl = wave_incl_scan_add(llen);
  const int total = readlane(incl, GM_WAVE - 1);
  if (total == 0) return;  // wave-uniform
  const int off = incl - llen;
  const int steps = bitlen(wave_max_nonneg(llen > 0 ? (s_len_flag & 0x3fffffff) : 0));
  L.desc[lane] = make_int4(key_base, off, s_base, s_len_flag);
  unsigned *m32 = reinterpret_cast<unsigned *>(L.marks);
  int carry = 0;
  for (int wb = 0; wb < total; wb += kMarkWindow) {
    const int wn = min(kMarkWindow, total - wb);
    const int nwords = ((wn + GM_WAVE * kTilesG - 1) / (GM_WAVE * kTilesG)) * (GM_WAVE * kTilesG / 4);
    for (int i = lane; i < nwords; i += GM_WAVE) m32[i] = 0u;
    wave_sync();
    if (llen > 0 && off >= wb && off < wb + kMarkWindow) L.marks[off - wb] = (unsigned char)(lane + 1);
    wave_sync();
    for (int t = 0; t < wn; t += GM_WAVE * kTilesG) {
      int own[kTilesG], key[kTilesG], kidx[kTilesG], sb[kTilesG], sl[kTilesG], fl[kTilesG], lo[kTilesG];
      bool in[kTilesG];
#pragma unroll
      for (int q = 0; q < kTilesG; ++q) own[q] = (int)L.marks[t + q * GM_WAVE + lane];
#pragma unroll
      for (int q = 0; q < kTilesG; ++q) {
        own[q] = max(wave_incl_scan_max(own[q]), carry);
        carry = readlane(own[q], GM_WAVE - 1);
      }
#pragma unroll
      for (int q = 0; q < kTilesG; ++q) {
        const int p = wb + t + q * GM_WAVE + lane;
        in[q] = p < total;
        const int4 d = L.desc[in[q] ? own[q] - 1 : 0];
        kidx[q] = p - d.y;
        sb[q] = in[q] ? d.z : 0;
        sl[q] = in[q] ? (d.w & 0x3fffffff) : 0;
        fl[q] = d.w >> 30;
        key[q] = col[in[q] ? d.x + kidx[q] : 0];  // unconditional load (select on the index): keeps vmcnt countable
        lo[q] = 0;
      }
      if constexpr (MODE == SEARCH_NONE) {
        act(in, key, own);  // the whole tile group at once: act(const bool in[], const int key[], const int own[] /* lane + 1 */)
        continue;
      } else {
      if (MODE == SEARCH_BITMAP || MODE == SEARCH_BITMAP_ROW) {
        unsigned wv[kTilesG];
#pragma unroll
        for (int q = 0; q < kTilesG; ++q)
          wv[q] = (MODE == SEARCH_BITMAP) ? bm[(unsigned)key[q] >> 5]
                                          : bm[(size_t)sl[q] * (size_t)bm_words + ((unsigned)key[q] >> 5)];
#pragma unroll
        for (int q = 0; q < kTilesG; ++q) {
          const bool f = in[q] & (((wv[q] >> ((unsigned)key[q] & 31u)) & 1u) != 0u) & (key[q] < sb[q]);
          act(f, own[q] - 1, kidx[q], 0, fl[q], key[q]);
        }
        continue;
      }
      // lower_bound by binary lifting: lo = #elements < key.  Written with non-short-circuit '&' and
      // always-executed loads on purpose: with '&&' the compiler sinks each load under its range
      // test and serialises the kTilesG chains behind s_waitcnt vmcnt(0).
      for (int s = steps - 1; s >= 0; --s) {
        int x[kTilesG], mid[kTilesG];
#pragma unroll
        for (int q = 0; q < kTilesG; ++q) {
          mid[q] = lo[q] + (1 << s);
          if (SLDS) {
            x[q] = stage[sb[q] + mid[q] - 1];  // may read past the list (never past LDS): masked below
          } else {
            x[q] = col[sb[q] + max(min(mid[q], sl[q]) - 1, 0)];
          }
        }
#pragma unroll
        for (int q = 0; q < kTilesG; ++q) {
          const bool take = (mid[q] <= sl[q]) & (x[q] < key[q]);
          lo[q] = take ? mid[q] : lo[q];
        }
      }
      int xf[kTilesG];
#pragma unroll
      for (int q = 0; q < kTilesG; ++q) {
        if (SLDS) xf[q] = stage[sb[q] + lo[q]];
        else xf[q] = col[sb[q] + max(min(lo[q], sl[q] - 1), 0)];
      }
#pragma unroll
      for (int q = 0; q < kTilesG; ++q) {
        const bool f = in[q] & (lo[q] < sl[q]) & (xf[q] == key[q]);
        act(f, own[q] - 1, kidx[q], lo[q], fl[q], key[q]);
      }
      }  // MODE != SEARCH_NONE
    }
    wave_sync();
  }
}

// Exact bisection of up to kTiles*64 queued candidates (key, edge) against the LDS-staged rows.
// Queue slots >= own_from belong to edge cur_owner (the long list being streamed: its owner is wave-uniform and is
// not written per candidate); slots below carry their owner in L.qown.
// BM: the candidates are verified with one probe of the dense bitmap `bm` of the (single) searched row instead of a
// bisection of the LDS stage (SPLIT chunks; positions are not reported, act gets pos = 0).
template <bool BM, bool IDX = false, class LT, class Act>
__device__ __forceinline__ void drain_candidates(LT &L, const int *__restrict__ stage, const int lane, const int n,
                                                 const int steps, const int own_from, const int cur_owner, Act act,
                                                 const unsigned *__restrict__ bm = nullptr) {
  int own[kTiles], key[kTiles], sb[kTiles], sl[kTiles], fl[kTiles], lo[kTiles], sidx[kTiles];
  bool in[kTiles];
#pragma unroll
  for (int q = 0; q < kTiles; ++q) {
    const int slot = q * GM_WAVE + lane;  // < kQueueCap: the reads below are always in bounds
    in[q] = slot < n;
    key[q] = L.qkey[slot];
    sidx[q] = 0;
    if constexpr (IDX) sidx[q] = (int)L.qidx[slot];
    own[q] = (int)L.qown[slot];
    own[q] = (slot >= own_from) ? cur_owner : own[q];
    own[q] = in[q] ? own[q] : 0;
  }
  if constexpr (BM) {
    unsigned wv[kTiles];
#pragma unroll
    for (int q = 0; q < kTiles; ++q) wv[q] = bm[(unsigned)(in[q] ? key[q] : 0) >> 5];  // unconditional loads
#pragma unroll
    for (int q = 0; q < kTiles; ++q) {
      const bool f = in[q] & (((wv[q] >> ((unsigned)key[q] & 31u)) & 1u) != 0u);
      act(f, own[q], 0, 0, L.desc[own[q]].w >> 30, key[q]);
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < kTiles; ++q) {
    const int4 d = L.desc[own[q]];
    sb[q] = d.z & 0xffff;
    sl[q] = in[q] ? (d.w & 0x3fffffff) : 0;
    fl[q] = d.w >> 30;
    lo[q] = 0;
  }
  for (int s = steps - 1; s >= 0; --s) {
    int x[kTiles], mid[kTiles];
#pragma unroll
    for (int q = 0; q < kTiles; ++q) {
      mid[q] = lo[q] + (1 << s);
      x[q] = stage[sb[q] + mid[q] - 1];
    }
#pragma unroll
    for (int q = 0; q < kTiles; ++q) {
      const bool take = (mid[q] <= sl[q]) & (x[q] < key[q]);
      lo[q] = take ? mid[q] : lo[q];
    }
  }
  int xf[kTiles];
#pragma unroll
  for (int q = 0; q < kTiles; ++q) xf[q] = stage[sb[q] + lo[q]];
#pragma unroll
  for (int q = 0; q < kTiles; ++q) {
    const bool f = in[q] & (lo[q] < sl[q]) & (xf[q] == key[q]);
    act(f, own[q], sidx[q], lo[q], fl[q], key[q]);  // (sidx: the candidate's index in its streamed list when IDX, else 0)
  }
}

// Pass X on a staged chunk with the hashed bit filter in front of the bisection. ~90 % of the streamed keys are
// not in the row they are tested against: they cost one hash + one ds_read_b32 here. The survivors are
// compacted (ballot + v_mbcnt) into a per-wave LDS queue and bisected 64..256 at a time with all lanes busy.
//   s_base carries the row's filter salt in bits 16..29.
// Two regimes share the queue:
//   * LONG lookup lists (>= kLongList keys) are streamed one edge at a time with a wave-uniform descriptor
//     (scalar base address, scalar salt): no owner marks, no scans -- this is where skewed graphs spend their time;
//   * the remaining short lists of the batch are flattened (owner marks + DPP max-scan), with a fast path for
//     tiles that contain no list boundary.
// (kLongList: gm_mine.h -- the setup of the edge supports' match masks sizes them by it)

template <int FL2 = kFilterLog2, bool BM = false, bool IDX = false, class LT, class Act>
__device__ __forceinline__ void flat_pass_filtered(LT &L, const int *__restrict__ stage, const unsigned *__restrict__ fbits,
                                                   const int *__restrict__ col, const int lane, const int llen_all,
                                                   const int key_base, const int s_base_salt, const int s_len_flag, const int dbg, Act act,
                                                   const unsigned *__restrict__ bm = nullptr) {
  if (wave_max_nonneg(llen_all) == 0) return;  // wave-uniform
  const int steps = bitlen(wave_max_nonneg(llen_all > 0 ? (s_len_flag & 0x3fffffff) : 0));
  const bool is_long = llen_all >= kLongList;
  const int llen = is_long ? 0 : llen_all;
  const int incl = wave_incl_scan_add(llen);
  const int total = readlane(incl, GM_WAVE - 1);
  const int off = incl - llen;
  L.desc[lane] = make_int4(key_base, off, s_base_salt, s_len_flag);
  int qcount = 0;  // wave-uniform number of queued candidates

  auto enqueue = [&](const bool cand, const int key, const int owner, const int sidx) {
    const unsigned long long m = __ballot(cand);
    if (__builtin_amdgcn_inverse_ballot_w64(m)) {  // (exec = the ballot: `if (cand)` made the compiler evaluate the bit test twice)
      const int slot = qcount + rank_below(m);
      L.qkey[slot] = key;
      L.qown[slot] = (unsigned char)owner;
      if constexpr (IDX) L.qidx[slot] = (unsigned short)sidx;
    }
    qcount += __popcll(m);
  };
  int own_from = kQueueCap, cur_owner = 0;  // wave-uniform: slots >= own_from belong to the long list being streamed
  auto enqueue_long = [&](const bool cand, const int key, const int sidx) {
    const unsigned long long m = __ballot(cand);
    if (__builtin_amdgcn_inverse_ballot_w64(m)) {
      const int slot = qcount + rank_below(m);
      L.qkey[slot] = key;
      if constexpr (IDX) L.qidx[slot] = (unsigned short)sidx;
    }
    qcount += __popcll(m);
  };
  auto drain_full_tiles = [&]() {
    if (qcount >= GM_WAVE) {  // wave-uniform
      wave_sync();
      const int n = qcount & ~(GM_WAVE - 1);
      if (!(dbg & 16)) drain_candidates<BM, IDX>(L, stage, lane, n, steps, own_from, cur_owner, act, bm);
      const int rest = qcount - n;  // < 64: move to the front, owners written out
      int k = 0;
      unsigned char o = 0;
      unsigned short si = 0;
      if (lane < rest) {
        k = L.qkey[n + lane];
        o = (n + lane >= own_from) ? (unsigned char)cur_owner : L.qown[n + lane];
        if constexpr (IDX) si = L.qidx[n + lane];
      }
      wave_sync();
      if (lane < rest) {
        L.qkey[lane] = k;
        L.qown[lane] = o;
        if constexpr (IDX) L.qidx[lane] = si;
      }
      qcount = rest;
      own_from = (own_from < kQueueCap) ? rest : kQueueCap;
    }
  };

  // ---- long lists: one edge at a time, wave-uniform descriptor -----------------------------------------
  unsigned long long lm = __ballot(is_long);
  while (lm) {
    const int src = __ffsll((long long)lm) - 1;
    lm &= lm - 1;
    const int base = readlane(key_base, src);
    const int n = readlane(llen_all, src);
    const unsigned salt = (unsigned)readlane(s_base_salt, src) >> 16;
    const int *__restrict__ kp = col + base;
    own_from = qcount;  // (< 64 queued candidates of earlier lists keep their written owners)
    cur_owner = src;
    // Software pipeline: the keys of the NEXT tile group are requested before the current group is hashed / filtered /
    // queued. (A deeper pipeline was measured and does not pay: 1 group ahead 11.16 ms, 2: 11.22, 3: 11.31, 4: 11.50.)
    // Loads are UNCONDITIONAL: a predicated load sits in its own exec-masked block, the compiler then cannot count how
    // many younger loads are in flight and waits with vmcnt(0) -- which also waits for the prefetch it has just issued.
    // The steady-state loop only sees FULL groups whose successor is full too: no range tests, no index clamps, and the
    // load address is scalar base + (lane * 4); the last one or two groups take the general form.
    auto process = [&](const int *key, const bool *in, const int t_base) {
      unsigned h[kTiles], fw[kTiles];
#pragma unroll
      for (int q = 0; q < kTiles; ++q) {
        h[q] = filter_hash<FL2>(key[q], salt);
        fw[q] = fbits[h[q] >> 5];
      }
#pragma unroll
      for (int q = 0; q < kTiles; ++q) enqueue_long(in[q] & (((fw[q] >> (h[q] & 31u)) & 1u) != 0u), key[q], t_base + q * GM_WAVE + lane);
      drain_full_tiles();
    };
    constexpr int G = GM_WAVE * kTiles;
    int nxt[kTiles];
#pragma unroll
    for (int q = 0; q < kTiles; ++q) nxt[q] = kp[min(q * GM_WAVE + lane, n - 1)];
    int t = 0;
    for (; t + 2 * G <= n; t += G) {
      int key[kTiles];
      bool in[kTiles];
#pragma unroll
      for (int q = 0; q < kTiles; ++q) {
        key[q] = nxt[q];
        in[q] = true;
      }
      const int *__restrict__ kn = kp + (t + G);  // wave-uniform
#pragma unroll
      for (int q = 0; q < kTiles; ++q) nxt[q] = kn[(unsigned)(q * GM_WAVE + lane)];
      process(key, in, t);
    }
    for (; t < n; t += G) {
      int key[kTiles];
      bool in[kTiles];
#pragma unroll
      for (int q = 0; q < kTiles; ++q) {
        key[q] = nxt[q];
        in[q] = (t + q * GM_WAVE + lane) < n;
      }
#pragma unroll
      for (int q = 0; q < kTiles; ++q) nxt[q] = kp[min(t + G + q * GM_WAVE + lane, n - 1)];
      process(key, in, t);
    }
    // the list is done: write the owner of what is still queued (< 64 entries), later candidates carry their own
    if (own_from + lane < qcount) L.qown[own_from + lane] = (unsigned char)src;
    own_from = kQueueCap;
  }

  // ---- short lists: flattened ---------------------------------------------------------------------------
  unsigned *m32 = reinterpret_cast<unsigned *>(L.marks);
  int carry = 0;
  for (int wb = 0; wb < total; wb += kMarkWindow) {
    const int wn = min(kMarkWindow, total - wb);
    const int nwords = ((wn + GM_WAVE * kTiles - 1) / (GM_WAVE * kTiles)) * (GM_WAVE * kTiles / 4);
    for (int i = lane; i < nwords; i += GM_WAVE) m32[i] = 0u;
    wave_sync();
    if (llen > 0 && off >= wb && off < wb + kMarkWindow) L.marks[off - wb] = (unsigned char)(lane + 1);
    wave_sync();
    for (int t = 0; t < wn; t += GM_WAVE * kTiles) {
      int own[kTiles], key[kTiles], sidx[kTiles];
      unsigned h[kTiles], fw[kTiles];
      bool in[kTiles];
#pragma unroll
      for (int q = 0; q < kTiles; ++q) own[q] = (int)L.marks[t + q * GM_WAVE + lane];
#pragma unroll
      for (int q = 0; q < kTiles; ++q) {
        if (__ballot(own[q] != 0) == 0ull) {
          own[q] = carry;  // no list starts inside this tile: every position belongs to the running owner
        } else {
          own[q] = max(wave_incl_scan_max(own[q]), carry);
          carry = readlane(own[q], GM_WAVE - 1);
        }
      }
#pragma unroll
      for (int q = 0; q < kTiles; ++q) {
        const int p = wb + t + q * GM_WAVE + lane;
        in[q] = p < total;
        const int4 d = L.desc[in[q] ? own[q] - 1 : 0];
        sidx[q] = p - d.y;
        key[q] = col[in[q] ? d.x + sidx[q] : 0];  // unconditional load (select on the index)
        h[q] = filter_hash<FL2>(key[q], (unsigned)d.z >> 16);
      }
#pragma unroll
      for (int q = 0; q < kTiles; ++q) fw[q] = (dbg & 32) ? 0u : fbits[h[q] >> 5];
#pragma unroll
      for (int q = 0; q < kTiles; ++q) enqueue(in[q] & (((fw[q] >> (h[q] & 31u)) & 1u) != 0u), key[q], own[q] - 1, sidx[q]);
      drain_full_tiles();
    }
    wave_sync();
  }
  if (qcount > 0) {
    wave_sync();
    drain_candidates<BM, IDX>(L, stage, lane, qcount, steps, kQueueCap, 0, act, bm);
  }
  wave_sync();
}

}  // namespace gm
