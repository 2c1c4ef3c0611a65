// gm_hset.h -- the rows of a task chunk as ONE HASHED SET of (local row, id) -> position in LDS, shared by the kernels that need the
// POSITION of a match in the host's row (gm_cbuild.hip: the bit of a k-clique matrix row; gm_sup.hip: the DAG entry of an edge).
// (gm_tch.hip keeps a simpler table of full ids: a triangle count needs membership only.)
//   * h = id * C mod 2^32 is a bijection of the ids; its top LB bits (XOR a salt that is injective in the local row, so that a hub id
//     staged for a hundred rows of the chunk spreads over a hundred buckets) pick one of 2^LB = STAGE buckets of four 32-bit slots, and
//     a slot holds the REMAINING 32 - LB bits of h above the entry's INDEX IN THE STAGE (LB bits: a chunk has <= STAGE entries);
//   * lookup of id x in row r = stage entries [lo, lo + len): XOR the four slots with h(x) << LB -- a slot with the same remainder
//     becomes its stage index, every other a number >= STAGE -- subtract lo, take the unsigned minimum: a hit is a minimum < len, and
//     it is the POSITION of x in the row.  Exact: inside one row the same bucket and the same remainder are the same h, hence the
//     same id; an entry of another row that happens to share bucket and remainder (its salt cancels the difference of the top bits)
//     has a stage index outside [lo, lo + len);
//   * empty = 0xffffffff and overflow marker = 0xfffffffe would read as stage indices STAGE - 1 / STAGE - 2 of some id: entries at
//     those indices (chunks of >= STAGE - 1 entries) live in the surplus list and their bucket carries the marker;
//   * a bucket that got more than four entries (0.4 % of them at this load) carries the marker in its last slot; its surplus entries
//     sit in a list of <= 128 (id, salt, position) triples that only the lanes missing in such a bucket consult.  A chunk that
//     overflows the list (adversarial ids) is looked up by bisection of the row in global memory -- slow, exact.
#pragma once
#include "gm_flat.h"

namespace gm {

constexpr int kHsOvfCap = 128;
constexpr unsigned kHsEmpty = 0xffffffffu, kHsMarker = 0xfffffffeu;
constexpr unsigned kHsMul = 0x9E3779B1u;
static_assert(kHsOvfCap == 2 * GM_WAVE, "the surplus list is scanned two entries per lane");

template <int STAGE>
struct HsHash {
  static constexpr int LB = STAGE == 1024 ? 10 : 11;
  static_assert((1 << LB) == STAGE, "one bucket per stage entry");
  static constexpr unsigned kMask = (unsigned)(STAGE - 1) << 4;
  static constexpr unsigned kPosLimit = (unsigned)STAGE - 2u;  // stage indices kept in the table (see above)
  static __device__ __forceinline__ unsigned hash(int x) { return (unsigned)x * kHsMul; }
  static __device__ __forceinline__ unsigned bucket(unsigned h, unsigned s) { return ((h >> (28 - LB)) & kMask) ^ s; }  // byte offset
  static __device__ __forceinline__ unsigned salt(int local_row) { return (((unsigned)local_row * 37u) & (unsigned)(STAGE - 1)) << 4; }
  static __device__ __forceinline__ int row_of(unsigned s) { return (int)(((s >> 4) * 941u) & (unsigned)(STAGE - 1)); }  // 37 * 941 = 1 mod 2048
};

// the part of a kernel's LDS block the set lives in
template <int STAGE>
struct alignas(16) HsTable {
  uint4 table[STAGE];              // buckets of four (hash remainder, position) slots
  int rpl[kMaxChunkVerts + 1];     // row offsets of the chunk's DAG rows (global entry indices)
  int ovf_key[kHsOvfCap];
  int ovf_salt[kHsOvfCap];
  int ovf_pos[kHsOvfCap];          // (position in the row)
  int n_ovf;
  int pad_[2];
};

__device__ __forceinline__ int hs_local_row(const int *rpl, const int nvl, const int e) {  // largest i with rpl[i] <= e
  int lo = 0, hi = nvl - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rpl[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Build the set of the chunk's entries col[eb .. eb + nel) (rows rp[ub .. ub + nvl]); all NT threads of the workgroup.  fill32: STAGE / 2
// words of scratch (packed 16-bit fill counters of the buckets).  Ends with a barrier; returns "the set is not usable" (surplus list
// overflowed, or the test switch): every lookup then bisects its row in global memory.
// Two passes over the entries, which stay in registers in between: (1) every entry takes a number in its bucket; (2) when the bucket's
// total is known, numbers 0..3 (0..2 if the bucket overflowed: its last slot holds the marker) go to the table, the rest to the surplus
// list.  (An entry at one of the two positions the table cannot hold counts as five and writes the marker itself.)
// task_rows (optional, LDS: nvl + 1 offsets): only the rows i with task_rows[i + 1] > task_rows[i] are staged -- nobody looks a key up in
// a row that hosts no task (round 6: a rank of eight finds ~70 tasks in a chunk of 2048 entries and staged all of them, in every chunk).
template <int STAGE, int NT>
__device__ __forceinline__ bool hs_build(HsTable<STAGE> &S, unsigned *fill32, const int *__restrict__ rp, const int *__restrict__ col,
                                         const int ub, const int nvl, const int eb, const int nel, const bool force_fallback, const int tid,
                                         const int *task_rows = nullptr) {
  using H = HsHash<STAGE>;
  for (int i = tid; i <= nvl; i += NT) S.rpl[i] = rp[ub + i];
  {
    const uint4 empty = make_uint4(kHsEmpty, kHsEmpty, kHsEmpty, kHsEmpty);
    for (int i = tid; i < STAGE; i += NT) S.table[i] = empty;
    for (int i = tid; i < STAGE / 2; i += NT) fill32[i] = 0u;
    if (tid < kHsOvfCap) {
      S.ovf_key[tid] = -1;
      S.ovf_salt[tid] = -1;
      S.ovf_pos[tid] = 0;
    }
    if (tid == 0) S.n_ovf = 0;
  }
  __syncthreads();
  unsigned *slots = reinterpret_cast<unsigned *>(S.table);
  constexpr int kU = 4, kIt = (STAGE + kU * NT - 1) / (kU * NT);  // entries requested together per thread; trips of a thread
  int xv[kIt][kU];
  unsigned pk[kIt][kU];  // local row | number in the bucket (capped at 15) << 8
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i0 = it * kU * NT + tid;
#pragma unroll
    for (int j = 0; j < kU; ++j) xv[it][j] = (i0 < nel) ? col[eb + min(i0 + j * NT, nel - 1)] : 0;
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int i = i0 + j * NT;
      pk[it][j] = 0xffffffffu;  // (not staged)
      if (i < nel) {
        const int lo = hs_local_row(S.rpl, nvl, eb + i);
        if (task_rows != nullptr && task_rows[lo + 1] == task_rows[lo]) continue;  // the row hosts nothing
        const unsigned b = H::bucket(H::hash(xv[it][j]), H::salt(lo)) >> 4;
        const unsigned shift = (b & 1u) * 16u;
        const unsigned num = (atomicAdd(&fill32[b >> 1], ((unsigned)i < H::kPosLimit ? 1u : 5u) << shift) >> shift) & 0xffffu;
        pk[it][j] = (unsigned)lo | (min(num, 15u) << 8);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int i = it * kU * NT + j * NT + tid;
      if (i < nel && pk[it][j] != 0xffffffffu) {
        const unsigned lo = pk[it][j] & 255u, num = pk[it][j] >> 8;
        const unsigned s = H::salt((int)lo), h = H::hash(xv[it][j]);
        const unsigned b = H::bucket(h, s) >> 4;
        const unsigned c = (fill32[b >> 1] >> ((b & 1u) * 16u)) & 0xffffu;
        const bool holds = (unsigned)i < H::kPosLimit;
        const unsigned at = (unsigned)(eb + i - S.rpl[lo]);  // position in its row
        if (holds && num < (c > 4u ? 3u : 4u)) {
          slots[(b << 2) + num] = (h << H::LB) | (unsigned)i;
        } else {
          if (num == 3u || !holds) slots[(b << 2) + 3] = kHsMarker;
          const int jo = atomicAdd(&S.n_ovf, 1);
          if (jo < kHsOvfCap) {
            S.ovf_key[jo] = xv[it][j];
            S.ovf_salt[jo] = (int)s;
            S.ovf_pos[jo] = (int)at;
          }
        }
      }
    }
  }
  __syncthreads();  // (also: the fill counters are dead)
  return S.n_ovf > kHsOvfCap || force_fallback;
}

// Where in the row their salt names -- stage entries [rlo, rlo + rlen) -- are the T keys of this lane?  hm = lanes whose key was found
// (at[q] = its position in the row), nm = lanes that missed in a bucket that overflowed (the surplus list decides: hs_surplus).
template <int STAGE, int T>
__device__ __forceinline__ void hs_probe(const HsTable<STAGE> &S, const int *__restrict__ col, const bool fallback, const int (&key)[T],
                                         const unsigned (&salt)[T], const unsigned (&rlo)[T], const unsigned (&rlen)[T],
                                         const unsigned long long (&inm)[T], unsigned (&at)[T], unsigned long long (&hm)[T],
                                         unsigned long long (&nm)[T]) {
  using H = HsHash<STAGE>;
  if (fallback) {  // wave-uniform
#pragma unroll
    for (int q = 0; q < T; ++q) {
      bool f = false;
      at[q] = 0u;
      if (__builtin_amdgcn_inverse_ballot_w64(inm[q])) {
        const int lo = H::row_of(salt[q]);
        const int rs = S.rpl[lo], rn = S.rpl[lo + 1] - rs;
        const int p = lower_bound(col + rs, rn, key[q]);
        f = p < rn && col[rs + p] == key[q];
        at[q] = (unsigned)p;
      }
      hm[q] = __ballot(f);
      nm[q] = 0ull;
    }
    return;
  }
  uint4 w[T];
  unsigned t[T];
#pragma unroll
  for (int q = 0; q < T; ++q) {
    const unsigned h = H::hash(key[q]);
    t[q] = h << H::LB;
    w[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(S.table) + H::bucket(h, salt[q]));
  }
#pragma unroll
  for (int q = 0; q < T; ++q) {
    at[q] = min(min((w[q].x ^ t[q]) - rlo[q], (w[q].y ^ t[q]) - rlo[q]), min((w[q].z ^ t[q]) - rlo[q], (w[q].w ^ t[q]) - rlo[q]));
    const unsigned long long m = __ballot(at[q] < rlen[q]);
    hm[q] = m & inm[q];
    nm[q] = __ballot(w[q].w == kHsMarker) & ~m & inm[q];
  }
}

// The surplus list is spread over the lanes, two entries each; a key that needs it is broadcast and compared by all lanes at once.
// act(q, src lane, position) for every key found (wave-uniform arguments).
template <int STAGE, int T, class Act>
__device__ __forceinline__ void hs_surplus(const HsTable<STAGE> &S, const int lane, const int (&key)[T], const unsigned (&salt)[T],
                                           const unsigned long long (&nm)[T], Act act) {
  const int k0 = S.ovf_key[lane], k1 = S.ovf_key[lane + GM_WAVE];
  const int s0 = S.ovf_salt[lane], s1 = S.ovf_salt[lane + GM_WAVE];
  const int p0 = S.ovf_pos[lane], p1 = S.ovf_pos[lane + GM_WAVE];
#pragma unroll
  for (int q = 0; q < T; ++q) {
    unsigned long long rest = nm[q];
    while (rest) {
      const int src = __ffsll((long long)rest) - 1;
      rest &= rest - 1;
      const int k = readlane(key[q], src), sv = readlane((int)salt[q], src);
      const bool e0 = (k0 == k) & (s0 == sv), e1 = (k1 == k) & (s1 == sv);
      const unsigned long long em = __ballot(e0 | e1);
      if (em != 0ull) act(q, src, readlane(e0 ? p0 : p1, __ffsll((long long)em) - 1));
    }
  }
}

// per-wave scratch of the flattened pass of the kernels built on the set
// BITWIN: flattened positions per mark window (a batch of 64 lists below kLongList keys has <= 12224, most have far fewer)
template <int BITWIN>
struct alignas(16) HsWaveT {
  int4 desc[GM_WAVE];              // per non-empty list of the batch: {key_base - offset among the flattened positions, salt, the kernel's two words}
  int2 rng[GM_WAVE];               // ... and where its host row sits in the stage: {first entry, entries}
  unsigned bits[BITWIN / 32 + 8];  // list-start marks of the flattened positions, one bit each (+ the over-read of the last group)
};
template <int STAGE>
using HsWave = HsWaveT<4096>;

// One batch of tasks against the set: lane's list = llen_all keys from col[key_base ..) (0 = no task), looked up in the row `salt_l`
// names = stage entries [rlo_l, rlo_l + rlen_l); word_l, word2_l = two per-task words handed back to the match handler.
//   hit(hm, word, word2, at, kidx, uniform): the lanes of hm found their key -- entry kidx of col -- at position `at` of the host row;
//     uniform (compile-time at every call site): the tile belongs to ONE task, word / word2 are wave-uniform (long lists)
//   hit1(word, word2, at, kidx): one key found through the surplus list (wave-uniform arguments)
template <int STAGE, int T, class Hit, class Hit1>
__device__ __forceinline__ void hs_pass(const HsTable<STAGE> &S, HsWave<STAGE> &L, const int *__restrict__ col, const bool fallback, const int lane,
                                        const int llen_all, const int key_base, const unsigned salt_l, const int rlo_l, const int rlen_in,
                                        const int word_l, const int word2_l, Hit hit, Hit1 hit1) {
  if (wave_max_nonneg(llen_all) == 0) return;  // wave-uniform
  // The table holds no entry at a stage index >= kPosLimit (those live in the surplus list), but an EMPTY slot / the overflow MARKER
  // decode to exactly those two indices for every key whose hash has its low 32 - LB bits all ones (ids such as 3461295 for LB = 10):
  // a row that covers index STAGE - 1 or STAGE - 2 must not accept them.  Clamped once per task: a table hit is only looked for
  // below kPosLimit; true entries at the two indices are found through their bucket's marker + the surplus list, which does not use rlen.
  const int rlen_l = min(rlen_in, max(0, (int)HsHash<STAGE>::kPosLimit - rlo_l));
  const bool is_long = llen_all >= kLongList;
  const int llen = is_long ? 0 : llen_all;
  // ---- long lists: one task at a time, wave-uniform base / salt / word; the keys of the NEXT tile group are requested before the
  // current group is looked up (unconditional, unclamped loads in the steady state) -----------------------------------------------------
  unsigned long long lm = __ballot(is_long);
  while (lm) {
    const int src = __ffsll((long long)lm) - 1;
    lm &= lm - 1;
    const int base = readlane(key_base, src);
    const int n = readlane(llen_all, src);
    const unsigned s_u = (unsigned)readlane((int)salt_l, src);
    const int w_u = readlane(word_l, src), w2_u = readlane(word2_l, src);
    const unsigned rlo_u = (unsigned)readlane(rlo_l, src), rlen_u = (unsigned)readlane(rlen_l, src);
    const int *__restrict__ kp = col + base;
    auto process = [&](const int (&key)[T], const unsigned long long (&inm)[T], const int t_base) {
      unsigned salt[T], at[T], rlo[T], rlen[T];
      int kidx[T];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        salt[q] = s_u;
        rlo[q] = rlo_u;
        rlen[q] = rlen_u;
        kidx[q] = base + t_base + q * GM_WAVE + lane;
      }
      unsigned long long hm[T], nm[T];
      hs_probe<STAGE, T>(S, col, fallback, key, salt, rlo, rlen, inm, at, hm, nm);
      unsigned long long any_need = 0ull;
#pragma unroll
      for (int q = 0; q < T; ++q) {
        hit(hm[q], w_u, w2_u, at[q], kidx[q], true);
        any_need |= nm[q];
      }
      if (any_need != 0ull)  // rare
        hs_surplus<STAGE, T>(S, lane, key, salt, nm, [&](const int q, const int sl, const int p) { hit1(w_u, w2_u, p, readlane(kidx[q], sl)); });
    };
    constexpr int G = GM_WAVE * T;
    int nxt[T];
#pragma unroll
    for (int q = 0; q < T; ++q) nxt[q] = kp[min(q * GM_WAVE + lane, n - 1)];
    int t = 0;
    for (; t + 2 * G <= n; t += G) {
      int key[T];
      unsigned long long inm[T];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        key[q] = nxt[q];
        inm[q] = ~0ull;
      }
      const int *__restrict__ kn = kp + (t + G);
#pragma unroll
      for (int q = 0; q < T; ++q) nxt[q] = kn[(unsigned)(q * GM_WAVE + lane)];
      process(key, inm, t);
    }
    for (; t < n; t += G) {
      int key[T];
      unsigned long long inm[T];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        key[q] = nxt[q];
        inm[q] = __ballot((t + q * GM_WAVE + lane) < n);
      }
#pragma unroll
      for (int q = 0; q < T; ++q) nxt[q] = kp[min(t + G + q * GM_WAVE + lane, n - 1)];
      process(key, inm, t);
    }
  }
  // ---- short lists: flattened.  Position p of the concatenated lists belongs to the LAST list that starts at or before it: every
  // list but the first of a window leaves one mark BIT at (its start - 1), the owner of p is the number of marks below p (one LDS read
  // per tile group, v_readlane + v_mbcnt per tile); the keys of the NEXT tile group are requested before the current group is looked
  // up (gm_tch.hip, where this form was measured first) -----------------------------------------------------------------------------
  const int incl = wave_incl_scan_add(llen);
  const int total = readlane(incl, GM_WAVE - 1);
  if (total == 0) return;  // wave-uniform
  const int offp = incl - llen;
  const unsigned long long nzm = __ballot(llen > 0);
  if (llen > 0) {  // (compacted: the k-th non-empty list)
    const int li = rank_below(nzm);
    L.desc[li] = make_int4(key_base - offp, (int)salt_l, word_l, word2_l);
    L.rng[li] = make_int2(rlo_l, rlen_l);
  }
  constexpr int G = GM_WAVE * T, kBitWin = (int)(sizeof(L.bits) / 4 - 8) * 32;
  static_assert(2 * T <= GM_WAVE, "the mark words of a tile group are read by its first lanes");
  struct Grp {
    int key[T], own[T];
    unsigned long long inm[T];
  };
  for (int wb = 0; wb < total; wb += kBitWin) {
    const int wn = min(kBitWin, total - wb);
    const int nw = ((wn + G - 1) / G) * (G / 32) + 2;
    for (int i = lane; i < nw; i += GM_WAVE) L.bits[i] = 0u;
    wave_sync();
    if (llen > 0 && offp > wb && offp < wb + kBitWin) atomicOr(&L.bits[(offp - 1 - wb) >> 5], 1u << ((offp - 1 - wb) & 31));
    int carry = __popcll(__ballot(llen > 0 && offp <= wb)) - 1;  // the list that owns position wb
    wave_sync();
    auto stage_a = [&](const int g, Grp &r) {  // owners and key loads of tile group g of the window
      const int mw = (int)L.bits[((g * G) >> 5) + (lane & (2 * T - 1))];  // lane l holds mark word l of the group
      int dx[T];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const unsigned long long m = ((unsigned long long)(unsigned)readlane(mw, 2 * q + 1) << 32) | (unsigned)readlane(mw, 2 * q);
        const bool in = (wb + g * G + q * GM_WAVE + lane) < total;
        r.own[q] = in ? carry + rank_below(m) : 0;
        carry += __popcll(m);
        r.inm[q] = __ballot(in);
        dx[q] = L.desc[r.own[q]].x;  // unconditional LDS read
      }
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const int pp = wb + g * G + q * GM_WAVE + lane;
        r.key[q] = col[(r.inm[q] >> lane) & 1ull ? dx[q] + pp : 0];  // unconditional load (select on the index)
      }
    };
    auto stage_b = [&](const int g, const Grp &r) {
      int word[T], word2[T], kidx[T];
      unsigned salt[T], at[T], rlo[T], rlen[T];
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const int4 dd = L.desc[r.own[q]];
        const int2 rr = L.rng[r.own[q]];
        const int pp = wb + g * G + q * GM_WAVE + lane;
        kidx[q] = (r.inm[q] >> lane) & 1ull ? dd.x + pp : 0;
        salt[q] = (unsigned)dd.y;
        word[q] = dd.z;
        word2[q] = dd.w;
        rlo[q] = (unsigned)rr.x;
        rlen[q] = (unsigned)rr.y;
      }
      unsigned long long hm[T], nm[T];
      hs_probe<STAGE, T>(S, col, fallback, r.key, salt, rlo, rlen, r.inm, at, hm, nm);
      unsigned long long any_need = 0ull;
#pragma unroll
      for (int q = 0; q < T; ++q) {
        hit(hm[q], word[q], word2[q], at[q], kidx[q], false);
        any_need |= nm[q];
      }
      if (any_need != 0ull)  // rare
        hs_surplus<STAGE, T>(S, lane, r.key, salt, nm, [&](const int q, const int sl, const int p) {
          hit1(readlane(word[q], sl), readlane(word2[q], sl), p, readlane(kidx[q], sl));
        });
    };
    const int ng = (wn + G - 1) / G;
    Grp cur;
    stage_a(0, cur);
    for (int g = 0; g + 1 < ng; ++g) {
      Grp nxt;
      stage_a(g + 1, nxt);
      stage_b(g, cur);
      cur = nxt;
    }
    stage_b(ng - 1, cur);
    wave_sync();
  }
}

}  // namespace gm
