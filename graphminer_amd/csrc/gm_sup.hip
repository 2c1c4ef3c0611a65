// gm_sup.hip -- EDGE SUPPORTS: t(e) = |N(u) ^ N(v)| for every edge e = {u, v} of the symmetric graph, from ONE pass over the
// triangles of its DAG, and the diamond count  sum_e C(t(e), 2)  from them.
//
// The reference's diamond kernel intersects the two SYMMETRIC neighbour lists of every edge (src/sgl/gpu_kernels/diamond_count.cuh:2-19:
// one warp per edge, n = |N(v0) ^ N(v1)|, count += n (n - 1) / 2), and so did this library until round 3 (gm_hrow.hip: hub rows as
// hashed sets, the partner lists streamed: 18.6 G keys on the LiveJournal stand-in).  But |N(u) ^ N(v)| is the number of triangles
// through the edge, and every triangle {a, b, c} of the graph is found exactly ONCE by the triangle kernel on the DAG (gm_tch.hip:
// 3.4 G keys on the same graph): as a match of one task edge while the shorter of two out-lists is streamed against the longer.
// A match knows its three edges as DAG entries:
//   * the task's own edge (one entry per task: tedge[], emitted with the task lists),
//   * host row start + the POSITION of the match in the host's row  (the hashed set of gm_hset.h returns it),
//   * the index of the streamed key in col[]  (the streamed list IS a DAG row, or the tail of one),
// so the kernel is the triangle kernel with three increments per match:
//   * the host-row edge into an LDS counter per stage entry, flushed once per chunk (part);
//   * the task's own edge into an LDS counter per batch lane, flushed once per task;
//   * the streamed edge by a global atomic (return-less, 4 bytes) -- a device-scope atomic costs about one 64-byte fabric transaction
//     (measured: with two of them per match R-MAT-22 took 20.2 ms for 750 M triangles, 3.35 ms without any).  The atomics are not
//     issued where the match is found: gfx950 counts loads, stores and return-less atomics in ONE counter (vmcnt) and the compiler
//     must assume they complete out of order, so a pending atomic turns the wait for the current tile group's keys into a wait for
//     everything -- the next group's prefetched keys included.  The entries of the matches are queued in LDS (256 per wave) and
//     flushed in bursts, whose latencies overlap each other.
// MATCH MASKS (round 5).  87 % of the triangles of an R-MAT graph are found by IN-EDGE tasks (the target s_i of u -> s_i hosts, the tail
// N+(u)[i + 1 ..) is streamed): the streamed edge (u, s_j) of such a match is an entry of row u right behind the task's own entry, and the
// task needs no atomic to report it -- it stores WHICH keys of its tail matched (bit k of its mask = key k), 64 keys per plain 8-byte
// store, and sup_cols_kernel sums the masks of a row by column: entry j of row u += sum_{i < j} bit (j - i - 1) of the mask of entry i.
// Masks: the in-edge tasks with tails of >= kSupMaskMinTail keys (ensure_sup_masks, gm_tables.hip: offsets per task / per entry); long
// tails (>= kLongList keys, one task at a time) store their tiles' ballots directly, the tails of the flattened pass collect theirs in six
// LDS words per batch lane.  Out-edge tasks, short tails and the surplus-list matches of unmasked tasks keep the atomics.  One GPU, a
// topologically numbered DAG; a rank's share of a larger job (gm_diamond_support_partial) runs without masks.
// A second kernel sums C(t, 2) over the entries.
// One GPU; rows beyond the 2048-entry stage send the caller back to the per-edge kernels (gm_launch.hip).
#include "gm_hset.h"

namespace gm {

constexpr int kSupTiles = 4;
#ifndef GM_SUP_WAVES
#define GM_SUP_WAVES 4
#endif
// (round 6 tried the blocked gather's recipe here -- 12-wave workgroups, two tiles in flight, 80 registers, six waves per SIMD: diamond R-MAT-22
// 5.71 against 5.11 ms, R-MAT-24 69.6 against 66.0; four waves with two tiles 5.27: this kernel wants its four tiles in flight more than it wants
// waves.  profiles/r06/ab_diamond_sup_occupancy.txt)
constexpr int kSupWaves = GM_SUP_WAVES;  // waves of a sup_kernel workgroup
#ifndef GM_SUP_TILES_SMALL
#define GM_SUP_TILES_SMALL 4
#endif
constexpr int kSupTilesSmall = GM_SUP_TILES_SMALL;  // (1024-bucket kernel)
constexpr int kSupQueue = 128, kSupFlushAt = kSupQueue - GM_WAVE;  // (a tile adds at most 64 entries; the key stream's tiles two per match)
constexpr int kSupMaskWords = (kLongList + 31) / 32;                // 32-bit LDS words of a flattened task's mask (tails below kLongList keys)
static_assert(kSupMaskWords % 2 == 0, "the LDS mask of a batch lane is copied out as 64-bit words");
static_assert(kSupTiles - 1 <= kSupMaskSpare && kSupTilesSmall - 1 <= kSupMaskSpare, "the mask of a long list has spare words for the tiles behind its end");

template <int STAGE>
struct alignas(16) SupLds {
  HsTable<STAGE> set;
  int trpl[kMaxChunkVerts + 1];  // row offsets of the chunk's task lists
  HsWave<STAGE> w[kSupWaves];      // (while the set is built: the fill counters of its buckets)
  // per batch lane: word 0 = the matches of its task -- or, for a task of the flattened pass that reports a match mask, the mask itself
  // (bit k = key k of its tail; its matches are the mask's popcount)
  unsigned mw[kSupWaves][GM_WAVE][kSupMaskWords];
  unsigned ecnt[STAGE];                   // per stage entry: matches found at it
  int hq[kSupWaves][kSupQueue];      // per wave: DAG entries (streamed edges) whose increment is still to be issued
  int next_batch;
  unsigned queue_pos;
  int next_group;  // key stream of the short lists: the next group of tiles
  int pad_;
};

template <int STAGE, bool MASKS>
__global__ __launch_bounds__((kSupWaves * GM_WAVE), (STAGE <= 1024 ? 4 : 2))
void sup_kernel(const MineParams p) {
  __shared__ SupLds<STAGE> B;
  using H = HsHash<STAGE>;
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int tid = threadIdx.x;
  constexpr int nthreads = kSupWaves * GM_WAVE;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int *__restrict__ trp = p.g.trp;
  const int2 *__restrict__ tdesc = p.g.tdesc;
  const int *__restrict__ tedge = p.g.tedge;
  unsigned *__restrict__ sup = p.scratch;
  HsWave<STAGE> &L = B.w[wave];
  unsigned (*mw)[kSupMaskWords] = B.mw[wave];
  unsigned long long *__restrict__ smask = p.smask;
  unsigned long long n_at = 0;  // wave-uniform: the increments this wave issued as global atomics from its queue (tooling: counters[2])
  for (;;) {
    if (tid == 0) B.queue_pos = atomicAdd(p.queue, (unsigned)p.grab);
    __syncthreads();
    const unsigned q = B.queue_pos;
    if (q >= (unsigned)p.count) break;
    const unsigned qe = min(q + (unsigned)p.grab, (unsigned)p.count);
    for (unsigned ci = q; ci < qe; ++ci) {
      const size_t pos = (size_t)p.first + (size_t)ci * (size_t)p.step;
      const size_t cid = p.order ? (size_t)p.order[pos] : pos;
      const ChunkRec r = p.chunks[cid];
      const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
      const int eb = r.e_begin, nel = r.e_end - r.e_begin;
      for (int i = tid; i <= nvl; i += nthreads) B.trpl[i] = trp[ub + i];
      for (int i = tid; i < nel; i += nthreads) B.ecnt[i] = 0u;
      if (tid == 0) {
        B.next_batch = 0;
        B.next_group = 0;
      }
      const bool fallback = hs_build<STAGE, nthreads>(B.set, reinterpret_cast<unsigned *>(&B.w[0]), rp, col, ub, nvl, eb, nel,
                                                       (p.flags & (1 << 22)) != 0, tid);  // (ends with a barrier)
      // ---- waves: the KEY STREAM of the chunk's short lists (GraphView::kst, gm_tch.hip) -- one contiguous range per chunk, one coalesced
      // load per 64 keys, no descriptor, no row search, no flattening.  What a match needs beyond the triangle count's -- the DAG entry of
      // the streamed key and of the task's own edge -- lies beside the stream (kst_et) and is read by the lanes that found their key.
      if (p.g.kst != nullptr) {
        constexpr int TS = STAGE <= 1024 ? kSupTilesSmall : kSupTiles, GS = TS * GM_WAVE;
        const int kb = p.g.kst_rp[ub], kn = p.g.kst_rp[ub + nvl] - kb;
        const unsigned *__restrict__ kp = p.g.kst + kb;
        int *hq = B.hq[wave];
        int qn = 0;  // wave-uniform: queued entries
        auto flush = [&]() {
          wave_sync();
          for (int i = lane; i < qn; i += GM_WAVE) atomicAdd(&sup[hq[i]], 1u);
          n_at += (unsigned long long)qn;
          qn = 0;
          wave_sync();
        };
        auto grab = [&]() {
          int gi = 0;
          if (lane == 0) gi = atomicAdd(&B.next_group, 1);
          return (readfirst(gi) * r.nparts + r.part) * GS;
        };
        unsigned nxt[TS];
        int g0 = kn > 0 ? grab() : 0;
        if (g0 < kn) {
#pragma unroll
          for (int q = 0; q < TS; ++q) nxt[q] = kp[min(g0 + q * GM_WAVE + lane, kn - 1)];
        }
        wave_sync();
        while (g0 < kn) {  // wave-uniform
          const int gc = g0;
          int key[TS];
          unsigned salt[TS], rlo[TS], rlen[TS], at[TS];
          unsigned long long inm[TS], hm[TS], nm[TS];
#pragma unroll
          for (int q = 0; q < TS; ++q) {
            key[q] = (int)(nxt[q] & 0xffffffu);
            const int lo = (int)(((nxt[q] >> 24) - (unsigned)ub) & 255u);  // host = ub + local row, local row < 256
            salt[q] = H::salt(lo);
            const int rs = B.set.rpl[lo];
            rlo[q] = (unsigned)(rs - eb);
            // (a table hit is only looked for below the stage indices the table holds: gm_hset.h hs_pass)
            rlen[q] = (unsigned)min(B.set.rpl[lo + 1] - rs, max(0, (int)H::kPosLimit - (rs - eb)));
            inm[q] = __ballot(gc + q * GM_WAVE + lane < kn);
          }
          g0 = grab();
          if (g0 < kn) {
#pragma unroll
            for (int q = 0; q < TS; ++q) nxt[q] = kp[min(g0 + q * GM_WAVE + lane, kn - 1)];
          }
          hs_probe<STAGE, TS>(B.set, col, fallback, key, salt, rlo, rlen, inm, at, hm, nm);
          unsigned long long any_need = 0ull;
#pragma unroll
          for (int q = 0; q < TS; ++q) {
            any_need |= nm[q];
            if (hm[q] == 0ull) continue;  // wave-uniform
            const int nh = 2 * (int)__popcll(hm[q]);
            if (qn + nh > kSupQueue) flush();
            if (__builtin_amdgcn_inverse_ballot_w64(hm[q])) {
              const int pos = kb + gc + q * GM_WAVE + lane;
              atomicAdd(&B.ecnt[rlo[q] + at[q]], 1u);
              const int slot = qn + 2 * rank_below(hm[q]);
              const int2 et = p.g.kst_et[pos];
              hq[slot] = et.x;
              hq[slot + 1] = et.y;
            }
            qn += nh;
          }
          if (any_need != 0ull)  // rare
            hs_surplus<STAGE, TS>(B.set, lane, key, salt, nm, [&](const int q, const int sl, const int pat) {
              if (qn + 2 > kSupQueue) flush();
              const int pos = kb + gc + q * GM_WAVE + sl;
              const unsigned r0 = (unsigned)readlane((int)rlo[q], sl);
              if (lane == 0) {
                atomicAdd(&B.ecnt[r0 + (unsigned)pat], 1u);
                const int2 et = p.g.kst_et[pos];
                hq[qn] = et.x;
                hq[qn + 1] = et.y;
              }
              qn += 2;
            });
        }
        flush();
      }
      const int tb = B.trpl[0], ntask = B.trpl[nvl] - tb;
      for (;;) {
        int bi = 0;
        if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
        bi = readfirst(bi) * r.nparts + r.part;
        const int t0 = bi * GM_WAVE;
        if (t0 >= ntask) break;
        const bool valid = t0 + lane < ntask;
        const int te = tb + min(t0 + lane, ntask - 1);
        const int2 d = tdesc[te];                      // {start, length} of the list to stream, coalesced
        const int own_e = tedge[te];                   // the task's own DAG entry
        const int lo = hs_local_row(B.trpl, nvl, te);  // the host row of this task
        const int ru = B.set.rpl[lo], a = B.set.rpl[lo + 1] - ru;
        const bool act = valid && d.y > 0 && a > 0;
        // tm: where this task's match mask goes (64-bit words into the arena), kNoMask = its streamed edges are reported by atomics
        const unsigned tm = (MASKS && act) ? p.g.tmoff[te] : kNoMask;
        const bool masked = MASKS && tm != kNoMask;
#pragma unroll
        for (int k = 0; k < (MASKS ? kSupMaskWords : 1); ++k) mw[lane][k] = 0u;
        int *hq = B.hq[wave];
        int qn = 0;  // wave-uniform: queued entries
        auto flush = [&]() {
          wave_sync();
          for (int i = lane; i < qn; i += GM_WAVE) atomicAdd(&sup[hq[i]], 1u);
          n_at += (unsigned long long)qn;
          qn = 0;
          wave_sync();
        };
        wave_sync();
        // The two per-task words handed back to the match handler (wave-uniform for the tiles of a long list):
        //   word  = where the host's row starts in the stage (bits 0..11) | the task's batch lane << 12 | bit 31: the task reports a mask
        //   word2 = a masked task of the flattened pass: the start of its list (bit k of its LDS mask = key k);
        //           a masked LONG list: tm - (start >> 6) -- the tile of 64 keys from col[kidx0 ..) is word (kidx0 >> 6) behind it, because
        //           kidx0 - start is a multiple of 64 (its mask has kSupTiles - 1 spare words: the last group looks all its tiles up)
        constexpr int kOwnerShift = 12;
        static_assert(STAGE <= (1 << kOwnerShift), "the stage index shares a word with the batch lane");
        auto hit = [&](const unsigned long long hm, const int word, const int word2, const unsigned at, const int kidx, const bool uniform) {
          if (uniform) {  // a tile of ONE task (long lists): word / word2 are wave-uniform
            const int wu = readfirst(word);
            const int row0 = wu & ((1 << kOwnerShift) - 1), owner = (wu >> kOwnerShift) & (GM_WAVE - 1);
            if (MASKS && wu < 0) {  // the tile's ballot IS 64 bits of the mask: one plain store, matches or not
              if (lane == 0) smask[(size_t)((unsigned)readfirst(word2) + ((unsigned)readfirst(kidx) >> 6))] = hm;
              if (hm == 0ull) return;
              if (__builtin_amdgcn_inverse_ballot_w64(hm)) atomicAdd(&B.ecnt[row0 + (int)at], 1u);
              return;  // (the task's own edge: the popcount of its mask, added by the second pass)
            }
            if (hm == 0ull) return;
            if (lane == 0) atomicAdd(&mw[owner][0], (unsigned)__popcll(hm));  // its count once (64 atomics on one LDS word would serialise)
            if (__builtin_amdgcn_inverse_ballot_w64(hm)) {
              atomicAdd(&B.ecnt[row0 + (int)at], 1u);
              hq[qn + rank_below(hm)] = kidx;
            }
            qn += __popcll(hm);
            if (qn > kSupFlushAt) flush();
            return;
          }
          if (hm == 0ull) return;  // wave-uniform
          const int row0 = word & ((1 << kOwnerShift) - 1), owner = (word >> kOwnerShift) & (GM_WAVE - 1);
          const unsigned long long am = MASKS ? hm & __ballot(word >= 0) : hm;  // the matches that are reported by atomics
          if (__builtin_amdgcn_inverse_ballot_w64(hm)) {
            atomicAdd(&B.ecnt[row0 + (int)at], 1u);
            if (MASKS && word < 0) {  // a masked task of the flattened pass: bit (kidx - start of its list) of the LDS mask of its batch lane
              const int k = kidx - word2;
              atomicOr(&mw[owner][k >> 5], 1u << (k & 31));
            } else {
              hq[qn + rank_below(am)] = kidx;
              atomicAdd(&mw[owner][0], 1u);
            }
          }
          qn += __popcll(am);
          if (qn > kSupFlushAt) flush();
        };
        auto hit1 = [&](const int word, const int word2, const int at, const int kidx) {  // one key found through the surplus list (wave-uniform)
          const int row0 = word & ((1 << kOwnerShift) - 1), owner = (word >> kOwnerShift) & (GM_WAVE - 1);
          if (MASKS && word < 0) {  // its bit joins the mask: in LDS (flattened pass) or behind the tile's store (long lists)
            if (lane == 0) {
              atomicAdd(&B.ecnt[row0 + at], 1u);
              const int start = readlane(d.x, owner);
              const int k = kidx - start;
              if (readlane(d.y, owner) >= kLongList) atomicOr(&smask[(size_t)((unsigned)word2 + ((unsigned)start >> 6) + ((unsigned)k >> 6))], 1ull << (k & 63));
              else atomicOr(&mw[owner][k >> 5], 1u << (k & 31));
            }
            return;
          }
          if (lane == 0) {
            atomicAdd(&B.ecnt[row0 + at], 1u);
            hq[qn] = kidx;
            atomicAdd(&mw[owner][0], 1u);
          }
          qn += 1;
          if (qn > kSupFlushAt) flush();
        };
        const int word_l = (ru - eb) | (lane << kOwnerShift) | (masked ? (int)0x80000000 : 0);
        const int word2_l = !masked ? 0 : (d.y >= kLongList ? (int)(tm - ((unsigned)d.x >> 6)) : d.x);
        hs_pass<STAGE, (STAGE <= 1024 ? kSupTilesSmall : kSupTiles)>(B.set, L, col, fallback, lane, act ? d.y : 0, d.x, H::salt(lo), ru - eb, a, word_l, word2_l, hit, hit1);
        flush();
        const unsigned c = masked ? 0u : mw[lane][0];  // (a masked task's own edge: the popcount of its mask, added by the second pass)
        if (masked && d.y < kLongList) {  // the mask of a flattened task goes out: ceil(len / 64) 64-bit words
          const unsigned long long *m64 = reinterpret_cast<const unsigned long long *>(mw[lane]);
#pragma unroll
          for (int k = 0; k < kSupMaskWords / 2; ++k)
            if (k * 64 < d.y) smask[(size_t)tm + (size_t)k] = m64[k];
        }
        if (valid && c) atomicAdd(&sup[own_e], c);
        wave_sync();
      }
      __syncthreads();  // every wave is done with the chunk: its per-entry counts go out, the set is rewritten by the next chunk
      for (int i = tid; i < nel; i += nthreads) {
        const unsigned c = B.ecnt[i];
        if (c) atomicAdd(&sup[eb + i], c);
      }
      __syncthreads();
    }
  }
  if (lane == 0 && n_at) atomicAdd(&p.counters[2], n_at);
}

// The masks summed by COLUMN into the supports: entry j of row u += sum over the masked entries i < j of bit (j - i - 1) of the mask of
// entry i (the streamed edges of the matches the in-edge tasks of row u found), and entry i += the popcount of its own mask (its task's
// own edge: a masked task adds nothing itself).  Two kernels, plain read-modify-writes -- the triangle pass is complete and every
// entry has one writer per kernel:
//   * NEAR (sup_near_kernel): word 0 of every mask, entry-parallel.  Bit k of the mask of entry e' belongs to entry e' + 1 + k of the SAME
//     row -- bits past the row's end are never set -- so entry e += sum_{t = 1 .. 64} bit (t - 1) of word0[e - t] needs no row bounds at
//     all: a 128-entry window of words in LDS per wave, t runs to the highest bit set in the window.  Covers every row of <= 65 entries.
//   * FAR (sup_far_kernel): the words beyond the first (tails of more than 64 keys), row by row: lane = row i of a 64-row block, word
//     after word of its mask, one LDS atomic per set bit into a counter per column.  The rows that have such tails are listed once per graph
//     (ensure_sup_masks), from the last ids of the DAG -- the widest rows under a numbering by degree -- down, and dealt out to the
//     waves round robin: a wave that took 64 consecutive wide rows in one dequeue WAS the kernel's duration (1.8 ms of 2.1).
constexpr int kSupColsWaves = 4;
__global__ __launch_bounds__(256) void sup_near_kernel(const SupColsParams p) {
  __shared__ unsigned long long win[4][2 * GM_WAVE];
  const int lane = threadIdx.x & (GM_WAVE - 1);
  unsigned long long *w = win[threadIdx.x >> 6];
  const long long nwin = (p.ne + GM_WAVE - 1) / GM_WAVE;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long wi = wave0; wi < nwin; wi += nwaves) {
    const long long e = wi * GM_WAVE + lane, ep = e - GM_WAVE;
    const unsigned oc = e < p.ne ? p.emoff[e] : kNoMask, op = ep >= 0 ? p.emoff[ep] : kNoMask;
    const unsigned long long mc = oc != kNoMask ? p.smask[oc] : 0ull, mp = op != kNoMask ? p.smask[op] : 0ull;
    w[lane] = mp;
    w[GM_WAVE + lane] = mc;
    const int hb = max(mc ? 64 - (int)__builtin_clzll(mc) : 0, mp ? 64 - (int)__builtin_clzll(mp) : 0);  // bit k reaches the entry k + 1 behind
    const int tmax = wave_max_nonneg(hb);
    wave_sync();
    unsigned acc = (unsigned)__popcll(mc);
    for (int t = 1; t <= tmax; ++t) acc += (unsigned)((w[GM_WAVE + lane - t] >> (t - 1)) & 1ull);
    if (acc) p.sup[e] += acc;  // (acc != 0 implies e < ne: a lane beyond the end has no word and no entry before it points past the end)
    wave_sync();
  }
}

__global__ __launch_bounds__(kSupColsWaves *GM_WAVE) void sup_far_kernel(const SupColsParams p) {
  __shared__ unsigned counters[kSupColsWaves][kTctStageMax];
  const int lane = threadIdx.x & (GM_WAVE - 1);
  unsigned *cc = counters[threadIdx.x >> 6];
  const int wave0 = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)(((long long)gridDim.x * blockDim.x) >> 6);
  // (the list runs from the widest ids down: row r of it to wave r mod W -- every wave gets its share of the wide rows, no dequeue)
  for (int r = wave0; r < p.n_far_rows; r += nwaves) {
    const int u = p.far_rows[r];
    const int r0 = p.rp[u], dd = p.rp[u + 1] - r0;
    for (int j = lane; j < dd; j += GM_WAVE) cc[j] = 0u;
    wave_sync();
    const int nrows = dd - 1 - GM_WAVE;  // entries [0, nrows) have tails of more than 64 keys
    for (int ib = 0; ib < nrows; ib += GM_WAVE) {
      const int i = ib + lane;
      const unsigned off = i < nrows ? p.emoff[r0 + i] : kNoMask;
      const int nw = off != kNoMask ? (dd - 1 - i + 63) >> 6 : 0;
      const int nwmax = wave_max_nonneg(nw);
      unsigned own = 0u;
      constexpr int kU = 8;  // words of a row requested together
      for (int w0 = 1; w0 < nwmax; w0 += kU) {
        unsigned long long xs[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) xs[k] = w0 + k < nw ? p.smask[(size_t)off + (size_t)(w0 + k)] : 0ull;
#pragma unroll
        for (int k = 0; k < kU; ++k) {
          unsigned long long x = xs[k];
          own += (unsigned)__popcll(x);
          const int base = i + 1 + 64 * (w0 + k);
          // (only bits of entries of this row count: a word the current launch did not rewrite -- a filtered or aborted launch -- must
          // not index past the row's counters; the arena starts zeroed, ensure_sup_masks)
          if (dd - base < 64) x &= dd > base ? (1ull << (dd - base)) - 1ull : 0ull;
          while (x) {
            atomicAdd(&cc[base + (int)__builtin_ctzll(x)], 1u);
            x &= x - 1;
          }
        }
      }
      if (own) atomicAdd(&cc[i], own);
    }
    wave_sync();
    for (int j = lane; j < dd; j += GM_WAVE) {
      const unsigned c = cc[j];
      if (c) p.sup[r0 + j] += c;
    }
    wave_sync();
  }
}
hipError_t launch_sup_cols(const SupColsParams &p, int cu_count, hipStream_t stream) {
  static_assert(sizeof(unsigned) * kSupColsWaves * kTctStageMax * 5 <= 163840, "five workgroups per CU");
  if (p.nv <= 0 || p.ne <= 0 || !p.emoff || !p.smask) return hipSuccess;
  const long long nwin = (p.ne + GM_WAVE - 1) / GM_WAVE;
  hipLaunchKernelGGL(sup_near_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>((nwin + 3) / 4, (long long)cu_count * 8))), dim3(256), 0, stream, p);
  if (p.n_far_rows > 0) {
    const long long blocks = std::min<long long>(((long long)p.n_far_rows + kSupColsWaves - 1) / kSupColsWaves, (long long)cu_count * 5);
    hipLaunchKernelGGL(sup_far_kernel, dim3((unsigned)std::max<long long>(1, blocks)), dim3(kSupColsWaves * GM_WAVE), 0, stream, p);
  }
  return hipGetLastError();
}

// sum over the entries [first, first + count) of C(t, 2): four entries per 16-byte load, ONE atomic per workgroup (same-address atomics are
// served one after the other, ~10 ns each: the 16 K waves of the round-4 launch spent 0.16 of its 0.22 ms queueing for the total)
__global__ __launch_bounds__(256) void sup_pairs_kernel(const unsigned *__restrict__ sup, long long first, long long count,
                                                        unsigned long long *__restrict__ out) {
  __shared__ unsigned long long part[4];
  auto c2 = [](const unsigned x) { return (unsigned long long)x * (unsigned long long)(x - (x ? 1u : 0u)) / 2ull; };
  const unsigned *__restrict__ p = sup + first;
  const long long head = std::min<long long>(count, (long long)((4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3));  // entries before the first 16-byte boundary
  const long long nvec = (count - head) >> 2;
  const uint4 *__restrict__ v = reinterpret_cast<const uint4 *>(p + head);
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (long long i = gid; i < nvec; i += stride) {
    const uint4 x = v[i];
    s += c2(x.x) + c2(x.y) + c2(x.z) + c2(x.w);
  }
  for (long long i = gid; i < head; i += stride) s += c2(p[i]);
  for (long long i = head + 4 * nvec + gid; i < count; i += stride) s += c2(p[i]);
  s = wave_sum_u64(s);
  if ((threadIdx.x & (GM_WAVE - 1)) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = part[0] + part[1] + part[2] + part[3];
    if (t) atomicAdd(out, t);
  }
}

// The out-edges of the FEW rows beyond the 2048-entry stage (no hashed set holds such a row; the triangle count sends them to the chunked
// kernel): one wave per task edge u -> v, the shorter of N+(u) -- beyond v under a topological numbering -- and N+(v) streamed by the lanes,
// the longer one bisected in global memory, the same three increments per match.  Slow per key, exact, a handful of rows per graph
// (R-MAT-26: a few dozen).  Several ranks take every world-th edge.
__global__ __launch_bounds__(256) void sup_long_kernel(const SupLongParams p) {
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long t = (long long)p.rank + wave0 * (long long)p.world; t < p.total; t += nwaves * (long long)p.world) {
    int lo = 0, hi = p.nrows - 1;  // the long row of edge t: largest r with prefix[r] <= t (wave-uniform)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (p.prefix[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int u = p.rows[lo], i = (int)(t - p.prefix[lo]);
    const int ru = p.rp[u], du = p.rp[u + 1] - ru, e = ru + i, v = p.col[e];
    const int rv = p.rp[v], dv = p.rp[v + 1] - rv;
    const int skip = p.topo ? i + 1 : 0;
    const int abase = ru + skip, a = du - skip;
    // stream the shorter list, bisect the longer
    const bool a_short = a <= dv;
    const int sbase = a_short ? abase : rv, sn = a_short ? a : dv, lbase = a_short ? rv : abase, ln = a_short ? dv : a;
    unsigned cnt = 0;
    for (int k = lane; k < sn; k += GM_WAVE) {
      const int key = p.col[sbase + k];
      const int pos = lower_bound(p.col + lbase, ln, key);
      if (pos < ln && p.col[lbase + pos] == key) {
        atomicAdd(&p.sup[sbase + k], 1u);
        atomicAdd(&p.sup[lbase + pos], 1u);
        ++cnt;
      }
    }
    const int total = wave_sum((int)cnt);
    if (lane == 0 && total) atomicAdd(&p.sup[e], (unsigned)total);
  }
}
hipError_t launch_sup_long(const SupLongParams &p, int cu_count, hipStream_t stream) {
  if (p.total <= 0 || p.nrows <= 0) return hipSuccess;
  const long long waves = (p.total + p.world - 1) / p.world;
  const long long blocks = std::min<long long>((waves + 3) / 4, (long long)cu_count * 8);
  hipLaunchKernelGGL(sup_long_kernel, dim3((unsigned)std::max<long long>(1, blocks)), dim3(256), 0, stream, p);
  return hipGetLastError();
}

int sup_per_cu(int stage) {
  const size_t lds = stage <= 1024 ? sizeof(SupLds<1024>) : sizeof(SupLds<kTctStageMax>);
  return (int)std::max<size_t>(1, std::min<size_t>(163840 / lds, 2048 / (kSupWaves * GM_WAVE)));
}
hipError_t launch_sup(const MineParams &p, int stage, int grid_blocks, hipStream_t stream) {
  const bool masks = p.g.tmoff != nullptr && p.smask != nullptr;
  static_assert(sizeof(SupLds<kTctStageMax>) <= 163840, "one workgroup per CU at least");
  static_assert(sizeof(HsWave<kTctStageMax>) * kSupWaves >= (size_t)kTctStageMax * 2, "fill counters alias the wave scratch");
  if (p.g.trp == nullptr || p.g.tdesc == nullptr || p.g.tedge == nullptr || p.scratch == nullptr) return hipErrorInvalidValue;
  const dim3 grid((unsigned)grid_blocks), block(kSupWaves * GM_WAVE);
  if (stage <= 1024 && masks) hipLaunchKernelGGL((sup_kernel<1024, true>), grid, block, 0, stream, p);
  else if (stage <= 1024) hipLaunchKernelGGL((sup_kernel<1024, false>), grid, block, 0, stream, p);
  else if (masks) hipLaunchKernelGGL((sup_kernel<kTctStageMax, true>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((sup_kernel<kTctStageMax, false>), grid, block, 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_sup_pairs(const unsigned *sup, long long first, long long count, unsigned long long *out, int cu_count, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  const long long blocks = std::min<long long>((count + 1023) / 1024, (long long)cu_count * 8);
  hipLaunchKernelGGL(sup_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, sup, first, count, out);
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_sup_kernel() {}
void gm_touch_sup() { hipLaunchKernelGGL(gm_touch_sup_kernel, dim3(1), dim3(1), 0, 0); }
