// gm_cbuild.hip -- k-clique (k = 4), first DFS level RE-HOSTED: every DAG edge u -> v is a task of the endpoint with the longer
// out-list, the other list is streamed against it, and the finished row of u's adjacency bit-matrix is stored into the matrix arena
// (see "k-clique, level 1 re-hosted" in gm_mine.h).  The reference's kernel re-intersects N+(v0) ^ N+(v1) per edge with the shorter
// list searched in the longer (src/clique/gpu_kernels/clique4_warp_edge.cuh:19-27) and never keeps the result; round 2 of this
// library streamed N+(v) of every edge whichever was longer (12.6 vs 5.0 G keys on the LiveJournal stand-in, gm_tct.hip).
//
// Kernel = the shorter-list-streams triangle kernel with positions: a chunk is a run of consecutive vertices whose DAG rows fit the LDS
// stage; its tasks are the task-list entries of those vertices, 64 per batch.  A match is either bit `position in the host's row`
// (type A) or bit `bit_off + index in the streamed list` (type B) of the task's row.  Rows are built in a per-wave LDS buffer of
// kCbRowBuf words -- a batch is processed in sub-batches of as many tasks as fit -- and stored to the arena by the wave that built
// them: no workgroup barrier, no device atomics.
//
// The chunk's rows are ONE HASHED SET of (row, id) -> position in LDS (round 3; before: sorted copy + bit filter + candidate queue +
// bisection, 80 VALU per 64 streamed keys with 30 % of the keys of the com-Orkut stand-in going through all of it):
//   * h = id * C mod 2^32 is a bijection of the ids; its top LB bits (XOR a salt that is injective in the local row: gm_tch.hip) pick
//     one of 2^LB = STAGE buckets of four 32-bit slots, and a slot holds the REMAINING 32 - LB bits of h above the entry's position
//     in its row (LB bits: rows have <= STAGE entries): bucket + slot are the (row, id) pair exactly, and one XOR with h << LB
//     turns the matching slot into the position and every other slot into a number >= STAGE -- the minimum of the four is the answer;
//   * empty = 0xffffffff and overflow marker = 0xfffffffe would read as positions STAGE - 1 / STAGE - 2 of some id: entries at those
//     positions (rows of >= STAGE - 1 entries) live in the surplus list, a hit is a minimum < STAGE - 2;
//   * surplus list / global-memory fallback as in gm_tch.hip.
#include "gm_flat.h"

namespace gm {

constexpr int kCbTiles = 4;  // 64-key tiles in flight per wave
constexpr int kCbOvfCap = 128;
constexpr unsigned kCbEmpty = 0xffffffffu, kCbMarker = 0xfffffffeu;
constexpr unsigned kCbMul = 0x9E3779B1u;

struct alignas(16) CbWave {
  int4 desc[GM_WAVE];                // per batch lane: {key_base - offset among the flattened positions, salt, row meta, offset}
  unsigned char marks[kMarkWindow];  // owner marks of the flattened positions
};

template <int STAGE>
struct alignas(16) CBuildLds {
  uint4 table[STAGE];              // buckets of four (hash remainder, position) slots
  int rpl[kMaxChunkVerts + 1];     // row offsets of the chunk's DAG rows (global entry indices)
  int trpl[kMaxChunkVerts + 1];    // row offsets of its task lists
  unsigned rows[kWavesPerBlock][kCbRowBuf];
  CbWave w[kWavesPerBlock];        // (while the table is built: packed 16-bit fill counters of the buckets)
  int ovf_key[kCbOvfCap];
  int ovf_salt[kCbOvfCap];
  int ovf_pos[kCbOvfCap];
  int n_ovf;
  int next_batch;
  unsigned queue_pos;
  int pad_;
};
static_assert(kCbOvfCap == 2 * GM_WAVE, "the surplus list is scanned two entries per lane");

template <int STAGE>
struct CbHash {
  static constexpr int LB = STAGE == 1024 ? 10 : 11;
  static_assert((1 << LB) == STAGE, "one bucket per stage entry");
  static constexpr unsigned kMask = (unsigned)(STAGE - 1) << 4;
  static constexpr unsigned kPosLimit = (unsigned)STAGE - 2u;  // positions kept in the table (see the header)
  static __device__ __forceinline__ unsigned hash(int x) { return (unsigned)x * kCbMul; }
  static __device__ __forceinline__ unsigned bucket(unsigned h, unsigned s) { return ((h >> (28 - LB)) & kMask) ^ s; }  // byte offset
  static __device__ __forceinline__ unsigned salt(int local_row) { return (((unsigned)local_row * 37u) & (unsigned)(STAGE - 1)) << 4; }
  static __device__ __forceinline__ int row_of(unsigned s) { return (int)(((s >> 4) * 941u) & (unsigned)(STAGE - 1)); }  // 37 * 941 = 1 mod 2048
};

__device__ __forceinline__ int cb_local_row(const int *rpl, const int nvl, const int e) {  // largest i with rpl[i] <= e
  int lo = 0, hi = nvl - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rpl[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Where in the row their salt names are the T keys of this lane?  pos[q] < kPosLimit: found there; nm = lanes that missed in a
// bucket that overflowed (the surplus list decides).
template <int STAGE, int T>
__device__ __forceinline__ void cb_probe(const CBuildLds<STAGE> &B, const int *__restrict__ col, const bool fallback, const int (&key)[T],
                                         const unsigned (&salt)[T], const unsigned long long (&inm)[T], unsigned (&pos)[T],
                                         unsigned long long (&hm)[T], unsigned long long (&nm)[T]) {
  using H = CbHash<STAGE>;
  if (fallback) {  // wave-uniform
#pragma unroll
    for (int q = 0; q < T; ++q) {
      bool f = false;
      pos[q] = 0u;
      if (__builtin_amdgcn_inverse_ballot_w64(inm[q])) {
        const int lo = H::row_of(salt[q]);
        const int rs = B.rpl[lo], rn = B.rpl[lo + 1] - rs;
        const int at = lower_bound(col + rs, rn, key[q]);
        f = at < rn && col[rs + at] == key[q];
        pos[q] = (unsigned)at;
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
    w[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(B.table) + H::bucket(h, salt[q]));
  }
#pragma unroll
  for (int q = 0; q < T; ++q) {
    pos[q] = min(min(w[q].x ^ t[q], w[q].y ^ t[q]), min(w[q].z ^ t[q], w[q].w ^ t[q]));
    const unsigned long long m = __ballot(pos[q] < H::kPosLimit);
    hm[q] = m & inm[q];
    nm[q] = __ballot(w[q].w == kCbMarker) & ~m & inm[q];
  }
}

template <int STAGE>
__global__ __launch_bounds__((kWavesPerBlock * GM_WAVE), (STAGE <= 1024 ? 5 : 3))
void cbuild_kernel(const CBuildParams p) {
  __shared__ CBuildLds<STAGE> B;
  using H = CbHash<STAGE>;
  constexpr int T = kCbTiles;
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int tid = threadIdx.x, nthreads = kWavesPerBlock * GM_WAVE;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int *__restrict__ trp = p.trp;
  const int4 *__restrict__ tasks = reinterpret_cast<const int4 *>(p.tasks);
  unsigned *__restrict__ mat = p.mat;
  CbWave &L = B.w[wave];
  unsigned *rb = B.rows[wave];
  for (;;) {
    if (tid == 0) B.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = B.queue_pos;
    if (q >= (unsigned)p.count) break;
    const ChunkRec r = p.chunks[p.order ? p.order[q] : (int)q];
    const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
    const int eb = r.e_begin, nel = r.e_end - r.e_begin;
    const int tb = trp[ub], ntask = trp[ub + nvl] - tb;
    if (ntask == 0) { __syncthreads(); continue; }  // (these vertices host nothing for this rank: nothing to stage)
    // ---- workgroup: the chunk's DAG rows into the set -------------------------------------------------------------------
    unsigned *fill32 = reinterpret_cast<unsigned *>(&B.w[0]);
    for (int i = tid; i <= nvl; i += nthreads) {
      B.rpl[i] = rp[ub + i];
      B.trpl[i] = trp[ub + i];
    }
    {
      const uint4 empty = make_uint4(kCbEmpty, kCbEmpty, kCbEmpty, kCbEmpty);
      for (int i = tid; i < STAGE; i += nthreads) B.table[i] = empty;
      for (int i = tid; i < STAGE / 2; i += nthreads) fill32[i] = 0u;
      if (tid < kCbOvfCap) {
        B.ovf_key[tid] = -1;
        B.ovf_salt[tid] = -1;
        B.ovf_pos[tid] = 0;
      }
      if (tid == 0) {
        B.n_ovf = 0;
        B.next_batch = 0;
      }
    }
    __syncthreads();
    unsigned *slots = reinterpret_cast<unsigned *>(B.table);
    // Two passes over the entries, which stay in registers in between: (1) every entry takes a number in its bucket; (2) when the
    // bucket's total is known, numbers 0..3 (0..2 if the bucket overflowed: its last slot holds the marker) go to the table, the rest
    // to the surplus list.  (An entry at one of the two positions the table cannot hold counts as five and writes the marker itself.)
    constexpr int kU = 4, kIt = STAGE / (kU * kWavesPerBlock * GM_WAVE);  // entries requested together per thread; trips of a thread
    static_assert(kIt * kU * kWavesPerBlock * GM_WAVE == STAGE, "the entries of a full stage are spread evenly over the threads");
    int xv[kIt][kU];
    unsigned pk[kIt][kU];  // local row | position << 8 | number in the bucket (capped at 15) << 20
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int i0 = it * kU * nthreads + tid;
#pragma unroll
      for (int j = 0; j < kU; ++j) xv[it][j] = (i0 < nel) ? col[eb + min(i0 + j * nthreads, nel - 1)] : 0;
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const int i = i0 + j * nthreads;
        pk[it][j] = 0u;
        if (i < nel) {
          const int lo = cb_local_row(B.rpl, nvl, eb + i);
          const unsigned at = (unsigned)(eb + i - B.rpl[lo]);  // position in its row
          const unsigned b = H::bucket(H::hash(xv[it][j]), H::salt(lo)) >> 4;
          const unsigned shift = (b & 1u) * 16u;
          const unsigned num = (atomicAdd(&fill32[b >> 1], (at < H::kPosLimit ? 1u : 5u) << shift) >> shift) & 0xffffu;
          pk[it][j] = (unsigned)lo | (at << 8) | (min(num, 15u) << 20);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const int i = it * kU * nthreads + j * nthreads + tid;
        if (i < nel) {
          const unsigned lo = pk[it][j] & 255u, at = (pk[it][j] >> 8) & 4095u, num = pk[it][j] >> 20;
          const unsigned s = H::salt((int)lo), h = H::hash(xv[it][j]);
          const unsigned b = H::bucket(h, s) >> 4;
          const unsigned c = (fill32[b >> 1] >> ((b & 1u) * 16u)) & 0xffffu;
          const bool holds = at < H::kPosLimit;
          if (holds && num < (c > 4u ? 3u : 4u)) {
            slots[(b << 2) + num] = (h << H::LB) | at;
          } else {
            if (num == 3u || !holds) slots[(b << 2) + 3] = kCbMarker;
            const int jo = atomicAdd(&B.n_ovf, 1);
            if (jo < kCbOvfCap) {
              B.ovf_key[jo] = xv[it][j];
              B.ovf_salt[jo] = (int)s;
              B.ovf_pos[jo] = (int)at;
            }
          }
        }
      }
    }
    __syncthreads();  // (also: the fill counters are dead, the waves may use their scratch)
    const bool fallback = B.n_ovf > kCbOvfCap || (p.flags & (1 << 22)) != 0;
    // ---- waves: batches of 64 tasks, sub-batches of as many rows as the wave's row buffer holds -----------------------------
    // (a chunk with few tasks -- the share of one rank of eight holds ~125 per chunk -- takes smaller batches, so that all four waves
    // get some: with 64 two waves of every workgroup idled, the build of a 1/8 share ran at 0.68 of its ideal)
    const int per_part = (ntask + r.nparts - 1) / r.nparts;
    const int bsz = per_part >= 8 * GM_WAVE ? GM_WAVE : (per_part >= 4 * GM_WAVE ? 32 : 16);
    for (;;) {
      int bi = 0;
      if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
      bi = readfirst(bi) * r.nparts + r.part;
      const int t0 = bi * bsz;
      if (t0 >= ntask) break;
      const int nvalid = min(bsz, ntask - t0);
      const bool valid = lane < nvalid;
      const int te = tb + min(t0 + lane, ntask - 1);
      const int4 Tk = tasks[te];                      // coalesced, 16 B per lane
      const int lo = cb_local_row(B.trpl, nvl, te);   // the host row of this task
      const int a = B.rpl[lo + 1] - B.rpl[lo];
      const unsigned salt_l = H::salt(lo);
      const unsigned fl = (unsigned)Tk.w;
      const int words = valid ? (int)((fl >> 20) & 127u) : 0;
      const int bit_off = (int)((fl >> 8) & 4095u);
      const bool type_b = (fl >> 31) != 0u;
      const unsigned long long off = ((unsigned long long)(fl & 255u) << 32) | (unsigned long long)(unsigned)Tk.z;
      const int incl_w = wave_incl_scan_add(words);
      int start = 0, consumed = 0;  // wave-uniform: first lane / first word of the current sub-batch
      while (start < nvalid) {
        const unsigned long long fit = __ballot(lane >= start && valid && incl_w - consumed <= kCbRowBuf);
        const int cnt = __popcll(fit);  // >= 1: a row is at most 64 words
        const bool in_sub = lane >= start && lane < start + cnt;
        const int row0 = incl_w - words - consumed;  // this lane's row inside the buffer
        const int used = readlane(incl_w, start + cnt - 1) - consumed;
        for (int i = lane; i < used; i += GM_WAVE) rb[i] = 0u;
        // row meta: bits 0..11 the row's first word in the buffer, 12..23 bit_off, 31 type B
        const int meta_l = (row0 & 4095) | (bit_off << 12) | (type_b ? (int)0x80000000u : 0);
        wave_sync();
        // a match of lane-key (tile q): position `at` in the host row, index `sidx` in the streamed list
        auto set_bits = [&](const unsigned long long hm, const int meta, const unsigned at, const int sidx) {
          if (hm == 0ull) return;  // wave-uniform
          if (__builtin_amdgcn_inverse_ballot_w64(hm)) {
            const int bit = (meta < 0) ? ((meta >> 12) & 4095) + sidx : (int)at;
            atomicOr(&rb[(meta & 4095) + (bit >> 5)], 1u << (bit & 31));
          }
        };
        auto surplus = [&](const int (&key)[T], const unsigned (&salt)[T], const int (&meta)[T], const int (&sidx)[T],
                           const unsigned long long (&nm)[T]) {
          const int k0 = B.ovf_key[lane], k1 = B.ovf_key[lane + GM_WAVE];
          const int s0 = B.ovf_salt[lane], s1 = B.ovf_salt[lane + GM_WAVE];
          const int p0 = B.ovf_pos[lane], p1 = B.ovf_pos[lane + GM_WAVE];
#pragma unroll
          for (int qq = 0; qq < T; ++qq) {
            unsigned long long rest = nm[qq];
            while (rest) {
              const int src = __ffsll((long long)rest) - 1;
              rest &= rest - 1;
              const int k = readlane(key[qq], src), sv = readlane((int)salt[qq], src);
              const bool e0 = (k0 == k) & (s0 == sv), e1 = (k1 == k) & (s1 == sv);
              const unsigned long long em = __ballot(e0 | e1);
              if (em != 0ull) {
                const int holder = __ffsll((long long)em) - 1;
                const int at = readlane(e0 ? p0 : p1, holder);
                const int mt = readlane(meta[qq], src), si = readlane(sidx[qq], src);
                if (lane == 0) {
                  const int bit = (mt < 0) ? ((mt >> 12) & 4095) + si : at;
                  atomicOr(&rb[(mt & 4095) + (bit >> 5)], 1u << (bit & 31));
                }
              }
            }
          }
        };
        const int llen_all = (in_sub && a > 0) ? Tk.y : 0;
        const int key_base = Tk.x;
        if (wave_max_nonneg(llen_all) != 0) {  // wave-uniform
          const bool is_long = llen_all >= kLongList;
          const int llen = is_long ? 0 : llen_all;
          // ---- long lists: one task at a time, wave-uniform base / salt / meta ---------------------------------------------
          unsigned long long lm = __ballot(is_long);
          while (lm) {
            const int src = __ffsll((long long)lm) - 1;
            lm &= lm - 1;
            const int base = readlane(key_base, src);
            const int n = readlane(llen_all, src);
            const unsigned s_u = (unsigned)readlane((int)salt_l, src);
            const int m_u = readlane(meta_l, src);
            const int *__restrict__ kp = col + base;
            auto process = [&](const int (&key)[T], const unsigned long long (&inm)[T], const int t_base) {
              unsigned salt[T], at[T];
              int meta[T], sidx[T];
#pragma unroll
              for (int qq = 0; qq < T; ++qq) {
                salt[qq] = s_u;
                meta[qq] = m_u;
                sidx[qq] = t_base + qq * GM_WAVE + lane;
              }
              unsigned long long hm[T], nm[T];
              cb_probe<STAGE, T>(B, col, fallback, key, salt, inm, at, hm, nm);
              unsigned long long any_need = 0ull;
#pragma unroll
              for (int qq = 0; qq < T; ++qq) {
                set_bits(hm[qq], m_u, at[qq], sidx[qq]);
                any_need |= nm[qq];
              }
              if (any_need != 0ull) surplus(key, salt, meta, sidx, nm);  // rare
            };
            constexpr int G = GM_WAVE * T;
            int nxt[T];
#pragma unroll
            for (int qq = 0; qq < T; ++qq) nxt[qq] = kp[min(qq * GM_WAVE + lane, n - 1)];
            int t = 0;
            for (; t + 2 * G <= n; t += G) {
              int key[T];
              unsigned long long inm[T];
#pragma unroll
              for (int qq = 0; qq < T; ++qq) {
                key[qq] = nxt[qq];
                inm[qq] = ~0ull;
              }
              const int *__restrict__ kn = kp + (t + G);
#pragma unroll
              for (int qq = 0; qq < T; ++qq) nxt[qq] = kn[(unsigned)(qq * GM_WAVE + lane)];
              process(key, inm, t);
            }
            for (; t < n; t += G) {
              int key[T];
              unsigned long long inm[T];
#pragma unroll
              for (int qq = 0; qq < T; ++qq) {
                key[qq] = nxt[qq];
                inm[qq] = __ballot((t + qq * GM_WAVE + lane) < n);
              }
#pragma unroll
              for (int qq = 0; qq < T; ++qq) nxt[qq] = kp[min(t + G + qq * GM_WAVE + lane, n - 1)];
              process(key, inm, t);
            }
          }
          // ---- short lists: flattened (owner marks + DPP max-scan; tiles without a list boundary skip the scan) ------------
          const int incl = wave_incl_scan_add(llen);
          const int total = readlane(incl, GM_WAVE - 1);
          if (total != 0) {  // wave-uniform
            const int offp = incl - llen;
            L.desc[lane] = make_int4(key_base - offp, (int)salt_l, meta_l, offp);
            unsigned *m32 = reinterpret_cast<unsigned *>(L.marks);
            int carry = 0;
            for (int wb = 0; wb < total; wb += kMarkWindow) {
              const int wn = min(kMarkWindow, total - wb);
              const int nwords = ((wn + GM_WAVE * T - 1) / (GM_WAVE * T)) * (GM_WAVE * T / 4);
              for (int i = lane; i < nwords; i += GM_WAVE) m32[i] = 0u;
              wave_sync();
              if (llen > 0 && offp >= wb && offp < wb + kMarkWindow) L.marks[offp - wb] = (unsigned char)(lane + 1);
              wave_sync();
              for (int t = 0; t < wn; t += GM_WAVE * T) {
                int own[T], key[T], meta[T], sidx[T];
                unsigned salt[T], at[T];
                unsigned long long inm[T];
#pragma unroll
                for (int qq = 0; qq < T; ++qq) own[qq] = (int)L.marks[t + qq * GM_WAVE + lane];
#pragma unroll
                for (int qq = 0; qq < T; ++qq) {
                  if (__ballot(own[qq] != 0) == 0ull) {
                    own[qq] = carry;  // no list starts inside this tile: every position belongs to the running owner
                  } else {
                    own[qq] = max(wave_incl_scan_max(own[qq]), carry);
                    carry = readlane(own[qq], GM_WAVE - 1);
                  }
                }
                int4 dd[T];
#pragma unroll
                for (int qq = 0; qq < T; ++qq) {
                  const bool in = (wb + t + qq * GM_WAVE + lane) < total;
                  inm[qq] = __ballot(in);
                  dd[qq] = L.desc[in ? own[qq] - 1 : 0];  // unconditional LDS read
                }
#pragma unroll
                for (int qq = 0; qq < T; ++qq) {
                  const int pp = wb + t + qq * GM_WAVE + lane;
                  const bool in = pp < total;
                  key[qq] = col[in ? dd[qq].x + pp : 0];  // unconditional load (select on the index)
                  salt[qq] = (unsigned)dd[qq].y;
                  meta[qq] = dd[qq].z;
                  sidx[qq] = pp - dd[qq].w;
                }
                unsigned long long hm[T], nm[T];
                cb_probe<STAGE, T>(B, col, fallback, key, salt, inm, at, hm, nm);
                unsigned long long any_need = 0ull;
#pragma unroll
                for (int qq = 0; qq < T; ++qq) {
                  set_bits(hm[qq], meta[qq], at[qq], sidx[qq]);
                  any_need |= nm[qq];
                }
                if (any_need != 0ull) surplus(key, salt, meta, sidx, nm);  // rare
              }
              wave_sync();
            }
          }
        }
        wave_sync();
        // store the finished rows (all of them: a row without a match is a row of zeros)
        const int maxw = wave_max_nonneg(in_sub ? words : 0);
        if (maxw <= 8) {  // short rows: every lane stores its own (consecutive tasks are mostly consecutive rows of one matrix)
          for (int i = 0; i < maxw; ++i)
            if (in_sub && i < words) mat[off + (unsigned long long)i] = rb[row0 + i];
        } else {
          for (int rr = start; rr < start + cnt; ++rr) {  // wave-uniform: one coalesced store per row
            const int w_r = readlane(words, rr), r0 = readlane(row0, rr);
            const unsigned long long o_r = ((unsigned long long)(unsigned)readlane((int)(off >> 32), rr) << 32) | (unsigned)readlane((int)(unsigned)off, rr);
            if (lane < w_r) mat[o_r + (unsigned long long)lane] = rb[r0 + lane];
          }
        }
        wave_sync();
        consumed += used;
        start += cnt;
      }
    }
    __syncthreads();  // the table is rewritten by the next chunk
  }
}

int cbuild_per_cu(int stage) { return stage <= 1024 ? 5 : 3; }
hipError_t launch_cbuild(const CBuildParams &p, int stage, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(CBuildLds<1024>) * 5 <= 163840, "five workgroups per CU");
  static_assert(sizeof(CBuildLds<kCbMaxDeg>) * 3 <= 163840, "three workgroups per CU");
  static_assert(sizeof(CbWave) * kWavesPerBlock >= (size_t)kCbMaxDeg * 2, "fill counters alias the wave scratch");
  static_assert(kCbMaxDeg <= 2048 && kCbRowBuf <= 4096, "positions are 11-bit fields of a slot, bit offsets and row offsets 12-bit fields");
  if (p.trp == nullptr || p.tasks == nullptr || p.mat == nullptr) return hipErrorInvalidValue;
  const dim3 grid((unsigned)grid_blocks), block(kWavesPerBlock * GM_WAVE);
  if (stage <= 1024) hipLaunchKernelGGL((cbuild_kernel<1024>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((cbuild_kernel<kCbMaxDeg>), grid, block, 0, stream, p);
  return hipGetLastError();
}

// ---- second level of the NARROW vertices ---------------------------------------------------------------------------------------
// The matrices of a narrow chunk's vertices are contiguous in the arena (vertex order): one coalesced copy into LDS, then one thread
// per row i:  sum_{j in M_i} popc(M_i & M_j)  (clique4_warp_edge.cuh:22-27 on bit rows) -- with a topological numbering M_j has no bit
// at or below j, so the words below j / 32 are skipped.
struct alignas(16) SmallLds {
  unsigned bits[kBitWords];
  int rpl[kMaxChunkVerts + 1];
  int boff[kMaxChunkVerts + 1];  // LDS word offset of every vertex' matrix
  unsigned queue_pos;
  int pad_[3];
};

__global__ __launch_bounds__(kWavesPerBlock *GM_WAVE, 8) void clique_small_kernel(const CliqueSmallParams p) {
  __shared__ SmallLds S;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1);
  constexpr int NT = kWavesPerBlock * GM_WAVE;
  unsigned long long tot = 0;
  for (;;) {
    if (tid == 0) S.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = S.queue_pos;
    if (q >= (unsigned)p.count) break;
    const size_t pos = (size_t)p.first + (size_t)q * (size_t)p.step;
    const ChunkRec r = p.chunks[p.order ? (size_t)p.order[pos] : pos];
    const int ub = r.u_begin, nvl = r.u_end - r.u_begin, eb = r.e_begin, nel = r.e_end - r.e_begin;
    const unsigned long long b0 = p.base[ub];
    const int nwords = (int)(p.base[ub + nvl] - b0);  // <= kBitWords: the chunk table was cut for that
    if (nwords == 0) { __syncthreads(); continue; }
    for (int i = tid; i <= nvl; i += NT) {
      S.rpl[i] = p.rp[ub + i];
      S.boff[i] = (int)(p.base[ub + i] - b0);
    }
    for (int i = tid; i < nwords; i += NT) S.bits[i] = p.mat[b0 + (unsigned long long)i];
    __syncthreads();
    unsigned c = 0;
    for (int le = tid; le < nel; le += NT) {
      const int lo = cb_local_row(S.rpl, nvl, eb + le);
      const int d = S.rpl[lo + 1] - S.rpl[lo];
      if (S.boff[lo + 1] == S.boff[lo]) continue;  // d < kCbMinDeg: no matrix
      const int s = (d + 31) >> 5, i = eb + le - S.rpl[lo];
      const unsigned *M = S.bits + S.boff[lo];
      const unsigned *Mi = M + i * s;
      for (int w = 0; w < s; ++w) {
        unsigned x = Mi[w];
        while (x) {
          const int bit = __ffs((int)x) - 1;
          x &= x - 1;
          const unsigned *Mj = M + (w * 32 + bit) * s;
          for (int w2 = p.topo ? w : 0; w2 < s; ++w2) c += (unsigned)__popc(Mi[w2] & Mj[w2]);  // (topological: M_j has no bit below word w)
        }
      }
    }
    tot += (unsigned long long)c;
    __syncthreads();
  }
  const unsigned long long s0 = wave_sum_u64(tot);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_clique_small(const CliqueSmallParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(clique_small_kernel, dim3((unsigned)grid_blocks), dim3(kWavesPerBlock * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_cbuild_kernel() {}
void gm_touch_cbuild() { hipLaunchKernelGGL(gm_touch_cbuild_kernel, dim3(1), dim3(1), 0, 0); }

