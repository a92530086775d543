// Split row plan (round 5): the row plan of the rows gradient as a two-pass MSD radix partition (digits row >> 9 | row & 511)
// whose OFFSETS are built before the forward -- from the row keys alone: per-row view counts and CSR pointers, all the
// forward needs -- and whose passes run in the backward ON THE 16-BYTE VIEW RECORDS themselves.  No permutation is ever
// written or read, and the rows gradient no longer fetches one random 128-byte line per view for 16 bytes of payload (PMC:
// 4.3 GB of the 9.0 GB of the permutation form).
//
//   build (keys only):  hist_hi -> scan_tiles -> bucket_starts -> scatter<LOWS> (9-bit low digits, bucket order)
//                       -> hist_lo -> scan_rows  => counts[R], row_ptr[R + 1], offA[bucket][tile], offB[tile][digit]
//   records:            scatter<REC_A> (view order -> bucket order), then either
//                         bucket_rows_grad (C <= 64): one workgroup per bucket of 512 rows ranks + stages the bucket's
//                           records in LDS like a second pass and CONSUMES them there (sums in registers), or
//                         scatter<REC_B> (bucket order -> plan order) + the segmented reduction of attention.hip (perm = NULL)
//
// All passes are stable (wave-striped tiles, per-wavefront digit counters advanced in item order, match-any ranking), so
// the views of a row stay in view order: deterministic sums (the REC_B form: those of the permutation plan, bit for bit).
// A tile is 4096 entries of one workgroup (512 threads x 8): with 512 buckets a tile leaves 8-entry = 128-byte runs per
// bucket, staged through LDS (64 KB; two workgroups per CU) so that every run is written by consecutive lanes.
// Row keys: 512 < n_rows <= 2^18 (high digit = key >> 9 in at most 512 buckets); anything else keeps dva_row_plan.
#include "dva_common.h"

namespace dva {
namespace ps {

constexpr int IPT = 8;             // entries per thread: a tile of TILE entries is one workgroup of TILE / 8 threads
constexpr int BINS = 512;
constexpr int LO_BITS = 9;
constexpr int HEAD_INTS = 4096;   // tot[512] | bucket_start[513] | tile_start[513] (padded) | order[512] at 2048

// DVA_PLAN_TILE = 4096 (default: 512 threads, 76 KB of LDS, two workgroups per CU whose load and write-out phases overlap,
// 128-byte runs that the XCD-aware tile order below pairs up in one L2) or 8192 (1024 threads, 148 KB, one workgroup per
// CU, 256-byte runs: pass A 0.29 against 0.23 ms, the build 0.24 against 0.26 ms; before the XCD-aware order the larger
// tile was the faster one); read once, build and record passes of a plan must agree
static inline int tile_size() {
  static const int t = tune_int("DVA_PLAN_TILE", 4096) == 8192 ? 8192 : 4096;
  return t;
}

struct Layout {
  int64_t nt, nb, ntb;
  size_t off_tot, off_bstart, off_tstart, off_order, off_a, off_b, off_desc, total;
};

static inline bool eligible(int64_t n_views, int64_t n_rows) {
  return n_views > 0 && n_views <= 0x7fffffffLL && n_rows > BINS && n_rows <= (int64_t)BINS * BINS;
}

static inline Layout layout(int64_t n, int64_t n_rows) {
  Layout L;
  const int TILE = tile_size();
  L.nt = (n + TILE - 1) / TILE;
  L.nb = (n_rows + BINS - 1) / BINS;
  L.ntb = L.nt + L.nb;
  L.off_tot = 0;
  L.off_bstart = 512 * 4;
  L.off_tstart = (512 + 520) * 4;
  L.off_order = 2048 * 4;
  L.off_a = HEAD_INTS * 4;
  L.off_b = L.off_a + (size_t)L.nt * BINS * 4;
  L.off_desc = L.off_b + (size_t)L.ntb * BINS * 4;      // int4 {bucket, first entry, entries, 0} per B tile
  L.total = L.off_desc + (size_t)L.ntb * 16;
  return L;
}

// exclusive scan of one int per thread over the block (NT threads, a multiple of 64); s_w: NT / 64 ints
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int* s_w, int* total = nullptr) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();                       // s_w may still be read by a previous scan
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    const int t = s_w[i];
    before += i < w ? t : 0;
    all += t;
  }
  if (total) *total = all;
  return before + inc - v;
}

__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int hi_digit(uint32_t key, int nb) {
  const int d = (int)(key >> LO_BITS);
  return d < nb ? d : nb - 1;            // keys >= n_rows are out of contract: they stay inside the tables
}

// ---- build -------------------------------------------------------------------------------------------------------
// offA is bucket-major, offA[d][tile]: the scan over the tiles of a bucket reads one contiguous row
template <int TILE>
__global__ __launch_bounds__(TILE / IPT) void hist_hi_kernel(const uint32_t* __restrict__ keys, int64_t n, int nb,
                                                             int64_t nt, int32_t* __restrict__ offA) {
  constexpr int THREADS = TILE / IPT;
  __shared__ int h[BINS];
  for (int i = threadIdx.x; i < BINS; i += THREADS) h[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * TILE;
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int64_t e = base + k * THREADS + threadIdx.x;
    if (e < n) atomicAdd(&h[hi_digit(keys[e], nb)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += THREADS) offA[(int64_t)i * nt + blockIdx.x] = h[i];
}

// row d of offA[nb][nt]: counts -> exclusive prefix over the tiles (in place, chunks of 256 tiles); tot[d] = row sum
__global__ __launch_bounds__(256) void scan_tiles_kernel(int32_t* __restrict__ off, int64_t nt, int32_t* __restrict__ tot) {
  __shared__ int s_w[4];
  int32_t* row = off + (int64_t)blockIdx.x * nt;
  int carry = 0;
  for (int64_t t0 = 0; t0 < nt; t0 += 256) {
    const int64_t t = t0 + threadIdx.x;
    const int v = t < nt ? row[t] : 0;
    int all;
    const int ex = block_excl_scan<256>(v, s_w, &all);
    if (t < nt) row[t] = carry + ex;
    carry += all;
  }
  if (threadIdx.x == 0) tot[blockIdx.x] = carry;
}

// bucket_start[0 .. BINS], tile_start[0 .. BINS] (B tiles: every bucket is cut into its own tiles of TILE entries) and the
// descriptor {bucket, first entry, entries} of every B tile (one load per tile in the passes instead of a search)
__global__ __launch_bounds__(BINS) void bucket_starts_kernel(const int32_t* __restrict__ tot, int nb, int TILE,
                                                             int32_t* __restrict__ bucket_start,
                                                             int32_t* __restrict__ tile_start, int4* __restrict__ desc,
                                                             int32_t* __restrict__ order) {
  __shared__ int s_w[BINS / 64];
  __shared__ int s_tot[BINS];
  const int d = threadIdx.x;
  const int v = d < nb ? tot[d] : 0;
  // buckets by decreasing size (ties: by index): the bucket rows gradient hands its workgroups out largest first, so that
  // a mapping whose views crowd into a few buckets does not leave one of them for the end
  s_tot[d] = d < nb ? v : -1;
  __syncthreads();
  if (d < nb) {
    int before = 0;
    for (int j = 0; j < nb; ++j) {
      const int t = s_tot[j];
      before += (t > v || (t == v && j < d)) ? 1 : 0;
    }
    order[before] = d;
  }
  int all;
  const int bs = block_excl_scan<BINS>(v, s_w, &all);
  bucket_start[d] = bs;
  if (d == BINS - 1) bucket_start[BINS] = all;
  int allt;
  const int nt_b = (v + TILE - 1) / TILE;
  const int ts = block_excl_scan<BINS>(nt_b, s_w, &allt);
  tile_start[d] = ts;
  if (d == BINS - 1) tile_start[BINS] = allt;
  for (int j = 0; j < nt_b; ++j) {
    const int left = v - j * TILE;
    desc[ts + j] = make_int4(d, bs + j * TILE, left < TILE ? left : TILE, 0);
  }
}

// geometry of B tile `tb`: its bucket, first entry and entry count
__device__ __forceinline__ void tile_b(const int4* __restrict__ desc, int tb, int& b, int64_t& start, int& count) {
  const int4 d = desc[tb];
  b = rfl_i(d.x);
  start = (int64_t)rfl_i(d.y);
  count = rfl_i(d.z);
}

template <int TILE>
__global__ __launch_bounds__(TILE / IPT) void hist_lo_kernel(const uint16_t* __restrict__ lows, int nb,
                                                             const int32_t* __restrict__ tile_start,
                                                             const int4* __restrict__ desc, int32_t* __restrict__ offB) {
  constexpr int THREADS = TILE / IPT;
  __shared__ int h[BINS];
  if ((int)blockIdx.x >= tile_start[nb]) return;
  int b, count;
  int64_t start;
  tile_b(desc, blockIdx.x, b, start, count);
  for (int i = threadIdx.x; i < BINS; i += THREADS) h[i] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int i = k * THREADS + threadIdx.x;
    if (i < count) atomicAdd(&h[lows[start + i]], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BINS; i += THREADS) offB[(int64_t)blockIdx.x * BINS + i] = h[i];
}

// one workgroup per bucket, one thread per low digit = per row: prefix over the bucket's tiles (in place), row counts,
// CSR pointers over ALL rows (rows nobody maps to get empty segments)
__global__ __launch_bounds__(BINS) void scan_rows_kernel(int32_t* __restrict__ offB, const int32_t* __restrict__ bucket_start,
                                                         const int32_t* __restrict__ tile_start, int64_t n_rows,
                                                         int64_t n_views, int32_t* __restrict__ row_ptr,
                                                         int32_t* __restrict__ counts) {
  __shared__ int s_w[BINS / 64];
  const int b = blockIdx.x, d = threadIdx.x;
  const int tb0 = tile_start[b], tb1 = tile_start[b + 1];
  int run = 0;
  for (int tb = tb0; tb < tb1; ++tb) {
    const int v = offB[(int64_t)tb * BINS + d];
    offB[(int64_t)tb * BINS + d] = run;
    run += v;
  }
  const int64_t r = (int64_t)b * BINS + d;
  if (counts && r < n_rows) counts[r] = run;
  const int excl = block_excl_scan<BINS>(run, s_w);
  if (r <= n_rows) row_ptr[r] = bucket_start[b] + excl;
  if (r + 1 == n_rows && d == BINS - 1) row_ptr[n_rows] = (int32_t)n_views;
}

// ---- the stable scatter of one tile ------------------------------------------------------------------------------
enum { MODE_LOWS = 0, MODE_REC_A = 1, MODE_REC_B = 2 };
template <int MODE> struct Elem { typedef uint4 type; };
template <> struct Elem<MODE_LOWS> { typedef uint32_t type; };
__device__ __forceinline__ uint32_t key_of(const uint4& e) { return e.w; }
__device__ __forceinline__ uint32_t key_of(uint32_t e) { return e; }

// lanes of the wavefront whose (valid) 9-bit digit equals this lane's
__device__ __forceinline__ uint64_t match_digit(int d, bool valid) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < LO_BITS; ++bit) {
    const bool one = (d >> bit) & 1;
    const uint64_t m = __ballot(one);
    peers &= one ? m : ~m;
  }
  return peers;
}

// keys NULL (REC_A): word 3 of the records already is the row key (dva_chain_attn_bwd writes it).
// One workgroup per tile, in an XCD-aware order (xcd_order, gridDim.x a multiple of 8): workgroup i runs on XCD i % 8, so
// XCD x takes the x-th eighth of the tiles and its CUs hold NEIGHBOURING tiles at any time -- the two halves of a 128-byte
// line shared by the runs of two neighbouring tiles then meet in one L2 instead of leaving two partial writes (0.66 ->
// 0.56 ms for the two record passes).  Persistent workgroups that request the next tile's entries under the write-out of
// the current one were measured slower (0.60 ms: 119 registers, and the tiles of a CU no longer neighbour those of the
// other CUs of its XCD in time).
// RW = 8 (round 6, MODE_REC_A only): the 32-byte records of the fp32 attention backward {point | 4 fp32 weights | 3 pad
// words}: five payload words travel, the row key (from `keys`) is written into word 7 of the record on the way, so that
// the bucket kernel finds it where the 16-byte records carry theirs (the last word).  Two 16-byte LDS planes.
template <int MODE, int TILE, int RW = 4>
__global__ __launch_bounds__(TILE / IPT) void scatter_kernel(const uint32_t* __restrict__ keys, const uint4* __restrict__ src,
                                                             void* __restrict__ dst, int64_t n, int nb, int64_t n_rows,
                                                             int64_t nt, const int32_t* __restrict__ bucket_start,
                                                             const int32_t* __restrict__ tile_start,
                                                             const int4* __restrict__ desc,
                                                             const int32_t* __restrict__ off,
                                                             const int32_t* __restrict__ row_ptr, int xcd_order) {
  constexpr int THREADS = TILE / IPT, WAVES = THREADS / 64;
  typedef typename Elem<MODE>::type E;
  static_assert(RW == 4 || (RW == 8 && MODE == MODE_REC_A), "32-byte records only go through pass A");
  __shared__ E s_stage[TILE];
  __shared__ uint4 s_stage_hi[RW == 8 ? TILE : 1];
  __shared__ uint16_t s_cnt[WAVES][BINS];
  __shared__ int s_lstart[BINS];
  __shared__ int s_base[BINS];
  __shared__ int s_w[WAVES];
  static_assert(THREADS >= BINS, "one thread per digit in the scan over the wavefronts");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n_tiles = MODE == MODE_REC_B ? tile_start[nb] : (int)nt;
  const int tile = xcd_order ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (tile >= n_tiles) return;
  int b = 0, count;
  int64_t start;
  if (MODE == MODE_REC_B) {
    tile_b(desc, tile, b, start, count);
  } else {
    start = (int64_t)tile * TILE;
    const int64_t left = n - start;
    count = (int)(left < TILE ? left : TILE);
  }
  // ---- loads first (8 independent requests per lane), tables while they fly
  uint32_t kk[IPT], p0[IPT], p1[IPT], p2[IPT];     // key | the three payload words of a record (scalars: no scratch)
  uint32_t p3[RW == 8 ? IPT : 1], p4[RW == 8 ? IPT : 1];
  bool ok[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int idx = w * (64 * IPT) + i * 64 + lane;
    ok[i] = idx < count;
    const int64_t g = start + (ok[i] ? idx : 0);
    if constexpr (MODE == MODE_LOWS) {
      kk[i] = keys[g];
      p0[i] = p1[i] = p2[i] = 0u;
    } else if constexpr (RW == 8) {
      const uint4 r = src[2 * g];
      p0[i] = r.x, p1[i] = r.y, p2[i] = r.z, p3[i] = r.w;
      p4[i] = src[2 * g + 1].x;
      kk[i] = keys[g];
    } else {
      const uint4 r = src[g];
      p0[i] = r.x, p1[i] = r.y, p2[i] = r.z;
      kk[i] = (MODE == MODE_REC_A && keys) ? keys[g] : r.w;      // (uniform: a kernel argument)
    }
  }
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);
#pragma unroll
    for (int k = 0; k < WAVES * BINS / 2 / THREADS; ++k) z[k * THREADS + tid] = 0u;
  }
  if (tid < BINS) {
    int base;
    if (MODE == MODE_REC_B) {
      const int64_t r = (int64_t)b * BINS + tid;
      base = (r <= n_rows ? row_ptr[r] : 0) + off[(int64_t)tile * BINS + tid];
    } else {
      base = tid < nb ? bucket_start[tid] + off[(int64_t)tid * nt + tile] : 0;
    }
    s_base[tid] = base;
  }
  __syncthreads();
  // ---- stable rank inside the wavefront's 512 entries: counter of the digit before this item + lanes below
  int dg[IPT], rank[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const uint32_t key = kk[i];
    dg[i] = MODE == MODE_REC_B ? (int)(key & (BINS - 1)) : hi_digit(key, nb);
    const uint64_t peers = match_digit(dg[i], ok[i]);
    const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
    const int prev = s_cnt[w][dg[i]];
    if (ok[i] && below == 0) s_cnt[w][dg[i]] = (uint16_t)(prev + __popcll(peers));
    rank[i] = prev + below;
  }
  __syncthreads();
  // ---- counters -> prefix over the wavefronts; digit starts inside the tile
  int run = 0;
  if (tid < BINS) {
#pragma unroll
    for (int k = 0; k < WAVES; ++k) {
      const int c = s_cnt[k][tid];
      s_cnt[k][tid] = (uint16_t)run;
      run += c;
    }
  }
  const int ls = block_excl_scan<THREADS>(tid < BINS ? run : 0, s_w);
  if (tid < BINS) s_lstart[tid] = ls;
  __syncthreads();
  // ---- stage in tile order of the digits
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    if (ok[i]) {
      const int pos = s_lstart[dg[i]] + s_cnt[w][dg[i]] + rank[i];
      if constexpr (MODE == MODE_LOWS) {
        s_stage[pos] = kk[i];
      } else if constexpr (RW == 8) {
        s_stage[pos] = make_uint4(p0[i], p1[i], p2[i], p3[i]);
        s_stage_hi[pos] = make_uint4(p4[i], 0u, 0u, kk[i]);
      } else {
        s_stage[pos] = make_uint4(p0[i], p1[i], p2[i], kk[i]);
      }
    }
  }
  __syncthreads();
  // ---- runs of one digit leave through consecutive lanes
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int j = k * THREADS + tid;
    if (j < count) {
      const E v = s_stage[j];
      uint4 vh = make_uint4(0u, 0u, 0u, 0u);
      if constexpr (RW == 8) vh = s_stage_hi[j];
      const uint32_t key = RW == 8 ? vh.w : key_of(v);
      const int d = MODE == MODE_REC_B ? (int)(key & (BINS - 1)) : hi_digit(key, nb);
      int64_t g = (int64_t)s_base[d] + (j - s_lstart[d]);
      g = g < n ? g : n - 1;
      if constexpr (MODE == MODE_LOWS) {
        reinterpret_cast<uint16_t*>(dst)[g] = (uint16_t)(key & (BINS - 1));
      } else if constexpr (RW == 8) {
        reinterpret_cast<uint4*>(dst)[2 * g] = v;
        reinterpret_cast<uint4*>(dst)[2 * g + 1] = vh;
      } else {
        reinterpret_cast<uint4*>(dst)[g] = v;
      }
    }
  }
}


// ---- the rows gradient straight from the bucket-ordered records (pass A's output) ---------------------------------
// One workgroup per bucket = 512 consecutive map rows, whose C fp32 accumulators ARE the workgroup's registers (1024
// threads x 32 at C = 64: lane team `grp` of C / 8 lanes owns rows grp, grp + GROUPS, ...).  The bucket's tiles pass
// through LDS one after the other: ranked by the low digit and staged in row order exactly like pass B -- but instead of
// being written back (16 bytes per view) and read again by the rows gradient (16 more), the staged records are
// consumed where they lie: per row its records in view order, the 128-byte grad_out row of each record's point
// gathered U at a time.  Deterministic (a row is summed by one lane team in view order); the order differs from the
// segmented reduction of attention.hip (which splits a row over 8 lane slots), so the two agree to fp32 rounding, not
// bit for bit.  bf16 in, bf16 out, C in {32, 64}.
template <int C, int BT>
__global__ __launch_bounds__(1024) void bucket_rows_grad_kernel(const uint4* __restrict__ rec, const bf16_t* __restrict__ gout,
                                                                bf16_t* __restrict__ grows, int64_t n_rows, int G,
                                                                const int32_t* __restrict__ bucket_start,
                                                                const int32_t* __restrict__ order) {
  // BT = records ranked + staged at a time (the kernel's own tile: the bucket is one contiguous, view-ordered range).
  // The 256 workgroups in flight walk their buckets side by side, so at any time they all gather grad_out rows of the
  // same window of points -- BT / 65536 of them at the headline's bucket size: a smaller BT narrows the window towards
  // what an L2 holds (and leaves LDS for nothing else: one workgroup per CU is set by the 128 registers anyway).
  constexpr int THREADS = 1024, WAVES = THREADS / 64, IPB = BT / THREADS;
  constexpr int LPR = C / 8, GROUPS = THREADS / LPR, RPT = BINS / GROUPS, U = BT >= 8192 ? 8 : 4;
  __shared__ uint4 s_stage[BT];
  __shared__ uint16_t s_cnt[WAVES][BINS];
  __shared__ int s_lstart[BINS + 1];
  __shared__ int s_w[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = order[blockIdx.x];          // largest bucket first
  const int grp = tid / LPR, cl = tid % LPR, gch = cl / (LPR / G);
  float acc[RPT][8];
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[rr][k] = 0.f;
  const int64_t b0 = bucket_start[b], b1 = bucket_start[b + 1];
  for (int64_t start = b0; start < b1; start += BT) {
    const int count = (int)(b1 - start < BT ? b1 - start : BT);
    uint32_t kk[IPB], p0[IPB], p1[IPB], p2[IPB];
    bool ok[IPB];
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      const int idx = w * (64 * IPB) + i * 64 + lane;
      ok[i] = idx < count;
      const uint4 r = rec[start + (ok[i] ? idx : 0)];
      p0[i] = r.x, p1[i] = r.y, p2[i] = r.z, kk[i] = r.w;
    }
    {
      uint32_t* z = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);
#pragma unroll
      for (int k = 0; k < WAVES * BINS / 2 / THREADS; ++k) z[k * THREADS + tid] = 0u;
    }
    __syncthreads();                       // (also: the previous tile's records are consumed)
    int dg[IPB], rank[IPB];
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      dg[i] = (int)(kk[i] & (BINS - 1));
      const uint64_t peers = match_digit(dg[i], ok[i]);
      const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
      const int prev = s_cnt[w][dg[i]];
      if (ok[i] && below == 0) s_cnt[w][dg[i]] = (uint16_t)(prev + __popcll(peers));
      rank[i] = prev + below;
    }
    __syncthreads();
    int run = 0;
    if (tid < BINS) {
#pragma unroll
      for (int k = 0; k < WAVES; ++k) {
        const int c = s_cnt[k][tid];
        s_cnt[k][tid] = (uint16_t)run;
        run += c;
      }
    }
    const int ls = block_excl_scan<THREADS>(tid < BINS ? run : 0, s_w);
    if (tid < BINS) s_lstart[tid] = ls;
    if (tid == BINS - 1) s_lstart[BINS] = ls + run;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      if (ok[i]) s_stage[s_lstart[dg[i]] + s_cnt[w][dg[i]] + rank[i]] = make_uint4(p0[i], p1[i], p2[i], kk[i]);
    }
    __syncthreads();
    // ---- consume: row d = grp + GROUPS * rr, its records [s_lstart[d], s_lstart[d + 1]) in view order
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int d = grp + GROUPS * rr;
      const int beg = s_lstart[d], end = s_lstart[d + 1];
      for (int i0 = beg; i0 < end; i0 += U) {
        uint4 raw[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool in = i0 + u < end;
          const uint32_t* rv = reinterpret_cast<const uint32_t*>(&s_stage[in ? i0 + u : beg]);
          const int64_t p = (int)rv[0];
          const uint32_t w2 = rv[1 + (gch >> 1)];
          sc[u] = in ? __uint_as_float((gch & 1) ? (w2 & 0xffff0000u) : (w2 << 16)) : 0.f;
          raw[u] = make_uint4(0u, 0u, 0u, 0u);
          if (in) raw[u] = *reinterpret_cast<const uint4*>(gout + p * C + cl * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint4 r = raw[u];
          acc[rr][0] = fmaf(__uint_as_float(r.x << 16), sc[u], acc[rr][0]);
          acc[rr][1] = fmaf(__uint_as_float(r.x & 0xffff0000u), sc[u], acc[rr][1]);
          acc[rr][2] = fmaf(__uint_as_float(r.y << 16), sc[u], acc[rr][2]);
          acc[rr][3] = fmaf(__uint_as_float(r.y & 0xffff0000u), sc[u], acc[rr][3]);
          acc[rr][4] = fmaf(__uint_as_float(r.z << 16), sc[u], acc[rr][4]);
          acc[rr][5] = fmaf(__uint_as_float(r.z & 0xffff0000u), sc[u], acc[rr][5]);
          acc[rr][6] = fmaf(__uint_as_float(r.w << 16), sc[u], acc[rr][6]);
          acc[rr][7] = fmaf(__uint_as_float(r.w & 0xffff0000u), sc[u], acc[rr][7]);
        }
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr) {
    const int64_t r = (int64_t)b * BINS + grp + GROUPS * rr;
    if (r < n_rows) {
      const uint4 o = {pack_bf16x2(acc[rr][0], acc[rr][1]), pack_bf16x2(acc[rr][2], acc[rr][3]),
                       pack_bf16x2(acc[rr][4], acc[rr][5]), pack_bf16x2(acc[rr][6], acc[rr][7])};
      *reinterpret_cast<uint4*>(grows + r * C + cl * 8) = o;
    }
  }
}

// The fp32 twin (round 6; the reference's default arithmetic, no autocast): 32-byte records in bucket order (pass A with
// RW = 8: {point | w0 w1 w2 | w3 0 0 key}), fp32 grad_out rows (256 bytes at C = 64: a lane team of C / 4 lanes, four
// channels per lane), fp32 rows out.  Same structure as the bf16 kernel: ranked by the low digit, staged in row order in
// two 16-byte LDS planes, consumed in view order by the team that owns the row (deterministic).
template <int C, int BT>
__global__ __launch_bounds__(1024) void bucket_rows_grad_f32_kernel(const uint4* __restrict__ rec, const float* __restrict__ gout,
                                                                    float* __restrict__ grows, int64_t n_rows, int G,
                                                                    const int32_t* __restrict__ bucket_start,
                                                                    const int32_t* __restrict__ order) {
  constexpr int THREADS = 1024, WAVES = THREADS / 64, IPB = BT / THREADS;
  constexpr int LPR = C / 4, GROUPS = THREADS / LPR, RPT = BINS / GROUPS, U = 4;
  __shared__ uint4 s_lo[BT], s_hi[BT];
  __shared__ uint16_t s_cnt[WAVES][BINS];
  __shared__ int s_lstart[BINS + 1];
  __shared__ int s_w[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = order[blockIdx.x];          // largest bucket first
  const int grp = tid / LPR, cl = tid % LPR, gch = cl / (LPR / G);
  float acc[RPT][4];
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[rr][k] = 0.f;
  const int64_t b0 = bucket_start[b], b1 = bucket_start[b + 1];
  for (int64_t start = b0; start < b1; start += BT) {
    const int count = (int)(b1 - start < BT ? b1 - start : BT);
    uint32_t kk[IPB], p0[IPB], p1[IPB], p2[IPB], p3[IPB], p4[IPB];
    bool ok[IPB];
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      const int idx = w * (64 * IPB) + i * 64 + lane;
      ok[i] = idx < count;
      const int64_t g = start + (ok[i] ? idx : 0);
      const uint4 r = rec[2 * g], rh = rec[2 * g + 1];
      p0[i] = r.x, p1[i] = r.y, p2[i] = r.z, p3[i] = r.w, p4[i] = rh.x, kk[i] = rh.w;
    }
    {
      uint32_t* z = reinterpret_cast<uint32_t*>(&s_cnt[0][0]);
#pragma unroll
      for (int k = 0; k < WAVES * BINS / 2 / THREADS; ++k) z[k * THREADS + tid] = 0u;
    }
    __syncthreads();                       // (also: the previous tile's records are consumed)
    int dg[IPB], rank[IPB];
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      dg[i] = (int)(kk[i] & (BINS - 1));
      const uint64_t peers = match_digit(dg[i], ok[i]);
      const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
      const int prev = s_cnt[w][dg[i]];
      if (ok[i] && below == 0) s_cnt[w][dg[i]] = (uint16_t)(prev + __popcll(peers));
      rank[i] = prev + below;
    }
    __syncthreads();
    int run = 0;
    if (tid < BINS) {
#pragma unroll
      for (int k = 0; k < WAVES; ++k) {
        const int c = s_cnt[k][tid];
        s_cnt[k][tid] = (uint16_t)run;
        run += c;
      }
    }
    const int ls = block_excl_scan<THREADS>(tid < BINS ? run : 0, s_w);
    if (tid < BINS) s_lstart[tid] = ls;
    if (tid == BINS - 1) s_lstart[BINS] = ls + run;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPB; ++i) {
      if (ok[i]) {
        const int pos = s_lstart[dg[i]] + s_cnt[w][dg[i]] + rank[i];
        s_lo[pos] = make_uint4(p0[i], p1[i], p2[i], p3[i]);
        s_hi[pos] = make_uint4(p4[i], 0u, 0u, kk[i]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int d = grp + GROUPS * rr;
      const int beg = s_lstart[d], end = s_lstart[d + 1];
      for (int i0 = beg; i0 < end; i0 += U) {
        float4 raw[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool in = i0 + u < end;
          const int at = in ? i0 + u : beg;
          const uint32_t* lo = reinterpret_cast<const uint32_t*>(&s_lo[at]);
          const int64_t p = (int)lo[0];
          const uint32_t wb = gch < 3 ? lo[1 + gch] : reinterpret_cast<const uint32_t*>(&s_hi[at])[0];
          sc[u] = in ? __uint_as_float(wb) : 0.f;
          raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (in) raw[u] = *reinterpret_cast<const float4*>(gout + p * C + cl * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc[rr][0] = fmaf(raw[u].x, sc[u], acc[rr][0]);
          acc[rr][1] = fmaf(raw[u].y, sc[u], acc[rr][1]);
          acc[rr][2] = fmaf(raw[u].z, sc[u], acc[rr][2]);
          acc[rr][3] = fmaf(raw[u].w, sc[u], acc[rr][3]);
        }
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr) {
    const int64_t r = (int64_t)b * BINS + grp + GROUPS * rr;
    if (r < n_rows)
      *reinterpret_cast<float4*>(grows + r * C + cl * 4) = make_float4(acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]);
  }
}

template <int C>
static void bucket_rows_grad(const uint4* rec, const bf16_t* gout, bf16_t* grows, int64_t n_rows, int G, int nb,
                             const int32_t* bstart, const int32_t* order, hipStream_t s) {
  static const int bt = tune_int("DVA_PLAN_BT", 8192);
  if (bt == 2048)
    hipLaunchKernelGGL((bucket_rows_grad_kernel<C, 2048>), dim3(nb), dim3(1024), 0, s, rec, gout, grows, n_rows, G, bstart, order);
  else if (bt == 4096)
    hipLaunchKernelGGL((bucket_rows_grad_kernel<C, 4096>), dim3(nb), dim3(1024), 0, s, rec, gout, grows, n_rows, G, bstart, order);
  else
    hipLaunchKernelGGL((bucket_rows_grad_kernel<C, 8192>), dim3(nb), dim3(1024), 0, s, rec, gout, grows, n_rows, G, bstart, order);
}

}  // namespace ps
}  // namespace dva

namespace dva {
namespace ps {
struct Tables {
  int32_t *tot, *bstart, *tstart, *order, *offA, *offB;
  int4* desc;
};
static inline Tables tables_of(void* tables, const Layout& L) {
  char* tb = (char*)tables;
  return {(int32_t*)(tb + L.off_tot), (int32_t*)(tb + L.off_bstart), (int32_t*)(tb + L.off_tstart),
          (int32_t*)(tb + L.off_order), (int32_t*)(tb + L.off_a), (int32_t*)(tb + L.off_b), (int4*)(tb + L.off_desc)};
}

// grid of a scatter pass: its tiles rounded up to a multiple of 8 for the XCD-aware order (the extra workgroups exit);
// DVA_PLAN_XCD=0: plain order, the A/B
static inline bool xcd_on() {
  static const int on = tune_int("DVA_PLAN_XCD", 1);
  return on != 0;
}
static inline unsigned scatter_grid(int64_t tiles) { return (unsigned)(xcd_on() ? (tiles + 7) / 8 * 8 : tiles); }

template <int TILE>
static void build(const uint32_t* keys, int64_t n, int64_t n_rows, int32_t* row_ptr, int32_t* counts, const Layout& L,
                  const Tables& T, void* scratch, hipStream_t s) {
  constexpr int THREADS = TILE / IPT;
  const int nb = (int)L.nb;
  hipLaunchKernelGGL(hist_hi_kernel<TILE>, dim3((unsigned)L.nt), dim3(THREADS), 0, s, keys, n, nb, L.nt, T.offA);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(nb), dim3(256), 0, s, T.offA, L.nt, T.tot);
  hipLaunchKernelGGL(bucket_starts_kernel, dim3(1), dim3(BINS), 0, s, T.tot, nb, TILE, T.bstart, T.tstart, T.desc, T.order);
  hipLaunchKernelGGL((scatter_kernel<MODE_LOWS, TILE>), dim3(scatter_grid(L.nt)), dim3(THREADS), 0, s, keys,
                     (const uint4*)nullptr, scratch, n, nb, n_rows, L.nt, T.bstart, T.tstart, T.desc, T.offA,
                     (const int32_t*)nullptr, (int)xcd_on());
  hipLaunchKernelGGL(hist_lo_kernel<TILE>, dim3((unsigned)L.ntb), dim3(THREADS), 0, s, (const uint16_t*)scratch, nb,
                     T.tstart, T.desc, T.offB);
  hipLaunchKernelGGL(scan_rows_kernel, dim3(nb), dim3(BINS), 0, s, T.offB, T.bstart, T.tstart, n_rows, n, row_ptr, counts);
}

template <int TILE>
static void sort_records(const uint32_t* keys, const uint4* rec, int64_t n, int64_t n_rows, const int32_t* row_ptr,
                         const Layout& L, const Tables& T, void* buf, void* out, hipStream_t s) {
  constexpr int THREADS = TILE / IPT;
  const int nb = (int)L.nb;
  hipLaunchKernelGGL((scatter_kernel<MODE_REC_A, TILE>), dim3(scatter_grid(L.nt)), dim3(THREADS), 0, s, keys, rec,
                     buf, n, nb, n_rows, L.nt, T.bstart, T.tstart, T.desc, T.offA, (const int32_t*)nullptr,
                     (int)xcd_on());
  if (!out) return;                 // pass A only: the caller consumes the bucket-ordered records (dva_plan_split_rows_grad)
  hipLaunchKernelGGL((scatter_kernel<MODE_REC_B, TILE>), dim3(scatter_grid(L.ntb)), dim3(THREADS), 0, s,
                     (const uint32_t*)nullptr, (const uint4*)buf, out, n, nb, n_rows, L.nt, T.bstart, T.tstart, T.desc,
                     T.offB, row_ptr, (int)xcd_on());
}
}  // namespace ps
}  // namespace dva

using namespace dva;

extern "C" {

int64_t dva_plan_split_table_bytes(int64_t n_views, int64_t n_rows) {
  if (n_views < 0 || n_rows < 0) return DVA_ERR_INVALID;
  if (!ps::eligible(n_views, n_rows)) return DVA_ERR_UNSUPPORTED;
  return (int64_t)ps::layout(n_views, n_rows).total;
}

int dva_plan_split_build(const int32_t* row_idx, int64_t n_views, int64_t n_rows, int32_t* row_ptr, int32_t* counts,
                         void* tables, int64_t tables_bytes, void* scratch, int64_t scratch_bytes, void* stream) {
  if (n_views < 0 || n_rows < 0) return DVA_ERR_INVALID;
  if (!ps::eligible(n_views, n_rows)) return DVA_ERR_UNSUPPORTED;
  if (!row_idx || !row_ptr || !tables || !scratch) return DVA_ERR_INVALID;
  const ps::Layout L = ps::layout(n_views, n_rows);
  if ((int64_t)L.total > tables_bytes || scratch_bytes < n_views * 2) return DVA_ERR_INVALID;
  const ps::Tables T = ps::tables_of(tables, L);
  if (ps::tile_size() == 4096)
    ps::build<4096>((const uint32_t*)row_idx, n_views, n_rows, row_ptr, counts, L, T, scratch, (hipStream_t)stream);
  else
    ps::build<8192>((const uint32_t*)row_idx, n_views, n_rows, row_ptr, counts, L, T, scratch, (hipStream_t)stream);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_plan_split_sort_records(const int32_t* row_idx, const void* rec, int64_t n_views, int64_t n_rows,
                                const int32_t* row_ptr, const void* tables, int64_t tables_bytes, void* buf,
                                void* rec_sorted, void* stream) {
  if (n_views < 0 || n_rows < 0) return DVA_ERR_INVALID;
  if (!ps::eligible(n_views, n_rows)) return DVA_ERR_UNSUPPORTED;
  if (!rec || !row_ptr || !tables || !buf || buf == rec || buf == rec_sorted) return DVA_ERR_INVALID;
  if (((uintptr_t)rec % 16) || ((uintptr_t)buf % 16) || ((uintptr_t)rec_sorted % 16)) return DVA_ERR_UNSUPPORTED;
  const ps::Layout L = ps::layout(n_views, n_rows);
  if ((int64_t)L.total > tables_bytes) return DVA_ERR_INVALID;
  const ps::Tables T = ps::tables_of(const_cast<void*>(tables), L);
  if (ps::tile_size() == 4096)
    ps::sort_records<4096>((const uint32_t*)row_idx, (const uint4*)rec, n_views, n_rows, row_ptr, L, T, buf, rec_sorted,
                           (hipStream_t)stream);
  else
    ps::sort_records<8192>((const uint32_t*)row_idx, (const uint4*)rec, n_views, n_rows, row_ptr, L, T, buf, rec_sorted,
                           (hipStream_t)stream);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// Pass A on the 32-BYTE records of the fp32 attention backward (dva_chain_attn_bwd_f32: {point | 4 fp32 weights | 3 pad
// words}, view order) -> bucket order in `buf` [n_views][32], the row key (row_idx, required) written into word 7.
int dva_plan_split_sort_records32(const int32_t* row_idx, const void* rec, int64_t n_views, int64_t n_rows,
                                  const void* tables, int64_t tables_bytes, void* buf, void* stream) {
  if (n_views < 0 || n_rows < 0) return DVA_ERR_INVALID;
  if (!ps::eligible(n_views, n_rows) || ps::tile_size() != 4096) return DVA_ERR_UNSUPPORTED;      // (two LDS planes of a tile)
  if (!row_idx || !rec || !tables || !buf || buf == rec) return DVA_ERR_INVALID;
  if (((uintptr_t)rec % 16) || ((uintptr_t)buf % 16)) return DVA_ERR_UNSUPPORTED;
  const ps::Layout L = ps::layout(n_views, n_rows);
  if ((int64_t)L.total > tables_bytes) return DVA_ERR_INVALID;
  const ps::Tables T = ps::tables_of(const_cast<void*>(tables), L);
  constexpr int TILE = 4096, THREADS = TILE / ps::IPT;
  hipLaunchKernelGGL((ps::scatter_kernel<ps::MODE_REC_A, TILE, 8>), dim3(ps::scatter_grid(L.nt)), dim3(THREADS), 0,
                     (hipStream_t)stream, (const uint32_t*)row_idx, (const uint4*)rec, buf, n_views, (int)L.nb, n_rows,
                     L.nt, T.bstart, T.tstart, T.desc, T.offA, (const int32_t*)nullptr, (int)ps::xcd_on());
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_plan_split_rows_grad(const void* grad_out, const void* bucket_rec, int64_t n_views, int64_t n_rows,
                             const void* tables, int64_t tables_bytes, void* grad_rows, int32_t C, int32_t G, int32_t dtype,
                             int32_t out_dtype, void* stream) {
  if (n_views < 0 || n_rows < 0 || C <= 0 || G <= 0) return DVA_ERR_INVALID;
  const bool f32 = dtype == DVA_F32 && out_dtype == DVA_F32;       // 32-byte records (dva_plan_split_sort_records32)
  if (!ps::eligible(n_views, n_rows) || (!f32 && (dtype != DVA_BF16 || out_dtype != DVA_BF16)) ||
      (C != 32 && C != 64) || (G != 1 && G != 2 && G != 4) || ((C / (f32 ? 4 : 8)) % G) != 0)
    return DVA_ERR_UNSUPPORTED;
  if (!grad_out || !bucket_rec || !tables || !grad_rows) return DVA_ERR_INVALID;
  if (((uintptr_t)bucket_rec % 16) || ((uintptr_t)grad_out % 16) || ((uintptr_t)grad_rows % 16)) return DVA_ERR_UNSUPPORTED;
  const ps::Layout L = ps::layout(n_views, n_rows);
  if ((int64_t)L.total > tables_bytes) return DVA_ERR_INVALID;
  const ps::Tables T = ps::tables_of(const_cast<void*>(tables), L);
  hipStream_t s = (hipStream_t)stream;
  const int nb = (int)L.nb;
  if (f32) {
    if (C == 64)
      hipLaunchKernelGGL((ps::bucket_rows_grad_f32_kernel<64, 4096>), dim3(nb), dim3(1024), 0, s, (const uint4*)bucket_rec,
                         (const float*)grad_out, (float*)grad_rows, n_rows, (int)G, T.bstart, T.order);
    else
      hipLaunchKernelGGL((ps::bucket_rows_grad_f32_kernel<32, 4096>), dim3(nb), dim3(1024), 0, s, (const uint4*)bucket_rec,
                         (const float*)grad_out, (float*)grad_rows, n_rows, (int)G, T.bstart, T.order);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (C == 64)
    ps::bucket_rows_grad<64>((const uint4*)bucket_rec, (const bf16_t*)grad_out, (bf16_t*)grad_rows, n_rows, (int)G, nb,
                             T.bstart, T.order, s);
  else
    ps::bucket_rows_grad<32>((const uint4*)bucket_rec, (const bf16_t*)grad_out, (bf16_t*)grad_rows, n_rows, (int)G, nb,
                             T.bstart, T.order, s);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
