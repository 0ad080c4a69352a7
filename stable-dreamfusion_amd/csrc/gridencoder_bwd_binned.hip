// gridencoder_bwd_binned.hip — table-gradient scatter of the hash-grid encoder without a global
// atomic per contribution (D = 3, C = 2: the configuration the -O path trains).
//
// Why: on MI355X a device-scope atomic is executed at the memory side of the fabric (the per-XCD
// L2s are not coherent with each other), at ~18 G atomics/s for random addresses and far less when
// neighbouring samples of a ray hammer the same coarse cell. The reference's scheme — one atomic per
// (sample, level, corner), gridencoder.cu:252-349 — needs 128 atomics per sample and took 9.9 ms per
// call at 4.2e5 samples (rocprof, profiles/r01_v1_*). This file replaces it by a two-kernel
// "bin, then reduce in LDS" scatter:
//
//   K1 bin     every (sample, level, corner) contribution becomes an item: float tables {row, float2} (12 bytes), half tables
//              one 12-byte item per x-PAIR of corners {row0 | (row0 ^ row1) << 20, half2, half2} (see Item<true>).
//              Lanes hold consecutive samples of a ray, so at coarse and middle levels runs of lanes
//              hit the same table row: a wave-level segmented scan folds each run into one item.
//              Items are binned by row range (2048 rows per bucket) with an LDS histogram and ONE
//              global atomic per (workgroup, bucket) that reserves a slice of the bucket's item list.
//   K2 reduce  one workgroup per bucket (several for over-full coarse buckets) sums its items into
//              per-wave 2048 x float2 accumulators in LDS (plain read-modify-writes, same-row lanes
//              serialised by a ticket byte; LDS float atomics are ~80x slower) and adds the result to
//              the table gradient with plain, coalesced read-modify-writes (the workgroup owns those rows).
//
// A bucket that is asked for more slots than its list holds (cursor > capacity) is complete for any input distribution
// too. Half tables (the -O path): K1 simply drops what does not fit; K2 ignores the list of such a bucket altogether, and
// two more kernels between K1 and K2 — which find nothing to do and exit in every launch of the training loop — zero the
// bucket's 64-bit fixed-point SPILL accumulator in global memory (k_grid_bwd_spill_zero) and re-derive ALL contributions
// of the overflowed buckets from the samples, adding them there with 64-bit integer atomics (k_grid_bwd_spill: K1's
// arithmetic, bit for bit). The table gradient is therefore the exactly summed, once-rounded value whatever the
// distribution of the samples and whatever the order of arrival — bit-reproducible (the reference rounds every
// contribution AND every partial sum to half, gridencoder.cu:334-340) — and K1's hot path carries no overflow code.
// Float tables fall back to float atomics, which is what the reference does for every contribution.
#include "grid_point.h"
#include "dev_stamps.h"

#include <map>
#include <mutex>

using namespace sdfx;
using namespace sdfx::grid;

namespace {

SDFX_DEV_CTL_DEFINE   // devtools build: per-workgroup timestamps, ablation bits (dev_stamps.h); nothing in the product build

#ifndef SDFX_BUCKET_LOG2
#define SDFX_BUCKET_LOG2 11   // measurement aid: -DSDFX_BUCKET_LOG2=12 builds the 4096-row variant (64 KiB of accumulators in K2)
#endif
constexpr uint32_t kBucketRowsLog2 = SDFX_BUCKET_LOG2;
constexpr uint32_t kBucketRows = 1u << kBucketRowsLog2;  // rows per bucket (16 KiB of float2 accumulators)
#ifndef SDFX_MAX_BUCKETS
#define SDFX_MAX_BUCKETS 512   // measurement aid: -DSDFX_MAX_BUCKETS=256 -DSDFX_BIN_THREADS=256 builds K1 with 256-sample tiles (levels up to 2^19 rows)
#endif
constexpr uint32_t kMaxBucketsPerLevel = SDFX_MAX_BUCKETS;   // levels up to 2^20 rows
#ifndef SDFX_BIN_THREADS
#define SDFX_BIN_THREADS 512   // measurement aid: -DSDFX_BIN_THREADS=1024 builds K1 with 1024-sample tiles (half as many reservations per item)
#endif
constexpr uint32_t kBinThreads = SDFX_BIN_THREADS;
constexpr uint32_t kPointsPerThread = 1;
static_assert(kMaxBucketsPerLevel <= kBinThreads, "K1 scans the bucket histogram with one thread per bucket");
constexpr uint32_t kReduceThreads = SDFX_BUCKET_LOG2 >= 12 ? 128 : 256;   // (float tables: per-wave private accumulators must fit the LDS)
constexpr uint32_t kItemsPerSplit = 131072;              // a bucket holding more items than this is reduced by several workgroups
constexpr uint32_t kItemsPerSplitCoarse = 65536;         // ... for levels of few buckets, each of which gets a large share of the batch
constexpr uint32_t kCoarseBuckets = 128u >> (SDFX_BUCKET_LOG2 - 11);   // (levels of at most 2^18 rows, whatever the bucket size)
constexpr uint32_t kMaxSplits = 512;
constexpr uint32_t kNoSharedAcc = 0xFFFFFFFFu;

struct BinPlan {
    uint32_t bucket_first[kMaxLevels + 1];  // first bucket id of each level (prefix sum)
    uint32_t split_first[kMaxLevels + 1];   // first K2 workgroup id of each level
    uint32_t splits[kMaxLevels];            // K2 workgroups per bucket of this level
    uint32_t cap[kMaxLevels];               // item capacity of one bucket of this level
    uint32_t per_split[kMaxLevels];         // items one K2 workgroup takes before a second one is brought in
    uint32_t acc_first[kMaxLevels];         // coarse levels: first row of the level in the shared fixed-point accumulator
                                            // (kNoSharedAcc otherwise)
    uint32_t item_first[kMaxLevels];        // first item slot (in units of 1024 items) of the level's bucket 0
    uint32_t merge_mask;                    // bit l: fold lane runs at level l before binning
    uint32_t levels;
};

template <bool HALF> struct Item;
// Half tables: one item carries the TWO corners of an x-pair, (x, y, z) and (x + 1, y, z). Their rows always lie in the same
// 2048-row bucket at a hashed level (x + 1 differs from x in a run of low bits, and the hash XORs y, z terms that the pair
// shares) and almost always at a dense one (rows r, r + 1), so 12 bytes {row0 | (row0 ^ row1) << 20, half2, half2} replace two
// 8-byte items: 6 bytes per contribution instead of 8 through HBM both ways — K2 streams the lists at the HBM read rate and K1's
// fine levels write them at close to the write rate (profiles/r03_scatter_pmc_wave_states.txt) — and half as many histogram
// atomics, staging slots and list entries. A pair that straddles a bucket boundary goes out as two items whose second value is 0.
template <> struct Item<true> {
    uint32_t rows;   // row0 in the level (20 bits) | (row0 ^ row1) << 20 (11 bits: same bucket)
    uint32_t val0, val1;
    __device__ __forceinline__ uint32_t row0() const { return rows & 0xFFFFFu; }
    __device__ __forceinline__ uint32_t row1() const { return (rows & 0xFFFFFu) ^ (rows >> 20); }
    __device__ __forceinline__ uint32_t bucket() const { return (rows & 0xFFFFFu) >> SDFX_BUCKET_LOG2; }
};
template <> struct Item<false> {  // {row in level, float2 contribution}
    uint32_t row;
    float a, b;
    static __device__ __forceinline__ Item make(uint32_t row, float a, float b) {
        Item it;
        it.row = row; it.a = a; it.b = b;
        return it;
    }
    __device__ __forceinline__ float2 value() const { return make_float2(a, b); }
    __device__ __forceinline__ uint32_t bucket() const { return row >> SDFX_BUCKET_LOG2; }
};

// value of lane (l - n) of the same 16-lane row, `self` where there is none (DPP row_shr: a VALU move, no LDS traffic)
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v, uint32_t self) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)self, (int)v, 0x110 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ float row_shr(float v, float self) {
    return __uint_as_float(row_shr<N>(__float_as_uint(v), __float_as_uint(self)));
}
// value of lane (l + 1) of the same row, `self` at the row's last lane (DPP row_shl:1)
__device__ __forceinline__ uint32_t row_shl1(uint32_t v, uint32_t self) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)self, (int)v, 0x101, 0xf, 0xf, false);
}

// Runs of neighbouring lanes (within a 16-lane DPP row) whose samples sit in the same grid cell: all 8 corners of
// such samples address the same 8 table rows, so the run structure is found ONCE per sample from the cell id and
// reused for every corner. Hillis-Steele segmented scan: step N adds the value N lanes down unless the lane's
// segment flag is already set; `take[N]` is that condition as an all-ones / zero word, so that the per-corner
// work is one masked DPP fetch per channel and step (v_and_b32_dpp) plus the add. After the scan the LAST lane
// of every run holds the run's sum (`tail`). Runs are cut at row boundaries, which only matters at the coarsest
// levels where a run could span more than 16 samples.
struct RunScan {
    uint32_t take1, take2, take4, take8;
    bool tail;
};

__device__ __forceinline__ RunScan scan_cell_runs(uint32_t cell, bool active, int lane) {
    const int rl = lane & 15;
    // inactive lanes get a key no active lane can have, so they break runs and are never emitted
    const uint32_t k = active ? cell : (0xFFFFFFFFu - (uint32_t)lane);
    const uint32_t prev = row_shr<1>(k, ~k);
    const bool head = (rl == 0) || (prev != k);
    RunScan r;
    uint32_t f = head ? 1u : 0u;  // lanes below N read the fill value 1, so after step N every lane < 2N is closed
    r.take1 = f ? 0u : 0xFFFFFFFFu; f |= row_shr<1>(f, 1u);
    r.take2 = f ? 0u : 0xFFFFFFFFu; f |= row_shr<2>(f, 1u);
    r.take4 = f ? 0u : 0xFFFFFFFFu; f |= row_shr<4>(f, 1u);
    r.take8 = f ? 0u : 0xFFFFFFFFu;
    // keep the masks opaque words in vector registers: knowing that they are 0 / ~0 the compiler turns `fetched & mask` into a
    // select on a scalar condition, which cannot be merged with the DPP fetch (v_mov_b32_dpp + v_cndmask_b32 instead of ONE
    // v_and_b32_dpp per channel and step)
    asm volatile("" : "+v"(r.take1), "+v"(r.take2), "+v"(r.take4), "+v"(r.take8));
    const uint32_t next_head = row_shl1(head ? 1u : 0u, 1u);
    r.tail = active && ((rl == 15) || (next_head != 0u));
    return r;
}

// (value N lanes down in the row, 0 outside it) & mask — compiles to ONE v_and_b32_dpp
template <int N>
__device__ __forceinline__ float masked_shr(float v, uint32_t mask) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x110 + N, 0xf, 0xf, true) & mask);
}

// one segmented-scan step on the two channels of a corner at once: two v_and_b32_dpp + one v_pk_add_f32
__device__ __forceinline__ void fold_runs(const RunScan& r, float2_t& v) {
    { const float2_t u = {masked_shr<1>(v.x, r.take1), masked_shr<1>(v.y, r.take1)}; v = v + u; }
    { const float2_t u = {masked_shr<2>(v.x, r.take2), masked_shr<2>(v.y, r.take2)}; v = v + u; }
    { const float2_t u = {masked_shr<4>(v.x, r.take4), masked_shr<4>(v.y, r.take4)}; v = v + u; }
    { const float2_t u = {masked_shr<8>(v.x, r.take8), masked_shr<8>(v.y, r.take8)}; v = v + u; }
}

__device__ __forceinline__ uint32_t round_half2(float2_t v) {   // both channels rounded to half (nearest even) by one instruction
    half2_t h;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(v.x), "v"(v.y));
    return __builtin_bit_cast(uint32_t, h);
}
struct BinLevels {
    LevelConst lv[kMaxLevels];   // per-level constants of the forward (grid_point.h): one 32-byte scalar load per workgroup
};

__device__ __forceinline__ Item<true> make_pair_item(uint32_t row0, uint32_t row1, uint32_t val0, uint32_t val1) {
    Item<true> it;
    it.rows = row0 | ((row0 ^ row1) << 20);
    it.val0 = val0; it.val1 = val1;
    return it;
}
// ---- the spill path of half tables: buckets asked for more slots than their list holds ------------------------
// Every bucket has a spill accumulator in the scratch, [kBucketRows][2 channels] 64-bit fixed point (the format of K2's LDS
// accumulators). It is zeroed by k_grid_bwd_spill_zero iff cursor > cap, filled by k_grid_bwd_spill, read by K2.
constexpr uint32_t kSpillWords = kBucketRows * 2;   // 64-bit words per bucket

// one rounded contribution (two halves) of row `row` of the level into the spill accumulator of its (overflowed) bucket
__device__ __forceinline__ void spill_add(unsigned long long* __restrict__ spill_acc, __half* gtab, uint32_t gbucket, uint32_t row, uint32_t val) {
    const uint32_t lo = val & 0xFFFFu, hi = val >> 16;
    if (((lo & 0x7C00u) == 0x7C00u) || ((hi & 0x7C00u) == 0x7C00u)) {   // inf / nan absorb whatever the order: straight to the table
        unsafeAtomicAdd(reinterpret_cast<__half2*>(gtab + (size_t)row * 2), *reinterpret_cast<const __half2*>(&val));
        return;
    }
    unsigned long long* a = spill_acc + (size_t)gbucket * kSpillWords + (size_t)(row & (kBucketRows - 1)) * 2;
    if (lo & 0x7FFFu) atomicAdd(a, (unsigned long long)half_to_fixed(lo));
    if (hi & 0x7FFFu) atomicAdd(a + 1, (unsigned long long)half_to_fixed(hi));
}

// ---------------------------------------------------------------------------------------------
// K1: contributions -> binned items
// ---------------------------------------------------------------------------------------------
// Instruction issue bounds this kernel (round 2, PMC: 560 VALU + 276 SALU instructions per wave and level, 76 % of its time),
// so it is written like the forward (gridencoder_fwd.hip): the grid's kind, interpolation and alignment are template
// parameters, the level's constants sit in scalar registers, the 8 row indices share their hash / stride terms
// (level_prepare), both channels of a corner travel as one float2 (v_pk_mul_f32, v_pk_add_f32, v_cvt_pk_f16_f32), the run
// folding is a workgroup-uniform template branch and nothing about a corner is decided by control flow.
// the contributions of this lane's sample at one level: 8 table rows, 8 float2 values (folded over the lane run that ends here
// when MERGE), and whether this lane emits them. Shared by K1 and by the spill kernel, which must reproduce K1's values bit for bit.
template <bool HALF> struct Contrib;
template <> struct Contrib<true> {     // half tables: the value is rounded to half2 (the reference's half(g * w)) as soon as it is final —
    uint32_t rows[8];                  // 8 registers instead of 16 live across the histogram, the scan and two barriers
    uint32_t h[8];
    bool emit;
    __device__ __forceinline__ void set(uint32_t idx, float2_t v) { h[idx] = round_half2(v); }
};
template <> struct Contrib<false> {
    uint32_t rows[8];
    float2_t v[8];
    bool emit;
    __device__ __forceinline__ void set(uint32_t idx, float2_t v_) { v[idx] = v_; }
};

// The inputs of one (level, tile) item for this thread: all ISSUED by tile_load — coordinates and gradient row together, the row
// limit read once into a scalar register before — and consumed by tile_compute. Rounds 1-4 paid three DEPENDENT round trips in front
// of every tile (row limit, then the coordinates, then the gradient, which was only loaded for in-range points): with the
// gradient load left out the kernel lost 21 % of its time (profiles/r05_xcd_timeline_one_tile_per_workgroup.txt).
struct TileIn {
    float x[3];      // inputs[b] (unit cube), or with a stencil source the base sample xyzs[b % M] (world)
    uint32_t g[2];   // the gradient row: one half2 word (half tables) or two floats
    bool ok;         // the row b = b0 + tile * kBinThreads + threadIdx.x is < b1 and not a padding row; nothing was loaded otherwise
};

// slab of row b of a [7, M, ...] stencil batch when b lies in the tile of kBinThreads rows starting at the (uniform) row `first`:
// the tile's first slab, decided on uniform values, unless the tile straddles a slab boundary
__device__ __forceinline__ uint32_t tile_slab(uint32_t first, uint32_t b, uint32_t M) {
    const uint32_t k0 = stencil_slab(__builtin_amdgcn_readfirstlane(first), M);
    if (__builtin_amdgcn_readfirstlane(first) + kBinThreads <= (k0 + 1u) * M || k0 == 6u) return k0;   // (uniform branch)
    return stencil_slab(b, M);
}

template <bool HALF>
__device__ __forceinline__ void tile_load(const typename Elem<HALF>::type* __restrict__ grad, const float* __restrict__ inputs, uint32_t B,
                                          uint32_t L, uint32_t b0, uint32_t b1, uint32_t level, uint32_t tile, int grad_layout,
                                          const RowLimitNow& rl, const StencilSrc& src, TileIn& t) {
    using T = typename Elem<HALF>::type;
    const uint32_t first = b0 + tile * kBinThreads, b = first + threadIdx.x;
    t.ok = b < b1 && row_live_tile(rl, first, threadIdx.x, kBinThreads);
    t.x[0] = t.x[1] = t.x[2] = 0.f;
    t.g[0] = t.g[1] = 0u;
    if (!t.ok) return;
    if (SDFX_ABLATE(32u)) {   // measurement: coordinates from the row number instead of from memory
        t.x[0] = (float)(b & 1023u) * (1.f / 1024.f); t.x[1] = (float)((b >> 10) & 1023u) * (1.f / 1024.f); t.x[2] = (float)(b >> 20) * (1.f / 16.f);
    } else {
        const float* xp = src.xyzs ? src.xyzs + (size_t)(b - tile_slab(first, b, src.M) * src.M) * 3 : inputs + (size_t)b * 3;
        t.x[0] = xp[0]; t.x[1] = xp[1]; t.x[2] = xp[2];
    }
    // the gradient row is loaded whether or not the point turns out to lie in the unit cube (the row exists either way): the load
    // does not wait for the coordinates
    const T* gp = grad_layout == 0 ? grad + ((size_t)level * B + b) * 2 : grad + ((size_t)b * L + level) * 2;
    if (SDFX_ABLATE(16u)) {   // measurement: no gradient load
        t.g[0] = HALF ? 0xA51F211Fu : __float_as_uint(0.01f); t.g[1] = __float_as_uint(-0.02f);
    } else if constexpr (HALF) {
        t.g[0] = *reinterpret_cast<const uint32_t*>(gp);
    } else {
        const uint2 f = *reinterpret_cast<const uint2*>(gp);
        t.g[0] = f.x; t.g[1] = f.y;
    }
}

template <bool HALF, uint32_t INTERP, bool ALIGN, bool HASHGRID, bool MERGE>
__device__ __forceinline__ void tile_compute(const TileIn& t, uint32_t b0, uint32_t tile, const LevelConst& lc, const StencilSrc& src,
                                             Contrib<HALF>& c) {
    constexpr uint32_t NCORN = 8;
    const int lane = lane_id();
    const uint32_t first = b0 + tile * kBinThreads, b = first + threadIdx.x;
    // ---- the sample: coordinates, gradient row, cell, weights, the 8 table rows ----
    bool valid = t.ok;
    float in[3] = {t.x[0], t.x[1], t.x[2]};
    if (src.xyzs && !SDFX_ABLATE(32u)) {   // sdfx_set_stencil_source: row b of the [7, M, 3] batch formed here (stencil_unit_row's arithmetic)
        float p[3];
        stencil_world(src, valid ? tile_slab(first, b, src.M) : 0u, t.x, p);
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) in[d] = (p[d] + src.bound) * src.inv;
    }
#pragma unroll
    for (uint32_t d = 0; d < 3; d++)
        if (in[d] < 0 || in[d] > 1) valid = false;  // gridencoder.cu:279-284
    float2_t g = {0.f, 0.f};
    if constexpr (HALF) {
        const half2_t h = __builtin_bit_cast(half2_t, t.g[0]);
        g.x = (float)h.x; g.y = (float)h.y;
    } else {
        g.x = __uint_as_float(t.g[0]); g.y = __uint_as_float(t.g[1]);
    }
    // nothing to scatter: samples behind a ray's early-termination cut, and the zero-gradient padding rows of
    // fixed-capacity sample buffers (which all sit in ONE cell and would overflow its bucket)
    if (g.x == 0.f && g.y == 0.f) valid = false;
    if (!valid) { g.x = 0.f; g.y = 0.f; }
    const float xs[3] = {valid ? in[0] : 0.f, valid ? in[1] : 0.f, valid ? in[2] : 0.f};
    LevelPoint p;
    level_prepare_uniform<INTERP, ALIGN, HASHGRID>(lc, xs, p);   // (one level per workgroup: the level's kind is control flow)
    // corner idx = xbit + 2 ybit + 4 zbit (gridencoder.cu:171-184); weight ((1 * a_x) * a_y) * a_z in that order
    const float ax[2] = {1 - p.ax1, p.ax1}, ay[2] = {1 - p.ay1, p.ay1}, az[2] = {1 - p.az1, p.az1};
    float2_t v[NCORN];
#pragma unroll
    for (uint32_t idx = 0; idx < NCORN; idx++) {
        const uint32_t k = idx >> 1;
        c.rows[idx] = (idx & 1u) ? p.r1[k] : p.r0[k];
        const float w = ((1 * ax[idx & 1u]) * ay[k & 1u]) * az[k >> 1];
        v[idx] = g * w;
    }
    c.emit = valid;
    if constexpr (MERGE) {
        // merged levels have res <= 640, so a cell id fits 10 bits per axis
        const RunScan runs = scan_cell_runs(p.cx | (p.cy << 10) | (p.cz << 20), valid, lane);
#pragma unroll
        for (uint32_t idx = 0; idx < NCORN; idx++) fold_runs(runs, v[idx]);
        c.emit = runs.tail;
    }
#pragma unroll
    for (uint32_t idx = 0; idx < NCORN; idx++) c.set(idx, v[idx]);
}

// load + compute in one go (the spill kernel)
template <bool HALF, uint32_t INTERP, bool ALIGN, bool HASHGRID, bool MERGE>
__device__ __forceinline__ void tile_contributions(const typename Elem<HALF>::type* __restrict__ grad, const float* __restrict__ inputs,
                                                   uint32_t B, uint32_t L, uint32_t b0, uint32_t b1, uint32_t level, uint32_t tile,
                                                   const LevelConst& lc, int grad_layout, const RowLimitNow& rl, const StencilSrc& src,
                                                   Contrib<HALF>& c) {
    TileIn t;
    tile_load<HALF>(grad, inputs, B, L, b0, b1, level, tile, grad_layout, rl, src, t);
    tile_compute<HALF, INTERP, ALIGN, HASHGRID, MERGE>(t, b0, tile, lc, src, c);
}

template <bool HALF, uint32_t INTERP, bool ALIGN, bool HASHGRID, bool MERGE>
__device__ __forceinline__ void bin_tile(const TileIn& in, typename Elem<HALF>::type* __restrict__ grad_table, uint32_t b0,
                                         uint32_t level, uint32_t tile, const LevelConst& lc, const BinPlan& bin, uint32_t* __restrict__ cursors,
                                         Item<HALF>* __restrict__ items, const StencilSrc& src, uint32_t* hist, uint32_t* gbase,
                                         uint32_t* boff, uint32_t* wave_tot, uint32_t* block_total, Item<HALF>* stage) {
    using T = typename Elem<HALF>::type;
    constexpr uint32_t C = 2, NCORN = 8;
    const int lane = lane_id();
    const uint32_t bucket0 = bin.bucket_first[level];
    const uint32_t nb = bin.bucket_first[level + 1] - bucket0;
    for (uint32_t i = threadIdx.x; i < nb; i += kBinThreads) hist[i] = 0;
    __syncthreads();

    Contrib<HALF> c;
    tile_compute<HALF, INTERP, ALIGN, HASHGRID, MERGE>(in, b0, tile, lc, src, c);
#ifdef SDFX_K1_PAD   // measurement aid (tools/build_variant.py ... -DSDFX_K1_PAD=n): n more full-rate vector instructions per thread and level,
    {                // results unchanged — does K1's time follow its vector instruction count? (profiles/r06_scatter_k1_valu_pad.txt)
        uint32_t pad = c.rows[0];
#pragma unroll
        for (int i = 0; i < SDFX_K1_PAD; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad));
        c.rows[0] = pad - (uint32_t)SDFX_K1_PAD;
    }
#endif
    const uint32_t (&rows)[NCORN] = c.rows;
    const bool emit = c.emit;
    // items of this lane: one per corner (float tables) or one per x-pair of corners (half tables, see Item<true>)
    constexpr uint32_t NIT = HALF ? NCORN / 2 : NCORN;
    uint32_t ibucket[NIT], rank[NIT];
    bool split[NIT];
#pragma unroll
    for (uint32_t i = 0; i < NIT; i++) {
        if constexpr (HALF) {
            ibucket[i] = rows[2 * i] >> kBucketRowsLog2;
            split[i] = (rows[2 * i + 1] >> kBucketRowsLog2) != ibucket[i];   // the pair straddles two buckets (dense levels, rarely)
            if constexpr (kBucketRowsLog2 > 12) split[i] = split[i] || ((rows[2 * i] ^ rows[2 * i + 1]) >> 12) != 0u;   // (12 bits in the item)
        } else {
            ibucket[i] = rows[i] >> kBucketRowsLog2;
            split[i] = false;
        }
    }
    if (emit) {
#pragma unroll
        for (uint32_t i = 0; i < NIT; i++) rank[i] = SDFX_ABLATE(8u) ? 0u : atomicAdd(&hist[ibucket[i]], 1u);  // LDS
    }
    __syncthreads();
    // one global atomic per (workgroup, non-empty bucket): reserve a slice of the bucket's list. Its result is not needed before the
    // write-out: it stays in a register while the histogram is scanned and the items are staged (the round trip of a returning
    // device-scope atomic is a microsecond or two). And an exclusive prefix sum of the histogram = where each bucket's items go in
    // the workgroup's LDS staging area
    uint32_t my_cnt = 0, my_base = 0;
    if (threadIdx.x < nb) {
        my_cnt = hist[threadIdx.x];
        if (my_cnt && !SDFX_ABLATE(4u)) my_base = atomicAdd(&cursors[bucket0 + threadIdx.x], my_cnt);
    }
    {   // nb <= kMaxBucketsPerLevel = kBinThreads: thread b scans bucket b (wave scan + per-wave totals)
        const uint32_t incl = wave_incl_sum_u32(my_cnt, lane);
        if (lane == (int)kWave - 1) wave_tot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t woff = 0;   // totals of the waves below this one: every thread reads all of them at once, no loop of dependent LDS reads
#pragma unroll                // (with the scan on the DPP network instead of six ds_bpermute round trips: K1 span 834-841 -> 807 us at
                              // B = 3.26 M, same box, profiles/r06_scatter_k1_scan_and_order.txt)
        for (uint32_t w = 0; w + 1 < kBinThreads / 64; w++) woff += w < (threadIdx.x >> 6) ? wave_tot[w] : 0u;
        if (threadIdx.x < nb) boff[threadIdx.x] = woff + incl - my_cnt;
        if (threadIdx.x == kBinThreads - 1) *block_total = woff + incl;
    }
    __syncthreads();

    // Stage the items in LDS grouped by bucket, then stream them out: consecutive staging slots of one bucket go to
    // consecutive slots of its list, so a wave store covers a few contiguous runs instead of 64 unrelated 8-byte
    // writes (the scattered version was bound by L2 write transactions: one per item).
    // A slot at or beyond the list's capacity: half tables drop the item — the cursor keeps counting, cursor > cap tells K2 to
    // take the bucket's sum from its spill accumulator, which k_grid_bwd_spill fills with ALL of the bucket's contributions
    // (exact); float tables add with the reference's float atomics.
    const uint32_t cap = bin.cap[level];
    Item<HALF>* level_items = items + (size_t)bin.item_first[level] * 1024u;
    if (emit && !SDFX_ABLATE(2u)) {
#pragma unroll
        for (uint32_t i = 0; i < NIT; i++) {
            if constexpr (HALF) {
                const uint32_t v0 = c.h[2 * i], v1 = c.h[2 * i + 1];
                stage[boff[ibucket[i]] + rank[i]] = make_pair_item(rows[2 * i], split[i] ? rows[2 * i] : rows[2 * i + 1], v0, split[i] ? 0u : v1);
                if (split[i]) {   // the second corner goes to its own bucket's list by a one-slot reservation of this lane
                    const uint32_t b1 = rows[2 * i + 1] >> kBucketRowsLog2;
                    const uint32_t slot = atomicAdd(&cursors[bucket0 + b1], 1u);
                    if (slot < cap) level_items[(size_t)b1 * cap + slot] = make_pair_item(rows[2 * i + 1], rows[2 * i + 1], v1, 0u);
                }
            } else {
                stage[boff[ibucket[i]] + rank[i]] = Item<false>::make(rows[i], c.v[i].x, c.v[i].y);
            }
        }
    }
    if (threadIdx.x < nb) gbase[threadIdx.x] = my_base;   // (the reservation's result is first touched here)
    __syncthreads();

    const uint32_t total = SDFX_ABLATE(3u) ? 0u : *block_total;
    for (uint32_t k = threadIdx.x; k < total; k += kBinThreads) {
        const Item<HALF> it = stage[k];
        const uint32_t bucket = it.bucket();
        const uint32_t slot = gbase[bucket] + (k - boff[bucket]);
        if (slot < cap) {
            level_items[(size_t)bucket * cap + slot] = it;
        } else if constexpr (!HALF) {
            T* dst = grad_table + ((size_t)lc.row0 + it.row) * C;
            unsafeAtomicAdd(dst, it.a);
            unsafeAtomicAdd(dst + 1, it.b);
        }
    }
}

// One (level, tile) item per workgroup; every XCD walks its own range of whole levels (make_plan). A workgroup's time is a chain of
// memory round trips around ~3.5 us of arithmetic (with the input loads, the reservation atomics and the list stores each left out
// in turn the round-4 kernel lost 21-25 % of its time, with all of them 55 %: profiles/r05_xcd_timeline_one_tile_per_workgroup.txt),
// so the chain is kept short: inputs issued together (tile_load), the reservation's result not waited for before the write-out
// (bin_tile), 48 registers without spills (Contrib<true>). Measured this round and NOT adopted, all bit-identical in their results:
//   * a tile LOOP with the next tile's loads in flight (commit 71d704a): 87 registers = 6 waves per SIMD, 1083 us against 1034 us at
//     B = 3.26 M (at 64 registers it spilled 24 words: 2158 us) — profiles/r05_xcd_timeline_k1_tile_loop_variant.txt;
//   * the same loop software-pipelined at 64 registers without spills (commit 6b153e7: previous tile's write-out and reservation
//     results one tile behind, one memory wait per iteration): 984-1056 us against this kernel's 898-926 us — the loop's own
//     overhead (item bookkeeping, 108 scalar-register spill moves, a higher arithmetic-only floor: 616 against 535 us) eats what the
//     overlap gains, and the list stores still cost 13 % with nobody waiting for them: they are throughput, not latency
//     (profiles/r05_xcd_timeline_k1_pipelined_loop_variant.txt);
//   * fewer, longer reservations — 4096-row buckets and / or 1024-sample tiles (half as many atomics, 192-byte runs): K1 unchanged
//     (896-920 us) or slower (978-1020 us), K2 385 us with 64 KB of accumulators against 340-366 (profiles/r05_scatter_bucket_tile_variants.txt);
//     tools/ubench/write_streams.hip says why the runs do not help yet: 96-byte and 192-byte runs write at 2.2-2.4 TB/s, the jump
//     to 4.8-5.4 TB/s comes at 384 bytes (profiles/r05_write_streams_ubench.txt).
//   * round 6: the VALUES of the items (weights, products, run folding, rounding) computed after the histogram and the reservation
//     atomics instead of before, so that the returning atomics fly behind them (56 registers): K1 span 838-855 us against 804-806 us
//     for this order on the same box; an XCD's two levels walked tile by tile instead of one after the other, and 2 / 4 items per
//     workgroup: no change (profiles/r06_scatter_k1_scan_and_order.txt, r06_scatter_k1_interleave_tpw.txt).
//   * round 6, what bounds it (profiles/r06_scatter_k1_bound.txt): 410 vector instructions per wave and level = 0.8 of the SIMD cycles
//     of its span — 64 dummy instructions more (-DSDFX_K1_PAD) cost 2-3 %, 128 cost 5-7 %: the chain leaves ~20 % of the vector issue
//     unused and no more; 256-sample tiles (8 workgroups of 4 waves per CU, -DSDFX_BIN_THREADS=256 -DSDFX_MAX_BUCKETS=256) live as long
//     as 512-sample ones: 1046 against 800 us; and the list stores: a 64-byte sector that two runs share leaves the L2 before the second
//     run arrives 7 times in 10 (17.4 M of 39.3 M sectors written back twice, 3.8 M partial requests, 4 x the write-request stalls of
//     a dense stream) — the L2 combines writes over a short window only, whatever the store's cache-policy bits.
// (second bound = waves per SIMD: 8, i.e. 4 workgroups per CU, which the 30 KB of LDS allow)
template <bool HALF, uint32_t INTERP, bool ALIGN, bool HASHGRID>
__global__ __launch_bounds__(kBinThreads, 8) void k_grid_bwd_bin(const typename Elem<HALF>::type* __restrict__ grad,
                                                               const float* __restrict__ inputs,
                                                               typename Elem<HALF>::type* __restrict__ grad_table,
                                                               uint32_t B, uint32_t L, uint32_t b0, uint32_t b1,
                                                               GridPlan plan, BinPlan bin, BinLevels lv, int grad_layout,
                                                               uint32_t* __restrict__ cursors,
                                                               Item<HALF>* __restrict__ items, RowLimit rl, StencilSrc src) {
    __shared__ uint32_t hist[kMaxBucketsPerLevel];
    __shared__ uint32_t gbase[kMaxBucketsPerLevel];
    __shared__ uint32_t boff[kMaxBucketsPerLevel];
    __shared__ uint32_t wave_tot[kBinThreads / 64];
    __shared__ uint32_t block_total;
    __shared__ Item<HALF> stage[kBinThreads * (HALF ? 4 : 8)];   // 24 KiB (half: 4 pair items per sample) / 48 KiB (float items)

    uint32_t level, tile;
    if (!plan_item(plan, blockIdx.x, level, tile)) return;   // wave-uniform (depends on blockIdx only)
    // a tile of padding rows (sdfx_set_row_limit) has nothing to scatter
    const RowLimitNow rln = row_limit_now(rl);
    if (rows_dead(rln, b0 + tile * kBinThreads, kBinThreads)) return;
    SDFX_STAMP_BEGIN
    TileIn in;
    tile_load<HALF>(grad, inputs, B, L, b0, b1, level, tile, grad_layout, rln, src, in);
    const LevelConst lc = lv.lv[level];
    if ((bin.merge_mask >> level) & 1u)   // workgroup-uniform; all lanes take part in the DPP exchanges
        bin_tile<HALF, INTERP, ALIGN, HASHGRID, true>(in, grad_table, b0, level, tile, lc, bin, cursors, items, src, hist, gbase, boff, wave_tot,
                                                      &block_total, stage);
    else
        bin_tile<HALF, INTERP, ALIGN, HASHGRID, false>(in, grad_table, b0, level, tile, lc, bin, cursors, items, src, hist, gbase, boff, wave_tot,
                                                       &block_total, stage);
    SDFX_STAMP_END_AT(2u, level, tile, level * plan.tiles + tile)
}

// ---------------------------------------------------------------------------------------------
// The spill path of half tables (see the head of the file). Both kernels exit at once unless some bucket overflowed.
// ---------------------------------------------------------------------------------------------
// since the library was loaded, on this device: [0] buckets that overflowed, [1] launches in which some bucket did
__device__ uint32_t g_spill_totals[2];

// one workgroup per bucket: zero the spill accumulator of an overflowed bucket and raise the launch's `any` flag
// (`bucket_lo`: first bucket of the launch's group of levels, `flag`: that group's word of `diag` — see the two-group launch below)
__global__ __launch_bounds__(256) void k_grid_bwd_spill_zero(BinPlan bin, const uint32_t* __restrict__ cursors,
                                                             unsigned long long* __restrict__ spill_acc, uint32_t* __restrict__ diag,
                                                             uint32_t bucket_lo, uint32_t flag) {
    const uint32_t gb = blockIdx.x + bucket_lo;
    uint32_t level = 0;
    while (level + 1 < bin.levels && gb >= bin.bucket_first[level + 1]) level++;
    if (cursors[gb] <= bin.cap[level]) return;
    for (uint32_t i = threadIdx.x; i < kSpillWords; i += 256) spill_acc[(size_t)gb * kSpillWords + i] = 0ull;
    if (threadIdx.x == 0) {
        diag[flag] = 1u;          // some bucket of the group overflowed (benign race: every writer stores 1)
        atomicAdd(&diag[1], 1u);  // how many (sdfx_grid_encode_backward_binned_stats)
        atomicAdd(&g_spill_totals[0], 1u);
    }
}

// K1's (level, tile) items once more, kSpillTilesPerGroup consecutive ones per workgroup: every contribution whose bucket
// overflowed goes to that bucket's spill accumulator. The values are K1's (same lane map, same run folding, same rounding).
constexpr uint32_t kSpillTilesPerGroup = 32;

template <uint32_t INTERP, bool ALIGN, bool HASHGRID, bool MERGE>
__device__ __forceinline__ void spill_tile(const __half* __restrict__ grad, const float* __restrict__ inputs, __half* __restrict__ grad_table,
                                           uint32_t B, uint32_t L, uint32_t b0, uint32_t b1, uint32_t level, uint32_t tile,
                                           const LevelConst& lc, const BinPlan& bin, int grad_layout, const uint32_t* __restrict__ cursors,
                                           const RowLimitNow& rl, const StencilSrc& src, unsigned long long* __restrict__ spill_acc) {
    Contrib<true> c;
    tile_contributions<true, INTERP, ALIGN, HASHGRID, MERGE>(grad, inputs, B, L, b0, b1, level, tile, lc, grad_layout, rl, src, c);
    if (!c.emit) return;
    const uint32_t bucket0 = bin.bucket_first[level], cap = bin.cap[level];
    __half* gtab = grad_table + (size_t)lc.row0 * 2;
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        const uint32_t gb = bucket0 + (c.rows[idx] >> kBucketRowsLog2);
        if (cursors[gb] > cap) spill_add(spill_acc, gtab, gb, c.rows[idx], c.h[idx]);
    }
}

template <uint32_t INTERP, bool ALIGN, bool HASHGRID>
__global__ __launch_bounds__(kBinThreads) void k_grid_bwd_spill(const __half* __restrict__ grad, const float* __restrict__ inputs,
                                                                __half* __restrict__ grad_table, uint32_t B, uint32_t L, uint32_t b0,
                                                                uint32_t b1, GridPlan plan, BinPlan bin, BinLevels lv, int grad_layout,
                                                                const uint32_t* __restrict__ cursors, RowLimit rl, StencilSrc src,
                                                                unsigned long long* __restrict__ spill_acc,
                                                                const uint32_t* __restrict__ diag, uint32_t k1_grid, uint32_t flag) {
    if (diag[flag] == 0u) return;   // no bucket overflowed: nearly every launch of the training loop ends here
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_spill_totals[1], 1u);
    const RowLimitNow rln = row_limit_now(rl);
    for (uint32_t i = 0; i < kSpillTilesPerGroup; i++) {
        const uint32_t vblock = blockIdx.x * kSpillTilesPerGroup + i;
        uint32_t level, tile;
        if (vblock >= k1_grid || !plan_item(plan, vblock, level, tile)) continue;   // workgroup-uniform
        if (rows_dead(rln, b0 + tile * kBinThreads, kBinThreads)) continue;
        const LevelConst lc = lv.lv[level];
        if ((bin.merge_mask >> level) & 1u)
            spill_tile<INTERP, ALIGN, HASHGRID, true>(grad, inputs, grad_table, B, L, b0, b1, level, tile, lc, bin, grad_layout, cursors, rln, src, spill_acc);
        else
            spill_tile<INTERP, ALIGN, HASHGRID, false>(grad, inputs, grad_table, B, L, b0, b1, level, tile, lc, bin, grad_layout, cursors, rln, src, spill_acc);
    }
}

// ---------------------------------------------------------------------------------------------
// K2: per-bucket reduction in LDS, then one read-modify-write per touched table row
// ---------------------------------------------------------------------------------------------
// LDS float atomics are the wrong tool here: ds_add_f32 with 64 scattered addresses keeps the LDS pipe busy for
// ~390 cycles per wave instruction on gfx950 (rocprofv3: SQ_LDS_IDX_ACTIVE 734 M cycles for 1.9 M ds_add_f32,
// SQ_WAIT_INST_LDS = 92 % of the wave cycles; profiles/r01_pmc_gridbwd.txt), while plain ds_read/ds_write and
// the INTEGER atomics of K1 cost 4-5 cycles.
//
//  * half items (the -O path): every finite half is an integer multiple of 2^-24 below 2^16, so value * 2^24 is
//    an integer below 2^40 and a 64-bit integer accumulator (ds_add_u64) sums millions of items EXACTLY, in any
//    order, however many lanes hit the same row. The bucket's sum is converted and rounded once at the flush.
//    Non-finite items (an overflowed AMP step) bypass the accumulator and go to the table with a global atomic.
//  * float items: every wave owns a PRIVATE 2048 x float2 accumulator and adds with ordinary read-modify-writes;
//    lanes of one wave instruction that target the same row are serialised by a one-byte ticket: all pending
//    lanes write their lane id to tag[row], read it back (LDS executes a wave's instructions in order), the lane
//    whose id survived adds, the others go round again.
#ifndef SDFX_REDUCE_THREADS
#define SDFX_REDUCE_THREADS (SDFX_BUCKET_LOG2 >= 12 ? 1024 : 512)
#endif
constexpr uint32_t kReduceThreadsFixed = SDFX_REDUCE_THREADS;
constexpr uint32_t kReduceWaves = kReduceThreads / 64;
constexpr uint32_t kReduceLdsBytes = kReduceWaves * kBucketRows * (sizeof(float2) + 1);
constexpr uint32_t kReduceFixedLdsBytes = SDFX_BUCKET_LOG2 > 12 ? kBucketRows * 2 * (uint32_t)sizeof(unsigned long long) : 0u;   // (dynamic part)

struct ReduceJob {
    uint32_t level, bucket, gbucket, split, used, begin, end, cap;
    bool spilled;   // the bucket was asked for more slots than its list has. Half tables: the list is ignored, the bucket's whole sum
                    // is in its spill accumulator; float tables: the list is full and the rest was added atomically by K1
};

// workgroup -> (level, bucket, split) and its slice of the bucket's item list; false if there is nothing to do
template <bool HALF>
__device__ __forceinline__ bool reduce_job(const BinPlan& bin, const uint32_t* __restrict__ cursors, ReduceJob& j, uint32_t wg_lo = 0) {
    const uint32_t wg = blockIdx.x + wg_lo;   // (a launch may cover the workgroups of a group of levels only)
    // (K2's workgroups are launched level by level, coarse to fine. Launching the longest-running ones — the finest levels, 4 items
    // per point — first, so that the short ones fill the tail, was measured this round and is 5 % SLOWER: K2 streams its lists at
    // the HBM read rate — 1.34 GB in 340 us — and the fine levels' workgroups all running at once slow each other, 95 -> 150-170 us
    // apiece: profiles/r05_scatter_k2_per_level.txt)
    uint32_t level = 0;
    while (level + 1 < bin.levels && wg >= bin.split_first[level + 1]) level++;
    const uint32_t splits = bin.splits[level];
    const uint32_t local = wg - bin.split_first[level];
    j.level = level;
    j.bucket = local / splits;
    j.gbucket = bin.bucket_first[level] + j.bucket;
    j.split = local - j.bucket * splits;
    j.cap = bin.cap[level];
    uint32_t n = cursors[j.gbucket];
    j.spilled = n > j.cap;
    if (j.spilled) n = HALF ? 0u : j.cap;
    // How many of the `splits` workgroups launched for this bucket actually share it is decided from the item count found at
    // run time: one workgroup (sole owner, plain read-modify-write flush) unless the bucket is heavy (a level of few buckets:
    // only those are launched with splits > 1, and they have a shared 64-bit accumulator, see make_bin_plan).
    const uint32_t per_split = bin.per_split[level];
    uint32_t used = (n + per_split - 1) / per_split;
    if (used < 1) used = 1;
    if (used > splits) used = splits;
    j.used = used;
    if (j.split >= used) return false;
    const uint32_t per = (n + used - 1) / used;
    j.begin = j.split * per;
    j.end = j.begin + per < n ? j.begin + per : n;
    return j.begin < j.end || (HALF && j.spilled);   // (then used == 1, split == 0: this workgroup flushes the spill accumulator)
}

// add (a, b) to a table row that this workgroup alone touches: plain read-modify-write
template <bool HALF>
__device__ __forceinline__ void flush_row(typename Elem<HALF>::type* dst, float a, float b) {
    if constexpr (HALF) {
        const __half2 o = *reinterpret_cast<const __half2*>(dst);
        *reinterpret_cast<__half2*>(dst) = __halves2half2(__float2half_rn(__low2float(o) + a), __float2half_rn(__high2float(o) + b));
    } else {
        float2 o = *reinterpret_cast<const float2*>(dst);
        o.x += a; o.y += b;
        *reinterpret_cast<float2*>(dst) = o;
    }
}

// half_to_fixed / fixed_to_float: sdfx_math.h (checked exhaustively on the host, tests/test_hostmath.py)

__global__ __launch_bounds__(kReduceThreadsFixed) void k_grid_bwd_reduce_fixed(__half* __restrict__ grad_table, GridPlan plan,
                                                                              BinPlan bin,
                                                                              const uint32_t* __restrict__ cursors,
                                                                              const Item<true>* __restrict__ items,
                                                                              unsigned long long* __restrict__ shared_acc,
                                                                              const unsigned long long* __restrict__ spill_acc, uint32_t wg_lo) {
#if SDFX_BUCKET_LOG2 > 12
    extern __shared__ __align__(16) unsigned long long acc[];   // 128 KiB at 8192 rows: one workgroup per CU (launched with kReduceFixedLdsBytes)
#else
    __shared__ unsigned long long acc[kBucketRows * 2];  // 32 KiB
#endif
    ReduceJob j;
    if (!reduce_job<true>(bin, cursors, j, wg_lo)) return;
    SDFX_STAMP_BEGIN
    for (uint32_t i = threadIdx.x; i < kBucketRows * 2; i += kReduceThreadsFixed) acc[i] = 0ull;
    __syncthreads();

    const uint32_t row0 = plan.off[j.level];
#ifndef SDFX_REDUCE_UNROLL
#define SDFX_REDUCE_UNROLL 8   // measurement aid
#endif
    constexpr uint32_t kUnroll = SDFX_REDUCE_UNROLL;  // independent loads in flight per thread
    const Item<true>* src = items + (size_t)bin.item_first[j.level] * 1024u + (size_t)j.bucket * j.cap;
    for (uint32_t base = j.begin; base < j.end; base += kUnroll * kReduceThreadsFixed) {
        Item<true> it[kUnroll];
        bool have[kUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; u++) {
            const uint32_t i = base + u * kReduceThreadsFixed + threadIdx.x;
            have[u] = i < j.end;
            it[u] = src[have[u] ? i : j.begin];
        }
        // one rounded contribution (two halves) of row `row` (in the level)
        auto add = [&](uint32_t v, uint32_t row) {
            const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
            if (((lo & 0x7C00u) == 0x7C00u) || ((hi & 0x7C00u) == 0x7C00u)) {  // inf / nan absorb whatever the order: straight to the table
                unsafeAtomicAdd(reinterpret_cast<__half2*>(grad_table + ((size_t)row0 + row) * 2), *reinterpret_cast<const __half2*>(&v));
                return;
            }
            // channel-major accumulators (acc[ch * kBucketRows + row]): a 64-bit slot covers two banks, so with the two channels of
            // a row side by side only rows 0..7 (mod 8) are distinct bank groups — 64 random rows collide 8 ways on average; with
            // one array per channel it is rows mod 16
            uint32_t r = row & (kBucketRows - 1);
            if (SDFX_ABLATE(64u)) r = (r ^ (threadIdx.x * 37u)) & (kBucketRows - 1);   // measurement: no two lanes of a wave on one row
            if (lo & 0x7FFFu) atomicAdd(&acc[r], (unsigned long long)half_to_fixed(lo));      // ds_add_u64
            if (hi & 0x7FFFu) atomicAdd(&acc[kBucketRows + r], (unsigned long long)half_to_fixed(hi));
        };
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; u++) {
            if (!have[u]) continue;
            if (SDFX_ABLATE(128u)) {   // measurement: the list stream alone — the items are consumed, nothing is added (wrong sums)
                if ((it[u].rows ^ it[u].val0 ^ it[u].val1) == 0x9E3779B9u) atomicAdd(&acc[0], 1ull);
                continue;
            }
            add(it[u].val0, it[u].row0());
            if (it[u].val1 & 0x7FFF7FFFu) add(it[u].val1, it[u].row1());
        }
    }
    __syncthreads();

    const uint32_t level_rows = plan.off[j.level + 1] - row0;
    const uint32_t first_row = j.bucket << kBucketRowsLog2;
    // an overflowed bucket: its whole sum is in the spill accumulator (exact 64-bit sums, [row][channel]); acc[] stayed zero
    const unsigned long long* spill = j.spilled ? spill_acc + (size_t)j.gbucket * kSpillWords : nullptr;
    if (j.used == 1) {   // sole owner of the bucket's rows: round the exact sums once, plain read-modify-write
        for (uint32_t r = threadIdx.x; r < kBucketRows; r += kReduceThreadsFixed) {
            long long ia = (long long)acc[r], ib = (long long)acc[kBucketRows + r];
            if (spill) { ia += (long long)spill[r * 2]; ib += (long long)spill[r * 2 + 1]; }
            if (ia == 0 && ib == 0) continue;
            const uint32_t row = first_row + r;
            if (row >= level_rows) continue;
            // |sum| < 2^63 * 2^-24; the double is exact up to 2^53, the float conversion rounds once
            flush_row<true>(grad_table + ((size_t)row0 + row) * 2, fixed_to_float(ia), fixed_to_float(ib));
        }
        SDFX_STAMP_END_AT(3u, j.level, j.bucket, blockIdx.x + wg_lo)
        return;
    }
    // Several workgroups share this bucket (a level of few buckets): they add their exact partial sums into a 64-bit
    // accumulator in global memory, which k_grid_bwd_finish rounds ONCE into the table. The result is therefore
    // independent of how the bucket was split and of the order of arrival. (A last-arriver flush inside this kernel
    // needs a device-scope release per workgroup, i.e. an L2 write-back on this multi-XCD part: +170 us measured.)
    unsigned long long* gacc = shared_acc + ((size_t)bin.acc_first[j.level] + first_row) * 2;
    for (uint32_t i = threadIdx.x; i < kBucketRows * 2; i += kReduceThreadsFixed) {   // the global accumulator stays [row][channel]
        unsigned long long v = acc[(i & 1u) * kBucketRows + (i >> 1)];
        if (spill) v += spill[i];
        if (v) atomicAdd(&gacc[i], v);
    }
    SDFX_STAMP_END_AT(3u, j.level, j.bucket | (j.split << 16), blockIdx.x + wg_lo)
}

// K3: one workgroup per bucket of the levels of few buckets; buckets that were reduced by a single workgroup are done already
__global__ __launch_bounds__(256) void k_grid_bwd_finish(__half* __restrict__ grad_table, GridPlan plan, BinPlan bin,
                                                         const uint32_t* __restrict__ cursors,
                                                         const unsigned long long* __restrict__ shared_acc) {
    uint32_t level = 0, b = blockIdx.x;
    for (;; level++) {  // these levels are the first ones of the plan
        const uint32_t nb = bin.bucket_first[level + 1] - bin.bucket_first[level];
        if (b < nb) break;
        b -= nb;
    }
    const uint32_t n = cursors[bin.bucket_first[level] + b];
    if (n > bin.cap[level]) return;   // overflowed: K2's one workgroup flushed the spill accumulator
    const uint32_t per_split = bin.per_split[level];
    uint32_t used = (n + per_split - 1) / per_split;
    if (used > bin.splits[level]) used = bin.splits[level];
    if (used <= 1) return;
    const uint32_t row0 = plan.off[level];
    const uint32_t level_rows = plan.off[level + 1] - row0;
    const uint32_t first_row = b << kBucketRowsLog2;
    const unsigned long long* gacc = shared_acc + ((size_t)bin.acc_first[level] + first_row) * 2;
    for (uint32_t r = threadIdx.x; r < kBucketRows; r += 256) {
        const uint32_t row = first_row + r;
        if (row >= level_rows) continue;
        const long long ia = (long long)gacc[r * 2], ib = (long long)gacc[r * 2 + 1];
        if (ia == 0 && ib == 0) continue;
        flush_row<true>(grad_table + ((size_t)row0 + row) * 2, fixed_to_float(ia), fixed_to_float(ib));
    }
}

typedef __attribute__((address_space(3))) volatile uint8_t lds_ticket_t;  // keeps ds_write_b8/ds_read_u8 (not flat_*)

__device__ __forceinline__ void ticket_add(float2* __restrict__ acc, lds_ticket_t* tag, uint32_t r, float2 v,
                                           bool pending, uint32_t lane) {
    while (__ballot(pending)) {
        if (pending) tag[r] = (uint8_t)lane;
        if (pending && tag[r] == (uint8_t)lane) {
            float2 a = acc[r];
            a.x += v.x;
            a.y += v.y;
            acc[r] = a;
            pending = false;
        }
    }
}

__global__ __launch_bounds__(kReduceThreads) void k_grid_bwd_reduce_ticket(float* __restrict__ grad_table, GridPlan plan,
                                                                            BinPlan bin,
                                                                            const uint32_t* __restrict__ cursors,
                                                                            const Item<false>* __restrict__ items) {
    extern __shared__ __align__(16) unsigned char reduce_lds[];
    float2* acc_all = reinterpret_cast<float2*>(reduce_lds);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    float2* acc = acc_all + wave * kBucketRows;
    lds_ticket_t* tag = (lds_ticket_t*)(reduce_lds + kReduceWaves * kBucketRows * sizeof(float2) + wave * kBucketRows);

    ReduceJob j;
    if (!reduce_job<false>(bin, cursors, j)) return;
    for (uint32_t i = threadIdx.x; i < kReduceWaves * kBucketRows; i += kReduceThreads) acc_all[i] = make_float2(0.f, 0.f);
    __syncthreads();

    constexpr uint32_t kUnroll = 8;
    const Item<false>* src = items + (size_t)bin.item_first[j.level] * 1024u + (size_t)j.bucket * j.cap;
    for (uint32_t base = j.begin; base < j.end; base += kUnroll * kReduceThreads) {
        Item<false> it[kUnroll];
        bool have[kUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; u++) {
            const uint32_t i = base + u * kReduceThreads + threadIdx.x;
            have[u] = i < j.end;
            it[u] = src[have[u] ? i : j.begin];
        }
#pragma unroll
        for (uint32_t u = 0; u < kUnroll; u++)
            ticket_add(acc, tag, it[u].row & (kBucketRows - 1), it[u].value(), have[u], lane);
    }
    __syncthreads();

    const uint32_t row0 = plan.off[j.level];
    const uint32_t level_rows = plan.off[j.level + 1] - row0;
    const uint32_t first_row = j.bucket << kBucketRowsLog2;
    for (uint32_t r = threadIdx.x; r < kBucketRows; r += kReduceThreads) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (uint32_t w = 0; w < kReduceWaves; w++) {
            const float2 p = acc_all[w * kBucketRows + r];
            a += p.x;
            b += p.y;
        }
        if (a == 0.f && b == 0.f) continue;
        const uint32_t row = first_row + r;
        if (row >= level_rows) continue;
        float* dst = grad_table + ((size_t)row0 + row) * 2;
        if (j.used == 1) {
            flush_row<false>(dst, a, b);
        } else {   // a heavy bucket shared by several workgroups: float atomics (what the reference does for every contribution)
            unsafeAtomicAdd(dst, a);
            unsafeAtomicAdd(dst + 1, b);
        }
    }
}

// host: bucket geometry for a chunk of `chunk` samples
BinPlan make_bin_plan(const GridPlan& plan, uint32_t levels, uint32_t chunk, bool pair_items, uint64_t* total_items_1024,
                      uint32_t* total_buckets, uint32_t* total_splits, uint32_t* shared_acc_rows = nullptr,
                      uint32_t* coarse_buckets = nullptr) {
    BinPlan b;
    memset(&b, 0, sizeof(b));
    b.levels = levels;
    uint64_t items = 0;
    uint32_t buckets = 0, wgs = 0, acc_rows = 0, coarse = 0;
    for (uint32_t l = 0; l < levels; l++) {
        const uint32_t rows = plan.off[l + 1] - plan.off[l];
        const uint32_t nb = (rows + kBucketRows - 1) >> kBucketRowsLog2;
        b.bucket_first[l] = buckets;
        b.split_first[l] = wgs;
        const uint64_t worst = (uint64_t)(pair_items ? 4 : 8) * chunk;  // every contribution of the chunk lands in this level (half tables: two per item)
        // uniform share + 25 % + slack; coarse levels rely on the run folding; beyond that: the spill accumulators (half tables,
        // exact) or float atomics (float tables)
        uint64_t cap = (worst + nb - 1) / nb;
        // Levels of few buckets are the DENSE ones (rows = cells, not hashes): the samples of a scene sit in a fraction of the
        // volume, so the buckets of that region get several times the uniform share even after the run folding — 3-4 % of the
        // training iterations overflowed a bucket there with + 25 % (round 4: the spill kernel's trace durations). They are few
        // buckets, so twice the share costs little memory; hashed levels spread whatever the scene is.
        cap = (nb <= kCoarseBuckets ? 2 * cap + cap / 2 : cap + cap / 4) + 256;
        b.cap[l] = (uint32_t)cap;
        // A level of a few buckets (the 16^3 level has two) receives all 8*B contributions in those few lists: with
        // 131072 items per workgroup only a few dozen workgroups would run (measured: 677 of 1756 us for that
        // level alone). Such levels are cut finer and flushed atomically.
        // SDFX_GRIDBWD_COARSE_SPLIT: items per K2 workgroup at the levels of few buckets (measurement aid)
        const uint32_t coarse_split = [] { const int v = dev_switch("SDFX_GRIDBWD_COARSE_SPLIT", 0); return v > 0 ? (uint32_t)v : kItemsPerSplitCoarse; }();
        // Half tables: only the levels of few buckets have the shared 64-bit accumulator that makes a split bucket's sum exact and
        // order-independent, so every other bucket is reduced by ONE workgroup however long its list (it is bounded by `cap`).
        b.per_split[l] = nb <= kCoarseBuckets ? coarse_split : (pair_items ? (1u << 30) : kItemsPerSplit);
        b.acc_first[l] = kNoSharedAcc;
        if (nb <= kCoarseBuckets) {
            b.acc_first[l] = acc_rows;
            acc_rows += nb * kBucketRows;
            if (coarse == buckets) coarse += nb;  // K3 covers the leading run of coarse levels
        }
        uint32_t splits = (uint32_t)((cap + b.per_split[l] - 1) / b.per_split[l]);
        if (splits < 1) splits = 1;
        if (splits > kMaxSplits) splits = kMaxSplits;
        b.splits[l] = splits;
        b.item_first[l] = (uint32_t)items;
        items += ((uint64_t)nb * cap + 1023) / 1024;
        buckets += nb;
        wgs += nb * splits;
        // fold lane runs where neighbouring samples (~1/600 of the unit cube apart) usually share a cell
        // (scan_cell_runs packs a cell id into 10 bits per axis: res <= 1023; SDFX_GRIDBWD_MERGE_RES moves the threshold for A/B runs)
        const uint32_t merge_res = [] { const int v = dev_switch("SDFX_GRIDBWD_MERGE_RES", 640); return (uint32_t)(v < 0 ? 0 : (v > 1023 ? 1023 : v)); }();
        if (plan.res[l] <= merge_res) b.merge_mask |= 1u << l;
    }
    b.bucket_first[levels] = buckets;
    b.split_first[levels] = wgs;
    *total_items_1024 = items;
    *total_buckets = buckets;
    *total_splits = wgs;
    if (shared_acc_rows) *shared_acc_rows = acc_rows;
    if (coarse_buckets) *coarse_buckets = coarse;
    return b;
}

bool binned_supported(uint32_t D, uint32_t C, uint32_t L, const int32_t* offsets_host) {
    if (D != 3 || C != 2 || L < 1 || L > kMaxLevels) return false;
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t rows = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        if (((rows + kBucketRows - 1) >> kBucketRowsLog2) > kMaxBucketsPerLevel) return false;
    }
    return true;
}

// scratch = [bucket cursors][diagnostics][shared 64-bit accumulators of the few-bucket levels]   <- cleared per launch
//           [spill accumulators, one per bucket (half tables); zeroed on demand][item lists]    <- never cleared
constexpr uint64_t kCursorBytes = (uint64_t)kMaxLevels * kMaxBucketsPerLevel * sizeof(uint32_t);
constexpr uint64_t kDiagOffset = kCursorBytes;   // uint32 diag[0] = some bucket overflowed, diag[1] = how many
constexpr uint64_t kDiagBytes = 64;
constexpr uint64_t kSharedAccOffset = kDiagOffset + kDiagBytes;

struct ScratchLayout {
    uint64_t cleared_bytes;   // [0, cleared_bytes) is zeroed at every launch
    uint64_t spill_offset, items_offset, total_bytes;
};

ScratchLayout scratch_layout(const GridPlan& plan, uint32_t levels, uint32_t chunk, bool half) {
    uint64_t items;
    uint32_t nb, wg, acc_rows;
    make_bin_plan(plan, levels, chunk, half, &items, &nb, &wg, &acc_rows);
    ScratchLayout s;
    s.cleared_bytes = kSharedAccOffset + (half ? (uint64_t)acc_rows * 2 * sizeof(unsigned long long) : 0);
    s.spill_offset = s.cleared_bytes;
    s.items_offset = s.spill_offset + (half ? (uint64_t)nb * kSpillWords * sizeof(unsigned long long) : 0);
    s.total_bytes = s.items_offset + items * 1024u * (half ? sizeof(Item<true>) : sizeof(Item<false>));
    return s;
}

// K1's per-XCD ranges cut by COST instead of count. The ranges stay contiguous in (virtual level, tile) order — the slices of a
// bucket's item list that share cache lines stay on one XCD's L2 (see the flat-mapping measurement below) — but every XCD gets the
// same share of the estimated work. cost[l] = relative cost of one tile of level l.
void balance_plan(GridPlan& p, uint32_t levels, const float* cost) {
    double cum[kMaxLevels + 1];
    cum[0] = 0.0;
    double mean = 0.0;
    for (uint32_t l = 0; l < levels; l++) mean += cost[l] > 0 ? cost[l] : 0;
    mean = mean > 0 ? mean / levels : 1.0;
    auto c_of = [&](uint32_t v) { const double c = cost[p.order[v]]; return c > 1e-3 * mean ? c : 1e-3 * mean; };
    for (uint32_t v = 0; v < levels; v++) cum[v + 1] = cum[v] + (double)p.tiles * c_of(v);
    const double total = cum[levels];
    uint32_t cut[kXcds + 1];
    cut[0] = 0;
    cut[kXcds] = levels * p.tiles;
    for (uint32_t k = 1; k < kXcds; k++) {
        const double target = total * k / kXcds;
        uint32_t v = 0;
        while (v + 1 < levels && cum[v + 1] <= target) v++;
        uint64_t off = (uint64_t)((target - cum[v]) / c_of(v));
        if (off > p.tiles) off = p.tiles;
        cut[k] = v * p.tiles + (uint32_t)off;
        if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
        if (cut[k] > cut[kXcds]) cut[k] = cut[kXcds];
    }
    for (uint32_t k = 0; k < kXcds; k++) { p.start[k] = cut[k]; p.end[k] = cut[k + 1]; }
}

// relative cost of one K1 tile per level for the -O configuration (16 levels, base 16, 2048 finest), from the kernel-trace
// durations per max_level in profiles/r02_gridbwd_balance.txt; SDFX_GRIDBWD_LEVEL_COST="c0,c1,..." overrides
const float* k1_level_cost(uint32_t levels) {
    static float table[kMaxLevels];
    static const bool init = [] {
        const float measured[16] = {51, 41, 52, 19, 5, 5, 29, 26, 27, 29, 32, 36, 25, 30, 34, 59};
        for (uint32_t l = 0; l < kMaxLevels; l++) table[l] = l < 16 ? measured[l] : 30.f;
        if (const char* e = dev_string("SDFX_GRIDBWD_LEVEL_COST")) {
            uint32_t l = 0;
            while (*e && l < kMaxLevels) {
                char* end = nullptr;
                const float v = strtof(e, &end);
                if (end == e) break;
                table[l++] = v;
                e = (*end == ',') ? end + 1 : end;
            }
        }
        return true;
    }();
    (void)init; (void)levels;
    return table;
}

// ---- K2 of the fine levels beside K1 of the coarse ones -------------------------------------------------------------------------
// One side stream and two events per device, created at first use (the trainer's eager warm-up iterations come before its graph
// captures; inside a capture the fork / join below makes the side stream part of the captured graph).
struct Overlap { hipStream_t side; hipEvent_t fork, join; };
Overlap* overlap_streams() {
    if (dev_switch("SDFX_GRIDBWD_OVERLAP", 0) == 0) return nullptr;   // product library: never (see the call site)
    static std::mutex m;
    static std::map<int, Overlap> per_device;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(m);
    auto it = per_device.find(dev);
    if (it == per_device.end()) {
        Overlap o{};
        if (hipStreamCreateWithFlags(&o.side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&o.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&o.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            o = Overlap{};
        }
        it = per_device.emplace(dev, o).first;
    }
    return it->second.side ? &it->second : nullptr;
}

// Can the plan be cut into a FINE group (levels >= lvl_split, walked first by every XCD) and a COARSE group (the rest, walked after)?
// True for the -O grid (16 levels: XCD k walks level 15 - k, then level k). `fine` / `coarse` = the plan restricted to each group.
// The levels whose buckets are split over several K2 workgroups (shared accumulators, rounded by K3) must all be coarse.
bool two_level_groups(const GridPlan& p, const BinPlan& bin, uint32_t levels, GridPlan& fine, GridPlan& coarse, uint32_t& lvl_split) {
    if (levels < 2 || p.tiles == 0) return false;
    lvl_split = levels / 2;
    fine = p; coarse = p;
    for (uint32_t k = 0; k < kXcds; k++) {
        uint32_t mid = p.end[k];
        bool seen_coarse = false;
        for (uint32_t item = p.start[k]; item < p.end[k];) {
            const uint32_t virt = item / p.tiles, level = p.order[virt];
            const uint32_t seg_end = (virt + 1) * p.tiles < p.end[k] ? (virt + 1) * p.tiles : p.end[k];
            if (level >= lvl_split) {
                if (seen_coarse) return false;            // fine after coarse on this XCD
            } else if (!seen_coarse) {
                seen_coarse = true;
                mid = item;
            }
            item = seg_end;
        }
        fine.end[k] = mid;
        coarse.start[k] = mid;
    }
    for (uint32_t l = lvl_split; l < levels; l++)
        if (bin.acc_first[l] != kNoSharedAcc) return false;
    return true;
}

// SDFX_GRIDBWD_BALANCE=1 (devtools build): cost-balanced ranges; default: equal tile counts
bool k1_balance_enabled() {
    return dev_switch("SDFX_GRIDBWD_BALANCE", 0) == 1;
}

}  // namespace

extern "C" {

// host-only: the per-XCD item ranges K1 would use (ranges[2k], ranges[2k+1] = [start, end) of XCD k in virtual-level-major
// (level order[v], tile) items), tiles per level in *tiles; balance: 0 equal counts, 1 cost-balanced. Returns the level count.
int sdfx_grid_backward_plan(const int32_t* offsets_host, uint32_t max_level, float S, uint32_t H, uint32_t B, int balance,
                            int32_t* ranges, uint32_t* tiles) {
    if (!offsets_host || !ranges || max_level < 1 || max_level > kMaxLevels) return SDFX_E_INVALID;
    GridPlan plan = make_plan(offsets_host, max_level, S, H, 2, 2, (uint64_t)div_up(B, kBinThreads * kPointsPerThread) * kTile);
    if (balance) balance_plan(plan, max_level, k1_level_cost(max_level));
    for (uint32_t k = 0; k < kXcds; k++) { ranges[2 * k] = (int32_t)plan.start[k]; ranges[2 * k + 1] = (int32_t)plan.end[k]; }
    if (tiles) *tiles = plan.tiles;
    return (int)max_level;
}

// bytes of scratch that let `chunk_points` samples be processed per pass (more samples are chunked)
uint64_t sdfx_grid_encode_backward_binned_scratch_bytes(const int32_t* offsets_host, uint32_t L, uint32_t max_level, float S,
                                                         uint32_t H, uint32_t chunk_points, int is_half) {
    if (!offsets_host || L < 1 || L > kMaxLevels || max_level < 1 || max_level > L) return 0;
    const GridPlan plan = make_plan(offsets_host, max_level, S, H, 2, is_half ? 2 : 4, chunk_points);
    return scratch_layout(plan, max_level, chunk_points, is_half != 0).total_bytes;
}

// Diagnostics (synchronises the stream), out[4]: [0] = buckets of the LAST chunk launched on `scratch` whose list overflowed, i.e.
// whose sums came from the spill accumulators (half tables; 0 for float tables), [1] = reserved (0), [2] / [3] = buckets / launches
// that overflowed on this device since the library was loaded.
int sdfx_grid_encode_backward_binned_stats(const void* scratch, uint32_t* out, sdfx_stream_t stream) {
    SDFX_REQUIRE(scratch && out, "grid_encode_backward_binned_stats: null pointer");
    hipStream_t st = as_stream(stream);
    uint32_t host[2] = {0, 0};
    if (hipMemcpyAsync(host, static_cast<const char*>(scratch) + kDiagOffset, sizeof(host), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("grid_encode_backward_binned_stats: copy failed");
        return SDFX_E_LAUNCH;
    }
    out[0] = host[1];
    out[1] = 0;
    uint32_t totals[2] = {0, 0};
    if (hipMemcpyFromSymbol(totals, HIP_SYMBOL(g_spill_totals), sizeof(totals)) != hipSuccess) {
        set_error("grid_encode_backward_binned_stats: reading the totals failed");
        return SDFX_E_LAUNCH;
    }
    out[2] = totals[0];
    out[3] = totals[1];
    return SDFX_OK;
}

// Same contract as sdfx_grid_encode_backward (table gradient only: D = 3, C = 2, no dy_dx), plus scratch.
// Returns SDFX_E_UNSUPPORTED for other shapes so the caller can take the atomic path.
int sdfx_grid_encode_backward_binned(const void* grad, const float* inputs, const int32_t* offsets_host,
                                     void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                     uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                     uint32_t interp, int is_half, int grad_layout, void* scratch, uint64_t scratch_bytes,
                                     sdfx_stream_t stream) {
    SDFX_REQUIRE(grad && offsets_host && grad_embeddings && scratch, "grid_encode_backward_binned: null pointer");
    {
        const StencilSrc src = stencil_src();   // sdfx_set_stencil_source: inputs may be NULL, B must be 7 M
        SDFX_REQUIRE(src.xyzs ? (B == 0 || (uint64_t)src.M * 7u == B) : inputs != nullptr,
                     "grid_encode_backward_binned: null inputs, or a stencil source whose 7 M differs from B = %u", B);
    }
    if (!binned_supported(D, C, L, offsets_host)) {
        set_error("grid_encode_backward_binned: only D=3, C=2, levels of at most %u rows", kMaxBucketsPerLevel * kBucketRows);
        return SDFX_E_UNSUPPORTED;
    }
    SDFX_REQUIRE(max_level >= 1 && max_level <= L, "grid_encode_backward_binned: max_level must be in [1, L]");
    SDFX_REQUIRE(gridtype <= 1 && interp <= 1 && (grad_layout == 0 || grad_layout == 1), "grid_encode_backward_binned: bad enum");
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(grad_embeddings) % (is_half ? 4 : 8)) == 0 &&
                     (reinterpret_cast<uintptr_t>(scratch) % 16) == 0,
                 "grid_encode_backward_binned: grad_embeddings / scratch misaligned");
    if (B == 0) return SDFX_OK;
    hipStream_t st = as_stream(stream);
    const uint32_t eb = is_half ? 2 : 4;
    {   // K2 needs more than the 64 KiB of LDS a kernel gets without asking
        static bool lds_ok = [] {
            if (kReduceFixedLdsBytes && hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_bwd_reduce_fixed),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kReduceFixedLdsBytes) != hipSuccess)
                return false;
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_bwd_reduce_ticket),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kReduceLdsBytes) == hipSuccess;
        }();
        SDFX_REQUIRE(lds_ok, "grid_encode_backward_binned: cannot reserve LDS for the reduce kernel");
    }

    // largest chunk (multiple of 512 samples) whose item lists fit the scratch
    uint32_t chunk = B;
    {
        const uint32_t gran = kBinThreads * kPointsPerThread;
        chunk = ((chunk + gran - 1) / gran) * gran;
        for (;;) {
            const GridPlan p = make_plan(offsets_host, max_level, S, H, 2, eb, chunk);
            if (scratch_layout(p, max_level, chunk, is_half != 0).total_bytes <= scratch_bytes) break;
            SDFX_REQUIRE(chunk > gran, "grid_encode_backward_binned: scratch too small (%llu bytes)",
                         (unsigned long long)scratch_bytes);
            chunk = ((chunk / 2 + gran - 1) / gran) * gran;
        }
    }
    dev_ctl_sync();
    char* const base = static_cast<char*>(scratch);
    uint32_t* cursors = static_cast<uint32_t*>(scratch);
    unsigned long long* shared_acc = reinterpret_cast<unsigned long long*>(base + kSharedAccOffset);

    for (uint32_t b0 = 0; b0 < B; b0 += chunk) {
        const uint32_t b1 = b0 + chunk < B ? b0 + chunk : B;
        const uint32_t n = b1 - b0;
        // one plan tile = one K1 workgroup = kBinThreads * kPointsPerThread samples
        GridPlan plan = make_plan(offsets_host, max_level, S, H, 2, eb,
                                  (uint64_t)div_up(n, kBinThreads * kPointsPerThread) * kTile);
        // Each XCD walks its own range of whole levels. The ranges are unbalanced for K1 (levels 15 + 0 on one XCD cost ~4x levels
        // 8 + 7 on another), but giving every XCD the same mix of levels was measured SLOWER twice: 784 -> 1091 us at B = 1.81 M
        // with one list per bucket (round 2), 851 -> 1411 us in the iteration with one sub-list per (bucket, XCD) (round 3,
        // profiles/r03_scatter_flat_sublists_ab.txt); that mapping is gone from the code.
        if (k1_balance_enabled()) balance_plan(plan, max_level, k1_level_cost(max_level));
        uint64_t items_1024;
        uint32_t nbuckets, nsplits, acc_rows, coarse_buckets;
        const BinPlan bin = make_bin_plan(plan, max_level, chunk, is_half != 0, &items_1024, &nbuckets, &nsplits, &acc_rows, &coarse_buckets);
        const ScratchLayout lay = scratch_layout(plan, max_level, chunk, is_half != 0);
        void* items = base + lay.items_offset;
        unsigned long long* spill_acc = reinterpret_cast<unsigned long long*>(base + lay.spill_offset);
        uint32_t* diag = reinterpret_cast<uint32_t*>(base + kDiagOffset);
        zero_device(scratch, lay.cleared_bytes, st);  // cursors, diagnostics, shared accumulators
        BinLevels lv;
        memset(&lv, 0, sizeof(lv));
        for (uint32_t l = 0; l < max_level; l++) lv.lv[l] = make_level_const(offsets_host, l, S, H);
        const int sel = (interp ? 4 : 0) | (align_corners ? 2 : 0) | (gridtype == 0 ? 1 : 0);
#define SDFX_BIN(HALF_, INTERP_, ALIGN_, HASH_, PLAN_, STREAM_)                                                                    \
    hipLaunchKernelGGL((k_grid_bwd_bin<HALF_, INTERP_, ALIGN_, HASH_>), dim3(plan_grid_size(PLAN_)), dim3(kBinThreads), 0, STREAM_, \
                       static_cast<const typename Elem<HALF_>::type*>(grad), inputs,                                              \
                       static_cast<typename Elem<HALF_>::type*>(grad_embeddings), B, L, b0, b1, PLAN_, bin, lv, grad_layout,       \
                       cursors, static_cast<Item<HALF_>*>(items), row_limit(), stencil_src())
#define SDFX_BIN_SEL(HALF_, PLAN_, STREAM_)                                                                                       \
    switch (sel) {                                                                                                                \
        case 0: SDFX_BIN(HALF_, 0u, false, false, PLAN_, STREAM_); break;                                                         \
        case 1: SDFX_BIN(HALF_, 0u, false, true, PLAN_, STREAM_); break;                                                          \
        case 2: SDFX_BIN(HALF_, 0u, true, false, PLAN_, STREAM_); break;                                                          \
        case 3: SDFX_BIN(HALF_, 0u, true, true, PLAN_, STREAM_); break;                                                           \
        case 4: SDFX_BIN(HALF_, 1u, false, false, PLAN_, STREAM_); break;                                                         \
        case 5: SDFX_BIN(HALF_, 1u, false, true, PLAN_, STREAM_); break;                                                          \
        case 6: SDFX_BIN(HALF_, 1u, true, false, PLAN_, STREAM_); break;                                                          \
        default: SDFX_BIN(HALF_, 1u, true, true, PLAN_, STREAM_); break;                                                          \
    }
#define SDFX_SPILL(INTERP_, ALIGN_, HASH_, PLAN_, STREAM_, FLAG_)                                                                 \
    hipLaunchKernelGGL((k_grid_bwd_spill<INTERP_, ALIGN_, HASH_>), dim3(div_up(plan_grid_size(PLAN_), kSpillTilesPerGroup)),      \
                       dim3(kBinThreads), 0, STREAM_, static_cast<const __half*>(grad), inputs,                                   \
                       static_cast<__half*>(grad_embeddings), B, L, b0, b1, PLAN_, bin, lv, grad_layout, cursors, row_limit(),    \
                       stencil_src(), spill_acc, diag, plan_grid_size(PLAN_), FLAG_)
        // K1 (+ the spill pair, which exits at once unless a bucket overflowed: see the head of the file) and K2 for the levels
        // [lvl_lo, lvl_hi) of plan PLAN_ on stream STREAM_; FLAG_ = that group's overflow word of `diag`
        auto half_group_k1 = [&](const GridPlan& gp, hipStream_t stream) { SDFX_BIN_SEL(true, gp, stream) };
        auto half_group_k2 = [&](const GridPlan& gp, hipStream_t stream, uint32_t lvl_lo, uint32_t lvl_hi, uint32_t flag) {
            const uint32_t bk_lo = bin.bucket_first[lvl_lo], bk_hi = bin.bucket_first[lvl_hi];
            const uint32_t wg_lo = bin.split_first[lvl_lo], wg_hi = bin.split_first[lvl_hi];
            if (bk_hi == bk_lo) return;
            hipLaunchKernelGGL(k_grid_bwd_spill_zero, dim3(bk_hi - bk_lo), dim3(256), 0, stream, bin, cursors, spill_acc, diag, bk_lo, flag);
            switch (sel) {
                case 0: SDFX_SPILL(0u, false, false, gp, stream, flag); break;
                case 1: SDFX_SPILL(0u, false, true, gp, stream, flag); break;
                case 2: SDFX_SPILL(0u, true, false, gp, stream, flag); break;
                case 3: SDFX_SPILL(0u, true, true, gp, stream, flag); break;
                case 4: SDFX_SPILL(1u, false, false, gp, stream, flag); break;
                case 5: SDFX_SPILL(1u, false, true, gp, stream, flag); break;
                case 6: SDFX_SPILL(1u, true, false, gp, stream, flag); break;
                default: SDFX_SPILL(1u, true, true, gp, stream, flag); break;
            }
            hipLaunchKernelGGL(k_grid_bwd_reduce_fixed, dim3(wg_hi - wg_lo), dim3(kReduceThreadsFixed), kReduceFixedLdsBytes, stream,
                               static_cast<__half*>(grad_embeddings), plan, bin, cursors, static_cast<const Item<true>*>(items),
                               shared_acc, spill_acc, wg_lo);
        };
        if (is_half) {
            // MEASUREMENT VARIANT (devtools library, SDFX_GRIDBWD_OVERLAP=1; the product library always takes the one-stream branch
            // below): two groups of levels when the plan allows (the -O grid: every XCD walks one level of the fine half, then one of
            // the coarse half), K2 of the FINE half — three quarters of the items, a stream of HBM reads into LDS atomics — on a side
            // stream beside K1 of the COARSE half (a latency chain writing short runs). Built on the idea that the two leave each
            // other's resources idle; measured in round 5: they do not — 1352-1395 us against 1180 us per call at B = 3.26 M (K1 span
            // 900 -> 1104 us, K2 346 -> 569 us: the list reads and the list writes fight over the same HBM), results identical
            // (profiles/r05_scatter_k1_k2_overlap.txt).
            //   main:  zero, K1(fine) -e1-> K1(coarse), [spill pair], ............. -e2-> K2(coarse), K3
            //   side:            -e1-> [spill pair], K2(fine) -e2->
            GridPlan fine, coarse;
            uint32_t lvl_split = 0;
            Overlap* ov = two_level_groups(plan, bin, max_level, fine, coarse, lvl_split) ? overlap_streams() : nullptr;
            if (ov && hipEventRecord(ov->fork, st) != hipSuccess) ov = nullptr;   // (nothing launched yet beyond the zeroing)
            if (ov) {
                half_group_k1(fine, st);
                (void)hipEventRecord(ov->fork, st);
                (void)hipStreamWaitEvent(ov->side, ov->fork, 0);
                half_group_k2(fine, ov->side, lvl_split, max_level, 0u);
                (void)hipEventRecord(ov->join, ov->side);
                half_group_k1(coarse, st);
                (void)hipStreamWaitEvent(st, ov->join, 0);
                half_group_k2(coarse, st, 0u, lvl_split, 2u);
            } else {
                half_group_k1(plan, st);
                half_group_k2(plan, st, 0u, max_level, 0u);
            }
            if (coarse_buckets)
                hipLaunchKernelGGL(k_grid_bwd_finish, dim3(coarse_buckets), dim3(256), 0, st, static_cast<__half*>(grad_embeddings),
                                   plan, bin, cursors, shared_acc);
        } else {
            SDFX_BIN_SEL(false, plan, st)
            hipLaunchKernelGGL(k_grid_bwd_reduce_ticket, dim3(nsplits), dim3(kReduceThreads), kReduceLdsBytes, st,
                               static_cast<float*>(grad_embeddings), plan, bin, cursors, static_cast<const Item<false>*>(items));
        }
#undef SDFX_SPILL
#undef SDFX_BIN_SEL
#undef SDFX_BIN
    }
    return check_launch("grid_encode_backward_binned");
}

}  // extern "C"
