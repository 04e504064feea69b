// tiles.hip -- column-sliced two-phase SpMV ("propagation blocking") for matrices
// whose x gather misses the caches.
//
// Why: a one-pass row-major SpMV moves a whole 128-B line per 8-byte x element
// that misses L1: 56 G nnz/s when x lives beyond L2, i.e.
// <= 9 % of the HBM roofline whatever the kernel.  This path replaces the random
// access by two streaming passes:
//   phase 1  pb_expand : the matrix is stored SLICE-MAJOR (slices of 16 Ki columns,
//            inside a slice sorted by row, then column).  A workgroup stages its
//            x slice in LDS (128 KiB), streams {value, 16-bit local column} and
//            writes the products back in the same order (coalesced).
//   phase 2  pb_reduce : a workgroup owns a bin of R consecutive rows whose partial
//            sums live in LDS (R x 8 B).  The bin's entries, taken tile by tile
//            (slice 0..S-1), form one dense BIN-MAJOR index space: the 16-bit local
//            rows are stored in that order and stream in; every lane maps its
//            bin-major position to the slice-major position of the product through
//            the bin's tile table (start, offset), so the products are read as runs
//            of one tile each with all 64 lanes busy; ds_add_f64 into LDS, and the
//            bin of y is stored at the end.
// ~28-30 B/nnz of pure streaming instead of 12 B/nnz + one line per nnz.  Measured on c5 (1.6 G nnz,
// profiles/r01_summary.md): phase 1 5.4 ms = 5.8 TB/s of the bytes it moves, phase 2 3.4 ms = 6.0 TB/s.
// Multi-GPU (one process per GPU): phase 2 can also deliver y to the other ranks -- stores fused into the
// kernel, or a push kernel beside it fed by per-bin ready flags.
//
// Relation to the reference: its CPU variants spmvTilesCSR / spmvTilesAllocdCSR
// (src/SpMV_CSR_OMP.c:101-226) use the same 2-D decomposition -- column
// partitions, partial results per tile, final reduction; this is the GPU form
// of that idea with the partition sizes dictated by LDS.
//
// Summation order: products of one row are added by LDS atomics in arrival
// order -> results agree with the serial oracle to rounding (not bitwise, and
// not bitwise run-to-run).  The parity gate and the 1e-13 * sum|a x| check hold.
//
// The slice-major format is built ON THE DEVICE from the device CSR (one-time
// analysis, like the row-block table): rocPRIM's stable radix sort by slice id
// keeps the (row, column) order of the CSR inside every slice.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#include <chrono>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <algorithm>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

namespace spmvhip {

constexpr uint32_t PB_CBITS   = 14;
constexpr uint32_t PB_C       = 1u << PB_CBITS;     // columns per slice: 16 Ki = 128 KiB of x in LDS (19 Ki = 152 KiB measured no faster: c3 1.075 vs 1.050 ms, c5 equal)
static_assert(PB_C % 1024 == 0 && PB_C * 8 <= 160 * 1024 && PB_C <= 65536, "slice width");
constexpr uint32_t PB_R_MAX   = 20000;             // rows per bin: <= 156 KiB of y in LDS (160 KiB per CU on gfx950)
constexpr uint32_t PB_CUS_DEFAULT = 256;           // compute units when the device does not say (phase 2 keeps one workgroup per CU)
constexpr uint32_t PB_CHUNK   = 1u << 17;           // entries of one slice handled by one phase-1 workgroup
constexpr int      PB_THREADS = 1024;
constexpr uint32_t PD_WAVES   = 4;                 // deterministic form: wavefronts of a phase-2 workgroup = sub-bins of a bin (c5: 4 -> 9.4 ms, 8 -> 10.1, 16 -> 11.2)
constexpr int      PD_THREADS = PD_WAVES * 64;
constexpr size_t   PB_RESIDENT_BYTES = 288ull << 20; // products up to this size stay in the 256 MiB Infinity Cache between the phases (c2, 244 MiB: 0.156 ms plain vs 0.176 ms nt stores)

// Rows -> bins.  Up to three zones of equal-height bins: n1 bins of a rows, n2 bins of b rows, the rest c rows
// each.  Uniform formats use the last zone only (n1 = n2 = 0, c = R).  The TAPERED form -- one round of low bins
// first and last, full-height bins between -- exists for the multi-GPU exchange: phase 2 finishes its bins in
// rounds and a row can only leave for the other ranks once its bin is finished, so the height of the FIRST round
// sets when the links start to work and the height of the LAST round what is still to be sent when the kernel
// ends, while the tile length (= efficiency) is set by the bins between.
struct BinMap {
    uint32_t n1 = 0, n2 = 0, a = 1, b = 1, c = 1;
    uint64_t z1 = 0, z2 = 0;                        // first row of zone 2 / zone 3
    __host__ __device__ uint32_t binOf(uint64_t row) const {
        return row < z1 ? (uint32_t)(row / a) : row < z2 ? n1 + (uint32_t)((row - z1) / b) : n1 + n2 + (uint32_t)((row - z2) / c);
    }
    __host__ __device__ uint64_t row0(uint32_t bin) const {
        return bin < n1 ? (uint64_t)bin * a : bin < n1 + n2 ? z1 + (uint64_t)(bin - n1) * b : z2 + (uint64_t)(bin - n1 - n2) * c;
    }
    __host__ __device__ uint32_t height(uint32_t bin) const { return bin < n1 ? a : bin < n1 + n2 ? b : c; }
    __host__ __device__ uint32_t maxHeight() const { uint32_t m = c; if (n1 && a > m) m = a; if (n2 && b > m) m = b; return m; }
};

struct TileFormat {
    uint32_t S = 0, B = 0, R = 0;                   // slices, bins, rows of the highest bin
    BinMap    bins;
    uint64_t nnz = 0;
    void*     slab = nullptr;                       // ONE allocation behind val / lcol / lrow (it doubles as sort buffer while the format is built)
    double*   val = nullptr;                        // slice-major values
    uint16_t* lcol = nullptr;                       // slice-major: column - slice*PB_C
    uint16_t* lrow = nullptr;                       // BIN-major: row - bin*R
    bool      det = false;                          // deterministic form: `bins` / B / binPos describe SUB-bins (PD_WAVES per bin of the API)
    uint32_t* pidx = nullptr;                       // deterministic form: BIN-major position -> slice-major position of its product
    size_t    tempBytes = 0;                        // peak of the temporaries of the build
    uint2*    tl = nullptr;                         // non-empty tiles in bin-major order: {first bin-major position,
                                                    // slice-major start - bin-major start (mod 2^32)}; sentinels follow
    uint32_t  nList = 0;
    uint32_t* binPos = nullptr;                     // [B+1] bin-major position where each bin starts
    uint32_t* waveTile = nullptr;                   // [B * waves] tile-list index where each wavefront of phase 2 starts
    uint3*    work = nullptr;                       // phase-1 work items {slice, begin, end}
    uint32_t  nWork = 0;
    bool      ntStore = true;                       // phase 1 stores the products non-temporally (streams larger than the Infinity Cache)
    uint32_t* ready = nullptr;                      // [B] epoch of the last step whose bin was stored (push kernel hand-off)
    uint32_t* pushFail = nullptr;                   // set by the push kernel when a flag never arrived
    uint32_t  epoch = 0;
    uint32_t  cus = PB_CUS_DEFAULT;                 // CUs of the device the format was built on
    size_t    bytes = 0;
    spmvTilesOpts opts{0, 0, -1, 0, 0};             // what the format was built with (0 / -1 = automatic)
    // what the API calls a bin (spmvHipTilesShape, hipSpMVTilesReduce): the deterministic form groups PD_WAVES sub-bins
    uint32_t apiBins() const { return det ? (B + PD_WAVES - 1) / PD_WAVES : B; }
    uint32_t apiRows() const { return det ? PD_WAVES * bins.c : R; }
    BinMap   apiMap() const { if (!det) return bins; BinMap m; m.c = PD_WAVES * bins.c; return m; }
    uint32_t  chunk = 0;                            // phase-1 work item size in entries actually used
    double    buildMs = 0;                          // wall time of the one-time build (events on the null stream around it)
    double    allocMs = 0;                          // ... of which the host spent in hipMalloc (format, product workspace, 36 B/entry of temporaries)
};

struct TileDst { double* p[SPMV_MAX_PEERS]; uint32_t n; };
struct TileSignal { uint32_t* ready; uint32_t epoch; };        // per-bin "this bin of y is stored" flags (push kernel)

namespace {

__device__ __forceinline__ uint64_t lin_block() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }

// ---- build kernels ---------------------------------------------------------------------------
// What travels through the sort with every entry: the sort moves {value, local column, row} itself, so that nothing
// has to be fetched through a permutation afterwards (a first version sorted indices and then gathered AS, JA and the
// row of every entry through them: 1.6 G x 3 isolated 128-B line fetches = 165 of the 217 ms of the c5 build).
// ... 12 bytes: the value as two words (the struct then needs 4-byte alignment only) and the row; the local column rides in the
// low bits of the 32-bit sort key (slice << 14 | local column, sorted on the slice bits only: stable, so the (row, column)
// order of the CSR survives inside every slice).
struct PbPay { uint32_t vlo, vhi, row; };
static_assert(sizeof(PbPay) == 12 && alignof(PbPay) == 4, "payload layout");
__host__ __device__ __forceinline__ PbPay pb_pack(double v, uint32_t row) {
    uint64_t b;
    memcpy(&b, &v, 8);
    return PbPay{(uint32_t)b, (uint32_t)(b >> 32), row};
}
__host__ __device__ __forceinline__ double pb_value(const PbPay& e) {
    const uint64_t b = ((uint64_t)e.vhi << 32) | e.vlo;
    double v;
    memcpy(&v, &b, 8);
    return v;
}

// one wavefront per row: slice id (the sort key) and payload of every entry, in CSR order
template <typename I>
__global__ __launch_bounds__(256) void pb_payload_kernel(uint64_t M, const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
                                                         const double* __restrict__ AS, uint32_t* __restrict__ keys,
                                                         PbPay* __restrict__ pay) {
    const uint64_t r = lin_block() * 4 + threadIdx.x / 64;
    if (r >= M) return;
    const uint64_t b = IRP[r], e = IRP[r + 1];
    for (uint64_t j = b + threadIdx.x % 64; j < e; j += 64) {
        const uint32_t c = JA[j];
        keys[j] = c;                                 // = slice << PB_CBITS | local column
        pay[j] = pb_pack(AS[j], (uint32_t)r);
    }
}

// mark where each tile starts in the sorted (slice-major) order.  tile id t = slice*B + bin is non-decreasing
// along that order, so the first entry of a tile also fills the start of every empty tile before it.
__global__ __launch_bounds__(256) void pb_bounds_kernel(
    uint64_t nnz, const PbPay* __restrict__ spay, const uint32_t* __restrict__ skeys,
    uint32_t B, BinMap bm, uint64_t nTiles, uint32_t* __restrict__ tileStart) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint64_t t = (uint64_t)(skeys[p] >> PB_CBITS) * B + bm.binOf(spay[p].row);
    uint64_t tPrev;                                  // tile of the previous entry, or "-1"
    if (p == 0) tPrev = ~0ull;
    else tPrev = (uint64_t)(skeys[p - 1] >> PB_CBITS) * B + bm.binOf(spay[p - 1].row);
    if (t != tPrev)
        for (uint64_t u = tPrev + 1; u <= t; ++u) tileStart[u] = (uint32_t)p;      // tPrev+1 wraps to 0 for p == 0
    if (p == nnz - 1)
        for (uint64_t u = t + 1; u <= nTiles; ++u) tileStart[u] = (uint32_t)nnz;
}

// tile lengths in BIN-major order (i = bin*S + slice); their exclusive scan is every tile's bin-major start
__global__ __launch_bounds__(256) void pb_lens_kernel(uint32_t S, uint32_t B, const uint32_t* __restrict__ tileStart,
                                                      uint32_t* __restrict__ lens, uint32_t* __restrict__ flags) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i >= (uint64_t)S * B) return;
    const uint32_t b = (uint32_t)(i / S), s = (uint32_t)(i % S);
    const uint64_t t = (uint64_t)s * B + b;
    const uint32_t len = tileStart[t + 1] - tileStart[t];
    lens[i] = len;
    flags[i] = len != 0;
}

// the list of non-empty tiles in bin-major order + where each bin starts
__global__ __launch_bounds__(256) void pb_list_kernel(uint32_t S, uint32_t B, uint64_t nnz, const uint32_t* __restrict__ tileStart,
                                                      const uint32_t* __restrict__ lens, const uint32_t* __restrict__ bmStart,
                                                      const uint32_t* __restrict__ listIdx, uint32_t nList,
                                                      uint2* __restrict__ tl,
                                                      uint32_t* __restrict__ binPos, uint32_t* __restrict__ binTile) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i >= (uint64_t)S * B) return;
    const uint32_t b = (uint32_t)(i / S), s = (uint32_t)(i % S);
    if (lens[i]) {
        const uint64_t t = (uint64_t)s * B + b;
        tl[listIdx[i]] = make_uint2(bmStart[i], tileStart[t] - bmStart[i]);
    }
    if (s == 0) { binPos[b] = bmStart[i]; binTile[b] = listIdx[i]; }
    if (i + 1 == (uint64_t)S * B) { binPos[B] = (uint32_t)nnz; binTile[B] = nList; }
}

// the sorted payload is already slice-major: split it into the value and local-column streams, and place the local
// rows in bin-major order
__global__ __launch_bounds__(256) void pb_place_kernel(
    uint64_t nnz, const PbPay* __restrict__ spay, const uint32_t* __restrict__ skeys,
    uint32_t S, uint32_t B, BinMap bm, const uint32_t* __restrict__ tileStart, const uint32_t* __restrict__ bmStart,
    double* __restrict__ val, uint16_t* __restrict__ lcol, uint16_t* __restrict__ lrow, uint32_t* __restrict__ pidx) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= nnz) return;
    const PbPay e = spay[p];
    const uint32_t key = skeys[p];
    const uint32_t bin = bm.binOf(e.row), slice = key >> PB_CBITS;
    val[p] = pb_value(e);
    lcol[p] = (uint16_t)(key & (PB_C - 1));
    const uint32_t within = (uint32_t)p - tileStart[(uint64_t)slice * B + bin];
    const uint32_t at = bmStart[(uint64_t)bin * S + slice] + within;
    lrow[at] = (uint16_t)(e.row - bm.row0(bin));
    if (pidx) pidx[at] = (uint32_t)p;
}

__global__ __launch_bounds__(256) void pb_fill_kernel(uint32_t* p, uint64_t n, uint32_t v) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- phase 1 -----------------------------------------------------------------------------------
// Each lane owns 2 consecutive entries per step, consecutive lanes consecutive pairs: every wavefront access is
// one dense span (1 KiB of values or products, 256 B of columns).  The work item's range is peeled to a
// multiple of 64 entries at the front, so every span starts on a cache line.  Both matter
// (scripts/microbench_expand.hip, ns per 1000 entries): 4 entries per lane with two 16-B accesses -- each
// wavefront instruction touching half of every 32 B -- 3.73 aligned / 3.97 when the range starts 32 B off a
// line; this layout 3.32 aligned / 3.82 unaligned.  Two batches of P1_DEPTH steps ping-pong in registers (a
// `cur = nxt` copy would wait for nxt's loads); loads use clamped addresses instead of a bounds branch so the
// compiler's vmcnt stays exact; the first batch is issued before the x slice is staged so the stream is
// already moving during the LDS fill.
typedef double   dbl2 __attribute__((ext_vector_type(2)));
typedef uint16_t ush2 __attribute__((ext_vector_type(2)));

constexpr int      P1_DEPTH = 8;                    // steps per batch
constexpr uint32_t P1_STEP  = 2 * PB_THREADS;       // entries one workgroup step covers
struct P1Regs { dbl2 a[P1_DEPTH]; ush2 c[P1_DEPTH]; };

// requires ve - 2 >= first entry of the vector range
// UNIT: every value of the matrix is `unitValue` (DevMat::unit) -- the 8 B/nnz value stream is not read
template <bool UNIT>
__device__ __forceinline__ void p1_load(P1Regs& r, uint32_t p, uint32_t ve, const double* __restrict__ val,
                                        const uint16_t* __restrict__ lcol, double unitValue) {
#pragma unroll
    for (int u = 0; u < P1_DEPTH; ++u) {
        const uint32_t q = min(p + u * P1_STEP, ve - 2u);
        if (UNIT) r.a[u] = dbl2{unitValue, unitValue};
        else      r.a[u] = __builtin_nontemporal_load((const dbl2*)(val + q));
        r.c[u] = __builtin_nontemporal_load((const ush2*)(lcol + q));
    }
}

template <bool NT>
__device__ __forceinline__ void p1_store(const P1Regs& r, uint32_t p, uint32_t ve, const double* xs, double* __restrict__ prod) {
#pragma unroll
    for (int u = 0; u < P1_DEPTH; ++u) {
        const uint32_t q = p + u * P1_STEP;
        if (q < ve) {
            dbl2 o;
            o.x = r.a[u].x * xs[r.c[u].x]; o.y = r.a[u].y * xs[r.c[u].y];
            if (NT) __builtin_nontemporal_store(o, (dbl2*)(prod + q)); else *(dbl2*)(prod + q) = o;
        }
    }
}

template <bool NT, bool UNIT>
__global__ __launch_bounds__(PB_THREADS) void pb_expand_kernel(
    const uint3* __restrict__ work, const double* __restrict__ val, const uint16_t* __restrict__ lcol,
    const double* __restrict__ x, uint64_t N, double* __restrict__ prod, double unitValue) {
    extern __shared__ double xs[];                  // PB_C doubles
    const uint3 w = work[lin_block()];
    const uint32_t begin = w.y, end = w.z;
    const uint32_t vb = min(end, (begin + 63u) & ~63u);         // first line-aligned entry inside the range
    const uint32_t ve = vb + ((end - vb) & ~1u);                // end of the whole pairs
    const bool vec = ve > vb;                                   // uniform

    uint32_t p = vb + 2 * threadIdx.x;
    P1Regs cur, nxt;
    const uint64_t col0 = (uint64_t)w.x * PB_C;
    // Order of the first requests: the x slice BEFORE the first stream batch.  vmcnt counts in issue order, so the LDS
    // fill below waits for the slice only while the batch behind it is still in flight; with the batch first (round 1)
    // the fill waited for both (c5: 8.13-8.20 against 8.26-8.46 ms over 5 fresh processes each,
    // profiles/r02_tiles_load_policy.log).
    {
        double xv[PB_C / PB_THREADS];
#pragma unroll
        for (uint32_t i = 0; i < PB_C / PB_THREADS; ++i) {
            const uint32_t k = threadIdx.x + i * PB_THREADS;
            xv[i] = (col0 + k < N) ? x[col0 + k] : 0.0;
        }
        if (vec) p1_load<UNIT>(cur, p, ve, val, lcol, unitValue);
#pragma unroll
        for (uint32_t i = 0; i < PB_C / PB_THREADS; ++i) xs[threadIdx.x + i * PB_THREADS] = xv[i];
    }
    __syncthreads();

    // scalar head (at most 63 entries) and tail (at most 1)
    if (threadIdx.x < vb - begin) { const uint32_t q = begin + threadIdx.x; prod[q] = (UNIT ? unitValue : val[q]) * xs[lcol[q]]; }
    if (threadIdx.x < end - ve)   { const uint32_t q = ve + threadIdx.x;    prod[q] = (UNIT ? unitValue : val[q]) * xs[lcol[q]]; }

    if (vec) {
        constexpr uint32_t BATCH = P1_DEPTH * P1_STEP;
        for (; p < ve; p += 2 * BATCH) {
            p1_load<UNIT>(nxt, p + BATCH, ve, val, lcol, unitValue);
            p1_store<NT>(cur, p, ve, xs, prod);
            p1_load<UNIT>(cur, p + 2 * BATCH, ve, val, lcol, unitValue);
            p1_store<NT>(nxt, p + BATCH, ve, xs, prod);
        }
    }
}

// ---- phase 2 -----------------------------------------------------------------------------------
// The bin's entries are the bin-major positions [binPos[bin], binPos[bin+1]); each of the 16 wavefronts owns a
// contiguous sub-range and walks it 64 positions at a time.  Position v belongs to the tile k with
// tl[k].start <= v < tl[k+1].start; its product sits at slice-major position v + tl[k].delta.  The walk through
// the tile table is wavefront-uniform, so it runs on the SCALAR unit: records come in through s_load (scalar
// cache, lgkmcnt) two tiles ahead of their use, the per-lane choice is one v_cndmask per tile boundary, and the
// vector memory pipeline carries nothing but the two streams.  (Keeping a window of records in vector registers,
// refilled with a vector load inside the loop, made the compiler wait for ALL outstanding loads at every step
// -- vmcnt is in-order -- and ran 34 % slower than the descriptor version this replaces.)
// Two batches of P2_DEPTH steps are in flight: the loads of the next batch are issued before the LDS atomics
// of the current one.
constexpr int      P2_DEPTH = 8;                    // steps per batch (4 / 6 / 8 / 12 equal within the noise)
constexpr uint32_t P2_WAVES = PB_THREADS / 64;
constexpr uint16_t P2_NONE  = 0xFFFF;               // "no entry" (local rows are < PB_R_MAX)
constexpr uint32_t P2_RUNS_FLAG = 0x80000000u;      // in waveTile[]: this wavefront's range holds runs of equal rows
constexpr uint32_t TL_PAD   = 4;                    // sentinel records behind the list (the cursor reads 2 ahead)

__host__ __device__ __forceinline__ uint32_t p2_sub(uint32_t len) { return ((len + P2_WAVES - 1) / P2_WAVES + 63u) & ~63u; }

struct P2Regs { double pv[P2_DEPTH]; uint16_t rv[P2_DEPTH]; };
struct P2Cursor { uint32_t k, delta; uint2 n1, n2; };            // all wavefront-uniform: tile k, its delta, records k+1, k+2

__device__ __forceinline__ void p2_fetch(P2Regs& r, P2Cursor& c, uint32_t vbase, uint32_t we, uint32_t lane,
                                         const uint2* __restrict__ tl, const double* __restrict__ prod,
                                         const uint16_t* __restrict__ lrow) {
    // No branch around the loads: positions past the end are clamped to the last entry (a valid, already cached
    // address) and marked P2_NONE afterwards, so the compiler can count the outstanding loads exactly and
    // waits only for the batch it consumes (with the loads under `if` it fell back to vmcnt(0) everywhere).
#pragma unroll
    for (int u = 0; u < P2_DEPTH; ++u) {
        const uint32_t vs = vbase + 64u * u;
        const uint32_t v = vs + lane, vlast = min(vs + 63u, we - 1u);
        uint32_t d = c.delta;
        while (c.n1.x <= vlast) {                    // the next tile starts inside this step
            d = (v >= c.n1.x) ? c.n1.y : d;
            c.delta = c.n1.y;
            c.n1 = c.n2;
            ++c.k;
            c.n2 = tl[c.k + 2];
        }
        const uint32_t vc = min(v, we - 1u);
        // The products are read with the DEFAULT cache policy, not non-temporally: a wavefront's 64 positions are 512
        // contiguous bytes at an arbitrary offset (a tile starts wherever the slice-major order put it), so consecutive
        // steps share the line they meet in.  Through L1 the second step hits; a non-temporal load bypasses L1 and the
        // line, already dropped by L2, comes from HBM a second time -- that was the "run-boundary" traffic of round 1
        // (phase 2 of c5: 3.19 -> 2.92 ms, the SpMV 8.49 -> 8.15 ms, medians of 6 fresh processes each;
        // profiles/r02_tiles_load_policy.log).  The row ids are read once: non-temporal.
        r.pv[u] = prod[(uint32_t)(vc + d)];
        const uint16_t row = __builtin_nontemporal_load(lrow + vc);
        r.rv[u] = v < we ? row : P2_NONE;
    }
}

// One step of 64 entries into the bin.  Entries of one row are neighbours in the bin-major order (a tile is sorted by
// row), so on matrices with clustered columns -- and in the heavy rows of the power-law ones -- many lanes of a
// wavefront carry the SAME row and their ds_add_f64 serialise on one LDS address (band +-16 Ki: phase 2 660 us
// against 376 us for the same matrix with uniform columns).  Runs of equal rows are therefore summed in registers
// first -- a segmented suffix sum inside each 16-lane DPP row, 4 shift-and-add steps -- and only the first lane of a
// run adds to LDS.  Whether a wavefront's range is worth it is decided when the format is built (at least one
// entry in eight repeats its predecessor's row) and kept as a flag in waveTile[]: ranges without runs -- all of them
// on uniform columns -- run the plain loop at no cost (testing every step in the kernel cost 2-7 % there).
template <int D>
__device__ __forceinline__ double p2_shl(double v) {            // value of lane + D inside the 16-lane row, else 0
    const uint64_t b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x100 + D, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x100 + D, 0xF, 0xF, false);
    return __longlong_as_double(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ uint32_t p2_prev_row(uint32_t row) {  // row of lane - 1 inside the 16-lane row; lane % 16 == 0 gets ~0
    return (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)row, 0x111, 0xF, 0xF, false);
}

__device__ __forceinline__ void p2_add_runs(double* yb, uint16_t rv, double pv, uint32_t lane) {
    const uint32_t row = rv;
    const bool follows = row == p2_prev_row(row) && row != P2_NONE;
    const uint64_t dup = __builtin_amdgcn_ballot_w64(follows);
    const uint64_t m = (~dup >> 1) >> lane;                      // run starts behind this lane
    const uint32_t e = m ? lane + 1u + (uint32_t)__builtin_ctzll(m) : 64u;     // where this lane's run ends
    double v = rv != P2_NONE ? pv : 0.0;
    double t;
    t = p2_shl<1>(v); v += lane + 1 < e ? t : 0.0;
    t = p2_shl<2>(v); v += lane + 2 < e ? t : 0.0;
    t = p2_shl<4>(v); v += lane + 4 < e ? t : 0.0;
    t = p2_shl<8>(v); v += lane + 8 < e ? t : 0.0;
    if (!follows && rv != P2_NONE) atomicAdd(&yb[rv], v);
}

// the finished bin: y, then -- by MODE -- the other ranks' copies of y or the ready flag of the push kernel
template <int MODE, int THREADS>
__device__ __forceinline__ void p2_finish(const double* yb, uint32_t R, uint64_t r0, uint64_t M, double* __restrict__ y,
                                          const TileDst& extra, const TileSignal& sig, uint32_t bin) {
    for (uint32_t k = threadIdx.x; k < R; k += THREADS)
        if (r0 + k < M) y[r0 + k] = yb[k];
    if (MODE == 1) {
        // destination-major (long contiguous runs per link); workgroups start at different destinations so that
        // all links carry traffic all the time
        for (uint32_t i = 0; i < extra.n; ++i) {
            double* __restrict__ dst = extra.p[(i + bin) % extra.n];
            for (uint32_t k = threadIdx.x; k < R; k += THREADS)
                if (r0 + k < M) dst[r0 + k] = yb[k];
        }
    }
    if (MODE == 2) {
        // hand the finished bin to the push kernel (another CU, maybe another XCD): every storing wave waits for its
        // stores, the workgroup meets, ONE lane writes the XCD's L2 back (agent release) and only then sets the flag
        // (MI355X_MICROARCH.md "inter-workgroup visibility", producer form; the explicit wait after the fence is the
        // documented guard against the compiler dropping it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(sig.ready + bin, sig.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// MULTI: the finished bin of y is also stored to `extra.n` further destinations -- the copies of y that the other
// ranks of a multi-GPU run hold, mapped into this process (peer windows over xGMI, peer.hip): the all-gather of y
// is fused into the producing kernel as point-to-point stores, one 512-B run per wavefront instruction.
template <int MODE>                                 // 0: y only; 1: y + extra destinations; 2: y + per-bin ready flag
__global__ __launch_bounds__(PB_THREADS) void pb_reduce_kernel(
    BinMap bm, uint32_t binBegin, uint32_t binEnd, uint64_t M, const uint32_t* __restrict__ binPos, const uint32_t* __restrict__ waveTile,
    const uint2* __restrict__ tl, const double* __restrict__ prod, const uint16_t* __restrict__ lrow,
    double* __restrict__ y, TileDst extra, TileSignal sig) {
    extern __shared__ double yb[];                  // R doubles
    const uint64_t bin = binBegin + lin_block();
    if (bin >= binEnd) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)), lane = threadIdx.x % 64;
    const uint32_t v0 = binPos[bin], v1 = binPos[bin + 1];
    const uint32_t sub = p2_sub(v1 - v0);
    const uint32_t wb = v0 + wave * sub, we = min(v1, wb + sub);
    P2Cursor c;
    P2Regs a, b;
    bool runs = false;                               // uniform per wavefront
    const bool busy = wave * sub < v1 - v0;          // uniform per wavefront
    if (busy) {
        const uint32_t wt = waveTile[bin * P2_WAVES + wave];
        runs = (wt & P2_RUNS_FLAG) != 0;
        c.k = wt & ~P2_RUNS_FLAG;
        c.delta = tl[c.k].y;
        c.n1 = tl[c.k + 1];
        c.n2 = tl[c.k + 2];
        p2_fetch(a, c, wb, we, lane, tl, prod, lrow);                   // moving before the bin is zeroed
    }
    const uint32_t R = bm.height((uint32_t)bin);
    for (uint32_t k = threadIdx.x; k < R; k += PB_THREADS) yb[k] = 0.0;
    __syncthreads();
    if (busy) {
        // ping-pong between the two register batches (an `a = b` copy would need b's loads to have landed, i.e.
        // a full wait at the end of every iteration); batches past the end hold nothing but P2_NONE
        if (!runs) {
            for (uint32_t v = wb; v < we; v += 2 * P2_DEPTH * 64) {
                p2_fetch(b, c, v + P2_DEPTH * 64, we, lane, tl, prod, lrow);
#pragma unroll
                for (int u = 0; u < P2_DEPTH; ++u)
                    if (a.rv[u] != P2_NONE) atomicAdd(&yb[a.rv[u]], a.pv[u]);
                p2_fetch(a, c, v + 2 * P2_DEPTH * 64, we, lane, tl, prod, lrow);
#pragma unroll
                for (int u = 0; u < P2_DEPTH; ++u)
                    if (b.rv[u] != P2_NONE) atomicAdd(&yb[b.rv[u]], b.pv[u]);
            }
        } else {                                     // this wavefront's entries come in runs of equal rows
            for (uint32_t v = wb; v < we; v += 2 * P2_DEPTH * 64) {
                p2_fetch(b, c, v + P2_DEPTH * 64, we, lane, tl, prod, lrow);
#pragma unroll
                for (int u = 0; u < P2_DEPTH; ++u) p2_add_runs(yb, a.rv[u], a.pv[u], lane);
                p2_fetch(a, c, v + 2 * P2_DEPTH * 64, we, lane, tl, prod, lrow);
#pragma unroll
                for (int u = 0; u < P2_DEPTH; ++u) p2_add_runs(yb, b.rv[u], b.pv[u], lane);
            }
        }
    }
    __syncthreads();
    p2_finish<MODE, PB_THREADS>(yb, R, bm.row0((uint32_t)bin), M, y, extra, sig, (uint32_t)bin);
}

// Deterministic form of phase 2.  A bin of the API is PD_WAVES consecutive SUB-bins; the format's tables (bins, binPos,
// local rows) are built on the sub-bins, and wavefront w of the bin's workgroup walks sub-bin w from its first entry to its
// last in bin-major order -- slice by slice, a tile sorted by (row, column) -- with plain ds_add_f64.  A row therefore
// receives its products from ONE wavefront in ascending column order (lanes of one instruction that meet in a row are served
// in lane order), which is the order of the serial oracle, whatever the scheduler does and however the rows were cut into
// bins or shards.
// With a sub-bin per wavefront the tiles are PD_WAVES times shorter than a bin's (c5: 20 entries), and the scalar walk
// through the tile table that serves the arrival-order kernel -- one dependent s_load per tile boundary -- becomes the
// bottleneck (first version: phase 2 of c5 8.7 ms against 3.0).  This form therefore stores, per bin-major
// position, the slice-major position of its product (`pidx`, +4 B/nnz) and runs as a three-stage software pipeline like
// the stripes kernel: stream (pidx, local row; non-temporal) two batches ahead, product gather one batch ahead (default
// policy: neighbouring sub-bins read the same lines), LDS adds -- vmcnt counts in issue order, so a gather is only
// waited for while younger stream loads are outstanding if those were issued after it.
constexpr int PD_DEPTH = 8;                          // steps of 64 entries per batch
struct PdStream { uint32_t idx[PD_DEPTH]; uint16_t row[PD_DEPTH]; uint32_t first; };
struct PdGather { double p[PD_DEPTH]; };

__device__ __forceinline__ void pd_stream(PdStream& s, uint32_t vb, uint32_t we, uint32_t lane, const uint32_t* __restrict__ pidx,
                                          const uint16_t* __restrict__ lrow) {
    s.first = vb;
#pragma unroll
    for (int u = 0; u < PD_DEPTH; ++u) {
        const uint32_t v = vb + 64u * u + lane, vc = min(v, we - 1u);       // clamped, not branched: vmcnt stays exact
        s.idx[u] = __builtin_nontemporal_load(pidx + vc);
        const uint16_t r = __builtin_nontemporal_load(lrow + vc);
        s.row[u] = v < we ? r : P2_NONE;
    }
}
__device__ __forceinline__ void pd_gather(PdGather& g, const PdStream& s, const double* __restrict__ prod) {
#pragma unroll
    for (int u = 0; u < PD_DEPTH; ++u) g.p[u] = prod[s.idx[u]];
}
__device__ __forceinline__ void pd_add(double* yw, const PdStream& s, const PdGather& g) {
#pragma unroll
    for (int u = 0; u < PD_DEPTH; ++u)
        if (s.row[u] != P2_NONE) atomicAdd(&yw[s.row[u]], g.p[u]);
}

template <int MODE>
__global__ __launch_bounds__(PD_THREADS) void pb_reduce_det_kernel(
    BinMap bm, uint32_t nSub, uint32_t binBegin, uint32_t binEnd, uint64_t M, const uint32_t* __restrict__ binPos,
    const uint32_t* __restrict__ pidx, const double* __restrict__ prod, const uint16_t* __restrict__ lrow,
    double* __restrict__ y, TileDst extra, TileSignal sig) {
    extern __shared__ double yb[];                  // PD_WAVES * (rows of a sub-bin) doubles
    const uint64_t bin = binBegin + lin_block();
    if (bin >= binEnd) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)), lane = threadIdx.x % 64;
    const uint32_t sb = (uint32_t)bin * PD_WAVES + wave;
    const uint32_t Rs = bm.c;                        // uniform sub-bin height (no tapered zones in this form)
    double* yw = yb + wave * Rs;
    uint32_t wb = 0, we = 0;
    if (sb < nSub) { wb = binPos[sb]; we = binPos[sb + 1]; }
    const bool busy = we > wb;                       // uniform per wavefront
    constexpr uint32_t BATCH = PD_DEPTH * 64;
    PdStream a, b, c, d;
    PdGather g0, g1;
    if (busy) {                                      // first three batches: moving before the bin is zeroed
        pd_stream(a, wb, we, lane, pidx, lrow);
        pd_stream(b, wb + BATCH, we, lane, pidx, lrow);
        pd_stream(c, wb + 2 * BATCH, we, lane, pidx, lrow);
    }
    const uint32_t R = PD_WAVES * Rs;
    for (uint32_t k = threadIdx.x; k < R; k += PD_THREADS) yb[k] = 0.0;
    __syncthreads();
    if (busy) {
        pd_gather(g0, a, prod);
        uint32_t next = wb + 3 * BATCH;
        // one stage: CUR is added, NXT gathered, FAR (the set CUR's predecessor freed) streamed
#define PD_STAGE(CUR, NXT, FAR, GC, GN)                   \
        if (CUR.first >= we) break;                       \
        pd_gather(GN, NXT, prod);                         \
        pd_stream(FAR, next, we, lane, pidx, lrow);       \
        next += BATCH;                                    \
        pd_add(yw, CUR, GC);
        for (;;) {
            PD_STAGE(a, b, d, g0, g1)
            PD_STAGE(b, c, a, g1, g0)
            PD_STAGE(c, d, b, g0, g1)
            PD_STAGE(d, a, c, g1, g0)
        }
#undef PD_STAGE
    }
    __syncthreads();
    p2_finish<MODE, PD_THREADS>(yb, R, bm.row0((uint32_t)bin * PD_WAVES), M, y, extra, sig, (uint32_t)bin);
}

// The push kernel: runs BESIDE phase 2 (own stream, a few wavefronts per CU, no LDS -- phase 2 leaves wave slots free)
// and copies every bin of y to the other ranks' vectors as soon as its flag shows this step's epoch.  Workgroup w takes
// the bins w, w + W, ... in ascending order (the order phase 2 produces them in) and serves ALL peers from one read of
// the bin: per lane PUSH_DEPTH 16-byte loads, then those registers stored to every peer (posted stores: with a link
// hop of microseconds what counts is bytes in flight -- a first version with one peer per workgroup and one load per
// store moved 9 GB/s per workgroup under phase 2's memory load).  Unlike the fused store the reduction workgroups never
// wait for a link, and the links work from the first finished bin on instead of in bursts at the end of every round of
// bins.  A flag that does not arrive within ~2 s marks the step failed (the caller then sees an incomplete y) instead
// of hanging the GPU.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
constexpr int      PUSH_THREADS = 256;
constexpr int      PUSH_DEPTH = 8;
constexpr uint32_t PUSH_WGS = 64;
constexpr uint32_t PUSH_SPIN_LIMIT = 1u << 20;
__global__ __launch_bounds__(PUSH_THREADS) void pb_push_kernel(
    BinMap bm, uint32_t B, uint64_t M, const uint32_t* __restrict__ ready, uint32_t epoch, const double* __restrict__ y,
    TileDst dst, uint32_t* __restrict__ fail) {
    __shared__ uint32_t ok;
    for (uint32_t bin = blockIdx.x; bin < B; bin += gridDim.x) {
        if (threadIdx.x == 0) {
            uint32_t spins = 0, seen;
            while ((seen = __hip_atomic_load(ready + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != epoch && ++spins < PUSH_SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(64);
            ok = seen == epoch;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // consumer form: one poll, one acquire, wait, barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) atomicExch(fail, 1u); return; }
        const uint64_t r0 = bm.row0(bin), r1 = min(r0 + bm.height(bin), M);
        // 16-byte accesses; y and the peers' vectors share their alignment (same row offset from 256-B aligned bases):
        // peel one row when the rank's first row is odd
        const uint64_t head = ((uintptr_t)(y + r0) & 15) ? 1 : 0;
        const uint64_t pairs = (r1 - r0 - head) / 2;
        const double* __restrict__ src = y + r0 + head;
        for (uint64_t i = threadIdx.x; i < pairs; i += (uint64_t)PUSH_THREADS * PUSH_DEPTH) {
            dbl2_t v[PUSH_DEPTH];
#pragma unroll
            for (int u = 0; u < PUSH_DEPTH; ++u) {
                const uint64_t q = i + (uint64_t)u * PUSH_THREADS;
                v[u] = *(const dbl2_t*)(src + 2 * (q < pairs ? q : i));
            }
            for (uint32_t k = 0; k < dst.n; ++k) {                      // peers in a bin-dependent order: all links busy
                double* __restrict__ out = dst.p[(k + bin) % dst.n] + r0 + head;
#pragma unroll
                for (int u = 0; u < PUSH_DEPTH; ++u) {
                    const uint64_t q = i + (uint64_t)u * PUSH_THREADS;
                    if (q < pairs) *(dbl2_t*)(out + 2 * q) = v[u];
                }
            }
        }
        if (threadIdx.x < dst.n) {
            double* __restrict__ out = dst.p[threadIdx.x];
            if (head && r1 > r0) out[r0] = y[r0];
            if ((r1 - r0 - head) & 1) out[r1 - 1] = y[r1 - 1];
        }
        __syncthreads();                            // `ok` is rewritten in the next round
    }
}

// where every wavefront of phase 2 starts in the tile list: the last tile of the bin that starts at or before
// the wavefront's first position
__global__ __launch_bounds__(256) void pb_wavetile_kernel(uint32_t B, const uint32_t* __restrict__ binPos,
                                                          const uint32_t* __restrict__ binTile, const uint2* __restrict__ tl,
                                                          const uint32_t* __restrict__ dupCount, uint32_t* __restrict__ waveTile) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i >= (uint64_t)B * P2_WAVES) return;
    const uint32_t b = (uint32_t)(i / P2_WAVES), wv = (uint32_t)(i % P2_WAVES);
    const uint32_t v0 = binPos[b], v1 = binPos[b + 1];
    const uint32_t sub = p2_sub(v1 - v0);
    const uint32_t wb = v0 + wv * sub;
    uint32_t lo = binTile[b], hi = binTile[b + 1];  // answer in [lo, hi)
    if (wb >= v1 || lo >= hi) { waveTile[i] = lo; return; }
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (tl[mid].x <= wb) lo = mid; else hi = mid;
    }
    const uint32_t len = min(v1, wb + sub) - wb;
    waveTile[i] = lo | (dupCount[i] * 8 >= len ? P2_RUNS_FLAG : 0u);
}

// entries that repeat the row of their predecessor in the bin-major order, counted per wavefront range of phase 2.
// The 64 positions of a wavefront fall into one or two of those ranges, so the lanes that share the range of the first
// counting lane add their count with ONE atomic (a banded matrix repeats the row in 9 entries out of 10: one atomic per
// entry on a few thousand counters took 166 ms of the format build, profiles/r02_default_bench_kernel_stats.csv).
__global__ __launch_bounds__(256) void pb_dupcount_kernel(uint32_t B, uint64_t nnz, const uint32_t* __restrict__ binPos,
                                                          const uint16_t* __restrict__ lrow, uint32_t* __restrict__ dupCount) {
    const uint64_t v = lin_block() * 256 + threadIdx.x;
    bool dup = v != 0 && v < nnz && lrow[v] == lrow[v - 1];
    uint64_t slot = 0;
    if (dup) {
        uint32_t lo = 0, hi = B;                    // bin with binPos[bin] <= v < binPos[bin + 1]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (binPos[mid] <= v) lo = mid; else hi = mid;
        }
        const uint32_t v0 = binPos[lo], v1 = binPos[lo + 1];
        if (v == v0) dup = false;                   // the predecessor belongs to another bin
        else slot = (uint64_t)lo * P2_WAVES + (uint32_t)((v - v0) / p2_sub(v1 - v0));
    }
    uint64_t todo = __builtin_amdgcn_ballot_w64(dup);
    while (todo) {                                  // wavefront-uniform: one round per distinct counter (mostly one)
        const int leader = __builtin_ctzll(todo);
        const uint64_t s0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(slot >> 32), leader) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)slot, leader);
        const uint64_t same = __builtin_amdgcn_ballot_w64(dup && slot == s0);
        if ((int)(threadIdx.x % 64) == leader) atomicAdd(&dupCount[s0], (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
}

#define PB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "libspmvhip: tiles: %s: %s\n", #expr, hipGetErrorString(e_)); return EXIT_FAILURE; } } while (0)

struct TempBuf {
    void* p = nullptr;
    ~TempBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 1)); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

// The products live in ONE workspace per device, shared by every matrix of the process: phase 1 of a matrix fills
// it, phase 2 of the same matrix empties it, and launches are ordered on the library stream.  Re-using the same
// addresses matters when a matrix is processed as several row groups whose products fit the 256 MiB Infinity
// Cache: the lines are overwritten while still cached instead of being written back to HBM and re-allocated
// (scripts/microbench_mall.hip: write-then-read of a 128 MiB buffer 6.6 TB/s, of a 2 GiB one 4.9-5.1 TB/s).
// used: work on the products may be unfinished; lastStream: where it was enqueued (compared, never handed to HIP again --
// its owner may have destroyed it meanwhile); marked: lastUse was recorded behind that work
struct ProdWorkspace { double* p = nullptr; void* raw = nullptr; size_t cap = 0; hipEvent_t lastUse = nullptr; hipStream_t lastStream = nullptr; bool used = false, marked = false; };
static ProdWorkspace g_prod[16];

static ProdWorkspace* prodSlot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    return &g_prod[dev];
}

static double* prodWorkspace(size_t n, bool grow) {
    ProdWorkspace* w = prodSlot();
    if (!w) return nullptr;
    if (n <= w->cap || !grow) return n <= w->cap ? w->p : nullptr;
    (void)hipDeviceSynchronize();                   // nothing may still read the old buffer
    w->used = w->marked = false;
    if (w->raw) (void)hipFree(w->raw);
    w->p = nullptr; w->raw = nullptr; w->cap = 0;
    if (hipMalloc(&w->raw, n * sizeof(double)) != hipSuccess) { w->raw = nullptr; return nullptr; }
    w->p = static_cast<double*>(w->raw);            // (placement relative to the other arrays, 0 ... 1 MiB offsets: no effect)
    w->cap = n;
    return w->p;
}

// The workspace is shared by every matrix of the device: a phase 1 must not start before the phase 2 that reads the
// previous products has finished.  On one stream that is stream order.  When the stream changes (spmvHipSetStream, two
// handles driven from two streams, the streams of successive shard objects) the new stream first waits for the event
// that prodMark() recorded behind the previous use.  The previous STREAM is not touched: by now it may be destroyed, and
// a HIP call on a dead handle is undefined (it happened to return an error; under the sanitizer build it crashed).
static int prodHandover(hipStream_t stream) {
    ProdWorkspace* w = prodSlot();
    if (!w) return EXIT_FAILURE;
    if (w->used && w->lastStream != stream) {
        if (!w->marked || hipStreamWaitEvent(stream, w->lastUse, 0) != hipSuccess) {
            (void)hipGetLastError();                // no mark (phase 1 without its phase 2, a use inside a graph capture): wait for the device
            if (hipDeviceSynchronize() != hipSuccess) return EXIT_FAILURE;
        }
    }
    w->lastStream = stream;
    w->used = true;
    w->marked = false;
    return EXIT_SUCCESS;
}

// behind the last kernel of a use of the products on `stream`
static void prodMark(hipStream_t stream) {
    ProdWorkspace* w = prodSlot();
    if (!w) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {   // an event recorded inside a capture belongs to the graph
        (void)hipGetLastError();
        return;
    }
    if (!w->lastUse && hipEventCreateWithFlags(&w->lastUse, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
    w->marked = hipEventRecord(w->lastUse, stream) == hipSuccess;
    if (!w->marked) (void)hipGetLastError();
}

// after a device synchronise: nothing is pending on the products
static void prodIdle() {
    if (ProdWorkspace* w = prodSlot()) w->used = w->marked = false;
}

void freeTilesWorkspace() {
    int keep = 0;
    (void)hipGetDevice(&keep);
    for (int d = 0; d < 16; ++d)
        if (g_prod[d].p || g_prod[d].lastUse) {
            (void)hipSetDevice(d);
            if (g_prod[d].raw) (void)hipFree(g_prod[d].raw);
            if (g_prod[d].lastUse) (void)hipEventDestroy(g_prod[d].lastUse);
            g_prod[d] = ProdWorkspace{};
        }
    (void)hipSetDevice(keep);
}

uint64_t tilesBinRow(const DevMat* d, uint32_t bin) {
    if (!d->tiles) return 0;
    const TileFormat* t = d->tiles;
    return bin >= t->apiBins() ? d->M : std::min<uint64_t>(d->M, t->apiMap().row0(bin));
}

void freeTiles(TileFormat* t) {
    if (!t) return;
    (void)hipFree(t->slab); (void)hipFree(t->tl); (void)hipFree(t->pidx);
    (void)hipFree(t->binPos); (void)hipFree(t->waveTile); (void)hipFree(t->work); (void)hipFree(t->ready);
    delete t;
}

// a handle holds at most one format of each form; the requested one becomes the active slot
void useTiles(DevMat* d, bool deterministic) {
    if (d->tiles && d->tiles->det == deterministic) return;
    if (d->tiles || d->tilesAlt) std::swap(d->tiles, d->tilesAlt);
    if (d->tiles && d->tiles->det != deterministic) std::swap(d->tiles, d->tilesAlt);      // (only one slot was filled, with the other form)
}

// `opts` == nullptr: automatic arrival-order format, kept if one exists.  Explicit options: the existing format of that FORM
// is replaced (the other form, if the handle holds it, is untouched).
int buildTiles(DevMat* d, const spmvTilesOpts* opts) {
    useTiles(d, opts && opts->deterministic);
    if (d->tiles && !opts) return EXIT_SUCCESS;
    if (d->kind != Kind::CSR) return EXIT_FAILURE;
    const spmvTilesOpts o = opts ? *opts : spmvTilesOpts{0, 0, -1, 0, 0};
    if (o.deterministic && o.taper) { fprintf(stderr, "libspmvhip: tiles: the deterministic form has no tapered bins\n"); return EXIT_FAILURE; }
    if (o.rowsPerBin != 0 && (o.rowsPerBin < 64 || o.rowsPerBin > PB_R_MAX)) {
        fprintf(stderr, "libspmvhip: tiles: rowsPerBin = %u is not 0 (automatic) or 64..%u\n", o.rowsPerBin, PB_R_MAX);
        return EXIT_FAILURE;
    }
    if (o.chunk != 0 && (o.chunk < 4096 || o.chunk > (1u << 24))) {
        fprintf(stderr, "libspmvhip: tiles: chunk = %u is not 0 (automatic) or 4096..2^24\n", o.chunk);
        return EXIT_FAILURE;
    }
    if (d->tiles) { freeTiles(d->tiles); d->tiles = nullptr; }
    const uint64_t nnz = d->NZ, M = d->M, N = d->N;
    if (nnz >= IRP32_LIMIT || nnz == 0) { fprintf(stderr, "libspmvhip: tiles: nnz = %lu unsupported (needs 0 < nnz < 2^32)\n", (unsigned long)nnz); return EXIT_FAILURE; }
    const uint64_t S64 = (N + PB_C - 1) / PB_C;
    if (S64 > 65535) { fprintf(stderr, "libspmvhip: tiles: %lu columns exceed 65535 slices\n", (unsigned long)N); return EXIT_FAILURE; }
    TileFormat* t = new TileFormat;
    struct Guard { TileFormat*& t; ~Guard() { if (t) freeTiles(t); } } guard{t};      // every early return frees the half-built format
    {   // one workgroup per CU of the CURRENT device in phases 1 and 2: rounds are counted in units of its CU count
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            t->cus = (uint32_t)cus;
    }
    const uint32_t PB_CUS = t->cus;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    PB_TRY(hipEventCreate(&ev0));
    PB_TRY(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evGuard{ev0, ev1};
    PB_TRY(hipEventRecord(ev0, nullptr));
    t->opts = o;
    t->S = (uint32_t)S64;
    // Rows per bin R (any value up to PB_R_MAX, not a power of two): phase 2 runs ONE workgroup per CU at a time (its
    // bin of y fills the LDS), so the bins are processed in rounds of PB_CUS and a partly filled last round leaves
    // most of the chip idle -- 610 bins of 16 Ki rows on 256 CUs are 2.4 rounds of work in the time of ~2.7.  For
    // matrices of up to a few thousand bins R is therefore chosen so that the bin count is a multiple of PB_CUS
    // (10 M rows: 512 bins of 19 532 rows = exactly two rounds, and 20 % longer tiles); beyond that the tail is
    // negligible and R is simply the largest that fits.  Measured (this box, ms): see DESIGN.md section 7.
    uint32_t R = PB_R_MAX;
    {
        const uint64_t bMin = (M + PB_R_MAX - 1) / PB_R_MAX;
        if (bMin <= 8 * PB_CUS) {
            const uint64_t b = (bMin + PB_CUS - 1) / PB_CUS * PB_CUS;
            R = (uint32_t)std::max<uint64_t>(64, ((M + b - 1) / b + 63) / 64 * 64);
            R = std::min(R, PB_R_MAX);
        }
    }
    if (o.rowsPerBin) R = o.rowsPerBin;
    t->bins = BinMap{};
    t->bins.c = R;
    const bool taper = o.taper != 0;
    if (taper && M >= (uint64_t)4 * PB_CUS * 1024) {
        // one round of quarter-height bins first and last, full-height bins (a multiple of PB_CUS of them) between
        const uint32_t low = std::max<uint32_t>(64, (R / 4 + 63) / 64 * 64);
        const uint64_t mid = M - (uint64_t)2 * PB_CUS * low;
        const uint64_t nb = ((mid + PB_R_MAX - 1) / PB_R_MAX + PB_CUS - 1) / PB_CUS * PB_CUS;
        const uint32_t high = std::min<uint32_t>(PB_R_MAX, (uint32_t)(((mid + nb - 1) / nb + 63) / 64 * 64));
        const uint64_t n2 = (mid + high - 1) / high;                       // the last full-height bin may reach into the low zone:
        t->bins.n1 = PB_CUS; t->bins.a = low;                              // zone 3 simply starts where zone 2 ends
        t->bins.n2 = (uint32_t)n2; t->bins.b = high;
        t->bins.c = low;
        t->bins.z1 = (uint64_t)PB_CUS * low;
        t->bins.z2 = t->bins.z1 + n2 * high;
        if (t->bins.z2 >= M) { t->bins.n2 = (uint32_t)((M - t->bins.z1 + high - 1) / high); t->bins.z2 = t->bins.z1 + (uint64_t)t->bins.n2 * high; }
    }
    t->det = o.deterministic != 0;
    if (t->det) {                                    // sub-bins: a quarter of the bin each, PD_WAVES of them share a workgroup's LDS
        const uint32_t Rs = o.rowsPerBin ? (o.rowsPerBin + PD_WAVES - 1) / PD_WAVES
                                         : std::min<uint32_t>(PB_R_MAX / PD_WAVES / 64 * 64, std::max<uint32_t>(64, ((R + PD_WAVES - 1) / PD_WAVES + 63) / 64 * 64));
        t->bins = BinMap{};
        t->bins.c = std::max<uint32_t>(1, std::min<uint32_t>(Rs, PB_R_MAX / PD_WAVES));
    }
    t->R = t->bins.maxHeight();
    t->B = t->bins.z2 >= M ? t->bins.n1 + t->bins.n2 : t->bins.n1 + t->bins.n2 + (uint32_t)((M - t->bins.z2 + t->bins.c - 1) / t->bins.c);
    t->nnz = nnz;
    const uint64_t nTiles = (uint64_t)t->S * t->B;
    if (nTiles >= (1ull << 32) - 2) { fprintf(stderr, "libspmvhip: tiles: too many tiles\n"); return EXIT_FAILURE; }

    // Memory.  The format's three arrays are ONE slab (val | lcol | lrow, each part 256-B aligned), and while the format is
    // built that slab and the device's product workspace (8 B/nnz, needed anyway) serve as sort buffers: the unsorted
    // payload lives in the slab, the two key buffers in the workspace, and the only temporary of size is the second payload
    // buffer, 12 B/nnz (round 2 allocated 36 B/nnz of its own plus, inside rocPRIM's pointer interface, 18 more: on c5
    // ~90 GB mapped for a 19 GB format, a second of hipMalloc).  rocPRIM's double-buffer interface sorts between the two
    // pairs of buffers without further full-size storage.
    TempBuf payB, sortTmp, tileStart;
    auto fail = [&](const char* what) { (void)hipGetLastError(); fprintf(stderr, "libspmvhip: tiles: %s failed\n", what); return EXIT_FAILURE; };
    PB_TRY(hipDeviceSynchronize());                  // nothing may still use the product workspace of this device
    prodIdle();
    const auto allocT0 = std::chrono::steady_clock::now();
    const size_t offLcol = (nnz * 8 + 255) / 256 * 256, offLrow = offLcol + (nnz * 2 + 255) / 256 * 256;
    const size_t slabBytes = std::max<size_t>(offLrow + nnz * 2, nnz * sizeof(PbPay));
    if (hipMalloc(&t->slab, slabBytes) || payB.alloc(nnz * sizeof(PbPay)) || tileStart.alloc((nTiles + 2) * 4))
        return fail("format / temporary allocation (12 B per entry of temporaries while the format is built)");
    double* const prodBuf = prodWorkspace(nnz, true);
    if (!prodBuf || hipMalloc(&t->binPos, ((size_t)t->B + 1) * 4) || hipMalloc(&t->waveTile, (size_t)t->B * (t->det ? 1 : P2_WAVES) * 4) ||
        (t->det && hipMalloc(&t->pidx, nnz * 4)))
        return fail("format allocation");
    t->allocMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - allocT0).count();
    t->val  = static_cast<double*>(t->slab);
    t->lcol = reinterpret_cast<uint16_t*>(static_cast<char*>(t->slab) + offLcol);
    t->lrow = reinterpret_cast<uint16_t*>(static_cast<char*>(t->slab) + offLrow);
    t->tempBytes = nnz * sizeof(PbPay) + (nTiles + 2) * 4;
    uint32_t* const keysA = reinterpret_cast<uint32_t*>(prodBuf);
    PbPay* const payA = static_cast<PbPay*>(t->slab);

    if (d->irpBytes == 4)
        hipLaunchKernelGGL((pb_payload_kernel<uint32_t>), grid2d((M + 3) / 4, 256), dim3(256), 0, nullptr, M, static_cast<const uint32_t*>(d->IRP), d->JA, d->AS,
                           keysA, payA);
    else
        hipLaunchKernelGGL((pb_payload_kernel<uint64_t>), grid2d((M + 3) / 4, 256), dim3(256), 0, nullptr, M, static_cast<const uint64_t*>(d->IRP), d->JA, d->AS,
                           keysA, payA);
    PB_TRY(hipGetLastError());

    unsigned bits = 1;
    while ((1u << bits) < t->S) ++bits;
    rocprim::double_buffer<uint32_t> dKeys(keysA, keysA + nnz);
    rocprim::double_buffer<PbPay>    dPay(payA, payB.as<PbPay>());
    size_t tmpBytes = 0;
    PB_TRY(rocprim::radix_sort_pairs(nullptr, tmpBytes, dKeys, dPay, (size_t)nnz, PB_CBITS, PB_CBITS + bits, (hipStream_t) nullptr));
    if (sortTmp.alloc(tmpBytes)) return fail("sort workspace");
    t->tempBytes += tmpBytes;
    PB_TRY(rocprim::radix_sort_pairs(sortTmp.p, tmpBytes, dKeys, dPay, (size_t)nnz, PB_CBITS, PB_CBITS + bits, (hipStream_t) nullptr));
    const uint32_t* const skeys = dKeys.current();
    if (dPay.current() == payA)                      // the sorted payload must not sit where val / lcol / lrow are about to be written
        PB_TRY(hipMemcpyAsync(payB.p, payA, nnz * sizeof(PbPay), hipMemcpyDeviceToDevice, nullptr));
    const PbPay* const spay = payB.as<PbPay>();

    hipLaunchKernelGGL(pb_bounds_kernel, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, spay, skeys, t->B, t->bins, nTiles,
                       tileStart.as<uint32_t>());
    PB_TRY(hipGetLastError());
    // bin-major view: tile lengths (bin-major) -> exclusive scans give every tile's bin-major start and its index
    // in the list of non-empty tiles
    {
        TempBuf lens, flags, bmStart, listIdx, scanTmp, binTile;
        if (lens.alloc(nTiles * 4) || flags.alloc(nTiles * 4) || bmStart.alloc(nTiles * 4) || listIdx.alloc(nTiles * 4) ||
            binTile.alloc(((size_t)t->B + 1) * 4))
            return fail("tile-list workspace");
        t->tempBytes += nTiles * 16;
        hipLaunchKernelGGL(pb_lens_kernel, grid2d((nTiles + 255) / 256, 256), dim3(256), 0, nullptr, t->S, t->B,
                           tileStart.as<uint32_t>(), lens.as<uint32_t>(), flags.as<uint32_t>());
        size_t scanBytes = 0;
        PB_TRY(rocprim::exclusive_scan(nullptr, scanBytes, lens.as<uint32_t>(), bmStart.as<uint32_t>(), 0u, (size_t)nTiles,
                                       rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
        if (scanTmp.alloc(scanBytes)) return fail("scan workspace");
        PB_TRY(rocprim::exclusive_scan(scanTmp.p, scanBytes, lens.as<uint32_t>(), bmStart.as<uint32_t>(), 0u, (size_t)nTiles,
                                       rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
        PB_TRY(rocprim::exclusive_scan(scanTmp.p, scanBytes, flags.as<uint32_t>(), listIdx.as<uint32_t>(), 0u, (size_t)nTiles,
                                       rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
        uint32_t lastIdx = 0, lastFlag = 0;
        PB_TRY(hipMemcpy(&lastIdx, listIdx.as<uint32_t>() + nTiles - 1, 4, hipMemcpyDeviceToHost));
        PB_TRY(hipMemcpy(&lastFlag, flags.as<uint32_t>() + nTiles - 1, 4, hipMemcpyDeviceToHost));
        t->nList = lastIdx + lastFlag;
        if (t->nList >= P2_RUNS_FLAG) return fail("tile list (more than 2^31 non-empty tiles)");
        if (hipMalloc(&t->tl, ((size_t)t->nList + TL_PAD) * sizeof(uint2))) return fail("tile-list allocation");
        hipLaunchKernelGGL(pb_fill_kernel, dim3(1), dim3(256), 0, nullptr, reinterpret_cast<uint32_t*>(t->tl + t->nList), 2 * TL_PAD, 0xFFFFFFFFu);
        hipLaunchKernelGGL(pb_list_kernel, grid2d((nTiles + 255) / 256, 256), dim3(256), 0, nullptr, t->S, t->B, nnz,
                           tileStart.as<uint32_t>(), lens.as<uint32_t>(), bmStart.as<uint32_t>(), listIdx.as<uint32_t>(), t->nList,
                           t->tl, t->binPos, binTile.as<uint32_t>());
        hipLaunchKernelGGL(pb_place_kernel, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, spay, skeys, t->S, t->B, t->bins,
                           tileStart.as<uint32_t>(), bmStart.as<uint32_t>(), t->val, t->lcol, t->lrow, t->pidx);
        if (t->det) {
            // every wavefront walks a whole sub-bin: its cursor starts at the sub-bin's first tile
            PB_TRY(hipMemcpyAsync(t->waveTile, binTile.p, (size_t)t->B * 4, hipMemcpyDeviceToDevice, nullptr));
        } else {
            TempBuf dupCount;
            if (dupCount.alloc((size_t)t->B * P2_WAVES * 4)) return fail("run-count workspace");
            PB_TRY(hipMemsetAsync(dupCount.p, 0, (size_t)t->B * P2_WAVES * 4, nullptr));
            hipLaunchKernelGGL(pb_dupcount_kernel, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, t->B, nnz, t->binPos, t->lrow,
                               dupCount.as<uint32_t>());
            hipLaunchKernelGGL(pb_wavetile_kernel, grid2d(((uint64_t)t->B * P2_WAVES + 255) / 256, 256), dim3(256), 0, nullptr, t->B,
                               t->binPos, binTile.as<uint32_t>(), t->tl, dupCount.as<uint32_t>(), t->waveTile);
            PB_TRY(hipGetLastError());
            PB_TRY(hipDeviceSynchronize());          // dupCount goes out of scope
        }
        PB_TRY(hipGetLastError());
        PB_TRY(hipDeviceSynchronize());
    }
    t->bytes = slabBytes + (t->det ? nnz * 4 : 0) + ((size_t)t->nList + TL_PAD) * 8 + ((size_t)t->B + 1) * 4 + (size_t)t->B * (t->det ? 1 : P2_WAVES) * 4;

    // phase-1 work list from the slice boundaries (tileStart[s*B])
    std::vector<uint32_t> sliceStart(t->S + 1);
    PB_TRY(hipMemcpy2D(sliceStart.data(), 4, tileStart.as<uint32_t>(), (size_t)t->B * 4, 4, t->S + 1, hipMemcpyDeviceToHost));
    std::vector<uint3> work;
    // Work item size.  One workgroup per CU at a time (the x slice fills the LDS), so the items are processed in rounds
    // of PB_CUS and what counts is (a) items of 50-100 k entries (c5: 524 k 8.88 ms, 262 k 8.82, 131 k 8.76, 65-98 k
    // 8.74) and (b) how full the LAST round is: on c3 (610 slices of 328 k entries) 3 pieces per slice = 7.15 rounds ran
    // at 1.081 ms, 2 / 4 / 6 pieces (4.77 / 9.53 / 14.3 rounds) at 1.041-1.046 ms.  The extra x-slice fills of more pieces
    // are nearly free (issued under the running stream, served by L2 / Infinity Cache).  Pieces per slice: within the
    // 50-100 k window, minimise  0.35 * idle share of the last round + a tenth of the relative fill traffic.
    uint32_t chunk = PB_CHUNK;
    if (o.chunk) chunk = o.chunk;
    else {
        uint64_t nonEmpty = 0;
        for (uint32_t s = 0; s < t->S; ++s) nonEmpty += sliceStart[s + 1] > sliceStart[s];
        const double avgLen = nonEmpty ? (double)nnz / (double)nonEmpty : 1.0;
        double bestCost = 1e300;
        const uint32_t pLo = (uint32_t)std::max(1.0, std::ceil(avgLen / 100000.0));
        const uint32_t pHi = (uint32_t)std::max((double)pLo, std::ceil(avgLen / 50000.0));
        for (uint32_t p = pLo; p <= std::min<uint32_t>(pHi, 1024); ++p) {
            const double rounds = (double)nonEmpty * p / PB_CUS;
            const double waste = (std::ceil(rounds) - rounds) / rounds;
            const double cost = 0.35 * waste + 0.1 * p * (PB_C * 8.0) / (avgLen * 18.0);
            if (cost < bestCost) { bestCost = cost; chunk = (uint32_t)std::min(16777216.0, std::ceil(avgLen / p * 1.02) + 64); }
        }
    }
    for (uint32_t s = 0; s < t->S; ++s) {
        // a slice is cut into equal pieces of at most PB_CHUNK entries (a fixed chunk size + remainder left one
        // short, fill-dominated work item per slice); inner boundaries fall on multiples of 64 entries so that
        // only a slice's first work item has a scalar head
        const uint32_t b0 = sliceStart[s], e0 = sliceStart[s + 1], len = e0 - b0;
        if (!len) continue;
        const uint32_t pieces = (len + chunk - 1) / chunk;
        const uint32_t piece = (len + pieces - 1) / pieces;
        uint32_t b = b0;
        for (uint32_t k = 1; k <= pieces && b < e0; ++k) {
            const uint32_t e = (k == pieces) ? e0 : std::min<uint32_t>(e0, (b0 + k * piece + 63u) & ~63u);
            if (e > b) work.push_back(make_uint3(s, b, e));
            b = e;
        }
    }
    // Order of the work items: workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and the pieces of
    // one slice all stage the same 128 KiB of x.  Listed one after the other they land on different XCDs and every
    // piece fetches the slice from the fabric (c5, 4 pieces: 2.6 GB of fills per SpMV); listed 8 apart -- groups of 8
    // slices, piece k of each, then piece k+1 of each -- they land on ONE XCD at almost the same time and all but the
    // first fill hit its L2.
    {
        std::vector<uint3> ordered;
        ordered.reserve(work.size());
        size_t i = 0;
        while (i < work.size()) {
            // the next (up to) 8 slices and their pieces
            size_t first[9];
            int ns = 0;
            size_t j = i;
            while (j < work.size() && ns < 8) {
                first[ns++] = j;
                const uint32_t sl = work[j].x;
                while (j < work.size() && work[j].x == sl) ++j;
            }
            first[ns] = j;
            size_t maxPieces = 0;
            for (int k = 0; k < ns; ++k) maxPieces = std::max(maxPieces, first[k + 1] - first[k]);
            for (size_t pc = 0; pc < maxPieces; ++pc)
                for (int k = 0; k < ns; ++k)
                    if (first[k] + pc < first[k + 1]) ordered.push_back(work[first[k] + pc]);
            i = j;
        }
        work.swap(ordered);
    }
    t->nWork = (uint32_t)work.size();
    PB_TRY(hipMalloc(&t->work, std::max<size_t>(work.size(), 1) * sizeof(uint3)));
    PB_TRY(hipMemcpy(t->work, work.data(), work.size() * sizeof(uint3), hipMemcpyHostToDevice));
    PB_TRY(hipDeviceSynchronize());

    // (set at every build: the attribute belongs to the current device, and a process may drive several)
    PB_TRY(hipFuncSetAttribute((const void*)pb_expand_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_C * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_expand_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_C * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_expand_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_C * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_expand_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_C * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_R_MAX * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_R_MAX * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_R_MAX * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_det_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_R_MAX * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_det_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_R_MAX * 8));
    PB_TRY(hipFuncSetAttribute((const void*)pb_reduce_det_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_R_MAX * 8));
    // products that fit the Infinity Cache (with room for x and the streams) are stored with the default policy so
    // that phase 2 finds them there; larger streams bypass it (microbench_mall.hip: resident 7.0 vs 5.8 TB/s with
    // non-temporal stores; non-resident 4.9 vs 5.1-5.4)
    t->ntStore = o.ntStore >= 0 ? o.ntStore != 0 : nnz * 8 > PB_RESIDENT_BYTES;
    t->chunk = chunk;
    PB_TRY(hipEventRecord(ev1, nullptr));
    PB_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    t->buildMs = ms;
    d->tiles = t;
    t = nullptr;                                    // the guard lets go
    return EXIT_SUCCESS;
}

size_t tilesBytes(const DevMat* d) { return (d->tiles ? d->tiles->bytes : 0) + (d->tilesAlt ? d->tilesAlt->bytes : 0); }

void tilesInfo(const DevMat* d, spmvTilesInfo* out) {
    memset(out, 0, sizeof *out);
    const TileFormat* t = d->tiles;
    if (!t) return;
    out->nBins = t->apiBins(); out->rowsPerBin = t->apiRows(); out->nSlices = t->S; out->taper = t->bins.n1 != 0; out->ntStore = t->ntStore;
    out->chunk = t->chunk; out->buildMs = t->buildMs; out->bytes = t->bytes; out->allocMs = t->allocMs;
    out->tempBytes = t->tempBytes; out->deterministic = t->det ? 1 : 0;
}

uint32_t tilesPhase2Threads(const DevMat* d) { return d->tiles && d->tiles->det ? (uint32_t)PD_THREADS : (uint32_t)PB_THREADS; }

void tilesShape(const DevMat* d, uint32_t* bins, uint32_t* rowsPerBin) {
    *bins = d->tiles ? d->tiles->apiBins() : 0;
    *rowsPerBin = d->tiles ? d->tiles->apiRows() : 0;
}

// phase 1 on `stream`: products of the whole matrix into the format's workspace
int enqueueTilesExpand(DevMat* d, const double* x, hipStream_t stream) {
    TileFormat* t = d->tiles;
    if (!t) return EXIT_FAILURE;
    double* prod = prodWorkspace(t->nnz, true);      // grows (after a device synchronise) only if it was released meanwhile
    if (!prod || prodHandover(stream)) return EXIT_FAILURE;
    if (t->nWork) {
#define PB_EXPAND(NT, UNIT) hipLaunchKernelGGL((pb_expand_kernel<NT, UNIT>), grid2d(t->nWork, PB_THREADS), dim3(PB_THREADS), PB_C * 8, stream, t->work, \
                                              t->val, t->lcol, x, d->N, prod, d->unitValue)
        if (d->unit) { if (t->ntStore) PB_EXPAND(true, true); else PB_EXPAND(false, true); }      // (the format keeps its value array: the slab is
        else         { if (t->ntStore) PB_EXPAND(true, false); else PB_EXPAND(false, false); }    //  the build's sort buffer; it is just not read)
#undef PB_EXPAND
    }
    return hipGetLastError() == hipSuccess ? EXIT_SUCCESS : EXIT_FAILURE;
}

// phase 2 on `stream` for the bins [binBegin, binEnd): rows [binBegin * R, min(binEnd * R, M)) of y, stored to y and
// to the nExtra further destinations (same row indexing)
int enqueueTilesReduce(DevMat* d, uint32_t binBegin, uint32_t binEnd, double* y, int nExtra, double* const* extra, hipStream_t stream) {
    TileFormat* t = d->tiles;
    if (!t || binBegin > binEnd || binEnd > t->apiBins() || nExtra < 0 || nExtra > SPMV_MAX_PEERS) return EXIT_FAILURE;
    if (binBegin == binEnd) return EXIT_SUCCESS;
    const dim3 grid = grid2d((uint64_t)((binEnd - binBegin + 7) / 8) * 8, PB_THREADS);
    const double* prod = prodWorkspace(t->nnz, false);
    if (!prod || prodHandover(stream)) return EXIT_FAILURE;
    TileDst dst{};
    dst.n = (uint32_t)nExtra;
    for (int i = 0; i < nExtra; ++i) dst.p[i] = extra[i];
    const TileSignal none{nullptr, 0};
    if (t->det) {
        const size_t lds = (size_t)8 * t->apiRows();
        if (nExtra)
            hipLaunchKernelGGL(pb_reduce_det_kernel<1>, grid, dim3(PD_THREADS), lds, stream, t->bins, t->B, binBegin, binEnd, d->M, t->binPos,
                               t->pidx, prod, t->lrow, y, dst, none);
        else
            hipLaunchKernelGGL(pb_reduce_det_kernel<0>, grid, dim3(PD_THREADS), lds, stream, t->bins, t->B, binBegin, binEnd, d->M, t->binPos,
                               t->pidx, prod, t->lrow, y, dst, none);
        const bool launched = hipGetLastError() == hipSuccess;
        prodMark(stream);
        return launched ? EXIT_SUCCESS : EXIT_FAILURE;
    }
    if (nExtra)
        hipLaunchKernelGGL(pb_reduce_kernel<1>, grid, dim3(PB_THREADS), (size_t)8 * t->R, stream,
                           t->bins, binBegin, binEnd, d->M, t->binPos, t->waveTile, t->tl, prod, t->lrow, y, dst, none);
    else
        hipLaunchKernelGGL(pb_reduce_kernel<0>, grid, dim3(PB_THREADS), (size_t)8 * t->R, stream,
                           t->bins, binBegin, binEnd, d->M, t->binPos, t->waveTile, t->tl, prod, t->lrow, y, dst, none);
    const bool launched = hipGetLastError() == hipSuccess;
    prodMark(stream);
    return launched ? EXIT_SUCCESS : EXIT_FAILURE;
}

// phase 2 over all bins on `stream` with the push kernel beside it on `side`: `side` first waits for what is enqueued on
// `stream` so far (phase 1); evJoin marks the end of the push kernel and `stream` does NOT wait for it here, so the
// next row group's phase 1 can run while this group's rows are still travelling.  y and the destinations as above.
int enqueueTilesReducePush(DevMat* d, double* y, int nExtra, double* const* extra, hipStream_t stream, hipStream_t side,
                           hipEvent_t evFork, hipEvent_t evJoin) {
    TileFormat* t = d->tiles;
    if (!t || nExtra < 1 || nExtra > SPMV_MAX_PEERS) return EXIT_FAILURE;
    const double* prod = prodWorkspace(t->nnz, true);
    if (!prod || prodHandover(stream)) return EXIT_FAILURE;
    const uint32_t nb = t->apiBins();
    if (!t->ready) {
        if (hipMalloc(&t->ready, ((size_t)nb + 1) * 4) != hipSuccess) return EXIT_FAILURE;
        if (hipMemset(t->ready, 0, ((size_t)nb + 1) * 4) != hipSuccess) return EXIT_FAILURE;   // [nb] doubles as the failure word
        t->pushFail = t->ready + nb;
    }
    const uint32_t epoch = ++t->epoch ? t->epoch : ++t->epoch;             // never 0 (the initial flag value)
    TileDst dst{};
    dst.n = (uint32_t)nExtra;
    for (int i = 0; i < nExtra; ++i) dst.p[i] = extra[i];
    const TileSignal sig{t->ready, epoch};
    if (hipEventRecord(evFork, stream) != hipSuccess || hipStreamWaitEvent(side, evFork, 0) != hipSuccess) return EXIT_FAILURE;
    hipLaunchKernelGGL(pb_push_kernel, dim3(std::min<uint32_t>(PUSH_WGS, nb)), dim3(PUSH_THREADS), 0, side, t->apiMap(), nb, d->M, t->ready,
                       epoch, y, dst, t->pushFail);
    if (t->det)
        hipLaunchKernelGGL(pb_reduce_det_kernel<2>, grid2d((uint64_t)((nb + 7) / 8) * 8, PB_THREADS), dim3(PD_THREADS), (size_t)8 * t->apiRows(), stream,
                           t->bins, t->B, 0u, nb, d->M, t->binPos, t->pidx, prod, t->lrow, y, TileDst{}, sig);
    else
        hipLaunchKernelGGL(pb_reduce_kernel<2>, grid2d((uint64_t)((t->B + 7) / 8) * 8, PB_THREADS), dim3(PB_THREADS), (size_t)8 * t->R, stream,
                           t->bins, 0u, t->B, d->M, t->binPos, t->waveTile, t->tl, prod, t->lrow, y, TileDst{}, sig);
    const bool launched = hipGetLastError() == hipSuccess;
    prodMark(stream);
    if (hipEventRecord(evJoin, side) != hipSuccess) return EXIT_FAILURE;          // the caller joins (tilesPushJoin) when it needs y delivered
    return launched ? EXIT_SUCCESS : EXIT_FAILURE;
}

// 1 when a push kernel of this matrix ever gave up waiting for a bin (synchronises the device)
int tilesPushFailed(DevMat* d) {
    TileFormat* t = d->tiles;
    if (!t || !t->pushFail) return 0;
    uint32_t f = 0;
    if (hipMemcpy(&f, t->pushFail, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return f != 0;
}

// enqueue both phases on `stream`
int enqueueTiles(DevMat* d, const double* x, double* y, hipStream_t stream) {
    TileFormat* t = d->tiles;
    if (!t) return EXIT_FAILURE;
    if (enqueueTilesExpand(d, x, stream)) return EXIT_FAILURE;
    return enqueueTilesReduce(d, 0, t->apiBins(), y, 0, nullptr, stream);
}

}  // namespace spmvhip
