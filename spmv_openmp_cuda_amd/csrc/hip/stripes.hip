// stripes.hip -- one-pass SpMV with y bins in LDS and x served by the XCD's L2 ("bin-wise CSC").
//
// Why: on a matrix whose columns have no locality a row-major kernel pays one 128-B line per 8-byte x element
// from the fabric (x lives beyond L2: <= 56 G gathers/s), and the two-phase kernel
// (tiles.hip) buys streaming accesses with 16 extra bytes per entry (products out and back).  This kernel
// keeps ONE pass of 12 B/nnz and moves the gather into L2 instead:
//   * rows are cut into bins of <= 20 000 consecutive rows with (nearly) equal entry counts; a workgroup owns a
//     bin and keeps its y in LDS (ds_add_f64); one workgroup per CU, persistent over the bins w, w + 256, ...;
//   * inside a bin the entries are stored in COLUMN order.  Every workgroup therefore sweeps x from column 0
//     to N-1 at the pace of its entry stream; the 32 workgroups resident on one XCD run the same sweep at about
//     the same pace (equal entry counts), so most lines of x are found in that XCD's 4 MiB L2, and the column
//     order lets neighbouring lanes share lines (a bin of R rows holds R.16.nnz/(M.N) entries per line of x:
//     0.64 on c3, 2 on c2, ~8 on the +-16 Ki band);
//   * a step = 128 consecutive entries of a bin, dealt to the lanes as (e, 64 + e): fp64 values and one 32-bit
//     word per entry, {17-bit column offset from the step's first column, 15-bit local row}; the step's first
//     column is a scalar load.  12.03 B/nnz.  A matrix with a step spanning >= 2^17 columns uses the WIDE
//     encoding (32-bit column + 16-bit local row, 14 B/nnz) -- same kernel, other template argument.
// HBM sees the 12 B/nnz stream, x once per XCD and round of bins and y once.
//
// Measured (profiles/r02_summary.md): c2 0.110 ms = 46 % of the 8 TB/s roofline (two-phase 0.149, SELL 0.237);
// c3 0.87 ms = 37 % (two-phase 1.03); band +-16 Ki 0.41 ms = 79 % (5.9 TB/s of algorithmic bytes).
// What bounds it when the columns have no locality is the CU's own L1: a 64-lane gather costs it ~28 clocks plus ~2.2
// clocks per distinct line it takes from L2 (the 64 B/clock fill path), each gather instruction touches ~47 lines on
// c3, and that adds up to the kernel time -- the kernel without its LDS adds is no faster, without its gathers it is
// the 12 B/nnz stream (0.41 ms), a bin takes the same ~415 us whether 64 or 256 CUs are working, and neither the
// cache-policy bits nor the address form of the gather change anything (DESIGN.md sections 4 and 8,
// profiles/r02_stripes_*.log).  Not HBM, and not L2 misses: see the throttle note below.  For very wide matrices the
// fabric takes over: every XCD re-reads the lines of x its 32 bins touch in every round of bins (c5: 0.08 entries
// per line and bin -> 50 B of line fills per entry), so the format pays off while x (N * 8 B) is small against the
// entry stream -- bench.py and hipSpMVAutoCSR try it while x fits the Infinity Cache; wider matrices are the
// two-phase kernel's.
//
// Relation to the reference: the computation of cudaSpMVWarpPerRowCSR (src/SpMV_CUDA.cu:52-73) -- lanes
// multiply entries of coalesced AS/JA spans with gathered x and the partial sums are reduced on chip -- with
// the reduction moved from a shuffle tree per row to LDS accumulators per bin of rows, which is what allows
// the column order.  Sums are added in arrival order: equal to the serial oracle to rounding, not bitwise.
//
// The format is built ON THE DEVICE from the device CSR (rocPRIM radix sort of (bin, column) keys; c3: ~60 ms).
//
// DETERMINISTIC forms (spmvStripesOpts.deterministic): a row's products are added in ascending column order -- the
// order of the serial oracle -- whatever the scheduler does; y is bit-identical to sgemvSerial.  Two ways to get there,
// neither always the faster, so the serial-order selection of hipSpMVRowsCSR measures both:
//   1 "owner wavefronts": every row of a bin is OWNED by one of the workgroup's wavefronts (local row mod 4); each
//     wavefront has its own column-ordered sub-stream of the bin and walks it in program order.  No ticket counter, no
//     hand-over.  The four wavefronts still sweep x together (each sub-stream spans all columns), but a gather instruction
//     covers 64 neighbours of a QUARTER of the bin's entries, so it touches more lines (c3: ~60 instead of ~47 per 64
//     entries: 1.69 ms against 0.87); on matrices with column locality it costs almost nothing (+-16 Ki band 0.44 against
//     0.42 ms, 3-D stencil 0.188 against 0.185).  Its format is a layout of its own (sub-stream-major).
//   2 "ordered tickets": the SAME format and the same shared stream as the arrival-order kernel (tickets, full-density
//     gathers), but the batches ADD in ticket order: a wavefront spins on an LDS word until the batch before its own has
//     been added and hands over after its adds (LDS operations of a wavefront execute in issue order).  Streams and
//     gathers stay in flight while it waits.  c3 1.05 ms, c2 0.135 against 0.154 -- but where lanes of one instruction
//     meet in a row (banded matrices: long add phases) the serial chain shows: +-16 Ki band 0.64 ms.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <vector>

#include "spmvHip.h"
#include "device_mat.hpp"

namespace spmvhip {

constexpr uint32_t SB_R_MAX   = 20000;              // rows per bin: <= 156.25 KiB of y in LDS
constexpr int      SB_THREADS = 256;                // 4 wavefronts: fewer, deeper wavefronts keep the gathers in flight close together in x
constexpr uint32_t SB_WAVES   = SB_THREADS / 64;
constexpr uint32_t SB_STEP    = 128;                // entries per wavefront step (two per lane)
constexpr uint32_t SB_ROWBITS = 15;
constexpr uint32_t SB_NONE    = (1u << SB_ROWBITS) - 1;     // local row of a padding entry (SB_R_MAX < SB_NONE)
constexpr uint32_t SB_DCOL_LIMIT = 1u << (32 - SB_ROWBITS);
constexpr int      SB_DEPTH   = 6;                  // steps per register batch
constexpr uint32_t SB_SPREAD  = 6;                  // 1/1024ths of a bin over which the sweeps of one XCD's workgroups start
constexpr uint64_t SB_MIN_BIN_NNZ = 16384;          // do not cut a small matrix into bins shorter than this

struct StripeFormat {
    uint32_t  B = 0, R = 0;                         // bins, rows of the highest bin
    uint32_t  subs = 1;                             // column-ordered sub-streams per bin: 1, or SB_WAVES in the deterministic form
    uint64_t  nnz = 0, nSteps = 0;
    bool      wide = false, det = false;            // det: the sub-stream-major layout of the owner-wavefront form
    double*   val = nullptr;                        // [nSteps * 128] sub-stream-major, column order inside a sub-stream, each padded to whole steps
    bool      unit = false;                         // every value of the matrix is `unitValue` (DevMat::unit): no `val` array, the kernel keeps it in a register
    double    unitValue = 0.0;
    uint32_t* cr = nullptr;                         // narrow: (column - stepBase) << 15 | local row;  wide: column
    uint16_t* lrowW = nullptr;                      // wide only: local row
    uint32_t* stepBase = nullptr;                   // narrow only: [nSteps] first column of the step
    uint32_t* binRow = nullptr;                     // [B+1] first row of each bin
    uint32_t* subStep = nullptr;                    // [B*subs+1] first step of each sub-stream
    uint32_t  grid = 1, spread = SB_SPREAD;         // persistent workgroups (one per CU of the device the format was built on), start spread
    size_t    bytes = 0;
    double    buildMs = 0;
    spmvStripesOpts opts{0, 0, -1, -1, 0};          // what the format was built with (0 / -1 = automatic)
};

namespace {

__device__ __forceinline__ uint64_t lin_block() { return (uint64_t)blockIdx.y * gridDim.x + blockIdx.x; }

typedef double   dbl2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

// ---- build kernels ---------------------------------------------------------------------------------------
// key = (bin * subs + owner) << colBits | column, payload = CSR position; one wavefront per row.  owner = the wavefront
// that adds this row in the deterministic form (local row mod subs), 0 otherwise.
template <typename I>
__global__ __launch_bounds__(256) void sb_keys_kernel(uint64_t M, const I* __restrict__ IRP, const uint32_t* __restrict__ JA,
                                                      const uint32_t* __restrict__ binRow, uint32_t B, uint32_t subs, unsigned colBits,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                      uint32_t* __restrict__ rowOf) {
    const uint64_t r = lin_block() * 4 + threadIdx.x / 64;
    if (r >= M) return;
    uint32_t lo = 0, hi = B;                         // bin with binRow[bin] <= r < binRow[bin + 1]
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (binRow[mid] <= r) lo = mid; else hi = mid;
    }
    const uint64_t group = (uint64_t)lo * subs + ((uint32_t)r - binRow[lo]) % subs;
    const uint64_t b = IRP[r], e = IRP[r + 1];
    for (uint64_t j = b + threadIdx.x % 64; j < e; j += 64) {
        keys[j] = group << colBits | JA[j];
        idx[j] = (uint32_t)j;
        rowOf[j] = (uint32_t)r;
    }
}

// where each group (sub-stream) starts in the sorted order: the group id is non-decreasing along it, so the first entry of
// a group also fills the start of every empty group before it
__global__ __launch_bounds__(256) void sb_bounds_kernel(uint64_t nnz, unsigned colBits, const uint64_t* __restrict__ skeys,
                                                        uint64_t nGroups, uint64_t* __restrict__ start) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint64_t g = skeys[p] >> colBits;
    const uint64_t gPrev = p ? skeys[p - 1] >> colBits : ~0ull;
    if (g != gPrev)
        for (uint64_t u = gPrev + 1; u <= g; ++u) start[u] = p;          // gPrev + 1 wraps to 0 for p == 0
    if (p == nnz - 1)
        for (uint64_t u = g + 1; u <= nGroups; ++u) start[u] = nnz;
}

// sorted position p -> padded position q of its sub-stream; values, encoded columns / rows, step bases
template <bool WIDE>
__global__ __launch_bounds__(256) void sb_scatter_kernel(
    uint64_t nnz, unsigned colBits, uint32_t subs, const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ perm,
    const uint32_t* __restrict__ rowOf, const double* __restrict__ AS, const uint32_t* __restrict__ binRow,
    const uint64_t* __restrict__ start, const uint32_t* __restrict__ subStep,
    double* __restrict__ val, uint32_t* __restrict__ cr, uint16_t* __restrict__ lrowW, uint32_t* __restrict__ stepBase,
    uint32_t* __restrict__ overflow) {
    const uint64_t p = lin_block() * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint64_t key = skeys[p];
    const uint64_t group = key >> colBits;
    const uint32_t bin = (uint32_t)(group / subs);
    const uint32_t col = (uint32_t)(key & ((1ull << colBits) - 1));
    const uint64_t pos = p - start[group];
    // inside a step the 128 column-ordered entries are dealt to the lanes as (e, 64 + e): lane l reads the pair at
    // positions 2l, 2l + 1 with one 16-byte load, and each of the step's two gather instructions covers 64 NEIGHBOURING
    // entries (lanes share lines inside an instruction; the two instructions touch different lines -- with (2l, 2l + 1)
    // pairs the second gather hit lines still pending from the first and stalled the L1: TCP_PENDING_STALL_CYCLES)
    const uint64_t e = pos % SB_STEP;
    const uint64_t q = (uint64_t)subStep[group] * SB_STEP + (pos - e) + (e < SB_STEP / 2 ? 2 * e : 2 * (e - SB_STEP / 2) + 1);
    const uint32_t j = perm[p];
    const uint32_t lrow = rowOf[j] - binRow[bin];
    if (val) val[q] = AS[j];                      // (no value array for a matrix whose values are all the same)
    if (WIDE) {
        cr[q] = col;
        lrowW[q] = (uint16_t)lrow;
    } else {
        const uint32_t base = (uint32_t)(skeys[p - e] & ((1ull << colBits) - 1));      // first entry of the step: same sub-stream
        const uint32_t d = col - base;
        if (d >= SB_DCOL_LIMIT) atomicOr(overflow, 1u);
        cr[q] = d << SB_ROWBITS | lrow;
        if (e == 0) stepBase[q / SB_STEP] = base;
    }
}

__global__ __launch_bounds__(256) void sb_fill32_kernel(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void sb_fill16_kernel(uint16_t* __restrict__ p, uint64_t n, uint16_t v) {
    const uint64_t i = lin_block() * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- the SpMV kernel -------------------------------------------------------------------------------------
// A batch = SB_DEPTH consecutive steps of a sub-stream.  In the default form the bin is ONE sub-stream and its batches are
// handed to the wavefronts of the bin's workgroup through a ticket counter in LDS, NOT by a fixed wavefront -> batch map:
// the scheduler favours the oldest wavefront of a workgroup, and with a fixed map wavefront 0 finished its share after a
// quarter of the bin's time while the youngest needed all of it -- several sweeps of x at different places, and the L2 kept
// none of them (c3: 52 % hits, 10 GB fetched for 2.6 GB; profiles/r02_stripes.md).  With tickets the workgroup has ONE
// frontier: a faster wavefront simply takes more batches.  In the deterministic form every wavefront walks its OWN
// sub-stream (the rows it owns) batch by batch in program order.
// Three stages per batch -- stream loads (values, column/row words; non-temporal), x gathers, LDS adds -- are kept
// apart by TWO batches of stream loads: vmcnt counts in issue order, so a gather may only be waited for while
// younger stream loads are outstanding if those were issued AFTER it.  Issue order of iteration k:
//     gather(k+1) [needs stream(k+1), issued two iterations ago], stream(k+3), add(k) [needs gather(k)]
// Four stream batches and two gather batches rotate through registers (the loop body is written out four times
// so that every batch is a named register set).  Steps past the sub-stream's end are clamped to its last step (a valid,
// cached address) and skipped when adding: no branch around a load, the compiler's vmcnt stays exact.
// (Measured and settled in round 2, logs under profiles/r02_stripes_*: `nt` stream loads beat plain ones by 8 %; plain
// gathers -- `nt` ones no longer allocate in L2, 2.3 ms; buffer / scalar-base address forms of the gather equal; the kernel
// without its LDS adds is no faster, without its gathers it is the 12 B/nnz stream.)
struct SbStream { dbl2 v[SB_DEPTH]; u32x2 c[SB_DEPTH]; u16x2 r[SB_DEPTH]; uint32_t base[SB_DEPTH]; uint32_t first, tk; };
struct SbGather { double x0[SB_DEPTH], x1[SB_DEPTH]; };

// stream loads of the batch with ticket `t` (wavefront-uniform).  A sub-stream has nb batches of SB_DEPTH steps; ticket t
// stands for batch (t + off) mod nb -- the workgroup starts its column sweep `off` batches into the bin and wraps around
// (default form only: the order in which a bin's entries are added is free there) -- and tickets >= nb for nothing
// (first = s1: loads clamped, adds skipped).
struct SbBin { uint32_t s0, s1, nb, off; };
template <bool WIDE, bool UNIT>
__device__ __forceinline__ void sb_stream(SbStream& s, uint32_t t, const SbBin bn, uint32_t lane, double unitValue,
                                          const double* __restrict__ val, const uint32_t* __restrict__ cr,
                                          const uint16_t* __restrict__ lrowW, const uint32_t* __restrict__ stepBase) {
    const uint32_t s1 = bn.s1;
    const uint32_t rb = t + bn.off;
    s.tk = t;
    s.first = t < bn.nb ? bn.s0 + (rb >= bn.nb ? rb - bn.nb : rb) * SB_DEPTH : s1;
#pragma unroll
    for (int u = 0; u < SB_DEPTH; ++u) {
        const uint32_t sc = min(s.first + u, s1 - 1u);
        const uint64_t q = (uint64_t)sc * SB_STEP + 2u * lane;
        if (UNIT) s.v[u] = dbl2{unitValue, unitValue};
        else      s.v[u] = __builtin_nontemporal_load((const dbl2*)(val + q));
        s.c[u] = __builtin_nontemporal_load((const u32x2*)(cr + q));
        if (WIDE) { s.r[u] = __builtin_nontemporal_load((const u16x2*)(lrowW + q)); s.base[u] = 0; }
        else      { s.base[u] = stepBase[sc]; s.r[u] = u16x2{0, 0}; }
    }
}

template <bool WIDE>
__device__ __forceinline__ void sb_gather(SbGather& g, const SbStream& s, const double* __restrict__ x) {
#pragma unroll
    for (int u = 0; u < SB_DEPTH; ++u) {
        const uint32_t c0 = WIDE ? s.c[u].x : s.base[u] + (s.c[u].x >> SB_ROWBITS);
        const uint32_t c1 = WIDE ? s.c[u].y : s.base[u] + (s.c[u].y >> SB_ROWBITS);
        g.x0[u] = x[c0];
        g.x1[u] = x[c1];
    }
}

template <bool WIDE>
__device__ __forceinline__ void sb_add(double* yb, const SbStream& s, const SbGather& g, uint32_t s1) {
#pragma unroll
    for (int u = 0; u < SB_DEPTH; ++u) {
        const uint32_t r0 = WIDE ? s.r[u].x : s.c[u].x & SB_NONE;
        const uint32_t r1 = WIDE ? s.r[u].y : s.c[u].y & SB_NONE;
        if (s.first + u < s1) {                      // wavefront-uniform
            if (r0 != SB_NONE) atomicAdd(&yb[r0], s.v[u].x * g.x0[u]);      // entries e = 0..63 of the step, then
            if (r1 != SB_NONE) atomicAdd(&yb[r1], s.v[u].y * g.x1[u]);      // e = 64..127: ascending columns
        }
    }
}

// next ticket of the workgroup (one LDS atomic by lane 0, broadcast)
__device__ __forceinline__ uint32_t sb_ticket(uint32_t* ctr, uint32_t lane) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(ctr, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

// Tried on top of this and removed again (commit 6ecc064, logs profiles/r02_stripes_throttle_*): keeping the 32
// workgroups of an XCD inside a window of each other by throttling the leaders (per-XCD progress line in global
// memory, plain stores / sc1 loads, refreshed asynchronously every 8th ticket) and aligning the rounds of bins.
// It does what it says -- workgroups within +-5 us of each other over a 480 us bin, rounds starting together --
// and buys nothing: c3 0.98-1.03 ms throttled against 0.93 free-running, band +-16 Ki 0.48 against 0.41.  A
// workgroup in a tight pack gathers no faster, because the gather is bounded by the lines a CU's L1 can have in
// flight to L2 (~0.27-0.38 lines per clock and CU here and in scripts/microbench_gather.hip), not by L2 misses.

template <bool WIDE, int DET, bool UNIT>
__global__ __launch_bounds__(SB_THREADS) void sb_spmv_kernel(
    uint32_t B, const uint32_t* __restrict__ binRow, const uint32_t* __restrict__ subStep, double unitValue,
    const double* __restrict__ val, const uint32_t* __restrict__ cr, const uint16_t* __restrict__ lrowW,
    const uint32_t* __restrict__ stepBase, const double* __restrict__ x, double* __restrict__ y, uint32_t ldsRows, uint32_t spread) {
    extern __shared__ double yb[];                  // ldsRows doubles (rows of the highest bin), then the ticket counter
    uint32_t* ctr = reinterpret_cast<uint32_t*>(yb + ldsRows);
    volatile uint32_t* turn = ctr + 1;               // DET == 2: the ticket whose batch may add to the bin now
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)), lane = threadIdx.x % 64;
    // persistent: one workgroup per CU walks the bins w, w + G, ... (its bin of y fills the LDS, so no second
    // workgroup could share the CU anyway, and a bin starts the moment the previous one is stored)
    for (uint64_t bin = lin_block(); bin < B; bin += (uint64_t)gridDim.x * gridDim.y) {
        const uint32_t row0 = binRow[bin], R = binRow[bin + 1] - row0;
        SbStream a, b, c, d;
        SbGather g0, g1;
        SbBin bn;
        uint32_t mine = 3;                           // DET: this wavefront's next batch
        if (DET == 1) {                              // this wavefront's own sub-stream, from its first batch on
            bn.s0 = subStep[bin * SB_WAVES + wave]; bn.s1 = subStep[bin * SB_WAVES + wave + 1];
            bn.off = 0;
        } else {
            bn.s0 = subStep[bin]; bn.s1 = subStep[bin + 1];
        }
        bn.nb = (bn.s1 - bn.s0 + SB_DEPTH - 1) / SB_DEPTH;
        // where this workgroup starts its sweep: the (up to) 32 workgroups of an XCD -- workgroups are dealt round-robin
        // to the 8 XCDs -- are spread over `spread`/1024 of the sweep
        bn.off = DET ? 0u : (uint32_t)(((uint64_t)((lin_block() / 8) % 32) * bn.nb * spread) >> 15);
        const uint32_t s1 = bn.s1;
        const bool any = s1 > bn.s0;                 // uniform per workgroup (default form) / per wavefront (deterministic form)
        if (any) {                                   // first three batches; moving before the bin is zeroed
            sb_stream<WIDE, UNIT>(a, DET == 1 ? 0u : wave, bn, lane, unitValue, val, cr, lrowW, stepBase);
            sb_stream<WIDE, UNIT>(b, DET == 1 ? 1u : wave + SB_WAVES, bn, lane, unitValue, val, cr, lrowW, stepBase);
            sb_stream<WIDE, UNIT>(c, DET == 1 ? 2u : wave + 2 * SB_WAVES, bn, lane, unitValue, val, cr, lrowW, stepBase);
        }
        for (uint32_t k = threadIdx.x; k < R; k += SB_THREADS) yb[k] = 0.0;
        if (DET != 1 && threadIdx.x == 0) { *ctr = 3 * SB_WAVES; *turn = 0; }
        __syncthreads();
        if (any) {
            sb_gather<WIDE>(g0, a, x);
            // one stage: CUR is added, NXT gathered, FAR (the set CUR's predecessor freed) streamed with a fresh ticket
#define SB_STAGE(CUR, NXT, FAR, GC, GN)                                                                          \
            if (CUR.first >= s1) break;                                                                          \
            sb_gather<WIDE>(GN, NXT, x);                                                                         \
            sb_stream<WIDE, UNIT>(FAR, DET == 1 ? mine++ : sb_ticket(ctr, lane), bn, lane, unitValue, val, cr, lrowW, stepBase);  \
            if (DET == 2) {       /* batches add in ticket order: wait for the batch before this one */        \
                while ((uint32_t)__builtin_amdgcn_readfirstlane((int)*turn) != CUR.tk) __builtin_amdgcn_s_sleep(1); \
            }                                                                                                    \
            sb_add<WIDE>(yb, CUR, GC, s1);                                                                       \
            if (DET == 2) {       /* LDS operations of one wavefront execute in issue order: the hand-over follows the adds */ \
                asm volatile("" ::: "memory");                                                                   \
                if (lane == 0) *turn = CUR.tk + 1;                                                               \
                asm volatile("" ::: "memory");                                                                   \
            }
            for (;;) {
                SB_STAGE(a, b, d, g0, g1)
                SB_STAGE(b, c, a, g1, g0)
                SB_STAGE(c, d, b, g0, g1)
                SB_STAGE(d, a, c, g1, g0)
            }
#undef SB_STAGE
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < R; k += SB_THREADS) y[(uint64_t)row0 + k] = yb[k];
        __syncthreads();                             // the next bin zeroes yb
    }
}

#define SB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "libspmvhip: stripes: %s: %s\n", #expr, hipGetErrorString(e_)); return EXIT_FAILURE; } } while (0)

struct TempBuf {
    void* p = nullptr;
    ~TempBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 1)); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

// Bins: consecutive rows, (nearly) equal entry counts, at most rMax rows.  Equal counts are what keeps the
// workgroups of one XCD at the same place of their column sweeps; the bin count is a multiple of the workgroup count
// (one workgroup per CU at a time: bins run in rounds) once the matrix is large enough for a full round.
bool planBins(const std::vector<uint64_t>& irp, uint64_t M, uint32_t rMax, uint32_t cus, std::vector<uint32_t>& binRow) {
    const uint64_t nnz = irp[M];
    uint64_t B = std::max<uint64_t>((M + rMax - 1) / rMax, std::min<uint64_t>(cus, (nnz + SB_MIN_BIN_NNZ - 1) / SB_MIN_BIN_NNZ));
    B = std::max<uint64_t>(1, std::min<uint64_t>(B, M));
    if (B > cus) B = (B + cus - 1) / cus * cus;
    if (B > M) B = M;
    if (B >= (1ull << 31) / SB_WAVES) return false;
    binRow.assign(B + 1, 0);
    uint64_t cur = 0;
    for (uint64_t b = 0; b < B; ++b) {
        const uint64_t left = B - b;                 // bins still to cut, this one included
        uint64_t next;
        if (left == 1) next = M;
        else {
            const uint64_t target = irp[cur] + (irp[M] - irp[cur] + left - 1) / left;
            next = (uint64_t)(std::lower_bound(irp.begin() + cur, irp.begin() + M + 1, target) - irp.begin());
            if (next > cur + 1 && target - irp[next - 1] < irp[next] - target) --next;     // the closer of the two cuts
            next = std::max(next, cur + 1);                                              // at least one row
            next = std::min(next, cur + rMax);                                           // y fits the LDS
            next = std::min(next, M - (left - 1));                                       // one row for every later bin
            const uint64_t cap = (left - 1) * (uint64_t)rMax;                            // ... and they can hold the rest
            if (M - next > cap) next = M - cap;
        }
        cur = next;
        binRow[b + 1] = (uint32_t)cur;
    }
    return cur == M;
}

int fillFormat(StripeFormat* f, bool wide, uint64_t nnz, unsigned colBits, const uint64_t* skeys, const uint32_t* perm,
               const uint32_t* rowOf, const double* AS, const uint64_t* dStart, uint32_t* dOverflow) {
    const uint64_t cells = f->nSteps * SB_STEP;
    f->wide = wide;
    (void)hipFree(f->cr); (void)hipFree(f->lrowW); (void)hipFree(f->stepBase);
    f->cr = nullptr; f->lrowW = nullptr; f->stepBase = nullptr;
    if (!f->val && !f->unit) SB_TRY(hipMalloc(&f->val, std::max<uint64_t>(cells, 1) * 8));
    SB_TRY(hipMalloc(&f->cr, std::max<uint64_t>(cells, 1) * 4));
    if (wide) SB_TRY(hipMalloc(&f->lrowW, std::max<uint64_t>(cells, 1) * 2));
    else      SB_TRY(hipMalloc(&f->stepBase, std::max<uint64_t>(f->nSteps, 1) * 4));
    // padding entries: value 0, local row SB_NONE (skipped by the kernel: 0 * x must not turn an Inf/NaN of x into a NaN of y)
    if (f->val) SB_TRY(hipMemsetAsync(f->val, 0, cells * 8, nullptr));
    if (cells) {
        if (wide) {
            SB_TRY(hipMemsetAsync(f->cr, 0, cells * 4, nullptr));
            hipLaunchKernelGGL(sb_fill16_kernel, grid2d((cells + 255) / 256, 256), dim3(256), 0, nullptr, f->lrowW, cells, (uint16_t)SB_NONE);
        } else {
            hipLaunchKernelGGL(sb_fill32_kernel, grid2d((cells + 255) / 256, 256), dim3(256), 0, nullptr, f->cr, cells, SB_NONE);
            SB_TRY(hipMemsetAsync(f->stepBase, 0, f->nSteps * 4, nullptr));
        }
    }
    SB_TRY(hipMemsetAsync(dOverflow, 0, 4, nullptr));
    if (wide)
        hipLaunchKernelGGL(sb_scatter_kernel<true>, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, colBits, f->subs, skeys, perm, rowOf, AS,
                           f->binRow, dStart, f->subStep, f->val, f->cr, f->lrowW, f->stepBase, dOverflow);
    else
        hipLaunchKernelGGL(sb_scatter_kernel<false>, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, colBits, f->subs, skeys, perm, rowOf, AS,
                           f->binRow, dStart, f->subStep, f->val, f->cr, f->lrowW, f->stepBase, dOverflow);
    SB_TRY(hipGetLastError());
    return EXIT_SUCCESS;
}

template <bool WIDE, int DET>
void launchSpmv(const StripeFormat* f, const double* x, double* y, hipStream_t stream) {
    if (f->unit)
        hipLaunchKernelGGL((sb_spmv_kernel<WIDE, DET, true>), dim3(f->grid), dim3(SB_THREADS), (size_t)8 * f->R + 16, stream, f->B, f->binRow, f->subStep,
                           f->unitValue, f->val, f->cr, f->lrowW, f->stepBase, x, y, f->R, f->spread);
    else
        hipLaunchKernelGGL((sb_spmv_kernel<WIDE, DET, false>), dim3(f->grid), dim3(SB_THREADS), (size_t)8 * f->R + 16, stream, f->B, f->binRow, f->subStep,
                           f->unitValue, f->val, f->cr, f->lrowW, f->stepBase, x, y, f->R, f->spread);
}

}  // namespace

void freeStripes(StripeFormat* f) {
    if (!f) return;
    (void)hipFree(f->val); (void)hipFree(f->cr); (void)hipFree(f->lrowW); (void)hipFree(f->stepBase);
    (void)hipFree(f->binRow); (void)hipFree(f->subStep);
    delete f;
}

size_t stripesBytes(const DevMat* d) { return (d->stripes ? d->stripes->bytes : 0) + (d->stripesAlt ? d->stripesAlt->bytes : 0); }

// a handle holds at most one format of each form; the requested one becomes the active slot
void useStripes(DevMat* d, bool deterministic) {
    if (d->stripes && d->stripes->det == deterministic) return;
    if (d->stripes || d->stripesAlt) std::swap(d->stripes, d->stripesAlt);
    if (d->stripes && d->stripes->det != deterministic) std::swap(d->stripes, d->stripesAlt);
}

void stripesInfo(const DevMat* d, spmvStripesInfo* out) {
    memset(out, 0, sizeof *out);
    const StripeFormat* f = d->stripes;
    if (!f) return;
    out->nBins = f->B; out->rowsPerBin = f->R; out->grid = f->grid; out->spread = f->spread;
    out->wide = f->wide ? 1 : 0; out->deterministic = f->det ? 1 : 0; out->buildMs = f->buildMs; out->bytes = f->bytes;
    // (deterministic = 1: the owner-wavefront layout; the caller overrides it with 2 when it runs this layout in ticket order)
}

// `opts` == nullptr: automatic format of the shared-stream layout, kept if one exists.  Explicit options: the existing format of
// that LAYOUT is replaced (the other layout, if the handle holds it, is untouched).  Layouts: opts->deterministic == 1 -> per-
// wavefront sub-streams; 0 and 2 -> one stream per bin (run in arrival order or in ticket order: a choice of the launch).
int buildStripes(DevMat* d, const spmvStripesOpts* opts) {
    useStripes(d, opts && opts->deterministic == 1);
    if (d->stripes && !opts) return EXIT_SUCCESS;
    if (d->kind != Kind::CSR) return EXIT_FAILURE;
    const spmvStripesOpts o = opts ? *opts : spmvStripesOpts{0, 0, -1, -1, 0};
    const uint64_t nnz = d->NZ, M = d->M, N = d->N;
    if (nnz >= IRP32_LIMIT || nnz == 0 || M == 0) {
        fprintf(stderr, "libspmvhip: stripes: nnz = %lu unsupported (needs 0 < nnz < 2^32)\n", (unsigned long)nnz);
        return EXIT_FAILURE;
    }
    int dev = 0, cusDev = 0;                         // one workgroup per CU of the CURRENT device (a process may drive several)
    SB_TRY(hipGetDevice(&dev));
    SB_TRY(hipDeviceGetAttribute(&cusDev, hipDeviceAttributeMultiprocessorCount, dev));
    cusDev = std::max(1, cusDev);
    if (o.rowsPerBin > SB_R_MAX || o.grid > (unsigned)cusDev || o.spread < -1 || o.spread > 1024 || o.deterministic < 0 || o.deterministic > 2) {
        fprintf(stderr, "libspmvhip: stripes: options out of range (rowsPerBin 0..%u, grid 0..%d, spread -1..1024, deterministic 0..2)\n", SB_R_MAX, cusDev);
        return EXIT_FAILURE;
    }
    if (d->stripes) { freeStripes(d->stripes); d->stripes = nullptr; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    SB_TRY(hipEventCreate(&ev0));
    SB_TRY(hipEventCreate(&ev1));
    struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evGuard{ev0, ev1};
    SB_TRY(hipEventRecord(ev0, nullptr));

    // row pointers on the host (bins are cut there)
    std::vector<uint64_t> irp(M + 1);
    if (d->irpBytes == 8) SB_TRY(hipMemcpy(irp.data(), d->IRP, (M + 1) * 8, hipMemcpyDeviceToHost));
    else {
        std::vector<uint32_t> tmp(M + 1);
        SB_TRY(hipMemcpy(tmp.data(), d->IRP, (M + 1) * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i <= M; ++i) irp[i] = tmp[i];
    }
    const uint32_t rMax = o.rowsPerBin ? o.rowsPerBin : SB_R_MAX;
    const uint32_t cus = o.grid ? o.grid : (uint32_t)cusDev;          // workgroups that will walk the bins: the bin count is a multiple of it
    std::vector<uint32_t> binRow;
    if (!planBins(irp, M, rMax, cus, binRow)) { fprintf(stderr, "libspmvhip: stripes: cannot cut %lu rows into bins\n", (unsigned long)M); return EXIT_FAILURE; }
    { std::vector<uint64_t>().swap(irp); }
    const uint32_t B = (uint32_t)binRow.size() - 1;
    uint32_t R = 0;
    for (uint32_t b = 0; b < B; ++b) R = std::max(R, binRow[b + 1] - binRow[b]);

    StripeFormat* f = new StripeFormat;
    f->opts = o;
    f->det = o.deterministic == 1;
    f->unit = d->unit; f->unitValue = d->unitValue;
    f->subs = f->det ? SB_WAVES : 1;
    f->B = B; f->R = R; f->nnz = nnz;
    f->grid = std::min<uint32_t>(B, cus);
    f->spread = o.spread >= 0 ? (uint32_t)o.spread : SB_SPREAD;          // (used by the arrival-order launch only)
    const uint64_t nGroups = (uint64_t)B * f->subs;
    auto fail = [&](const char* what) { (void)hipGetLastError(); fprintf(stderr, "libspmvhip: stripes: %s failed\n", what); freeStripes(f); return EXIT_FAILURE; };
    TempBuf keys, keysOut, idx, perm, rowOf, sortTmp, dStart, dOverflow;
    if (hipMalloc(&f->binRow, ((size_t)B + 1) * 4) || hipMalloc(&f->subStep, ((size_t)nGroups + 1) * 4) || dStart.alloc(((size_t)nGroups + 1) * 8) ||
        dOverflow.alloc(4))
        return fail("table allocation");
    if (hipMemcpy(f->binRow, binRow.data(), ((size_t)B + 1) * 4, hipMemcpyHostToDevice)) return fail("table upload");
    if (keys.alloc(nnz * 8) || keysOut.alloc(nnz * 8) || idx.alloc(nnz * 4) || perm.alloc(nnz * 4) || rowOf.alloc(nnz * 4))
        return fail("temporary allocation");

    unsigned colBits = 1, groupBits = 1;
    while (colBits < 32 && (1ull << colBits) < N) ++colBits;
    while ((1ull << groupBits) < nGroups) ++groupBits;
    if (d->irpBytes == 4)
        hipLaunchKernelGGL((sb_keys_kernel<uint32_t>), grid2d((M + 3) / 4, 256), dim3(256), 0, nullptr, M, static_cast<const uint32_t*>(d->IRP), d->JA,
                           f->binRow, B, f->subs, colBits, keys.as<uint64_t>(), idx.as<uint32_t>(), rowOf.as<uint32_t>());
    else
        hipLaunchKernelGGL((sb_keys_kernel<uint64_t>), grid2d((M + 3) / 4, 256), dim3(256), 0, nullptr, M, static_cast<const uint64_t*>(d->IRP), d->JA,
                           f->binRow, B, f->subs, colBits, keys.as<uint64_t>(), idx.as<uint32_t>(), rowOf.as<uint32_t>());
    if (hipGetLastError() != hipSuccess) return fail("key kernel");
    // (rocPRIM's double-buffer interface: the sort ping-pongs between the two pairs of buffers given here instead of
    // allocating a third full-size pair inside its temporary storage)
    rocprim::double_buffer<uint64_t> dKeys(keys.as<uint64_t>(), keysOut.as<uint64_t>());
    rocprim::double_buffer<uint32_t> dIdx(idx.as<uint32_t>(), perm.as<uint32_t>());
    size_t tmpBytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmpBytes, dKeys, dIdx, (size_t)nnz, 0, colBits + groupBits, (hipStream_t) nullptr) != hipSuccess ||
        sortTmp.alloc(tmpBytes))
        return fail("sort workspace");
    if (rocprim::radix_sort_pairs(sortTmp.p, tmpBytes, dKeys, dIdx, (size_t)nnz, 0, colBits + groupBits, (hipStream_t) nullptr) != hipSuccess)
        return fail("sort");
    const uint64_t* const skeys = dKeys.current();
    const uint32_t* const sperm = dIdx.current();

    // where every sub-stream starts in the sorted order -> its first step (each sub-stream is padded to whole steps)
    hipLaunchKernelGGL(sb_bounds_kernel, grid2d((nnz + 255) / 256, 256), dim3(256), 0, nullptr, nnz, colBits, skeys, nGroups,
                       dStart.as<uint64_t>());
    if (hipGetLastError() != hipSuccess) return fail("bounds kernel");
    std::vector<uint64_t> start(nGroups + 1);
    if (hipMemcpy(start.data(), dStart.p, (nGroups + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail("bounds download");
    std::vector<uint32_t> subStep(nGroups + 1);
    uint64_t steps = 0;
    for (uint64_t g = 0; g < nGroups; ++g) {
        subStep[g] = (uint32_t)steps;
        steps += (start[g + 1] - start[g] + SB_STEP - 1) / SB_STEP;
    }
    if (steps >= (1ull << 32) - (1u << 16)) return fail("step count (too many steps)");
    subStep[nGroups] = (uint32_t)steps;
    f->nSteps = steps;
    if (hipMemcpy(f->subStep, subStep.data(), (nGroups + 1) * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("table upload");

    bool wide = o.wide > 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (fillFormat(f, wide, nnz, colBits, skeys, sperm, rowOf.as<uint32_t>(), d->AS,
                       dStart.as<uint64_t>(), dOverflow.as<uint32_t>()))
            return fail("scatter");
        uint32_t ovf = 0;
        if (hipMemcpy(&ovf, dOverflow.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail("overflow flag");
        if (!ovf || wide) break;
        wide = true;                                  // some step spans >= 2^17 columns: 32-bit columns + 16-bit rows
    }
    // (set at every build: the attribute belongs to the current device, and a process may drive several)
    const void* kernels[] = {(const void*)sb_spmv_kernel<false, 0, false>, (const void*)sb_spmv_kernel<true, 0, false>, (const void*)sb_spmv_kernel<false, 1, false>,
                             (const void*)sb_spmv_kernel<true, 1, false>,  (const void*)sb_spmv_kernel<false, 2, false>, (const void*)sb_spmv_kernel<true, 2, false>,
                             (const void*)sb_spmv_kernel<false, 0, true>,  (const void*)sb_spmv_kernel<true, 0, true>,  (const void*)sb_spmv_kernel<false, 1, true>,
                             (const void*)sb_spmv_kernel<true, 1, true>,   (const void*)sb_spmv_kernel<false, 2, true>,  (const void*)sb_spmv_kernel<true, 2, true>};
    bool attrFailed = false;
    for (const void* k : kernels) attrFailed |= hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SB_R_MAX * 8 + 16) != hipSuccess;
    if (attrFailed)
        return fail("kernel attribute");
    const uint64_t cells = steps * SB_STEP;
    f->bytes = cells * ((f->wide ? 14 : 12) - (f->unit ? 8 : 0)) + (f->wide ? 0 : steps * 4) + ((size_t)B + 1) * 4 + ((size_t)nGroups + 1) * 4;
    if (hipEventRecord(ev1, nullptr) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess) return fail("synchronise");
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    f->buildMs = ms;
    d->stripes = f;
    return EXIT_SUCCESS;
}

// One persistent workgroup per CU (its bin of y fills the LDS).  Start of the sweeps: the workgroups of one XCD begin
// SB_SPREAD/1024 of the bin apart in all (0.6 %, wrapping around).  Started together they all ask L2 for the same lines in
// the same microsecond and every one of them waits for the fabric; a few microseconds apart the first to arrive pays and
// the rest hit: the first quarter of a bin 113 instead of 123 us and the bin 404 instead of 417 us on c3.  Past ~2 % the
// tail of the pack outlives its lines in the 4 MiB L2 (c3: 6.4 % 1.05 ms, 100 % 2.96 ms; profiles/r02_stripes_spread.log).
// mode 0: arrival order; 1: owner wavefronts (needs the sub-stream layout); 2: ordered tickets (needs the shared-stream layout)
int enqueueStripes(DevMat* d, const double* x, double* y, hipStream_t stream, int mode, dim3* grid, dim3* block) {
    const StripeFormat* f = d->stripes;
    if (!f || mode < 0 || mode > 2 || (mode == 1) != f->det) return EXIT_FAILURE;
    if (grid) *grid = dim3(f->grid);
    if (block) *block = dim3(SB_THREADS);
    if (f->wide) { if (mode == 2) launchSpmv<true, 2>(f, x, y, stream); else if (mode == 1) launchSpmv<true, 1>(f, x, y, stream); else launchSpmv<true, 0>(f, x, y, stream); }
    else         { if (mode == 2) launchSpmv<false, 2>(f, x, y, stream); else if (mode == 1) launchSpmv<false, 1>(f, x, y, stream); else launchSpmv<false, 0>(f, x, y, stream); }
    return hipGetLastError() == hipSuccess ? EXIT_SUCCESS : EXIT_FAILURE;
}

}  // namespace spmvhip
