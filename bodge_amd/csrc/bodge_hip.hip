// Host side of libbodge_hip.so: the C ABI declared in include/bodge_hip.h.
//
// One `bdg_system` is one BSR Hamiltonian resident in the HBM of one GPU plus
// the work buffers of the Chebyshev recurrence.  All launches go to a private
// non-blocking HIP stream; timing uses HIP events recorded on that stream.
// rocSOLVER (dense path) and RCCL (multi-GPU moments) are loaded with dlopen
// on first use so that the core library has no load-time dependency on them.

#include "bodge_hip.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <rocsolver/rocsolver.h>

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "host_assembly.hpp"
#include "kernels.hpp"
#include "sweep.hpp"

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t err__ = (expr);                                                      \
        if (err__ != hipSuccess)                                                        \
            return fail(err__ == hipErrorOutOfMemory ? BDG_ENOMEM : BDG_EDEVICE,        \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__),       \
                        __FILE__, __LINE__);                                            \
    } while (0)

template <typename T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    int reserve(size_t n) {
        if (n <= count) return BDG_OK;
        release();
        hipError_t err = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
        if (err != hipSuccess) {
            ptr = nullptr;
            return fail(BDG_ENOMEM, "hipMalloc of %zu bytes failed: %s", n * sizeof(T),
                        hipGetErrorString(err));
        }
        count = n;
        return BDG_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
};

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

struct bdg_system;
namespace {
// A Lanczos run keeps pointers into the handle's vector buffers; any other call that refills or
// reallocates them ends the run first (bdg_lanczos_advance then reports that begin is needed).
void lanczos_free(bdg_system* sys);
}

struct ExchangePeer {
    int rank = 0;            // peer's rank (RCCL) or member index (same-process group)
    int64_t send_begin = 0;  // offset into send_rows / send buffer rows
    int64_t send_count = 0;
    int64_t recv_col = 0;    // first local column of the rows received from this peer
    int64_t recv_begin = 0;  // offset into the receive buffer rows
    int64_t recv_count = 0;
};

struct bdg_system {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<hipEvent_t> ev_pool;  // extra (start, stop) pairs: one per reduction chunk of a call
    int64_t nb = 0, nnzb = 0;
    int64_t ncols = 0;       // block rows of the vector buffers: nb owned + halo
    int64_t row_offset = 0;  // global block row of local row 0 (slab mode)
    std::vector<ExchangePeer> peers;
    DeviceBuffer<int64_t> send_rows;
    DeviceBuffer<double2> send_buf, recv_buf;
    int64_t send_total = 0, recv_total = 0;
    bdg_comm* slab_comm = nullptr;  // RCCL transport for the halo exchange (not owned)
    // agreed over slab_comm in bdg_slab_set_exchange: every rank must choose the same arithmetic
    // mode and batch width, or the ncclSend/ncclRecv counts of the halo exchange do not match
    bool slab_all_real = false;
    int64_t slab_max_ncols = 0;
    // overlap of the halo exchange with the rows that do not need it
    std::vector<uint8_t> row_needs_halo;  // host: block row reads at least one halo column
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_step_done = nullptr, ev_halo_ready = nullptr;
    DeviceBuffer<int> tiles_interior, tiles_boundary;
    int split_rows_per_tile = 0, n_interior = 0, n_boundary = 0;
    void* lanczos = nullptr;  // LanczosState of a run in progress (defined with the driver)
    int max_row_blocks = 0;
    int64_t bandwidth = 0;  // max |column - row| over the stored blocks (square matrices)
    int num_cus = 0;
    int lanes_override = 0;
    DeviceBuffer<int> indptr, indices;
    DeviceBuffer<double2> blocks;
    DeviceBuffer<double2> packed[4];   // re-packed blocks per storage mode, built on first use
    bool is_real = false;              // imag(H) == 0 everywhere (checked at upload)
    bool is_ph = false;                // every block is [[A, B], [C, -conj(A)]] (checked at upload)
    double gershgorin = 0.0;           // max over scalar rows of sum |H_rc| (bound on |H|)
    // dictionary form: the distinct blocks and one id per stored block (0 entries = not used)
    int n_unique = 0;
    int dict_skipped = 0;  // why there is no dictionary: 0 = there is one, 1 = > 256 distinct blocks,
                           // 2 = more than 2^24 block columns (the packed word holds 24 bits), 3 = switched off
    DeviceBuffer<int> dict_ids;
    DeviceBuffer<int> dict_diagonal;      // per distinct block: 1 = diagonal as a 4x4 matrix (stencil kernels)
    DeviceBuffer<double2> dict_full;      // n_unique x 16 complex entries
    DeviceBuffer<double2> dict_table[4];  // packed per storage mode, built on first use
    DeviceBuffer<double2> vec_a, vec_b;
    DeviceBuffer<double2> vec_c, vec_d;  // two-steps-per-sweep form: t_{n+1}, t_{n+2} are written out of place
    // lattice-stencil form of the matrix (sweep.hpp): 0 = not examined, 1 = 5-point table built (planes
    // are lines), 2 = 7-point table built (3-D), -1 = not a stencil
    DeviceBuffer<uint2> stencil;
    int stencil_state = 0;
    bool stencil_wrap_p = false, stencil_wrap_x = false;  // periodic edge blocks: planes / stack of planes are rings
    DeviceBuffer<double> partial, dots;
    double* host_dots = nullptr;  // pinned staging for the dot products (sized like `dots`)
    size_t host_dots_count = 0;
    DeviceBuffer<int64_t> rows;
    // lattice geometry hint (rows = z + lz*(y + ly*x)) and the cached strip-major tile order
    int shape[3] = {0, 0, 0};
    DeviceBuffer<int> tile_order;
    int order_rows_per_tile = 0, order_strip_rows = 0;
    bdg_perf perf{};
};

struct bdg_comm {
    int device = 0;
    int n_ranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DeviceBuffer<double> scratch;
};

struct bdg_group {
    std::vector<bdg_system*> members;
    std::vector<hipEvent_t> packed, copied;  // per member
};

namespace {

// ------------------------------------------------------------ kernel dispatch
using StepKernel = void (*)(bdg::StepArgs);

using bdg::ComplexMode;
using bdg::ComplexPHMode;
using bdg::RealMode;
using bdg::RealPHMode;

// Storage / arithmetic mode of a launch.  id indexes bdg_system::packed.
struct ModeInfo {
    int id;          // 0 complex full, 1 real full, 2 complex PH, 3 real PH
    bool real, ph;
    int per_lane;    // vectors per lane
    int stride;      // LDS slots (16 B) per staged block
    double block_bytes;  // HBM bytes per stored block, index included
    double entry_bytes;  // HBM bytes per (site, vector) and launch: read t_n, read t_{n-1}, write t_{n+1}
};

ModeInfo mode_info(bool real, bool ph) {
    if (real && ph) return {3, true, true, RealPHMode::kVec, RealPHMode::kBlockStride, 100.0, 96.0};
    if (real) return {1, true, false, RealMode::kVec, RealMode::kBlockStride, 132.0, 96.0};
    if (ph) return {2, false, true, ComplexPHMode::kVec, ComplexPHMode::kBlockStride, 196.0, 192.0};
    return {0, false, false, ComplexMode::kVec, ComplexMode::kBlockStride, 260.0, 192.0};
}

template <typename Mode>
StepKernel generic_kernel(int rl) {
    switch (rl) {
        case 4: return bdg::cheb_step<Mode, 4>;
        case 8: return bdg::cheb_step<Mode, 8>;
        case 16: return bdg::cheb_step<Mode, 16>;
        case 32: return bdg::cheb_step<Mode, 32>;
        case 64: return bdg::cheb_step<Mode, 64>;
    }
    return nullptr;
}

template <typename Mode>
StepKernel generic_cols_kernel(int rl) {
    switch (rl) {
        case 4: return bdg::cheb_step<Mode, 4, true>;
        case 8: return bdg::cheb_step<Mode, 8, true>;
        case 16: return bdg::cheb_step<Mode, 16, true>;
        case 32: return bdg::cheb_step<Mode, 32, true>;
        case 64: return bdg::cheb_step<Mode, 64, true>;
    }
    return nullptr;
}

// Kernels taking per-column scalars (Lanczos): every (mode, lanes) in the generic form, and the
// dictionary form for 4 lanes per row (the usual 4..8 start vectors).
StepKernel step_cols_kernel(const ModeInfo& mode, int rl) {
    switch (mode.id) {
        case 1: return generic_cols_kernel<RealMode>(rl);
        case 2: return generic_cols_kernel<ComplexPHMode>(rl);
        case 3: return generic_cols_kernel<RealPHMode>(rl);
    }
    return generic_cols_kernel<ComplexMode>(rl);
}

template <int MAXB>
StepKernel dict_cols_for(const ModeInfo& mode) {
    switch (mode.id) {
        case 1: return bdg::cheb_step_dict<RealMode, 4, MAXB, true>;
        case 2: return bdg::cheb_step_dict<ComplexPHMode, 4, MAXB, true>;
        case 3: return bdg::cheb_step_dict<RealPHMode, 4, MAXB, true>;
    }
    return bdg::cheb_step_dict<ComplexMode, 4, MAXB, true>;
}

StepKernel step_kernel(const ModeInfo& mode, int rl) {
    switch (mode.id) {
        case 1: return generic_kernel<RealMode>(rl);
        case 2: return generic_kernel<ComplexPHMode>(rl);
        case 3: return generic_kernel<RealPHMode>(rl);
    }
    return generic_kernel<ComplexMode>(rl);
}

// Pipelined kernels exist for rows of at most 3 / 5 / 7 blocks (the 1-D / 2-D /
// 3-D cubic stencils) and 8..64 lanes per row (complex) or 4..32 (real, two
// vectors per lane); anything else runs the generic form.
template <typename CMode, typename RMode, int MAXB>
StepKernel pipelined_pair(bool real, int rl) {
    if (real) {
        switch (rl) {
            case 4: return bdg::cheb_step_pipelined<RMode, 4, MAXB>;
            case 8: return bdg::cheb_step_pipelined<RMode, 8, MAXB>;
            case 16: return bdg::cheb_step_pipelined<RMode, 16, MAXB>;
            case 32: return bdg::cheb_step_pipelined<RMode, 32, MAXB>;
        }
        return nullptr;
    }
    switch (rl) {
        case 8: return bdg::cheb_step_pipelined<CMode, 8, MAXB>;
        case 16: return bdg::cheb_step_pipelined<CMode, 16, MAXB>;
        case 32: return bdg::cheb_step_pipelined<CMode, 32, MAXB>;
        case 64: return bdg::cheb_step_pipelined<CMode, 64, MAXB>;
    }
    return nullptr;
}

template <int MAXB>
StepKernel pipelined_for(const ModeInfo& mode, int rl) {
    return mode.ph ? pipelined_pair<ComplexPHMode, RealPHMode, MAXB>(mode.real, rl)
                   : pipelined_pair<ComplexMode, RealMode, MAXB>(mode.real, rl);
}

template <typename CMode, typename RMode, int MAXB>
StepKernel dict_pair(bool real, int rl) {
    if (real) {
        switch (rl) {
            case 4: return bdg::cheb_step_dict<RMode, 4, MAXB>;
            case 8: return bdg::cheb_step_dict<RMode, 8, MAXB>;
            case 16: return bdg::cheb_step_dict<RMode, 16, MAXB>;
            case 32: return bdg::cheb_step_dict<RMode, 32, MAXB>;
        }
        return nullptr;
    }
    switch (rl) {
        case 4: return bdg::cheb_step_dict<CMode, 4, MAXB>;
        case 8: return bdg::cheb_step_dict<CMode, 8, MAXB>;
        case 16: return bdg::cheb_step_dict<CMode, 16, MAXB>;
        case 32: return bdg::cheb_step_dict<CMode, 32, MAXB>;
        case 64: return bdg::cheb_step_dict<CMode, 64, MAXB>;
    }
    return nullptr;
}

template <int MAXB>
StepKernel dict_for(const ModeInfo& mode, int rl) {
    return mode.ph ? dict_pair<ComplexPHMode, RealPHMode, MAXB>(mode.real, rl)
                   : dict_pair<ComplexMode, RealMode, MAXB>(mode.real, rl);
}

constexpr size_t kDictLdsLimit = 32 * 1024;  // bytes of LDS the block table may take per workgroup

// Dictionary kernel if the matrix has few enough distinct blocks for the table to sit in LDS.
StepKernel dict_kernel(const bdg_system* sys, const ModeInfo& mode, int rl) {
    const char* env = getenv("BODGE_AMD_DICT");
    if (env && env[0] == '0') return nullptr;
    if (sys->n_unique <= 0 || (size_t)sys->n_unique * mode.stride * sizeof(double2) > kDictLdsLimit)
        return nullptr;
    if (sys->max_row_blocks <= 3) return dict_for<3>(mode, rl);
    if (sys->max_row_blocks <= 5) return dict_for<5>(mode, rl);
    if (sys->max_row_blocks <= 7) return dict_for<7>(mode, rl);
    return nullptr;
}

StepKernel pipelined_kernel(const ModeInfo& mode, int rl, int max_row_blocks, int* maxb_out) {
    const char* env = getenv("BODGE_AMD_KERNEL");
    if (env && std::string(env) == "generic") return nullptr;
    if (max_row_blocks <= 3) { *maxb_out = 3; return pipelined_for<3>(mode, rl); }
    if (max_row_blocks <= 5) { *maxb_out = 5; return pipelined_for<5>(mode, rl); }
    if (max_row_blocks <= 7) { *maxb_out = 7; return pipelined_for<7>(mode, rl); }
    return nullptr;
}

struct StepPlan {
    int rl = 0;
    int rows_per_tile = 0;
    int n_tiles = 0;
    int grid = 0;
    size_t lds_bytes = 0;      // dynamic LDS to request at launch
    size_t lds_footprint = 0;  // what one workgroup occupies (reported)
    bool pipelined = false;
    bool dictionary = false;
    int stage_blocks = 1;      // generic form: blocks per wave staging region
    ModeInfo mode{};
    StepKernel kernel = nullptr;
};

int make_plan(bdg_system* sys, int rl, const ModeInfo& mode, StepPlan* plan, bool col_scalars = false) {
    plan->rl = rl;
    plan->mode = mode;
    const int block_stride = mode.stride;
    const int lane_doubles = 2 * mode.per_lane;
    const int rows_per_wave = bdg::kWave / rl;
    plan->rows_per_tile = rows_per_wave * bdg::kWavesPerBlock;
    plan->n_tiles = (int)((sys->nb + plan->rows_per_tile - 1) / plan->rows_per_tile);
    int maxb = 0;
    plan->kernel = dict_kernel(sys, mode, rl);
    if (col_scalars) {  // Lanczos: same kernel families, instantiations with per-column scalars
        if (plan->kernel && rl == 4)
            plan->kernel = sys->max_row_blocks <= 3   ? dict_cols_for<3>(mode)
                           : sys->max_row_blocks <= 5 ? dict_cols_for<5>(mode)
                                                      : dict_cols_for<7>(mode);
        else
            plan->kernel = nullptr;
    }
    if (plan->kernel) {
        plan->dictionary = true;
        // table of distinct blocks + 4 own t_n entries per lane (16 KiB per workgroup)
        const size_t table = (size_t)sys->n_unique * block_stride * sizeof(double2) +
                             (size_t)bdg::kBlockThreads * 4 * sizeof(double2);
        const size_t reduce = (size_t)bdg::kWavesPerBlock * rl * lane_doubles * sizeof(double);
        plan->lds_bytes = plan->lds_footprint = std::max(table, reduce);
    } else if (!col_scalars && (plan->kernel = pipelined_kernel(mode, rl, sys->max_row_blocks, &maxb))) {
        plan->pipelined = true;
        plan->lds_bytes = 0;
        plan->lds_footprint = (size_t)bdg::kWavesPerBlock * rows_per_wave * maxb * block_stride *
                                  sizeof(double2) +
                              (size_t)bdg::kWavesPerBlock * rl * lane_doubles * sizeof(double);
    } else {
        plan->kernel = col_scalars ? step_cols_kernel(mode, rl) : step_kernel(mode, rl);
        if (!plan->kernel) return fail(BDG_EINVAL, "unsupported lanes-per-row %d", rl);
        // a wave stages its tile's blocks in LDS; tiles that do not fit a quarter of the 160 KB
        // (long rows of general matrices) pass through in chunks
        const int tile_blocks = rows_per_wave * std::max(1, sys->max_row_blocks);
        const int cap = (int)((160 * 1024 / bdg::kWavesPerBlock) / (block_stride * sizeof(double2)));
        plan->stage_blocks = std::max(1, std::min(tile_blocks, cap));
        const size_t stage = (size_t)bdg::kWavesPerBlock * plan->stage_blocks * block_stride * sizeof(double2);
        const size_t reduce = (size_t)bdg::kWavesPerBlock * rl * lane_doubles * sizeof(double);
        plan->lds_bytes = plan->lds_footprint = std::max(stage, reduce);
        if (plan->lds_bytes > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(plan->kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)plan->lds_bytes));
    }
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &per_cu, reinterpret_cast<const void*>(plan->kernel), bdg::kBlockThreads, plan->lds_bytes));
    per_cu = std::max(1, std::min(per_cu, 8));
    if (const char* cap = getenv("BODGE_AMD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(cap));
    int grid = std::min(plan->n_tiles, per_cu * sys->num_cus);
    plan->grid = std::max(8, (grid + 7) / 8 * 8);
    return BDG_OK;
}

// Algorithmic HBM bytes of one recurrence launch: every stored block and index
// once, and per (site, vector) one read of t_n, one read of t_{n-1}, one write
// of t_{n+1} (260 nnzb + 4 (nb+1) + 192 R nb).
// The other modes store and move less (real: half; particle-hole packed: 12 of 16 entries),
// and are charged with their own byte counts.
// In the dictionary form a stored block costs one packed word (column | id, 4 B); the table
// of distinct blocks is read once per workgroup from L2 and is charged once.
double algorithmic_bytes(const bdg_system* sys, int vectors, const ModeInfo& mode, bool dictionary) {
    const double per_block = dictionary ? 4.0 : mode.block_bytes;
    const double table = dictionary ? (mode.block_bytes - 4.0) * sys->n_unique : 0.0;
    return per_block * (double)sys->nnzb + 4.0 * (double)(sys->nb + 1) + table +
           mode.entry_bytes * (double)vectors * (double)sys->nb;
}

// Table of distinct blocks in the layout `mode` reads.
int ensure_dict_table(bdg_system* sys, const ModeInfo& mode, const void** out) {
    DeviceBuffer<double2>& buf = sys->dict_table[mode.id];
    if (!buf.ptr) {
        const int entries = mode.ph ? 12 : 16;
        const size_t doubles = (size_t)sys->n_unique * entries * (mode.real ? 1 : 2);
        if (int rc = buf.reserve((doubles + 1) / 2)) return rc;
        bdg::pack_blocks<<<(sys->n_unique * entries + 255) / 256, 256, 0, sys->stream>>>(
            sys->dict_full.ptr, buf.ptr, sys->n_unique, entries, mode.real ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    *out = buf.ptr;
    return BDG_OK;
}

// Blocks in the layout `mode` reads (built on the device from the uploaded copy, once).
int ensure_blocks(bdg_system* sys, const ModeInfo& mode, const void** out) {
    if (mode.id == 0) {
        *out = sys->blocks.ptr;
        return BDG_OK;
    }
    DeviceBuffer<double2>& buf = sys->packed[mode.id];
    if (!buf.ptr) {
        const int entries = mode.ph ? 12 : 16;
        const size_t doubles = (size_t)std::max<int64_t>(1, sys->nnzb) * entries * (mode.real ? 1 : 2);
        if (int rc = buf.reserve((doubles + 1) / 2)) return rc;
        const int64_t total = sys->nnzb * entries;
        const int grid = (int)std::min<int64_t>(8192, (total + 255) / 256 + 1);
        bdg::pack_blocks<<<grid, 256, 0, sys->stream>>>(sys->blocks.ptr, buf.ptr, sys->nnzb, entries,
                                                        mode.real ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    *out = buf.ptr;
    return BDG_OK;
}

// Matrix-side kernel arguments for `plan` (block data or dictionary, sizes).  Every launch of
// a step kernel goes through here so that no pointer the chosen kernel reads is left unset.
constexpr size_t kStreamVectorBytes = (size_t)256 << 20;

int matrix_args(bdg_system* sys, const StepPlan& plan, bdg::StepArgs* args) {
    *args = bdg::StepArgs{};
    args->indptr = sys->indptr.ptr;
    args->indices = sys->indices.ptr;
    if (plan.dictionary) {
        if (int rc = ensure_dict_table(sys, plan.mode, &args->dict_table)) return rc;
        args->dict_ids = sys->dict_ids.ptr;
        args->n_unique = sys->n_unique;
    } else if (int rc = ensure_blocks(sys, plan.mode, &args->blocks)) {
        return rc;
    }
    args->nb = (int)sys->nb;
    args->ncols = (int)sys->ncols;
    args->n_tiles = plan.n_tiles;
    args->max_row_blocks = sys->max_row_blocks;
    args->stage_blocks = std::max(1, plan.stage_blocks);
    return BDG_OK;
}

// Strip-major tile order for lattice matrices.  Block rows are numbered
// z + lz*(y + ly*x): neighbours along x are a whole plane (ly*lz rows) apart, so a
// sweep in natural order re-touches a t_n line only after 2*ly*lz rows of other
// traffic.  When that exceeds what the XCD's 4 MB L2 keeps, the planes are cut
// into strips of `strip_rows` consecutive rows and the sweep runs along x inside
// one strip before moving to the next; the re-use distance becomes 2*strip_rows.
// Returns nullptr (natural order) when no geometry is known or one strip suffices.
int prepare_tile_order(bdg_system* sys, int rows_per_tile, int n_tiles, double row_bytes,
                       const int** order_out, int* strip_out) {
    *order_out = nullptr;
    *strip_out = 0;
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    if (plane <= 0 || (int64_t)sys->shape[0] * plane != sys->nb) return BDG_OK;
    double budget = 1024.0 * 1024.0;  // bytes of t_n lines an XCD should have to hold between re-uses
    if (const char* env = getenv("BODGE_AMD_L2_BUDGET")) budget = atof(env);
    int64_t strip = (int64_t)(budget / (2.0 * row_bytes));
    strip = std::max<int64_t>(rows_per_tile, strip / rows_per_tile * rows_per_tile);
    if (strip >= plane || budget <= 0) return BDG_OK;
    *strip_out = (int)strip;
    if (sys->order_rows_per_tile == rows_per_tile && sys->order_strip_rows == strip) {
        *order_out = sys->tile_order.ptr;
        return BDG_OK;
    }
    // Tile t (first row r0 = t * rows_per_tile) belongs to plane x = r0 / plane and strip
    // (r0 % plane) / strip; emit strip by strip, plane by plane, ascending inside.  O(n_tiles).
    std::vector<int> order;
    order.reserve(n_tiles);
    const int64_t T = rows_per_tile;
    for (int64_t lo_w = 0; lo_w < plane; lo_w += strip) {
        const int64_t hi_w = std::min(plane, lo_w + strip);
        for (int64_t x = 0; x < sys->shape[0]; ++x) {
            const int64_t first = (x * plane + lo_w + T - 1) / T;  // first tile starting in the window
            const int64_t last = (x * plane + hi_w + T - 1) / T;   // one past the last such tile
            for (int64_t t = first; t < last && t < n_tiles; ++t) order.push_back((int)t);
        }
    }
    if ((int)order.size() != n_tiles)
        return fail(BDG_EDEVICE, "internal error: tile order has %zu of %d tiles", order.size(), n_tiles);
    if (int rc = sys->tile_order.reserve((size_t)n_tiles)) return rc;
    HIP_TRY(hipMemcpy(sys->tile_order.ptr, order.data(), sizeof(int) * n_tiles, hipMemcpyHostToDevice));
    sys->order_rows_per_tile = rows_per_tile;
    sys->order_strip_rows = (int)strip;
    *order_out = sys->tile_order.ptr;
    return BDG_OK;
}

// ------------------------------------------------------- two steps per sweep (sweep.hpp)
using SweepKernel = void (*)(bdg::SweepArgs);

template <typename Mode>
SweepKernel sweep_kernel_for(int lanes, bool reverse) {
    switch (lanes) {
        case 1: return reverse ? bdg::cheb_sweep<Mode, 1, true> : bdg::cheb_sweep<Mode, 1, false>;
        case 2: return reverse ? bdg::cheb_sweep<Mode, 2, true> : bdg::cheb_sweep<Mode, 2, false>;
        case 4: return reverse ? bdg::cheb_sweep<Mode, 4, true> : bdg::cheb_sweep<Mode, 4, false>;
    }
    return nullptr;
}

SweepKernel sweep_kernel(const ModeInfo& mode, int lanes, bool reverse) {
    switch (mode.id) {
        case 1: return sweep_kernel_for<RealMode>(lanes, reverse);
        case 2: return sweep_kernel_for<ComplexPHMode>(lanes, reverse);
        case 3: return sweep_kernel_for<RealPHMode>(lanes, reverse);
    }
    return sweep_kernel_for<ComplexMode>(lanes, reverse);
}

// Segments along x for the marching kernels.  The waves of a launch take the (segment, window)
// units in rounds, so the launch lasts  ceil(units / waves) x (planes per segment + the planes a
// unit recomputes at its ends);  fewer, longer segments also re-read less.  Smallest count within
// 3 % of the best duration.
int choose_segments(int n_cols, int lx, int waves, int extra_planes, int min_planes) {
    int best = 1;
    double best_cost = 0.0;
    for (int segs = 1; segs <= std::max(1, lx / min_planes); ++segs) {
        const int64_t units = (int64_t)n_cols * segs;
        const double rounds = (double)((units + waves - 1) / waves);
        const double cost = rounds * ((double)((lx + segs - 1) / segs) + extra_planes);
        if (segs == 1 || cost < 0.97 * best_cost) {
            best = segs;
            best_cost = cost;
        }
    }
    return best;
}

template <typename Mode>
SweepKernel sweep3_kernel_for(int lanes, bool reverse) {
    switch (lanes) {
        case 2: return reverse ? bdg::cheb_sweep3<Mode, 2, true> : bdg::cheb_sweep3<Mode, 2, false>;
        case 4: return reverse ? bdg::cheb_sweep3<Mode, 4, true> : bdg::cheb_sweep3<Mode, 4, false>;
    }
    return nullptr;
}

// cheb_sweep3 that makes the random start block itself (first sweep of a run; marches forward)
template <typename Mode>
SweepKernel sweep3_gen_kernel_for(int lanes) {
    switch (lanes) {
        case 2: return bdg::cheb_sweep3<Mode, 2, false, true>;
        case 4: return bdg::cheb_sweep3<Mode, 4, false, true>;
    }
    return nullptr;
}

SweepKernel sweep3_gen_kernel(const ModeInfo& mode, int lanes) {
    switch (mode.id) {
        case 1: return sweep3_gen_kernel_for<RealMode>(lanes);
        case 2: return sweep3_gen_kernel_for<ComplexPHMode>(lanes);
        case 3: return sweep3_gen_kernel_for<RealPHMode>(lanes);
    }
    return sweep3_gen_kernel_for<ComplexMode>(lanes);
}

SweepKernel sweep3_kernel(const ModeInfo& mode, int lanes, bool reverse) {
    switch (mode.id) {
        case 1: return sweep3_kernel_for<RealMode>(lanes, reverse);
        case 2: return sweep3_kernel_for<ComplexPHMode>(lanes, reverse);
        case 3: return sweep3_kernel_for<RealPHMode>(lanes, reverse);
    }
    return sweep3_kernel_for<ComplexMode>(lanes, reverse);
}

struct SweepPlan {
    int lanes = bdg::kSweepLanes;
    int depth = 2;  // recurrence steps per sweep: 2 (cheb_sweep) or 3 (cheb_sweep3)
    SweepKernel kernel = nullptr, kernel_reverse = nullptr;
    SweepKernel kernel_gen = nullptr;  // depth 3: first sweep of a random-start run, t_0 made in registers
    int grid = 0;
    size_t lds_bytes = 0;
    bdg::SweepArgs args{};
};

// Smallest lattices the stencil kernels are chosen for by default.  Below, the x-segments get so
// short that the planes each wave recomputes at their ends eat the saving, and the one-step
// kernels work from the Infinity Cache with wide batches.  Measured with 64-vector calls
// (profiles/r02_sweep_experiments.log): 300x300 one-step 725 k vector-steps/s vs 504 k, 400x400 407 k
// vs 458 k, 500x500 252 k vs 326 k, 700x700 130 k vs 193 k, 1000x1000 57 k vs 104 k.
// BODGE_AMD_SWEEP=0 never, =1 whenever the matrix qualifies.
constexpr int64_t kSweepMinSites = 150000;   // 2-D: multi-step sweeps (K7, K7b)
constexpr int64_t kRollMinSites = 600000;    // 3-D: rolling one-step kernel (K8)
constexpr int64_t kSweepTwoLaneSites = 450000;  // from here on 2 lanes per site beat 4

// Stencil table of the matrix (built once per lattice shape).  *kind = 1: 5-point stencil whose
// planes are lines (2-D lattice: the two-steps-per-sweep kernel applies), 2: 7-point stencil of a
// 3-D lattice (one-step kernel with the x-neighbours in registers), 0: neither.
int ensure_stencil(bdg_system* sys, int* kind) {
    *kind = 0;
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    if (sys->stencil_state == 0) {
        sys->stencil_state = -1;
        const bool shaped = plane >= 2 * bdg::kSweepOwned && sys->shape[0] >= 8 &&
                            (int64_t)sys->shape[0] * plane == sys->nb;
        const bool three_d = sys->shape[1] > 1 && sys->shape[2] > 1;
        if (shaped && sys->ncols == sys->nb && sys->n_unique > 0 && sys->n_unique < (int)bdg::kNoBlock &&
            sys->max_row_blocks <= (three_d ? 7 : 5) && sys->nnzb > 0) {
            if (int rc = sys->stencil.reserve((size_t)sys->nb)) return rc;
            DeviceBuffer<int> bad;
            if (int rc = bad.reserve(3)) return rc;
            int host_bad[3] = {1, 0, 0};  // {not a stencil, periodic inside the planes, periodic across the planes}
            auto body = [&]() -> int {
                HIP_TRY(hipMemsetAsync(bad.ptr, 0, 3 * sizeof(int), sys->stream));
                const unsigned grid = (unsigned)std::min<int64_t>(4096, (sys->nb + 255) / 256);
                if (three_d)
                    bdg::build_stencil3<<<grid, 256, 0, sys->stream>>>(sys->indptr.ptr, sys->dict_ids.ptr,
                                                                        sys->dict_diagonal.ptr, (int)sys->nb, (int)plane,
                                                                        sys->shape[2], sys->stencil.ptr, bad.ptr);
                else
                    bdg::build_stencil<<<grid, 256, 0, sys->stream>>>(sys->indptr.ptr, sys->dict_ids.ptr,
                                                                       sys->dict_diagonal.ptr, (int)sys->nb, (int)plane,
                                                                       sys->stencil.ptr, bad.ptr);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(host_bad, bad.ptr, 3 * sizeof(int), hipMemcpyDeviceToHost, sys->stream));
                HIP_TRY(hipStreamSynchronize(sys->stream));
                return BDG_OK;
            };
            const int rc = body();
            bad.release();
            if (rc) return rc;
            if (host_bad[0] == 0) {
                sys->stencil_state = three_d ? 2 : 1;
                sys->stencil_wrap_p = host_bad[1] != 0;
                sys->stencil_wrap_x = host_bad[2] != 0;
            } else {
                sys->stencil.release();
            }
        }
    }
    *kind = std::max(0, sys->stencil_state);
    return BDG_OK;
}

// Should this batch run a stencil form, and which (see ensure_stencil)?  Whole square matrix,
// random start vectors (unit vectors use the band-limited one-step sweeps), no per-column scalars.
int sweep_wanted(bdg_system* sys, bool random_start, bool col_scalars, int* kind) {
    *kind = 0;
    const char* env = getenv("BODGE_AMD_SWEEP");
    if ((env && env[0] == '0') || !random_start || col_scalars || !sys->peers.empty()) return BDG_OK;
    const bool forced = env && env[0] == '1';
    if (!forced && sys->nb < std::min(kSweepMinSites, kRollMinSites)) return BDG_OK;
    const char* dict_env = getenv("BODGE_AMD_DICT");
    if (dict_env && dict_env[0] == '0') return BDG_OK;
    if (int rc = ensure_stencil(sys, kind)) return rc;
    if (!forced && sys->nb < (*kind == 2 ? kRollMinSites : kSweepMinSites)) *kind = 0;
    return BDG_OK;
}

int make_sweep_plan(bdg_system* sys, const ModeInfo& mode, int lanes, int depth, SweepPlan* plan) {
    plan->lanes = lanes;
    plan->depth = depth;
    plan->kernel = depth == 3 ? sweep3_kernel(mode, lanes, false) : sweep_kernel(mode, lanes, false);
    plan->kernel_reverse = depth == 3 ? sweep3_kernel(mode, lanes, true) : sweep_kernel(mode, lanes, true);
    plan->kernel_gen = depth == 3 ? sweep3_gen_kernel(mode, lanes) : nullptr;
    if (!plan->kernel) return fail(BDG_EINVAL, "the sweep kernel has 1, 2 or 4 lanes per site, not %d", lanes);
    const size_t table = (size_t)sys->n_unique * mode.stride * sizeof(double2);
    if (table > kDictLdsLimit) return fail(BDG_EINVAL, "block table too large for the sweep kernel");
    const size_t rows = (size_t)bdg::kSweepWaves * depth * bdg::kWave * 4 * sizeof(double2);
    plan->lds_bytes = table + rows;
    if (plan->lds_bytes > 64 * 1024)
        for (SweepKernel k : {plan->kernel, plan->kernel_reverse, plan->kernel_gen})
            if (k) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)plan->lds_bytes));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(plan->kernel),
                                                         bdg::kSweepThreads, plan->lds_bytes));
    per_cu = std::max(1, std::min(per_cu, 2 * bdg::kWavesPerBlock / bdg::kSweepWaves));
    if (const char* cap = getenv("BODGE_AMD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(cap));
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    bdg::SweepArgs& a = plan->args;
    a = bdg::SweepArgs{};
    a.stencil = sys->stencil.ptr;
    if (int rc = ensure_dict_table(sys, mode, &a.dict_table)) return rc;
    a.n_unique = sys->n_unique;
    a.nb = (int)sys->nb;
    a.plane = (int)plane;
    a.lx = sys->shape[0];
    const int owned = depth == 3 ? bdg::sweep3_owned(lanes) : bdg::sweep_owned(lanes);
    a.n_cols = (int)((plane + owned - 1) / owned);
    // one unit (segment x window) per resident wave, segments of at least 8 planes
    const int waves = per_cu * sys->num_cus * bdg::kSweepWaves;
    int n_segs = choose_segments(a.n_cols, a.lx, waves, 2 * depth, 8);
    if (const char* env = getenv("BODGE_AMD_SWEEP_SEGMENTS")) n_segs = atoi(env);
    a.n_segs = std::max(1, std::min(n_segs, a.lx / 8));
    a.zigzag = 1;
    if (const char* env = getenv("BODGE_AMD_SWEEP_ZIGZAG")) a.zigzag = atoi(env) != 0;
    a.wrap_p = sys->stencil_wrap_p ? 1 : 0;
    a.wrap_x = sys->stencil_wrap_x ? 1 : 0;
    const int64_t units = (int64_t)a.n_cols * a.n_segs;
    const int grid = (int)std::min<int64_t>((int64_t)per_cu * sys->num_cus, (units + bdg::kSweepWaves - 1) / bdg::kSweepWaves);
    plan->grid = std::max(8, (grid + 7) / 8 * 8);
    return BDG_OK;
}

// Algorithmic HBM bytes of one two-step sweep: one 8-byte stencil word per site, the block table
// once, and four passes over 4 x RL 16-byte payloads per site (read t_n, t_{n-1}; write t_{n+1},
// t_{n+2}).  The halo slots and segment-end planes the waves recompute are NOT counted.
double sweep_bytes(const bdg_system* sys, const ModeInfo& mode, int lanes) {
    return 8.0 * (double)sys->nb + (mode.block_bytes - 4.0) * sys->n_unique +
           4.0 * (4.0 * lanes * sizeof(double2)) * (double)sys->nb;
}

// Lanes per site (vectors per launch) of the sweep kernel.  4 lanes move the fewest redundant
// bytes per vector-step; with 1 lane (2 real / 1 complex vector per launch) the four buffers of a
// run are a quarter the size, and when they then fit the 256 MB Infinity Cache together
// (4 x 64 B x sites + the stencil words <= ~252 MB: up to ~10^6 sites) every launch after the
// first streams from that cache instead of HBM.  BODGE_AMD_SWEEP_LANES overrides.
// Steps per sweep: 3 (cheb_sweep3, 4 lanes per site only) moves 4/9 of the one-step kernels'
// bytes against 2/3 for 2.  BODGE_AMD_SWEEP_STEPS=2|3 overrides.
int sweep_depth_for(int lanes) {
    int depth = lanes >= 2 ? 3 : 2;
    if (const char* env = getenv("BODGE_AMD_SWEEP_STEPS")) {
        const int forced = atoi(env);
        if (forced == 2 || (forced == 3 && lanes >= 2)) depth = forced;
    }
    return depth;
}

// Default lanes per site: 2.  Fewer lanes mean wider windows (the 3-step kernel owns 26 of 32
// slots with 2 lanes, 10 of 16 with 4: less recomputed halo per useful site) at the price of
// shorter x-segments.  Measured on 1000x1000, 8 real vectors: 2 lanes 106.6 k vector-steps/s,
// 4 lanes 99.9 k (profiles/r02_sweep_experiments.log).
int sweep_lanes_for(const bdg_system* sys, int n_active, int per_lane) {
    if (const char* env = getenv("BODGE_AMD_SWEEP_LANES")) {
        const int forced = atoi(env);
        if (forced == 1 || forced == 2 || forced == 4) return forced;
    }
    (void)n_active;
    (void)per_lane;
    return sys->nb >= kSweepTwoLaneSites ? 2 : 4;  // small lattices: more work per launch matters more
}

// ---- 3-D: one step per launch with the x-neighbours in registers (cheb_roll3)
using RollKernel = void (*)(bdg::RollArgs);

RollKernel roll_kernel(const ModeInfo& mode) {
    switch (mode.id) {
        case 1: return bdg::cheb_roll3<RealMode>;
        case 2: return bdg::cheb_roll3<ComplexPHMode>;
        case 3: return bdg::cheb_roll3<RealPHMode>;
    }
    return bdg::cheb_roll3<ComplexMode>;
}

struct RollPlan {
    RollKernel kernel = nullptr;
    int grid = 0;
    size_t lds_bytes = 0;
    bdg::RollArgs args{};
};

int make_roll_plan(bdg_system* sys, const ModeInfo& mode, RollPlan* plan) {
    plan->kernel = roll_kernel(mode);
    const size_t table = (size_t)sys->n_unique * mode.stride * sizeof(double2);
    if (table > kDictLdsLimit) return fail(BDG_EINVAL, "block table too large for the rolling kernel");
    plan->lds_bytes = table + (size_t)bdg::kWavesPerBlock * bdg::kWave * 4 * sizeof(double2);
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(plan->kernel),
                                                         bdg::kBlockThreads, plan->lds_bytes));
    per_cu = std::max(1, std::min(per_cu, 2));
    if (const char* cap = getenv("BODGE_AMD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(cap));
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    bdg::RollArgs& a = plan->args;
    a = bdg::RollArgs{};
    a.stencil = sys->stencil.ptr;
    if (int rc = ensure_dict_table(sys, mode, &a.dict_table)) return rc;
    a.n_unique = sys->n_unique;
    a.nb = (int)sys->nb;
    a.plane = (int)plane;
    a.lz = sys->shape[2];
    a.lx = sys->shape[0];
    a.n_cols = (int)((plane + bdg::kRollOwned - 1) / bdg::kRollOwned);
    const int waves = per_cu * sys->num_cus * bdg::kWavesPerBlock;
    int n_segs = choose_segments(a.n_cols, a.lx, waves, 2, 4);
    if (const char* env = getenv("BODGE_AMD_SWEEP_SEGMENTS")) n_segs = atoi(env);
    a.n_segs = std::max(1, std::min(n_segs, a.lx / 4));
    const int64_t units = (int64_t)a.n_cols * a.n_segs;
    const int grid = (int)std::min<int64_t>((int64_t)per_cu * sys->num_cus,
                                            (units + bdg::kWavesPerBlock - 1) / bdg::kWavesPerBlock);
    plan->grid = std::max(8, (grid + 7) / 8 * 8);
    return BDG_OK;
}

// Algorithmic bytes of one launch of the rolling kernel: stencil word + three passes per site.
double roll_bytes(const bdg_system* sys, const ModeInfo& mode) {
    return 8.0 * (double)sys->nb + (mode.block_bytes - 4.0) * sys->n_unique +
           3.0 * (4.0 * bdg::kSweepLanes * sizeof(double2)) * (double)sys->nb;
}

enum class StartKind { Random, Unit };

struct StartSpec {
    StartKind kind;
    uint64_t seed = 0, first_id = 0;
    int vec_kind = 0;
    const int64_t* rows = nullptr;  // host
};

void dots_to_moments(const double* d, const double* e, int n_steps, int n_vectors, double* mu) {
    // mu[m][r]; mu_2n = 2 d_n - mu_0, mu_2n+1 = 2 e_n - mu_1
    for (int n = 0; n < n_steps; ++n)
        for (int r = 0; r < n_vectors; ++r) {
            const double d0 = d[r], e0 = e[r];
            const double dn = d[(size_t)n * n_vectors + r], en = e[(size_t)n * n_vectors + r];
            mu[(size_t)(2 * n) * n_vectors + r] = n == 0 ? d0 : 2.0 * dn - d0;
            mu[(size_t)(2 * n + 1) * n_vectors + r] = n == 0 ? e0 : 2.0 * en - e0;
        }
}

// ------------------------------------------------------------- lazy libraries
struct SolverApi {
    void* blas = nullptr;
    void* solver = nullptr;
    decltype(&rocblas_create_handle) create_handle = nullptr;
    decltype(&rocblas_destroy_handle) destroy_handle = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocsolver_zheevd) zheevd = nullptr;
    decltype(&rocsolver_zheev) zheev = nullptr;
    decltype(&rocsolver_zheevj) zheevj = nullptr;
    decltype(&rocsolver_dsyevd) dsyevd = nullptr;
    decltype(&rocsolver_dsyevj) dsyevj = nullptr;
};

// Reading a shared object through the page cache before dlopen.  librocsolver.so is 931 MB; on a
// machine whose page cache does not hold it yet, dlopen + first use fault it in a few KB at a
// time in link order.  Measured on fresh boxes (profiles/r02_rocsolver_cold.log): 1.5-9 minutes
// whichever way the bytes are asked for - the lazily provisioned root disk delivers ~2-3 MB/s
// for data nobody has touched, sequential or not, and parallel readers only slow it down - and
// well under a second once cached.  So the cost cannot be removed, only moved: `SolverPrefetch`
// streams the files on a background thread (bdg_dense_prefetch) so that the read overlaps with
// assembly, upload and whatever else the caller does before the first dense eigensolve above
// 4N = 2048; load_solver() waits for it.  Pure I/O: no symbol is used from the files.
void warm_page_cache(const char* path) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return;
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    std::vector<char> chunk((size_t)8 << 20);
    while (read(fd, chunk.data(), chunk.size()) > 0) {
    }
    close(fd);
}

struct FilePrefetch {
    std::vector<const char*> paths;
    FilePrefetch* after = nullptr;  // read only once that one is done (the disk serves one stream best)
    std::mutex lock;
    std::condition_variable changed;
    bool started = false, done = false, reported = false;
    double seconds = 0.0;
    FilePrefetch(std::vector<const char*> files, FilePrefetch* first) : paths(std::move(files)), after(first) {}
    void start() {
        std::lock_guard<std::mutex> guard(lock);
        if (started) return;
        started = true;
        if (getenv("BODGE_AMD_NO_PREFETCH")) {
            done = true;
            return;
        }
        // detached: a process that ends before the read has finished must not wait for it
        // (the objects themselves are never destroyed, see below)
        std::thread([this] {
            if (after && after->is_started()) (void)after->wait(-1.0);
            const auto t0 = std::chrono::steady_clock::now();
            for (const char* path : paths) warm_page_cache(path);
            std::lock_guard<std::mutex> inner(lock);
            seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            done = true;
            changed.notify_all();
        }).detach();
    }
    bool is_started() {
        std::lock_guard<std::mutex> guard(lock);
        return started;
    }
    // true once the files have been read; waits at most `timeout_s` (negative: no limit)
    bool wait(double timeout_s) {
        start();
        std::unique_lock<std::mutex> guard(lock);
        if (timeout_s < 0) changed.wait(guard, [this] { return done; });
        else changed.wait_for(guard, std::chrono::duration<double>(timeout_s), [this] { return done; });
        if (done && !reported && getenv("BODGE_AMD_TRACE")) {
            reported = true;
            fprintf(stderr, "[bdg] %s%s read in %.1f s\n", paths[0], paths.size() > 1 ? " ..." : "", seconds);
        }
        return done;
    }
};
// deliberately immortal: they outlive every exit path.  One stream at a time - side by side the two
// reads take as long as one after the other (the cold storage delivers ~2.5 MB/s in total) - and the
// dense-solver objects first when both are wanted: a diagonalize() call is waiting for those.
FilePrefetch& g_solver_prefetch =
    *new FilePrefetch({"/opt/rocm/lib/librocblas.so", "/opt/rocm/lib/librocsolver.so"}, nullptr);
FilePrefetch& g_rccl_prefetch = *new FilePrefetch({"/opt/rocm/lib/librccl.so"}, &g_solver_prefetch);

int load_solver(SolverApi** out) {
    static SolverApi api;
    static bool tried = false, ok = false;
    if (!tried) {
        tried = true;
        g_solver_prefetch.wait(-1.0);
        api.blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        if (!api.blas) api.blas = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        api.solver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        if (!api.solver) api.solver = dlopen("/opt/rocm/lib/librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        if (api.blas && api.solver) {
            api.create_handle =
                reinterpret_cast<decltype(api.create_handle)>(dlsym(api.blas, "rocblas_create_handle"));
            api.destroy_handle =
                reinterpret_cast<decltype(api.destroy_handle)>(dlsym(api.blas, "rocblas_destroy_handle"));
            api.set_stream =
                reinterpret_cast<decltype(api.set_stream)>(dlsym(api.blas, "rocblas_set_stream"));
            api.zheevd = reinterpret_cast<decltype(api.zheevd)>(dlsym(api.solver, "rocsolver_zheevd"));
            api.zheev = reinterpret_cast<decltype(api.zheev)>(dlsym(api.solver, "rocsolver_zheev"));
            api.zheevj = reinterpret_cast<decltype(api.zheevj)>(dlsym(api.solver, "rocsolver_zheevj"));
            api.dsyevd = reinterpret_cast<decltype(api.dsyevd)>(dlsym(api.solver, "rocsolver_dsyevd"));
            api.dsyevj = reinterpret_cast<decltype(api.dsyevj)>(dlsym(api.solver, "rocsolver_dsyevj"));
            ok = api.create_handle && api.destroy_handle && api.set_stream && api.zheevd && api.zheev &&
                 api.zheevj && api.dsyevd && api.dsyevj;
        }
    }
    if (!ok) return fail(BDG_ELIBRARY, "rocSOLVER/rocBLAS could not be loaded: %s", dlerror());
    *out = &api;
    return BDG_OK;
}

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
};

int load_rccl(RcclApi** out) {
    static RcclApi api;
    static bool tried = false, ok = false;
    if (!tried) {
        tried = true;
        (void)g_rccl_prefetch.wait(-1.0);  // 573 MB: streamed in before dlopen faults it in piecemeal
        api.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) api.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) {
            api.get_unique_id =
                reinterpret_cast<decltype(api.get_unique_id)>(dlsym(api.lib, "ncclGetUniqueId"));
            api.comm_init_rank =
                reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(api.lib, "ncclCommInitRank"));
            api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(api.lib, "ncclAllReduce"));
            api.comm_destroy =
                reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.lib, "ncclCommDestroy"));
            api.error_string =
                reinterpret_cast<decltype(api.error_string)>(dlsym(api.lib, "ncclGetErrorString"));
            api.send = reinterpret_cast<decltype(api.send)>(dlsym(api.lib, "ncclSend"));
            api.recv = reinterpret_cast<decltype(api.recv)>(dlsym(api.lib, "ncclRecv"));
            api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(api.lib, "ncclGroupStart"));
            api.group_end = reinterpret_cast<decltype(api.group_end)>(dlsym(api.lib, "ncclGroupEnd"));
            api.comm_count = reinterpret_cast<decltype(api.comm_count)>(dlsym(api.lib, "ncclCommCount"));
            ok = api.get_unique_id && api.comm_init_rank && api.all_reduce && api.comm_destroy &&
                 api.error_string && api.send && api.recv && api.group_start && api.group_end && api.comm_count;
        }
    }
    if (!ok) return fail(BDG_ELIBRARY, "RCCL could not be loaded: %s", dlerror());
    *out = &api;
    return BDG_OK;
}

#define NCCL_TRY(api, expr)                                                                  \
    do {                                                                                     \
        ncclResult_t res__ = (expr);                                                         \
        if (res__ != ncclSuccess)                                                            \
            return fail(BDG_ELIBRARY, "%s failed: %s", #expr, (api)->error_string(res__));   \
    } while (0)

int comm_allreduce(bdg_comm* comm, double* buf, int64_t count, ncclRedOp_t op) {
    if (!comm || !buf || count < 0) return fail(BDG_EINVAL, "bad all-reduce arguments");
    RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    HIP_TRY(hipSetDevice(comm->device));
    if (int rc = comm->scratch.reserve((size_t)count)) return rc;
    HIP_TRY(hipMemcpyAsync(comm->scratch.ptr, buf, sizeof(double) * count, hipMemcpyHostToDevice,
                           comm->stream));
    NCCL_TRY(api, api->all_reduce(comm->scratch.ptr, comm->scratch.ptr, (size_t)count, ncclDouble,
                                  op, comm->comm, comm->stream));
    HIP_TRY(hipMemcpyAsync(buf, comm->scratch.ptr, sizeof(double) * count, hipMemcpyDeviceToHost,
                           comm->stream));
    HIP_TRY(hipStreamSynchronize(comm->stream));
    return BDG_OK;
}


// ------------------------------------------------------------------ recurrence
// One batch = up to 64 start vectors advanced together on one handle.  The three
// phases are separate so that a group of slabs can be driven in lock step:
//   begin()  choose kernel + mode, allocate, write t_0 (own rows) and zero t_{-1}
//   step(n)  one launch of K1 (after the caller has refreshed the halo of t_n)
//   finish() reduce partials (done per chunk inside step), copy dots to the host
struct Batch {
    bdg_system* sys = nullptr;
    StepPlan plan;
    bdg::StepArgs args{};
    bool real = false;
    bool alternate = false;  // dictionary kernel: sweep direction flips every launch
    // unit start vectors: block rows that can be non-zero after n steps are within
    // (n + 1) * bandwidth of [band_lo, band_hi]; -1 = no band (random vectors, slabs, strip order)
    int64_t band_lo = -1, band_hi = -1;
    ModeInfo mode{};
    int rl = 0, rv = 0, n_active = 0, n_steps = 0, chunk = 1, strip_rows = 0;
    size_t width = 0, per_step = 0, vec_count = 0;
    double scale = 1.0;
    double2* cur = nullptr;
    double2* prev = nullptr;
    float kernel_ms = 0.f;
    int n_chunks = 0;
    // two-steps-per-sweep form
    bool sweep = false, roll = false;
    bool gen_start = false;  // the first sweep makes the random start block itself (no fill kernel)
    // A call cut into several batches enqueues them back to back and waits once: batch `slot` of
    // `n_slots` has its own timing events and its own piece of the pinned result buffer.
    int slot = 0, n_slots = 1, ev_base = 0;
    size_t host_stride = 0;
    SweepPlan splan;
    RollPlan rplan;
    double2 *spare1 = nullptr, *spare2 = nullptr;
    int launch_grid = 0;   // workgroups whose dot partials one recurrence step leaves behind
    int n_launches = 0;

    int begin(bdg_system* system, double scale_in, int steps, int active, const StartSpec& start,
              int force_real /* -1 auto, 0 complex, 1 real */, bool col_scalars = false) {
        sys = system;
        scale = scale_in;
        n_steps = steps;
        n_active = active;
        HIP_TRY(hipSetDevice(sys->device));
        // Real arithmetic applies when H has no imaginary part and the start vectors are real
        // (±1 or unit vectors): every t_n then stays real.  BODGE_AMD_REAL=0 forces complex.
        const bool start_is_real = start.kind == StartKind::Unit || start.vec_kind == BDG_VEC_RADEMACHER;
        const char* real_env = getenv("BODGE_AMD_REAL");
        const bool matrix_real = sys->slab_comm ? sys->slab_all_real : sys->is_real;  // slabs: agreed over all ranks
        real = matrix_real && start_is_real && !(real_env && real_env[0] == '0');
        if (force_real >= 0) real = force_real != 0;
        const char* ph_env = getenv("BODGE_AMD_PH");
        mode = mode_info(real, sys->is_ph && !(ph_env && ph_env[0] == '0'));
        const int per_lane = mode.per_lane;
        // Fewer than 4 lanes per row would put 32-64 rows' blocks into one wave's LDS
        // region with no reuse; small batches run with zero-padded columns.
        rl = std::max(4, next_pow2((n_active + per_lane - 1) / per_lane));
        if (sys->lanes_override * per_lane >= n_active && sys->lanes_override >= 4 &&
            sys->lanes_override * per_lane <= 64)
            rl = sys->lanes_override;
        // lattice-stencil kernels (sweep.hpp): K7 runs with 4, 2 or 1 lanes per site, K8 with 4
        sweep = roll = false;
        int stencil_kind = 0;
        if (sys->lanes_override == 0 && rl == 4)
            if (int rc = sweep_wanted(sys, start.kind == StartKind::Random, col_scalars, &stencil_kind)) return rc;
        if (stencil_kind == 1) {
            const int lanes = sweep_lanes_for(sys, n_active, per_lane);
            if (n_active <= lanes * per_lane) {
                sweep = true;
                rl = lanes;
            }
        } else if (stencil_kind == 2) {
            roll = true;
        }
        rv = rl * per_lane;  // vector columns in the buffers
        if (sys->slab_comm && sys->slab_comm->n_ranks > 1) {
            // the halo messages are 4 * rl payloads per row: a rank with another rl would hang or mis-unpack
            double probe[2] = {(double)(rl * 2 + (real ? 1 : 0)), -(double)(rl * 2 + (real ? 1 : 0))};
            if (int rc = comm_allreduce(sys->slab_comm, probe, 2, ncclMax)) return rc;
            if (probe[0] != -probe[1])
                return fail(BDG_EINVAL, "slab ranks chose different kernel configurations (lanes x mode %d here)",
                            rl * 2 + (real ? 1 : 0));
        }
        if (sweep) {
            // (no one-step plan: the generic kernels start at 4 lanes per row; the odd last step of a
            // run goes through the sweep kernel with its second step switched off)
            plan = StepPlan{};
            plan.rl = rl;
            plan.mode = mode;
            plan.dictionary = true;
            args = bdg::StepArgs{};
            if (int rc = make_sweep_plan(sys, mode, rl, sweep_depth_for(rl), &splan)) return rc;
        } else {
            if (int rc = make_plan(sys, rl, mode, &plan, col_scalars)) return rc;
            if (int rc = matrix_args(sys, plan, &args)) return rc;
            if (roll && !plan.dictionary) roll = false;
            if (roll)
                if (int rc = make_roll_plan(sys, mode, &rplan)) return rc;
        }
        launch_grid = sweep ? splan.grid : roll ? rplan.grid : plan.grid;
        n_launches = 0;

        vec_count = (size_t)4 * sys->ncols * rl;  // 16-byte lane payloads
        // t_n and t_{n-1} together beyond the 256 MB Infinity Cache: the write of t_{n+1} and the
        // read of t_{n-1} are hinted non-temporal (+6 % at 10^6 sites x 8 vectors); smaller buffers
        // stay resident from one launch to the next and are faster with plain accesses
        // (profiles/r01_stream_probe.log)
        args.stream_vectors = 2 * vec_count * sizeof(double2) > kStreamVectorBytes ? 3 : 0;
        if (const char* env = std::getenv("BODGE_AMD_STREAM_VECTORS")) args.stream_vectors = std::atoi(env);
        alternate = true;
        if (const char* env = std::getenv("BODGE_AMD_ALTERNATE")) alternate = std::atoi(env) != 0;
        if (int rc = sys->vec_a.reserve(vec_count)) return rc;
        if (int rc = sys->vec_b.reserve(vec_count)) return rc;
        if (sweep) {
            if (int rc = sys->vec_c.reserve(vec_count)) return rc;
            if (int rc = sys->vec_d.reserve(vec_count)) return rc;
            spare1 = sys->vec_c.ptr;
            spare2 = sys->vec_d.ptr;
            splan.args.stream = 2 * vec_count * sizeof(double2) > kStreamVectorBytes ? 1 : 0;
            if (const char* env = std::getenv("BODGE_AMD_SWEEP_STREAM")) splan.args.stream = std::atoi(env);
        }
        width = (size_t)2 * rv;
        if (int rc = prepare_overlap()) return rc;
        // Dot partials are reduced every `chunk` launches.  Buffer sizes do not depend on
        // n_steps (up to 1024), so a short warm-up call leaves nothing to allocate later.
        per_step = (size_t)(overlapped ? grid_interior + grid_boundary : launch_grid) * width;
        constexpr int kChunk = 64;
        chunk = std::min(n_steps, sweep && splan.depth == 3 ? 63 : kChunk);  // a sweep must not straddle two chunks
        if (int rc = sys->partial.reserve((size_t)kChunk * per_step)) return rc;
        const size_t dots_count = (size_t)std::max(n_steps, 1024) * width;
        if (int rc = sys->dots.reserve(dots_count)) return rc;
        host_stride = dots_count;
        ev_base = slot * ((n_steps + chunk - 1) / chunk);
        if (sys->host_dots_count < dots_count * n_slots) {
            HIP_TRY(hipStreamSynchronize(sys->stream));  // (an earlier batch of this call may still be copying into it)
            if (sys->host_dots) (void)hipHostFree(sys->host_dots);
            sys->host_dots = nullptr;
            sys->host_dots_count = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sys->host_dots), dots_count * n_slots * sizeof(double), 0));
            sys->host_dots_count = dots_count * n_slots;
        }
        if (sys->send_total > 0)
            if (int rc = sys->send_buf.reserve((size_t)sys->send_total * 4 * rl)) return rc;
        if (sys->recv_total > 0)
            if (int rc = sys->recv_buf.reserve((size_t)sys->recv_total * 4 * rl)) return rc;

        hipStream_t st = sys->stream;
        const int fill_grid = (int)std::min<size_t>(4096, (vec_count + 255) / 256);
        // Three-step sweeps make a random t_0 in registers during the first sweep (cheb_sweep3 GEN):
        // no fill kernel, and vec_a is only ever a spare buffer.  BODGE_AMD_SWEEP_GEN=0: fill and read.
        gen_start = sweep && splan.depth == 3 && splan.kernel_gen && start.kind == StartKind::Random &&
                    sys->row_offset == 0 && sys->ncols == sys->nb;
        if (const char* env = getenv("BODGE_AMD_SWEEP_GEN")) gen_start = gen_start && atoi(env) != 0;
        if (gen_start) {
            splan.args.gen_seed = start.seed;
            splan.args.gen_first_id = start.first_id;
            splan.args.gen_kind = real ? BDG_VEC_RADEMACHER : start.vec_kind;
            splan.args.gen_active = n_active;
        } else if (start.kind == StartKind::Random) {
            if (real)
                bdg::fill_random_real<<<fill_grid, 256, 0, st>>>(
                    reinterpret_cast<double*>(sys->vec_a.ptr), sys->nb, sys->ncols, rv, n_active,
                    start.seed, start.first_id, sys->row_offset);
            else
                bdg::fill_random<<<fill_grid, 256, 0, st>>>(sys->vec_a.ptr, sys->nb, sys->ncols, rv,
                                                            n_active, start.seed, start.first_id,
                                                            start.vec_kind, sys->row_offset);
        } else {
            if (int rc = sys->rows.reserve(64)) return rc;
            HIP_TRY(hipMemcpyAsync(sys->rows.ptr, start.rows, sizeof(int64_t) * n_active,
                                   hipMemcpyHostToDevice, st));
            if (sys->ncols == sys->nb && !getenv("BODGE_AMD_NO_BAND")) {
                band_lo = sys->nb;
                band_hi = 0;
                for (int r = 0; r < n_active; ++r) {
                    band_lo = std::min<int64_t>(band_lo, start.rows[r] >> 2);
                    band_hi = std::max<int64_t>(band_hi, start.rows[r] >> 2);
                }
            }
            bdg::fill_zero<<<fill_grid, 256, 0, st>>>(sys->vec_a.ptr, (int64_t)vec_count);
            if (real)
                bdg::set_unit_real<<<1, 64, 0, st>>>(reinterpret_cast<double*>(sys->vec_a.ptr), sys->nb,
                                                     sys->ncols, rv, n_active, sys->rows.ptr,
                                                     sys->row_offset);
            else
                bdg::set_unit<<<1, 64, 0, st>>>(sys->vec_a.ptr, sys->nb, sys->ncols, rv, n_active,
                                                sys->rows.ptr, sys->row_offset);
        }
        if (!sweep)  // (the sweep kernels are told that t_{-1} = 0 instead of reading 256 MB of zeros)
            bdg::fill_zero<<<fill_grid, 256, 0, st>>>(sys->vec_b.ptr, (int64_t)vec_count);
        HIP_TRY(hipGetLastError());

        // bytes of t_n per block row that neighbouring rows re-read: 4 entries per vector
        if (!sweep)
            if (int rc = prepare_tile_order(sys, plan.rows_per_tile, plan.n_tiles,
                                            (real ? 32.0 : 64.0) * rv, &args.tile_order, &strip_rows))
                return rc;
        if (args.tile_order) band_lo = band_hi = -1;  // the band is a range of naturally ordered tiles
        if (sweep || roll) {
            strip_rows = 0;
            band_lo = band_hi = -1;
        }
        cur = sys->vec_a.ptr;
        prev = sys->vec_b.ptr;
        kernel_ms = 0.f;
        n_chunks = 0;
        return BDG_OK;
    }

    // Halo exchange, split so that a same-process group can interleave its members.
    int pack(hipStream_t st = nullptr) {
        if (sys->send_total == 0) return BDG_OK;
        if (!st) st = sys->stream;
        HIP_TRY(hipSetDevice(sys->device));
        const int64_t total = sys->send_total * 4 * rl;
        bdg::halo_pack<<<(unsigned)std::min<int64_t>(2048, (total + 255) / 256), 256, 0, st>>>(
            cur, sys->send_rows.ptr, sys->send_total, sys->ncols, rl, sys->send_buf.ptr);
        HIP_TRY(hipGetLastError());
        return BDG_OK;
    }
    int unpack(hipStream_t st = nullptr) {
        if (!st) st = sys->stream;
        HIP_TRY(hipSetDevice(sys->device));
        for (const ExchangePeer& peer : sys->peers) {
            if (peer.recv_count == 0) continue;
            const int64_t total = peer.recv_count * 4 * rl;
            bdg::halo_unpack<<<(unsigned)std::min<int64_t>(2048, (total + 255) / 256), 256, 0, st>>>(
                cur, peer.recv_col, peer.recv_count, sys->ncols, rl,
                sys->recv_buf.ptr + (size_t)peer.recv_begin * 4 * rl);
        }
        HIP_TRY(hipGetLastError());
        return BDG_OK;
    }

    // ---- overlap: rows that read no halo column ("interior") do not have to wait for the
    // exchange.  The workgroup tiles are split into two lists; per launch of the recurrence
    //   comm stream   : wait(previous step) -> pack -> ncclSend/Recv -> unpack -> ev_halo_ready
    //   compute stream: K1(interior tiles) -> wait(ev_halo_ready) -> K1(boundary tiles) -> ev_step_done
    // so the transfer hides behind the interior launch.  Hazards: the exchange only reads owned
    // rows of t_n and writes halo rows of the same buffer, which nothing but the boundary launch of
    // this step reads; both launches write owned rows of the other buffer.
    int grid_interior = 0, grid_boundary = 0;
    bool overlapped = false;

    int prepare_overlap() {
        overlapped = false;
        const char* env = getenv("BODGE_AMD_OVERLAP");
        if (sys->peers.empty() || !sys->slab_comm || sys->row_needs_halo.empty() || (env && env[0] == '0'))
            return BDG_OK;
        if (sys->split_rows_per_tile != plan.rows_per_tile) {
            std::vector<int> interior, boundary;
            for (int t = 0; t < plan.n_tiles; ++t) {
                bool needs = false;
                const int64_t r0 = (int64_t)t * plan.rows_per_tile;
                for (int64_t i = r0; i < std::min<int64_t>(sys->nb, r0 + plan.rows_per_tile); ++i)
                    needs = needs || sys->row_needs_halo[(size_t)i];
                (needs ? boundary : interior).push_back(t);
            }
            if (int rc = sys->tiles_interior.reserve(std::max<size_t>(1, interior.size()))) return rc;
            if (int rc = sys->tiles_boundary.reserve(std::max<size_t>(1, boundary.size()))) return rc;
            if (!interior.empty())
                HIP_TRY(hipMemcpy(sys->tiles_interior.ptr, interior.data(), sizeof(int) * interior.size(),
                                  hipMemcpyHostToDevice));
            if (!boundary.empty())
                HIP_TRY(hipMemcpy(sys->tiles_boundary.ptr, boundary.data(), sizeof(int) * boundary.size(),
                                  hipMemcpyHostToDevice));
            sys->n_interior = (int)interior.size();
            sys->n_boundary = (int)boundary.size();
            sys->split_rows_per_tile = plan.rows_per_tile;
        }
        if (sys->n_interior == 0 || sys->n_boundary == 0) return BDG_OK;  // nothing to hide behind
        if (!sys->comm_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&sys->comm_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&sys->ev_step_done, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sys->ev_halo_ready, hipEventDisableTiming));
        }
        auto grid_for = [&](int tiles) { return std::max(8, (std::min(tiles, plan.grid) + 7) / 8 * 8); };
        grid_interior = grid_for(sys->n_interior);
        grid_boundary = grid_for(sys->n_boundary);
        overlapped = true;
        return BDG_OK;
    }

    int rccl_transfer(hipStream_t st) {
        bdg_comm* comm = sys->slab_comm;
        RcclApi* api = nullptr;
        if (int rc = load_rccl(&api)) return rc;
        NCCL_TRY(api, api->group_start());
        for (const ExchangePeer& peer : sys->peers) {
            const size_t unit = (size_t)4 * rl * 2;  // doubles per exchanged block row
            if (peer.send_count > 0)
                NCCL_TRY(api, api->send(sys->send_buf.ptr + (size_t)peer.send_begin * 4 * rl,
                                        (size_t)peer.send_count * unit, ncclDouble, peer.rank, comm->comm, st));
            if (peer.recv_count > 0)
                NCCL_TRY(api, api->recv(sys->recv_buf.ptr + (size_t)peer.recv_begin * 4 * rl,
                                        (size_t)peer.recv_count * unit, ncclDouble, peer.rank, comm->comm, st));
        }
        NCCL_TRY(api, api->group_end());
        return BDG_OK;
    }

    int step_overlapped(int n) {
        HIP_TRY(hipSetDevice(sys->device));
        hipStream_t st = sys->stream, cs = sys->comm_stream;
        const int in_chunk = n % chunk;
        const int chunk_id = n / chunk;
        while ((int)sys->ev_pool.size() < 2 * (ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            sys->ev_pool.push_back(ev);
        }
        // exchange of t_n on the communication stream, after everything that produced t_n
        HIP_TRY(hipEventRecord(sys->ev_step_done, st));
        HIP_TRY(hipStreamWaitEvent(cs, sys->ev_step_done, 0));
        if (int rc = pack(cs)) return rc;
        if (int rc = rccl_transfer(cs)) return rc;
        if (int rc = unpack(cs)) return rc;
        HIP_TRY(hipEventRecord(sys->ev_halo_ready, cs));

        if (in_chunk == 0) HIP_TRY(hipEventRecord(sys->ev_pool[2 * (ev_base + chunk_id)], st));
        args.cur = cur;
        args.prev = prev;
        args.coef = (n == 0 ? 1.0 : 2.0) / scale;
        double* slot = sys->partial.ptr + (size_t)in_chunk * per_step;
        bdg::StepArgs part = args;
        part.tile_order = sys->tiles_interior.ptr;
        part.n_tiles = sys->n_interior;
        part.partial = slot;
        part.reverse = alternate ? (n & 1) : 0;  // same cache-aware sweep as the plain step
        plan.kernel<<<grid_interior, bdg::kBlockThreads, plan.lds_bytes, st>>>(part);
        HIP_TRY(hipStreamWaitEvent(st, sys->ev_halo_ready, 0));
        part.tile_order = sys->tiles_boundary.ptr;
        part.n_tiles = sys->n_boundary;
        part.partial = slot + (size_t)grid_interior * width;
        plan.kernel<<<grid_boundary, bdg::kBlockThreads, plan.lds_bytes, st>>>(part);
        std::swap(cur, prev);
        if (in_chunk == chunk - 1 || n == n_steps - 1) {
            const int s0 = n - in_chunk;
            HIP_TRY(hipEventRecord(sys->ev_pool[2 * (ev_base + chunk_id) + 1], st));
            bdg::reduce_partials<<<in_chunk + 1, 256, 0, st>>>(sys->partial.ptr, sys->dots.ptr + (size_t)s0 * width,
                                                              grid_interior + grid_boundary, (int)width);
            HIP_TRY(hipGetLastError());
            n_chunks = chunk_id + 1;
        }
        return BDG_OK;
    }

    // RCCL transport without overlap: grouped send/recv of the packed rows on the compute stream.
    int exchange_rccl() {
        if (sys->peers.empty()) return BDG_OK;
        bdg_comm* comm = sys->slab_comm;
        if (!comm) return fail(BDG_EINVAL, "slab handle has exchange peers but no communicator");
        RcclApi* api = nullptr;
        if (int rc = load_rccl(&api)) return rc;
        if (int rc = pack()) return rc;
        NCCL_TRY(api, api->group_start());
        for (const ExchangePeer& peer : sys->peers) {
            const size_t unit = (size_t)4 * rl * 2;  // doubles per exchanged block row
            if (peer.send_count > 0)
                NCCL_TRY(api, api->send(sys->send_buf.ptr + (size_t)peer.send_begin * 4 * rl,
                                        (size_t)peer.send_count * unit, ncclDouble, peer.rank, comm->comm,
                                        sys->stream));
            if (peer.recv_count > 0)
                NCCL_TRY(api, api->recv(sys->recv_buf.ptr + (size_t)peer.recv_begin * 4 * rl,
                                        (size_t)peer.recv_count * unit, ncclDouble, peer.rank, comm->comm,
                                        sys->stream));
        }
        NCCL_TRY(api, api->group_end());
        return unpack();
    }

    int step(int n) {
        HIP_TRY(hipSetDevice(sys->device));
        hipStream_t st = sys->stream;
        const int in_chunk = n % chunk;
        const int chunk_id = n / chunk;
        // one event pair per chunk, read back in finish(): the host never waits inside the loop
        while ((int)sys->ev_pool.size() < 2 * (ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            sys->ev_pool.push_back(ev);
        }
        if (in_chunk == 0) HIP_TRY(hipEventRecord(sys->ev_pool[2 * (ev_base + chunk_id)], st));
        args.cur = cur;
        args.prev = prev;
        args.coef = (n == 0 ? 1.0 : 2.0) / scale;
        args.partial = sys->partial.ptr + (size_t)in_chunk * per_step;
        args.reverse = alternate ? (n & 1) : 0;
        args.tile_base = 0;
        args.n_tiles = plan.n_tiles;
        if (band_lo >= 0) {
            // t_{n+1} can be non-zero only where t_n or a neighbour within the bandwidth was
            const int64_t reach = (int64_t)(n + 1) * sys->bandwidth;
            const int64_t lo = std::max<int64_t>(0, band_lo - reach);
            const int64_t hi = std::min<int64_t>(sys->nb, band_hi + reach + 1);
            const int first = (int)(lo / plan.rows_per_tile);
            const int last = (int)((hi + plan.rows_per_tile - 1) / plan.rows_per_tile);
            args.tile_base = first;
            args.n_tiles = std::min(plan.n_tiles, last) - first;
        }
        if (roll) {
            bdg::RollArgs& ra = rplan.args;
            ra.cur = cur;
            ra.prev = prev;
            ra.coef = args.coef;
            ra.partial = args.partial;
            ra.reverse = args.reverse;
            ra.stream = args.stream_vectors;
            rplan.kernel<<<rplan.grid, bdg::kBlockThreads, rplan.lds_bytes, st>>>(ra);
        } else {
            plan.kernel<<<plan.grid, bdg::kBlockThreads, plan.lds_bytes, st>>>(args);
        }
        std::swap(cur, prev);
        if (in_chunk == chunk - 1 || n == n_steps - 1) {
            const int s0 = n - in_chunk;
            HIP_TRY(hipEventRecord(sys->ev_pool[2 * (ev_base + chunk_id) + 1], st));
            bdg::reduce_partials<<<in_chunk + 1, 256, 0, st>>>(
                sys->partial.ptr, sys->dots.ptr + (size_t)s0 * width, launch_grid, (int)width);
            HIP_TRY(hipGetLastError());
            n_chunks = chunk_id + 1;
        }
        return BDG_OK;
    }

    // Steps n .. n + depth - 1 in one sweep (sweep.hpp); what is left at the end of a run in a
    // shorter one.  Returns the number of steps made.  Buffers rotate:
    // (t_n, t_{n-1}, spare, spare) -> (t_{n+k}, t_{n+k-1}, spare, spare).
    int step_sweep(int n, int* made) {
        HIP_TRY(hipSetDevice(sys->device));
        hipStream_t st = sys->stream;
        const int in_chunk = n % chunk;
        const int chunk_id = n / chunk;
        const int now = std::min({splan.depth, n_steps - n, chunk - in_chunk});
        *made = now;
        while ((int)sys->ev_pool.size() < 2 * (ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            sys->ev_pool.push_back(ev);
        }
        if (in_chunk == 0) HIP_TRY(hipEventRecord(sys->ev_pool[2 * (ev_base + chunk_id)], st));
        bdg::SweepArgs& a = splan.args;
        a.cur = cur;
        a.prev = n == 0 ? nullptr : prev;
        a.out1 = spare1;
        a.out2 = spare2;
        a.coef1 = (n == 0 ? 1.0 : 2.0) / scale;
        a.coef2 = 2.0 / scale;
        a.two = now >= 2 ? 1 : 0;
        a.steps = now;
        a.partial1 = sys->partial.ptr + (size_t)in_chunk * per_step;
        a.partial2 = a.partial1 + per_step;
        a.partial3 = a.partial2 + per_step;
        const SweepKernel kernel = n == 0 && gen_start ? splan.kernel_gen
                                   : alternate && (n_launches & 1) ? splan.kernel_reverse : splan.kernel;
        if (n == 0 && gen_start) a.cur = nullptr;  // (never read)
        kernel<<<splan.grid, bdg::kSweepThreads, splan.lds_bytes, st>>>(a);
        ++n_launches;
        double2* old_cur = cur;
        double2* old_prev = prev;
        if (splan.depth == 2 && now == 1) {  // cheb_sweep writes a lone step to out1
            cur = spare1;
            prev = old_cur;
            spare1 = old_prev;
        } else if (now == 1) {               // cheb_sweep3 writes the last level to out2, the one before to out1
            cur = spare2;
            prev = old_cur;
            spare2 = old_prev;
        } else {
            cur = spare2;
            prev = spare1;
            spare1 = old_prev;
            spare2 = old_cur;
        }
        const int last = n + now - 1;
        const int last_in_chunk = last % chunk;
        if (last_in_chunk == chunk - 1 || last == n_steps - 1) {
            const int s0 = last - last_in_chunk;
            HIP_TRY(hipEventRecord(sys->ev_pool[2 * (ev_base + chunk_id) + 1], st));
            bdg::reduce_partials<<<last_in_chunk + 1, 256, 0, st>>>(
                sys->partial.ptr, sys->dots.ptr + (size_t)s0 * width, launch_grid, (int)width);
            HIP_TRY(hipGetLastError());
            n_chunks = chunk_id + 1;
        }
        return BDG_OK;
    }

    // d/e of this handle's rows into columns [col0, col0 + n_active) of (n_steps x ld) arrays;
    // accumulate = true adds to what is there (summing the slabs of a group).
    int finish(double* d_out, double* e_out, int ld, int col0, bool accumulate, bool first_batch) {
        if (int rc = finish_enqueue()) return rc;
        HIP_TRY(hipStreamSynchronize(sys->stream));
        return finish_collect(d_out, e_out, ld, col0, accumulate, first_batch);
    }
    // copy of the batch's dot products to the host, enqueued behind its last reduction
    int finish_enqueue() {
        HIP_TRY(hipSetDevice(sys->device));
        // pinned: a pageable target costs ~8 ms on its first use
        HIP_TRY(hipMemcpyAsync(sys->host_dots + (size_t)slot * host_stride, sys->dots.ptr,
                               (size_t)n_steps * width * sizeof(double), hipMemcpyDeviceToHost, sys->stream));
        return BDG_OK;
    }
    // after the stream has been waited for
    int finish_collect(double* d_out, double* e_out, int ld, int col0, bool accumulate, bool first_batch) {
        HIP_TRY(hipSetDevice(sys->device));
        const double* host = sys->host_dots + (size_t)slot * host_stride;
        for (int c = 0; c < n_chunks; ++c) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, sys->ev_pool[2 * (ev_base + c)], sys->ev_pool[2 * (ev_base + c) + 1]));
            kernel_ms += ms;
        }
        for (int n = 0; n < n_steps; ++n)
            for (int r = 0; r < n_active; ++r) {
                double& d = d_out[(size_t)n * ld + col0 + r];
                double& e = e_out[(size_t)n * ld + col0 + r];
                const double dv = host[(size_t)n * width + 2 * r], ev = host[(size_t)n * width + 2 * r + 1];
                d = accumulate ? d + dv : dv;
                e = accumulate ? e + ev : ev;
            }
        bdg_perf& p = sys->perf;
        if (first_batch) p = bdg_perf{};
        p.kernel_ms += kernel_ms;
        p.launches += sweep ? n_launches : n_steps;
        p.vector_steps += (int64_t)n_steps * n_active;
        p.bytes_per_launch = sweep  ? sweep_bytes(sys, mode, rl)
                             : roll ? roll_bytes(sys, mode)
                                    : algorithmic_bytes(sys, rv, mode, plan.dictionary);
        p.steps_per_launch = sweep ? splan.depth : 1;
        p.rolling = roll ? 1 : 0;
        p.dict_skipped = sys->dict_skipped;
        p.lanes_per_row = rl;
        p.vectors_per_launch = rv;
        p.real_arithmetic = real ? 1 : 0;
        p.ph_packed = mode.ph ? 1 : 0;
        p.dict_blocks = plan.dictionary ? sys->n_unique : 0;
        p.strip_rows = strip_rows;
        p.grid = launch_grid;
        p.lds_bytes = (int32_t)(sweep ? splan.lds_bytes : roll ? rplan.lds_bytes : plan.lds_footprint);
        p.pipelined = plan.pipelined ? 1 : 0;
        return BDG_OK;
    }
};

int check_recurrence_args(const void* sys, double scale, int n_steps, int n_vectors, const double* d_out,
                          const double* e_out) {
    if (!sys) return fail(BDG_EINVAL, "null system handle");
    if (!(scale > 0.0)) return fail(BDG_EINVAL, "scale must be positive");
    if (n_steps < 1 || n_vectors < 1) return fail(BDG_EINVAL, "n_steps and n_vectors must be >= 1");
    if (!d_out || !e_out) return fail(BDG_EINVAL, "null output buffer");
    return BDG_OK;
}

StartSpec batch_start(const StartSpec& start, int col) {
    StartSpec batch = start;
    if (start.kind == StartKind::Random) batch.first_id = start.first_id + col;
    else batch.rows = start.rows + col;
    return batch;
}

// Vectors advanced together.  Wide batches amortise launch latency and the matrix stream, narrow
// ones keep t_n and t_{n-1} close to the caches: measured optimum (wall time per vector-step of
// 64 vectors, profiles/r01_batch_width.log) is ~2.5 M site-vectors per launch, i.e. 64 vectors up
// to 200x200 sites, 32 at 300x300, 16 at 400x400, 8 from 64^3 on (10^6 sites: 18.0 us per
// vector-step at 8 per batch, 21.3 us at 64).  Rule: the largest power of two that keeps one
// vector buffer within 96 MB, at least one full lane group (8 real / 4 complex), at most 64.
int batch_width(bdg_system* sys, const StartSpec& start, int n_vectors) {
    if (const char* env = getenv("BODGE_AMD_BATCH")) return std::clamp(atoi(env), 1, 64);
    const bool start_is_real = start.kind == StartKind::Unit || start.vec_kind == BDG_VEC_RADEMACHER;
    const char* real_env = getenv("BODGE_AMD_REAL");
    const bool real = (sys->slab_comm ? sys->slab_all_real : sys->is_real) && start_is_real &&
                      !(real_env && real_env[0] == '0');
    int stencil_kind = 0;
    if (sys->lanes_override == 0 && sweep_wanted(sys, start.kind == StartKind::Random, false, &stencil_kind) == BDG_OK &&
        stencil_kind != 0) {
        const int per_lane = real ? 2 : 1;
        const int lanes = stencil_kind == 1 ? sweep_lanes_for(sys, n_vectors, per_lane) : bdg::kSweepLanes;
        return std::min(lanes * per_lane, std::max(n_vectors, 1));  // one lane group per launch
    }
    // (slabs: the widest slab of the run decides, so that every rank cuts the same batches)
    const double per_vector = (double)std::max(sys->ncols, sys->slab_max_ncols) * 4 * (real ? 8.0 : 16.0);
    const int granule = 8;  // (the register-pipelined complex kernels start at 8 lanes per row)
    constexpr double kBufferTarget = 96.0 * 1024 * 1024;
    int width = 64;
    while (width > granule && width * per_vector > kBufferTarget) width >>= 1;
    return std::min(width, std::max(n_vectors, 1));
}

// Single handle (whole matrix, or one slab of a multi-process run with RCCL halos).
int run_recurrence(bdg_system* sys, double scale, int n_steps, int n_vectors, StartSpec start,
                   double* d_out, double* e_out) {
    if (int rc = check_recurrence_args(sys, scale, n_steps, n_vectors, d_out, e_out)) return rc;
    lanczos_free(sys);
    const bool trace = getenv("BODGE_AMD_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int width = batch_width(sys, start, n_vectors);
    // Batches are enqueued back to back and waited for once (whole matrices; a slab's batches are
    // paced by its halo exchange anyway): the GPU does not idle while the host turns a batch around.
    const int n_batches = (n_vectors + width - 1) / width;
    const size_t staged = (size_t)n_batches * std::max(n_steps, 1024) * 2 * (size_t)width;  // doubles of pinned memory
    const bool pipelined = n_batches > 1 && n_batches <= 64 && staged <= ((size_t)4 << 20) && sys->ncols == sys->nb &&
                           !getenv("BODGE_AMD_NO_BATCH_PIPELINE");
    std::vector<Batch> queued(pipelined ? (size_t)n_batches : 0);
    size_t stride0 = 0;
    for (int col = 0, index = 0; col < n_vectors; col += width, ++index) {
        Batch single;
        Batch& batch = pipelined ? queued[(size_t)index] : single;
        if (pipelined) {
            batch.slot = index;
            batch.n_slots = n_batches;
        }
        const auto t0 = now();
        if (int rc = batch.begin(sys, scale, n_steps, std::min(width, n_vectors - col),
                                 batch_start(start, col), -1))
            return rc;
        if (pipelined) {  // one spacing of the result pieces for the whole call: the first batch is the widest
            if (index == 0) stride0 = batch.host_stride;
            batch.host_stride = stride0;
        }
        const auto t1 = now();
        for (int n = 0; n < n_steps; ++n) {
            if (batch.sweep) {
                int made = 1;
                if (int rc = batch.step_sweep(n, &made)) return rc;
                n += made - 1;
                continue;
            }
            if (batch.overlapped) {
                if (int rc = batch.step_overlapped(n)) return rc;
                continue;
            }
            if (int rc = batch.exchange_rccl()) return rc;
            if (int rc = batch.step(n)) return rc;
        }
        const auto t2 = now();
        if (pipelined) {
            if (int rc = batch.finish_enqueue()) return rc;
        } else if (int rc = batch.finish(d_out, e_out, n_vectors, col, false, col == 0)) {
            return rc;
        }
        if (trace)
            fprintf(stderr, "[bdg] begin %.3f ms, steps %.3f ms (kernels %.3f), finish %.3f ms\n", ms(t0, t1),
                    ms(t1, t2), batch.kernel_ms, ms(t2, now()));
    }
    if (pipelined) {
        HIP_TRY(hipSetDevice(sys->device));
        HIP_TRY(hipStreamSynchronize(sys->stream));
        for (int index = 0; index < n_batches; ++index)
            if (int rc = queued[(size_t)index].finish_collect(d_out, e_out, n_vectors, index * width, false, index == 0))
                return rc;
    }
    return BDG_OK;
}

// Same-process group of slabs (one handle per slab, on one or several devices of this
// process): the halo rows travel by device-to-device copies ordered with events.
//   stream m:  [wait until my previous send buffer was consumed] pack -> ev packed[m]
//              for each peer p: wait packed[p]; copy p.send segment -> my recv segment
//              -> ev copied[m]; unpack; K1
int run_group(bdg_group* group, double scale, int n_steps, int n_vectors, StartSpec start,
              double* d_out, double* e_out) {
    if (int rc = check_recurrence_args(group, scale, n_steps, n_vectors, d_out, e_out)) return rc;
    const size_t n_members = group->members.size();
    bool all_real = true;
    for (bdg_system* m : group->members) {
        all_real = all_real && m->is_real;
        lanczos_free(m);
    }
    // (storage packing is per member: it changes what a member reads, not what it exchanges)
    const bool start_is_real = start.kind == StartKind::Unit || start.vec_kind == BDG_VEC_RADEMACHER;
    const char* real_env = getenv("BODGE_AMD_REAL");
    const int force_real = (all_real && start_is_real && !(real_env && real_env[0] == '0')) ? 1 : 0;

    for (int col = 0; col < n_vectors; col += 64) {
        std::vector<Batch> batch(n_members);
        for (size_t m = 0; m < n_members; ++m)
            if (int rc = batch[m].begin(group->members[m], scale, n_steps, std::min(64, n_vectors - col),
                                        batch_start(start, col), force_real))
                return rc;
        for (size_t m = 1; m < n_members; ++m)
            if (batch[m].rl != batch[0].rl)
                return fail(BDG_EINVAL, "group members chose different kernel configurations");
        const size_t unit = (size_t)4 * batch[0].rl * sizeof(double2);  // bytes per exchanged block row
        for (int n = 0; n < n_steps; ++n) {
            for (size_t m = 0; m < n_members; ++m) {
                bdg_system* sys = group->members[m];
                HIP_TRY(hipSetDevice(sys->device));
                if (n > 0)
                    for (const ExchangePeer& peer : sys->peers)
                        HIP_TRY(hipStreamWaitEvent(sys->stream, group->copied[peer.rank], 0));
                if (int rc = batch[m].pack()) return rc;
                HIP_TRY(hipEventRecord(group->packed[m], sys->stream));
            }
            for (size_t m = 0; m < n_members; ++m) {
                bdg_system* sys = group->members[m];
                HIP_TRY(hipSetDevice(sys->device));
                for (const ExchangePeer& peer : sys->peers) {
                    if (peer.recv_count == 0) continue;
                    bdg_system* src = group->members[peer.rank];
                    const ExchangePeer* back = nullptr;
                    for (const ExchangePeer& q : src->peers)
                        if (q.rank == (int)m) back = &q;
                    HIP_TRY(hipStreamWaitEvent(sys->stream, group->packed[peer.rank], 0));
                    HIP_TRY(hipMemcpyPeerAsync(
                        reinterpret_cast<char*>(sys->recv_buf.ptr) + (size_t)peer.recv_begin * unit,
                        sys->device,
                        reinterpret_cast<const char*>(src->send_buf.ptr) + (size_t)back->send_begin * unit,
                        src->device, (size_t)peer.recv_count * unit, sys->stream));
                }
                HIP_TRY(hipEventRecord(group->copied[m], sys->stream));
                if (int rc = batch[m].unpack()) return rc;
                if (int rc = batch[m].step(n)) return rc;
            }
        }
        for (size_t m = 0; m < n_members; ++m)
            if (int rc = batch[m].finish(d_out, e_out, n_vectors, col, m > 0, col == 0)) return rc;
    }
    return BDG_OK;
}

// --------------------------------------------------------------------- Lanczos
// Lanczos process on A = H^2 for the eigenvalues of H closest to zero (the excitation gap).
// v_0 = v/|v|;  per iteration j, with unnormalised W_j = beta_j v_j kept in memory:
//   a.  U = H v_j               K1(cur = W_j,  prev = scratch, coef = 1/beta_j, pscale = 0)
//   b.  R = H U - beta_j v_{j-1}  K1(cur = U, prev = W_{j-1}, coef = 1, pscale = beta_j/beta_{j-1});
//       its d-dot |U|^2 = <v_j|H^2|v_j> = alpha_j
//   c.  W_{j+1} = R - alpha_j v_j = R - (alpha_j/beta_j) W_j,  beta_{j+1} = |W_{j+1}|
// All scalars stay on the device (lanczos_scalars); the host only enqueues and finally reads
// alpha/beta.  Several start vectors run as independent columns of the same launches.
struct LanczosState {
    Batch batch;                 // kernel plan and matrix arguments of the run
    int cols = 0, iter = 0, max_iter = 0, n_active = 0;
    DeviceBuffer<double2> work;  // third vector buffer (U = H v_j)
    DeviceBuffer<double> scalars, sums, norm_partial;
    double2 *w_cur = nullptr, *w_prev = nullptr;
    bdg::LanczosScalars z{};
};

void lanczos_free(bdg_system* sys) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return;
    lz->work.release();
    lz->scalars.release();
    lz->sums.release();
    lz->norm_partial.release();
    delete lz;
    sys->lanczos = nullptr;
}

constexpr int kNormGrid = 512;

int lanczos_norms(bdg_system* sys, LanczosState* lz, const double2* vec, const double2* other, bool combine) {
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    if (combine) {
        if (b.real)
            bdg::lanczos_combine<2><<<kNormGrid, 256, 0, st>>>(const_cast<double2*>(vec), other, lz->z.g, sys->nb,
                                                               sys->ncols, b.rl, lz->norm_partial.ptr);
        else
            bdg::lanczos_combine<1><<<kNormGrid, 256, 0, st>>>(const_cast<double2*>(vec), other, lz->z.g, sys->nb,
                                                               sys->ncols, b.rl, lz->norm_partial.ptr);
    } else {
        if (b.real)
            bdg::column_norms<2><<<kNormGrid, 256, 0, st>>>(vec, sys->nb, sys->ncols, b.rl, lz->norm_partial.ptr);
        else
            bdg::column_norms<1><<<kNormGrid, 256, 0, st>>>(vec, sys->nb, sys->ncols, b.rl, lz->norm_partial.ptr);
    }
    bdg::reduce_partials<<<1, 256, 0, st>>>(lz->norm_partial.ptr, lz->sums.ptr, kNormGrid, lz->cols);
    HIP_TRY(hipGetLastError());
    return BDG_OK;
}

int lanczos_begin(bdg_system* sys, int n_vectors, const StartSpec& start, int max_iter) {
    if (sys->ncols != sys->nb) return fail(BDG_EINVAL, "Lanczos needs a whole (square) matrix, not a slab");
    if (n_vectors < 1 || n_vectors > 64) return fail(BDG_EINVAL, "Lanczos runs 1..64 start vectors");
    if (max_iter < 1 || max_iter > (1 << 20)) return fail(BDG_EINVAL, "bad iteration limit");
    lanczos_free(sys);
    LanczosState* lz = new LanczosState();
    // the handle learns about the run only once everything below has succeeded
    struct Guard {
        LanczosState* lz;
        bool keep = false;
        ~Guard() {
            if (keep) return;
            lz->work.release();
            lz->scalars.release();
            lz->sums.release();
            lz->norm_partial.release();
            delete lz;
        }
    } guard{lz};
    Batch& b = lz->batch;
    if (int rc = b.begin(sys, 1.0, 1, n_vectors, start, -1, /*col_scalars=*/true)) return rc;  // W_0 in vec_a
    lz->cols = b.rv;
    lz->n_active = n_vectors;
    lz->max_iter = max_iter;
    lz->iter = 0;
    if (int rc = lz->work.reserve(b.vec_count)) return rc;
    const size_t cols = (size_t)lz->cols;
    if (int rc = lz->scalars.reserve((2 * ((size_t)max_iter + 1) + 5) * cols)) return rc;
    if (int rc = lz->sums.reserve(2 * cols)) return rc;
    if (int rc = lz->norm_partial.reserve((size_t)kNormGrid * cols)) return rc;
    HIP_TRY(hipMemsetAsync(lz->scalars.ptr, 0, sizeof(double) * lz->scalars.count, sys->stream));
    HIP_TRY(hipMemsetAsync(lz->work.ptr, 0, sizeof(double2) * b.vec_count, sys->stream));
    double* base = lz->scalars.ptr;
    lz->z.beta_hist = base;
    lz->z.alpha_hist = base + ((size_t)max_iter + 1) * cols;
    double* tail = base + 2 * ((size_t)max_iter + 1) * cols;
    lz->z.coef_a = tail;
    lz->z.pscale_a = tail + cols;
    lz->z.coef_b = tail + 2 * cols;
    lz->z.pscale_b = tail + 3 * cols;
    lz->z.g = tail + 4 * cols;
    lz->w_cur = sys->vec_a.ptr;
    lz->w_prev = sys->vec_b.ptr;
    if (int rc = lanczos_norms(sys, lz, lz->w_cur, nullptr, false)) return rc;
    bdg::lanczos_scalars<<<1, 128, 0, sys->stream>>>(lz->z, lz->sums.ptr, lz->cols, 0, 0);
    HIP_TRY(hipGetLastError());
    guard.keep = true;
    sys->lanczos = lz;
    return BDG_OK;
}

// One iteration j of the process (steps a-c above); W_j is in lz->w_cur on entry, W_{j+1} on exit.
int lanczos_iterate(bdg_system* sys, LanczosState* lz, bdg::StepArgs& args, int j) {
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    args.cur = lz->w_cur;  // a. U = H v_j
    args.prev = lz->work.ptr;
    args.col_coef = lz->z.coef_a;
    args.col_pscale = lz->z.pscale_a;
    b.plan.kernel<<<b.plan.grid, bdg::kBlockThreads, b.plan.lds_bytes, st>>>(args);
    args.cur = lz->work.ptr;  // b. R = H U - beta_j v_{j-1}, alpha_j = |U|^2
    args.prev = lz->w_prev;
    args.col_coef = lz->z.coef_b;
    args.col_pscale = lz->z.pscale_b;
    b.plan.kernel<<<b.plan.grid, bdg::kBlockThreads, b.plan.lds_bytes, st>>>(args);
    bdg::reduce_partials<<<1, 256, 0, st>>>(sys->partial.ptr, lz->sums.ptr, b.plan.grid, (int)b.width);
    bdg::lanczos_scalars<<<1, 128, 0, st>>>(lz->z, lz->sums.ptr, lz->cols, j, 1);
    // c. W_{j+1} = R - alpha_j v_j, beta_{j+1}
    if (int rc = lanczos_norms(sys, lz, lz->w_prev, lz->w_cur, true)) return rc;
    bdg::lanczos_scalars<<<1, 128, 0, st>>>(lz->z, lz->sums.ptr, lz->cols, j + 1, 0);
    std::swap(lz->w_cur, lz->w_prev);
    return BDG_OK;
}

// Second pass: repeat the first n_iter iterations of a freshly begun process (same start vectors:
// the Lanczos vectors are reproduced bit for bit) and accumulate, for every level l and column c,
//   y_{l,c} = Σ_j coef[j][l][c] v_j^{(c)}
// i.e. the Ritz vectors whose tridiagonal coordinates the host computed from the first pass.
// y_out[l][c] is a site-major complex vector of 4*nb entries.
int lanczos_ritz_vectors(bdg_system* sys, int n_iter, int n_levels, const double* coef, double* y_out) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return fail(BDG_EINVAL, "bdg_lanczos_begin has not been called");
    if (lz->iter != 0) return fail(BDG_EINVAL, "the Ritz-vector pass starts from a freshly begun process");
    if (n_iter < 1 || n_iter > lz->max_iter || n_levels < 1 || n_levels > 64)
        return fail(BDG_EINVAL, "bad iteration or level count");
    HIP_TRY(hipSetDevice(sys->device));
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    const size_t cols = (size_t)lz->cols, count = b.vec_count;
    DeviceBuffer<double2> y, host_order;
    DeviceBuffer<double> dev_coef;
    auto body = [&]() -> int {
        if (int rc = y.reserve((size_t)n_levels * count)) return rc;
        if (int rc = host_order.reserve((size_t)4 * sys->nb)) return rc;
        if (int rc = dev_coef.reserve((size_t)n_iter * n_levels * cols)) return rc;
        // coefficients padded to the buffer's column count (inactive columns: 0)
        std::vector<double> padded((size_t)n_iter * n_levels * cols, 0.0);
        for (int j = 0; j < n_iter; ++j)
            for (int l = 0; l < n_levels; ++l)
                for (int c = 0; c < lz->n_active; ++c)
                    padded[((size_t)j * n_levels + l) * cols + c] = coef[((size_t)j * n_levels + l) * lz->n_active + c];
        HIP_TRY(hipMemcpyAsync(dev_coef.ptr, padded.data(), sizeof(double) * padded.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(y.ptr, 0, sizeof(double2) * n_levels * count, st));
        bdg::StepArgs args = b.args;
        args.partial = sys->partial.ptr;
        const int grid = (int)std::min<size_t>(2048, (count + 255) / 256);
        for (int j = 0; j < n_iter; ++j) {
            const double* beta_j = lz->z.beta_hist + (size_t)j * cols;
            const double* coef_j = dev_coef.ptr + (size_t)j * n_levels * cols;
            if (b.real)
                bdg::lanczos_accumulate<2><<<grid, 256, 0, st>>>(lz->w_cur, beta_j, coef_j, n_levels, (int)cols, b.rl,
                                                                 (int64_t)count, y.ptr);
            else
                bdg::lanczos_accumulate<1><<<grid, 256, 0, st>>>(lz->w_cur, beta_j, coef_j, n_levels, (int)cols, b.rl,
                                                                 (int64_t)count, y.ptr);
            if (int rc = lanczos_iterate(sys, lz, args, j)) return rc;
        }
        HIP_TRY(hipGetLastError());
        lz->iter = n_iter;
        const size_t n = (size_t)4 * sys->nb;
        const int cgrid = (int)std::min<size_t>(4096, (n + 255) / 256);
        for (int l = 0; l < n_levels; ++l)
            for (int c = 0; c < lz->n_active; ++c) {
                const double2* src = y.ptr + (size_t)l * count;
                if (b.real)
                    bdg::sitemajor_from_planar_real<<<cgrid, 256, 0, st>>>(src, host_order.ptr, sys->nb, b.rl, c);
                else
                    bdg::sitemajor_from_planar<<<cgrid, 256, 0, st>>>(src, host_order.ptr, sys->nb, b.rl, c);
                HIP_TRY(hipMemcpyAsync(y_out + 2 * n * ((size_t)l * lz->n_active + c), host_order.ptr, sizeof(double2) * n,
                                       hipMemcpyDeviceToHost, st));
            }
        HIP_TRY(hipStreamSynchronize(st));
        return BDG_OK;
    };
    const int rc = body();
    y.release();
    host_order.release();
    dev_coef.release();
    return rc;
}

int lanczos_advance(bdg_system* sys, int n_iter, double* alpha_out, double* beta_out) {
    LanczosState* lz = static_cast<LanczosState*>(sys->lanczos);
    if (!lz) return fail(BDG_EINVAL, "bdg_lanczos_begin has not been called");
    if (n_iter < 1 || lz->iter + n_iter > lz->max_iter)
        return fail(BDG_EINVAL, "iteration count exceeds the limit given to bdg_lanczos_begin");
    HIP_TRY(hipSetDevice(sys->device));
    Batch& b = lz->batch;
    hipStream_t st = sys->stream;
    bdg::StepArgs args = b.args;
    args.partial = sys->partial.ptr;
    const int first = lz->iter;
    for (int j = first; j < first + n_iter; ++j)
        if (int rc = lanczos_iterate(sys, lz, args, j)) return rc;
    HIP_TRY(hipGetLastError());
    lz->iter += n_iter;
    const size_t cols = (size_t)lz->cols;
    std::vector<double> alpha((size_t)n_iter * cols), beta((size_t)n_iter * cols);
    HIP_TRY(hipMemcpyAsync(alpha.data(), lz->z.alpha_hist + (size_t)first * cols, sizeof(double) * alpha.size(),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(beta.data(), lz->z.beta_hist + ((size_t)first + 1) * cols, sizeof(double) * beta.size(),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int j = 0; j < n_iter; ++j)
        for (int r = 0; r < lz->n_active; ++r) {
            alpha_out[(size_t)j * lz->n_active + r] = alpha[(size_t)j * cols + r];
            beta_out[(size_t)j * lz->n_active + r] = beta[(size_t)j * cols + r];
        }
    return BDG_OK;
}

// ------------------------------------------------------------- dense eigensolver
// One-sided Jacobi on the GPU for matrices up to kJacobiLimit (no external library: the first
// use of rocSOLVER on a fresh machine pages in ~1 GB and was measured at 1.5-7.5 minutes).
constexpr int64_t kJacobiLimit = 256 * bdg::kJacobiElems;  // 2048: a column pair fits the registers of a workgroup
// With 16 elements per thread the same kernels reach 4096 (n = 3600: ~3 s against rocSOLVER's 0.13 s):
// used while the rocSOLVER object has not arrived from cold storage yet, which takes minutes.
constexpr int64_t kJacobiWideLimit = 256 * bdg::kJacobiElemsWide;

inline void scatter_for_jacobi(bdg_system* sys, double2* G, hipStream_t st) {
    bdg::scatter_dense<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, G,
                                                           (int)sys->nb);
}
inline void scatter_for_jacobi(bdg_system* sys, double* G, hipStream_t st) {
    bdg::scatter_dense_real<<<(unsigned)sys->nb, 128, 0, st>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr,
                                                                G, (int)sys->nb);
}

// T = double2 (Hermitian) or double (real symmetric, when imag(H) = 0: half the bytes per round)
template <typename T>
int eigh_jacobi_typed(bdg_system* sys, double* w_out, double* z_out) {
    const int64_t n = 4 * sys->nb;
    if (n > kJacobiWideLimit)
        return fail(BDG_EINVAL, "the Jacobi kernels hold a column pair in registers: 4*nb <= %d", (int)kJacobiWideLimit);
    const bool wide = n > kJacobiLimit;
    hipStream_t st = sys->stream;
    DeviceBuffer<T> G, V;
    DeviceBuffer<double> eig;
    DeviceBuffer<int> counter;
    auto body = [&]() -> int {
        if (int rc = G.reserve((size_t)n * n)) return rc;
        if (z_out)
            if (int rc = V.reserve((size_t)n * n)) return rc;
        if (int rc = eig.reserve((size_t)n)) return rc;
        if (int rc = counter.reserve(1)) return rc;
        const double shift = 1.5 * sys->gershgorin + 1.0;  // spectrum of G in [0.5 b + 1, 2.5 b + 1]
        HIP_TRY(hipMemsetAsync(G.ptr, 0, sizeof(T) * n * n, st));
        scatter_for_jacobi(sys, G.ptr, st);
        bdg::jacobi_setup<T><<<(unsigned)std::min<int64_t>(4096, (n * n + 255) / 256), 256, 0, st>>>(
            G.ptr, z_out ? V.ptr : nullptr, (int)n, shift);
        HIP_TRY(hipGetLastError());
        const int max_sweeps = 40;
        int sweep = 0;
        for (; sweep < max_sweeps; ++sweep) {
            HIP_TRY(hipMemsetAsync(counter.ptr, 0, sizeof(int), st));
            for (int round = 0; round < n - 1; ++round) {
                if (wide)
                    bdg::jacobi_round<T, bdg::kJacobiElemsWide><<<(unsigned)(n / 2), 256, 0, st>>>(
                        G.ptr, z_out ? V.ptr : nullptr, (int)n, round, 1e-15, counter.ptr);
                else
                    bdg::jacobi_round<T, bdg::kJacobiElems><<<(unsigned)(n / 2), 256, 0, st>>>(
                        G.ptr, z_out ? V.ptr : nullptr, (int)n, round, 1e-15, counter.ptr);
            }
            int rotations = 0;
            HIP_TRY(hipMemcpyAsync(&rotations, counter.ptr, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (rotations == 0) break;
        }
        if (sweep == max_sweeps) return fail(BDG_ELIBRARY, "Jacobi eigensolver did not converge in %d sweeps", max_sweeps);
        bdg::jacobi_eigenvalues<T><<<(unsigned)n, 256, 0, st>>>(G.ptr, (int)n, shift, eig.ptr);
        HIP_TRY(hipGetLastError());
        std::vector<double> vals((size_t)n);
        HIP_TRY(hipMemcpyAsync(vals.data(), eig.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<int64_t> order((size_t)n);
        for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return vals[(size_t)a] < vals[(size_t)b]; });
        for (int64_t i = 0; i < n; ++i) w_out[i] = vals[(size_t)order[(size_t)i]];
        if (z_out) {
            constexpr size_t kScalars = sizeof(T) / sizeof(double);  // 2 complex, 1 real
            std::vector<double> cols(kScalars * n * n);
            HIP_TRY(hipMemcpy(cols.data(), V.ptr, sizeof(T) * n * n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; ++i) {
                const double* src = cols.data() + kScalars * n * order[(size_t)i];
                double* dst = z_out + 2 * n * i;
                if (kScalars == 2) {
                    memcpy(dst, src, sizeof(double) * 2 * n);
                } else {
                    for (int64_t k = 0; k < n; ++k) {
                        dst[2 * k] = src[k];
                        dst[2 * k + 1] = 0.0;
                    }
                }
            }
        }
        return BDG_OK;
    };
    const int rc = body();
    G.release();
    V.release();
    eig.release();
    counter.release();
    return rc;
}

int eigh_jacobi(bdg_system* sys, double* w_out, double* z_out) {
    bool real_route = sys->is_real;
    if (const char* env = getenv("BODGE_AMD_EIGH_REAL")) real_route = real_route && atoi(env) != 0;
    return real_route ? eigh_jacobi_typed<double>(sys, w_out, z_out) : eigh_jacobi_typed<double2>(sys, w_out, z_out);
}

}  // namespace

// =========================================================================== ABI
extern "C" {

const char* bdg_last_error(void) { return g_error.c_str(); }
const char* bdg_version(void) { return "bodge_hip 0.1 (gfx950)"; }

int bdg_device_count(int* count) {
    if (!count) return fail(BDG_EINVAL, "null count pointer");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return BDG_OK;
}

int bdg_create(int device, int64_t nb, int64_t nnzb, const int32_t* indptr, const int32_t* indices,
               const double* data, bdg_system** out) {
    return bdg_create_slab(device, nb, nb, nnzb, indptr, indices, data, 0, out);
}

int bdg_create_slab(int device, int64_t nb, int64_t ncols, int64_t nnzb, const int32_t* indptr,
                    const int32_t* indices, const double* data, int64_t row_offset, bdg_system** out) {
    if (!out) return fail(BDG_EINVAL, "null output handle");
    *out = nullptr;
    if (nb < 1 || ncols < nb || nnzb < 0 || row_offset < 0 || !indptr || (nnzb > 0 && (!indices || !data)))
        return fail(BDG_EINVAL, "bad matrix arguments (nb=%lld ncols=%lld nnzb=%lld)", (long long)nb,
                    (long long)ncols, (long long)nnzb);
    if (ncols > (1ll << 29) || nnzb > (1ll << 30))
        return fail(BDG_EINVAL, "matrix too large for 32-bit block indexing");
    if (indptr[0] != 0 || indptr[nb] != nnzb) return fail(BDG_EINVAL, "indptr does not span the blocks");
    int max_row = 0;
    int64_t bandwidth = 0;
    for (int64_t i = 0; i < nb; ++i)
        if (indptr[i + 1] < indptr[i] || indptr[i + 1] > nnzb)
            return fail(BDG_EINVAL, "indptr is not monotone within [0, nnzb] at row %lld", (long long)i);
    for (int64_t i = 0; i < nb; ++i) {
        const int len = indptr[i + 1] - indptr[i];
        max_row = std::max(max_row, len);
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            if (indices[k] < 0 || indices[k] >= ncols)
                return fail(BDG_EINVAL, "column index %d out of range in row %lld", indices[k],
                            (long long)i);
            if (k > indptr[i] && indices[k] == indices[k - 1])
                return fail(BDG_EINVAL, "row %lld has a duplicate column", (long long)i);
            if (ncols == nb && k > indptr[i] && indices[k] < indices[k - 1])
                return fail(BDG_EINVAL, "row %lld is not sorted", (long long)i);
            bandwidth = std::max<int64_t>(bandwidth, std::llabs((long long)indices[k] - (long long)i));
        }
    }
    // Distinct blocks (exact, bitwise).  Gives up as soon as there are too many to be useful.
    // A lattice matrix repeats a handful of blocks, mostly in runs: the four most recent ones are
    // compared directly before the hash map is asked.
    constexpr int kMaxDistinct = 256;  // table index shares a 32-bit word with the 24-bit column
    int dict_skipped = 0;
    std::vector<int> ids;
    std::vector<double> distinct;  // n_unique x 32 doubles
    {
        const char* env = getenv("BODGE_AMD_DICT");
        bool wanted = !(env && env[0] == '0') && nnzb > 0 && ncols <= (1 << 24);
        dict_skipped = (env && env[0] == '0') ? 3 : ncols > (1 << 24) ? 2 : 0;
        if (wanted) {
            std::unordered_map<std::string_view, int> seen;
            const double* recent_key[4] = {nullptr, nullptr, nullptr, nullptr};
            int recent_id[4] = {0, 0, 0, 0};
            int recent_next = 0;
            ids.resize((size_t)nnzb);
            for (int64_t k = 0; k < nnzb; ++k) {
                const double* block = data + 32 * k;
                int id = -1;
                for (int m = 0; m < 4 && id < 0; ++m)
                    if (recent_key[m] && memcmp(recent_key[m], block, 256) == 0) id = recent_id[m];
                if (id < 0) {
                    std::string_view key(reinterpret_cast<const char*>(block), 256);
                    auto it = seen.find(key);
                    if (it == seen.end()) {
                        if ((int)seen.size() == kMaxDistinct) {
                            wanted = false;
                            dict_skipped = 1;
                            break;
                        }
                        it = seen.emplace(key, (int)seen.size()).first;
                        distinct.insert(distinct.end(), block, block + 32);
                    }
                    id = it->second;
                    recent_key[recent_next] = block;
                    recent_id[recent_next] = id;
                    recent_next = (recent_next + 1) & 3;
                }
                ids[(size_t)k] = (int)((unsigned)indices[k] | ((unsigned)id << 24));
            }
        }
        if (!wanted) {
            ids.clear();
            distinct.clear();
        }
    }
    // Properties of the matrix: imag(H) == 0, particle-hole form of every block (lower-right 2x2 ==
    // -conj(upper-left 2x2), exactly), Gershgorin bound.  With a block dictionary they follow from
    // the distinct blocks and the ids; otherwise every stored block is scanned.
    auto block_is_real = [](const double* blk) {
        for (int e = 0; e < 16; ++e)
            if (blk[2 * e + 1] != 0.0) return false;
        return true;
    };
    auto block_is_ph = [](const double* blk) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                const double* a = blk + 2 * (i * 4 + j);
                const double* d = blk + 2 * ((i + 2) * 4 + (j + 2));
                if (d[0] != -a[0] || d[1] != a[1]) return false;
            }
        return true;
    };
    auto block_row_sums = [](const double* blk, double out[4]) {
        for (int r = 0; r < 4; ++r) {
            out[r] = 0.0;
            for (int c = 0; c < 4; ++c) out[r] += std::hypot(blk[2 * (4 * r + c)], blk[2 * (4 * r + c) + 1]);
        }
    };
    bool is_real = true, is_ph = true;
    double gershgorin = 0.0;
    if (!ids.empty()) {
        const size_t n_distinct = distinct.size() / 32;
        std::vector<double> sums(4 * n_distinct);
        for (size_t d = 0; d < n_distinct; ++d) {
            is_real = is_real && block_is_real(distinct.data() + 32 * d);
            is_ph = is_ph && block_is_ph(distinct.data() + 32 * d);
            block_row_sums(distinct.data() + 32 * d, sums.data() + 4 * d);
        }
        for (int64_t i = 0; i < nb; ++i) {
            double row_sum[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
                const double* sum = sums.data() + 4 * ((unsigned)ids[(size_t)k] >> 24);
                for (int r = 0; r < 4; ++r) row_sum[r] += sum[r];
            }
            for (double v : row_sum) gershgorin = std::max(gershgorin, v);
        }
    } else {
        for (int64_t k = 0; k < nnzb && (is_real || is_ph); ++k) {
            is_real = is_real && block_is_real(data + 32 * k);
            is_ph = is_ph && block_is_ph(data + 32 * k);
        }
        for (int64_t i = 0; i < nb; ++i) {
            double row_sum[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
                double sum[4];
                block_row_sums(data + 32 * (int64_t)k, sum);
                for (int r = 0; r < 4; ++r) row_sum[r] += sum[r];
            }
            for (double v : row_sum) gershgorin = std::max(gershgorin, v);
        }
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        (void)hipGetLastError();
        return fail(BDG_EDEVICE, "no HIP device is visible");
    }
    if (device < 0 || device >= n_dev) return fail(BDG_EINVAL, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));

    bdg_system* sys = new bdg_system();
    sys->device = device;
    sys->nb = nb;
    sys->ncols = ncols;
    sys->row_offset = row_offset;
    sys->nnzb = nnzb;
    sys->max_row_blocks = std::max(1, max_row);
    sys->bandwidth = bandwidth;
    if (ncols > nb) {
        sys->row_needs_halo.assign((size_t)nb, 0);
        for (int64_t i = 0; i < nb; ++i)
            for (int k = indptr[i]; k < indptr[i + 1]; ++k)
                if (indices[k] >= nb) sys->row_needs_halo[(size_t)i] = 1;
    }
    sys->dict_skipped = dict_skipped;
    sys->is_real = is_real;
    sys->is_ph = is_ph;
    sys->gershgorin = gershgorin;
    hipDeviceProp_t prop;
    auto cleanup = [&](int rc) {
        bdg_destroy(sys);
        return rc;
    };
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return cleanup(fail(BDG_EDEVICE, "hipGetDeviceProperties failed"));
    sys->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&sys->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&sys->ev_start) != hipSuccess || hipEventCreate(&sys->ev_stop) != hipSuccess)
        return cleanup(fail(BDG_EDEVICE, "stream/event creation failed"));
    if (int rc = sys->indptr.reserve((size_t)nb + 1)) return cleanup(rc);
    if (int rc = sys->indices.reserve((size_t)std::max<int64_t>(1, nnzb))) return cleanup(rc);
    if (int rc = sys->blocks.reserve((size_t)std::max<int64_t>(1, nnzb) * 16)) return cleanup(rc);
    if (hipMemcpy(sys->indptr.ptr, indptr, sizeof(int) * (nb + 1), hipMemcpyHostToDevice) != hipSuccess ||
        (nnzb > 0 &&
         (hipMemcpy(sys->indices.ptr, indices, sizeof(int) * nnzb, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(sys->blocks.ptr, data, sizeof(double2) * 16 * nnzb, hipMemcpyHostToDevice) !=
              hipSuccess)))
        return cleanup(fail(BDG_EDEVICE, "upload of the BSR arrays failed"));
    // First use of the device-to-host copy path for more than a few KB costs ~7 ms
    // (measured in the first 256-launch call).  Take it here: allocate the dot-product
    // buffers now and pull them once.
    {
        const size_t dots_count = (size_t)1024 * 128;  // 1024 launches x 64 vectors x {d, e}
        if (int rc = sys->dots.reserve(dots_count)) return cleanup(rc);
        if (hipHostMalloc(reinterpret_cast<void**>(&sys->host_dots), dots_count * sizeof(double), 0) != hipSuccess)
            return cleanup(fail(BDG_ENOMEM, "pinned host allocation failed"));
        sys->host_dots_count = dots_count;
        if (hipMemcpyAsync(sys->host_dots, sys->dots.ptr, dots_count * sizeof(double), hipMemcpyDeviceToHost,
                           sys->stream) != hipSuccess ||
            hipStreamSynchronize(sys->stream) != hipSuccess)
            return cleanup(fail(BDG_EDEVICE, "device-to-host warm-up copy failed"));
    }
    if (!ids.empty()) {
        sys->n_unique = (int)(distinct.size() / 32);
        if (int rc = sys->dict_ids.reserve(ids.size())) return cleanup(rc);
        if (int rc = sys->dict_full.reserve((size_t)sys->n_unique * 16)) return cleanup(rc);
        std::vector<int> diagonal((size_t)sys->n_unique, 1);
        for (int d = 0; d < sys->n_unique; ++d)
            for (int e = 0; e < 16; ++e)
                if ((e >> 2) != (e & 3) && (distinct[(size_t)32 * d + 2 * e] != 0.0 || distinct[(size_t)32 * d + 2 * e + 1] != 0.0))
                    diagonal[(size_t)d] = 0;
        if (getenv("BODGE_AMD_NO_DIAGONAL_BLOCKS")) std::fill(diagonal.begin(), diagonal.end(), 0);
        if (int rc = sys->dict_diagonal.reserve((size_t)sys->n_unique)) return cleanup(rc);
        if (hipMemcpy(sys->dict_diagonal.ptr, diagonal.data(), sizeof(int) * diagonal.size(), hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(BDG_EDEVICE, "upload of the block dictionary failed"));
        if (hipMemcpy(sys->dict_ids.ptr, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(sys->dict_full.ptr, distinct.data(), sizeof(double) * distinct.size(),
                      hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(BDG_EDEVICE, "upload of the block dictionary failed"));
    }
    *out = sys;
    return BDG_OK;
}

int bdg_slab_set_exchange(bdg_system* sys, bdg_comm* comm, int32_t n_peers, const int32_t* peer_rank,
                          const int64_t* send_count, const int64_t* send_rows, const int64_t* recv_col,
                          const int64_t* recv_count) {
    if (!sys) return fail(BDG_EINVAL, "null system handle");
    if (n_peers < 0 || (n_peers > 0 && (!peer_rank || !send_count || !recv_col || !recv_count)))
        return fail(BDG_EINVAL, "bad exchange arguments");
    std::vector<ExchangePeer> peers((size_t)n_peers);
    int64_t send_total = 0, recv_total = 0;
    for (int p = 0; p < n_peers; ++p) {
        ExchangePeer& e = peers[p];
        e.rank = peer_rank[p];
        e.send_begin = send_total;
        e.send_count = send_count[p];
        e.recv_col = recv_col[p];
        e.recv_begin = recv_total;
        e.recv_count = recv_count[p];
        if (e.rank < 0 || e.send_count < 0 || e.recv_count < 0)
            return fail(BDG_EINVAL, "negative count or rank for peer %d", p);
        if (e.recv_count > 0 && (e.recv_col < sys->nb || e.recv_col + e.recv_count > sys->ncols))
            return fail(BDG_EINVAL, "receive range of peer %d is outside the halo columns", p);
        send_total += e.send_count;
        recv_total += e.recv_count;
    }
    if (send_total > 0 && !send_rows) return fail(BDG_EINVAL, "null send_rows");
    for (int64_t k = 0; k < send_total; ++k)
        if (send_rows[k] < 0 || send_rows[k] >= sys->nb)
            return fail(BDG_EINVAL, "send row %lld is not an owned row", (long long)send_rows[k]);
    HIP_TRY(hipSetDevice(sys->device));
    if (send_total > 0) {
        if (int rc = sys->send_rows.reserve((size_t)send_total)) return rc;
        HIP_TRY(hipMemcpy(sys->send_rows.ptr, send_rows, sizeof(int64_t) * send_total, hipMemcpyHostToDevice));
    }
    sys->peers = std::move(peers);
    sys->send_total = send_total;
    sys->recv_total = recv_total;
    sys->slab_comm = comm;
    sys->slab_all_real = sys->is_real;
    sys->slab_max_ncols = sys->ncols;
    if (comm && comm->n_ranks > 1) {
        // collective: every rank of the communicator sets its exchange lists at this point.
        // min over ranks of "my slab is real" and max of the buffer rows decide mode and batch width for all.
        double agree[2] = {sys->is_real ? 0.0 : 1.0, (double)sys->ncols};
        if (int rc = comm_allreduce(comm, agree, 2, ncclMax)) return rc;
        sys->slab_all_real = agree[0] == 0.0;
        sys->slab_max_ncols = (int64_t)agree[1];
    }
    return BDG_OK;
}

int bdg_group_create(bdg_system** members, int32_t n_members, bdg_group** out) {
    if (!members || n_members < 1 || !out) return fail(BDG_EINVAL, "bad group arguments");
    *out = nullptr;
    for (int m = 0; m < n_members; ++m) {
        if (!members[m]) return fail(BDG_EINVAL, "null group member %d", m);
        for (const ExchangePeer& peer : members[m]->peers) {
            if (peer.rank >= n_members) return fail(BDG_EINVAL, "member %d names peer %d outside the group", m, peer.rank);
            const ExchangePeer* back = nullptr;
            for (const ExchangePeer& q : members[peer.rank]->peers)
                if (q.rank == m) back = &q;
            if (!back || back->send_count != peer.recv_count || back->recv_count != peer.send_count)
                return fail(BDG_EINVAL, "exchange lists of members %d and %d do not match", m, peer.rank);
        }
    }
    bdg_group* group = new bdg_group();
    group->members.assign(members, members + n_members);
    group->packed.resize(n_members);
    group->copied.resize(n_members);
    for (int m = 0; m < n_members; ++m) {
        (void)hipSetDevice(members[m]->device);
        if (hipEventCreateWithFlags(&group->packed[m], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&group->copied[m], hipEventDisableTiming) != hipSuccess) {
            delete group;
            return fail(BDG_EDEVICE, "event creation failed");
        }
    }
    *out = group;
    return BDG_OK;
}

int bdg_group_destroy(bdg_group* group) {
    if (!group) return BDG_OK;
    for (size_t m = 0; m < group->members.size(); ++m) {
        (void)hipSetDevice(group->members[m]->device);
        (void)hipStreamSynchronize(group->members[m]->stream);
        if (group->packed[m]) (void)hipEventDestroy(group->packed[m]);
        if (group->copied[m]) (void)hipEventDestroy(group->copied[m]);
    }
    delete group;
    return BDG_OK;
}

int bdg_group_dots_random(bdg_group* group, double scale, int32_t n_steps, int32_t n_vectors,
                          uint64_t seed, uint64_t first_vec_id, int32_t vec_kind, double* d_out,
                          double* e_out) {
    if (vec_kind != BDG_VEC_RADEMACHER && vec_kind != BDG_VEC_Z4)
        return fail(BDG_EINVAL, "unknown start-vector kind %d", vec_kind);
    StartSpec start{StartKind::Random};
    start.seed = seed;
    start.first_id = first_vec_id;
    start.vec_kind = vec_kind;
    return run_group(group, scale, n_steps, n_vectors, start, d_out, e_out);
}

int bdg_group_dots_unit(bdg_group* group, double scale, int32_t n_steps, int32_t n_vectors,
                        const int64_t* rows, double* d_out, double* e_out) {
    if (!rows) return fail(BDG_EINVAL, "null rows pointer");
    StartSpec start{StartKind::Unit};
    start.rows = rows;
    return run_group(group, scale, n_steps, n_vectors, start, d_out, e_out);
}

int bdg_destroy(bdg_system* sys) {
    if (!sys) return BDG_OK;
    (void)hipSetDevice(sys->device);
    if (sys->stream) (void)hipStreamSynchronize(sys->stream);
    lanczos_free(sys);
    sys->indptr.release();
    sys->indices.release();
    sys->blocks.release();
    for (auto& buf : sys->packed) buf.release();
    for (auto& buf : sys->dict_table) buf.release();
    sys->dict_ids.release();
    sys->dict_diagonal.release();
    sys->dict_full.release();
    sys->vec_a.release();
    sys->vec_b.release();
    sys->vec_c.release();
    sys->vec_d.release();
    sys->stencil.release();
    sys->partial.release();
    sys->dots.release();
    sys->rows.release();
    if (sys->host_dots) (void)hipHostFree(sys->host_dots);
    sys->host_dots = nullptr;
    sys->tile_order.release();
    sys->send_rows.release();
    sys->tiles_interior.release();
    sys->tiles_boundary.release();
    if (sys->comm_stream) (void)hipStreamDestroy(sys->comm_stream);
    if (sys->ev_step_done) (void)hipEventDestroy(sys->ev_step_done);
    if (sys->ev_halo_ready) (void)hipEventDestroy(sys->ev_halo_ready);
    sys->send_buf.release();
    sys->recv_buf.release();
    if (sys->ev_start) (void)hipEventDestroy(sys->ev_start);
    if (sys->ev_stop) (void)hipEventDestroy(sys->ev_stop);
    for (hipEvent_t ev : sys->ev_pool) (void)hipEventDestroy(ev);
    if (sys->stream) (void)hipStreamDestroy(sys->stream);
    delete sys;
    return BDG_OK;
}

int bdg_set_lattice_shape(bdg_system* sys, int32_t lx, int32_t ly, int32_t lz) {
    if (!sys) return fail(BDG_EINVAL, "null system handle");
    if (lx < 0 || ly < 0 || lz < 0) return fail(BDG_EINVAL, "negative lattice extent");
    if ((int64_t)lx * ly * lz != sys->nb && (lx | ly | lz) != 0)
        return fail(BDG_EINVAL, "lattice %dx%dx%d does not have %lld sites", lx, ly, lz,
                    (long long)sys->nb);
    sys->shape[0] = lx;
    sys->shape[1] = ly;
    sys->shape[2] = lz;
    sys->order_rows_per_tile = 0;
    sys->stencil_state = 0;
    return BDG_OK;
}

int bdg_set_lanes_per_row(bdg_system* sys, int32_t lanes) {
    if (!sys) return fail(BDG_EINVAL, "null system handle");
    if (lanes != 0 && !step_kernel(mode_info(false, false), lanes)) return fail(BDG_EINVAL, "lanes must be 4, 8, 16, 32 or 64");
    sys->lanes_override = lanes;
    return BDG_OK;
}

int bdg_spmv(bdg_system* sys, const double* x, double* y) {
    if (!sys || !x || !y) return fail(BDG_EINVAL, "null argument");
    if (sys->ncols != sys->nb) return fail(BDG_EINVAL, "bdg_spmv needs a whole (square) matrix, not a slab");
    lanczos_free(sys);
    HIP_TRY(hipSetDevice(sys->device));
    constexpr int kCols = 4;  // narrowest kernel configuration; columns 1..3 stay zero
    StepPlan plan;
    if (int rc = make_plan(sys, kCols, mode_info(false, false), &plan)) return rc;
    const size_t n = (size_t)4 * sys->nb;
    if (int rc = sys->vec_a.reserve(n * kCols)) return rc;
    if (int rc = sys->vec_b.reserve(n * kCols)) return rc;
    if (int rc = sys->partial.reserve((size_t)plan.grid * 2 * kCols)) return rc;
    DeviceBuffer<double2> host_order;
    if (int rc = host_order.reserve(n)) return rc;
    hipStream_t st = sys->stream;
    const int grid = (int)std::min<size_t>(4096, (n * kCols + 255) / 256);
    auto body = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(host_order.ptr, x, sizeof(double2) * n, hipMemcpyHostToDevice, st));
        bdg::fill_zero<<<grid, 256, 0, st>>>(sys->vec_a.ptr, (int64_t)(n * kCols));
        bdg::fill_zero<<<grid, 256, 0, st>>>(sys->vec_b.ptr, (int64_t)(n * kCols));
        bdg::planar_from_sitemajor<<<grid, 256, 0, st>>>(host_order.ptr, sys->vec_a.ptr, sys->nb, kCols, 0);
        bdg::StepArgs args{};
        if (int rc = matrix_args(sys, plan, &args)) return rc;
        args.cur = sys->vec_a.ptr;
        args.prev = sys->vec_b.ptr;
        args.partial = sys->partial.ptr;
        args.coef = 1.0;
        plan.kernel<<<plan.grid, bdg::kBlockThreads, plan.lds_bytes, st>>>(args);
        bdg::sitemajor_from_planar<<<grid, 256, 0, st>>>(sys->vec_b.ptr, host_order.ptr, sys->nb, kCols, 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(y, host_order.ptr, sizeof(double2) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return BDG_OK;
    };
    int rc = body();
    host_order.release();
    return rc;
}

int bdg_cheb_dots_random(bdg_system* sys, double scale, int32_t n_steps, int32_t n_vectors,
                         uint64_t seed, uint64_t first_vec_id, int32_t vec_kind, double* d_out,
                         double* e_out) {
    if (vec_kind != BDG_VEC_RADEMACHER && vec_kind != BDG_VEC_Z4)
        return fail(BDG_EINVAL, "unknown start-vector kind %d", vec_kind);
    StartSpec start{StartKind::Random};
    start.seed = seed;
    start.first_id = first_vec_id;
    start.vec_kind = vec_kind;
    return run_recurrence(sys, scale, n_steps, n_vectors, start, d_out, e_out);
}

int bdg_cheb_dots_unit(bdg_system* sys, double scale, int32_t n_steps, int32_t n_vectors,
                       const int64_t* rows, double* d_out, double* e_out) {
    if (!sys) return fail(BDG_EINVAL, "null system handle");
    if (!rows) return fail(BDG_EINVAL, "null rows pointer");
    for (int r = 0; r < n_vectors; ++r)
        if (rows[r] < 0 || (sys->ncols == sys->nb && rows[r] >= 4 * sys->nb))
            return fail(BDG_EINVAL, "start row %lld out of range", (long long)rows[r]);
    StartSpec start{StartKind::Unit};
    start.rows = rows;
    return run_recurrence(sys, scale, n_steps, n_vectors, start, d_out, e_out);
}

int bdg_cheb_moments(bdg_system* sys, bdg_comm* comm, double scale, int32_t n_moments,
                     int32_t n_vectors, uint64_t seed, uint64_t first_vec_id, int32_t vec_kind,
                     double* mu_out) {
    if (n_moments < 2 || (n_moments & 1)) return fail(BDG_EINVAL, "n_moments must be even and >= 2");
    if (!mu_out) return fail(BDG_EINVAL, "null output buffer");
    const int n_steps = n_moments / 2;
    std::vector<double> d((size_t)n_steps * n_vectors), e(d.size()), mu((size_t)n_moments * n_vectors);
    if (int rc = bdg_cheb_dots_random(sys, scale, n_steps, n_vectors, seed, first_vec_id, vec_kind,
                                      d.data(), e.data()))
        return rc;
    dots_to_moments(d.data(), e.data(), n_steps, n_vectors, mu.data());
    for (int m = 0; m < n_moments; ++m) {
        double tot = 0.0;
        for (int r = 0; r < n_vectors; ++r) tot += mu[(size_t)m * n_vectors + r];
        mu_out[m] = tot;
    }
    if (comm) return comm_allreduce(comm, mu_out, n_moments, ncclSum);
    return BDG_OK;
}

int bdg_cheb_diag_moments(bdg_system* sys, double scale, int32_t n_moments, int32_t n_vectors,
                          const int64_t* rows, double* mu_out) {
    if (n_moments < 2 || (n_moments & 1)) return fail(BDG_EINVAL, "n_moments must be even and >= 2");
    if (!mu_out) return fail(BDG_EINVAL, "null output buffer");
    const int n_steps = n_moments / 2;
    std::vector<double> d((size_t)n_steps * n_vectors), e(d.size());
    if (int rc = bdg_cheb_dots_unit(sys, scale, n_steps, n_vectors, rows, d.data(), e.data())) return rc;
    dots_to_moments(d.data(), e.data(), n_steps, n_vectors, mu_out);
    return BDG_OK;
}

int bdg_random_vector(bdg_system* sys, uint64_t seed, uint64_t vec_id, int32_t vec_kind,
                      double* v_out) {
    if (!sys || !v_out) return fail(BDG_EINVAL, "null argument");
    if (vec_kind != BDG_VEC_RADEMACHER && vec_kind != BDG_VEC_Z4)
        return fail(BDG_EINVAL, "unknown start-vector kind %d", vec_kind);
    lanczos_free(sys);
    HIP_TRY(hipSetDevice(sys->device));
    const size_t n = (size_t)4 * sys->nb;
    if (int rc = sys->vec_a.reserve(n)) return rc;
    DeviceBuffer<double2> host_order;
    if (int rc = host_order.reserve(n)) return rc;
    const int grid = (int)std::min<size_t>(4096, (n + 255) / 256);
    auto body = [&]() -> int {
        bdg::fill_random<<<grid, 256, 0, sys->stream>>>(sys->vec_a.ptr, sys->nb, sys->nb, 1, 1, seed,
                                                        vec_id, vec_kind, sys->row_offset);
        bdg::sitemajor_from_planar<<<grid, 256, 0, sys->stream>>>(sys->vec_a.ptr, host_order.ptr,
                                                                  sys->nb, 1, 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(v_out, host_order.ptr, sizeof(double2) * n, hipMemcpyDeviceToHost,
                               sys->stream));
        HIP_TRY(hipStreamSynchronize(sys->stream));
        return BDG_OK;
    };
    int rc = body();
    host_order.release();
    return rc;
}

int bdg_eigh_dense(bdg_system* sys, double* w_out, double* z_out) {
    if (!sys || !w_out) return fail(BDG_EINVAL, "null argument");
    if (sys->ncols != sys->nb) return fail(BDG_EINVAL, "bdg_eigh_dense needs a whole (square) matrix, not a slab");
    lanczos_free(sys);
    HIP_TRY(hipSetDevice(sys->device));
    const int64_t n = 4 * sys->nb;
    {
        const char* forced = getenv("BODGE_AMD_EIGH");
        bool own = forced ? std::string(forced) == "jacobi" : n <= kJacobiLimit;
        if (!forced && n > kJacobiLimit && n <= kJacobiWideLimit) {
            // the library solves this size in 0.1-0.2 s once loaded, but from cold storage its 931 MB take
            // minutes to arrive: until they have (read on in the background), the own kernels serve
            g_solver_prefetch.start();
            own = !g_solver_prefetch.wait(0.0);
        }
        if (own) return eigh_jacobi(sys, w_out, z_out);
    }
    SolverApi* api = nullptr;
    if (int rc = load_solver(&api)) return rc;
    if (n > 46000) return fail(BDG_EINVAL, "dense path limited to 4*nb <= 46000 (32-bit LAPACK sizes)");
    DeviceBuffer<double2> dense;
    DeviceBuffer<double> dense_real;
    DeviceBuffer<double> eig, offdiag;
    DeviceBuffer<int> info;
    rocblas_handle handle = nullptr;
    const rocblas_evect evect = z_out ? rocblas_evect_original : rocblas_evect_none;
    // imag(H) = 0 everywhere (checked at upload): real symmetric drivers, half the memory and a
    // quarter of the arithmetic of the Hermitian ones (BASELINE config 5 names dsyevd).
    bool real_route = sys->is_real;
    if (const char* env = getenv("BODGE_AMD_EIGH_REAL")) real_route = real_route && atoi(env) != 0;

    // One attempt with the named rocSOLVER driver: "evd" divide & conquer, "evj" Jacobi, "ev" QL/QR.
    // `copy_bad`: copy the results out even if they contain non-finite values
    auto attempt = [&](const std::string& algo, bool copy_bad, bool* nonfinite) -> int {
        rocblas_status st;
        if (real_route) {
            HIP_TRY(hipMemsetAsync(dense_real.ptr, 0, sizeof(double) * n * n, sys->stream));
            bdg::scatter_dense_real<<<(unsigned)sys->nb, 128, 0, sys->stream>>>(
                sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, dense_real.ptr, (int)sys->nb);
            HIP_TRY(hipGetLastError());
            if (algo == "evj") {
                st = api->dsyevj(handle, rocblas_esort_ascending, evect, rocblas_fill_lower, (rocblas_int)n,
                                 dense_real.ptr, (rocblas_int)n, 0.0, offdiag.ptr, 100,
                                 reinterpret_cast<rocblas_int*>(offdiag.ptr + 1), eig.ptr, info.ptr);
            } else if (algo == "evd") {
                st = api->dsyevd(handle, evect, rocblas_fill_lower, (rocblas_int)n, dense_real.ptr,
                                 (rocblas_int)n, eig.ptr, offdiag.ptr, info.ptr);
            } else {
                return fail(BDG_EINVAL, "real symmetric route has drivers evd and evj, not '%s'", algo.c_str());
            }
        } else {
            HIP_TRY(hipMemsetAsync(dense.ptr, 0, sizeof(double2) * n * n, sys->stream));
            bdg::scatter_dense<<<(unsigned)sys->nb, 128, 0, sys->stream>>>(
                sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, dense.ptr, (int)sys->nb);
            HIP_TRY(hipGetLastError());
            auto* a_ptr = reinterpret_cast<rocblas_double_complex*>(dense.ptr);
            if (algo == "ev") {
                st = api->zheev(handle, evect, rocblas_fill_lower, (rocblas_int)n, a_ptr, (rocblas_int)n,
                                eig.ptr, offdiag.ptr, info.ptr);
            } else if (algo == "evj") {
                // offdiag doubles as {residual, n_sweeps} scratch
                st = api->zheevj(handle, rocblas_esort_ascending, evect, rocblas_fill_lower, (rocblas_int)n,
                                 a_ptr, (rocblas_int)n, 0.0, offdiag.ptr, 100,
                                 reinterpret_cast<rocblas_int*>(offdiag.ptr + 1), eig.ptr, info.ptr);
            } else {
                st = api->zheevd(handle, evect, rocblas_fill_lower, (rocblas_int)n, a_ptr, (rocblas_int)n,
                                 eig.ptr, offdiag.ptr, info.ptr);
            }
        }
        if (st != rocblas_status_success)
            return fail(BDG_ELIBRARY, "rocsolver eigensolver (%s) returned %d", algo.c_str(), (int)st);
        // convergence flag and a finite-ness scan of the results, both on the device: nothing is
        // copied to the host before the result is known to be usable
        const double* vec_ptr = real_route ? dense_real.ptr : reinterpret_cast<const double*>(dense.ptr);
        const int64_t vec_doubles = z_out ? (real_route ? n * n : 2 * n * n) : 0;
        HIP_TRY(hipMemsetAsync(info.ptr + 1, 0, sizeof(int), sys->stream));
        bdg::count_nonfinite<<<64, 256, 0, sys->stream>>>(eig.ptr, n, info.ptr + 1);
        if (vec_doubles > 0)
            bdg::count_nonfinite<<<4096, 256, 0, sys->stream>>>(vec_ptr, vec_doubles, info.ptr + 1);
        HIP_TRY(hipGetLastError());
        int host_info[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(host_info, info.ptr, 2 * sizeof(int), hipMemcpyDeviceToHost, sys->stream));
        HIP_TRY(hipStreamSynchronize(sys->stream));
        if (host_info[0] != 0)
            return fail(BDG_ELIBRARY, "eigensolver (%s) did not converge (info=%d)", algo.c_str(), host_info[0]);
        *nonfinite = host_info[1] != 0;
        if (*nonfinite && !copy_bad) return BDG_OK;
        HIP_TRY(hipMemcpyAsync(w_out, eig.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, sys->stream));
        if (z_out) {
            if (real_route) {
                // real eigenvectors land in the back half of z_out and are widened in place
                double* tmp = z_out + (size_t)n * n;
                HIP_TRY(hipMemcpyAsync(tmp, dense_real.ptr, sizeof(double) * n * n, hipMemcpyDeviceToHost,
                                       sys->stream));
                HIP_TRY(hipStreamSynchronize(sys->stream));
                for (size_t i = 0, total = (size_t)n * n; i < total; ++i) {
                    const double v = tmp[i];
                    z_out[2 * i] = v;
                    z_out[2 * i + 1] = 0.0;
                }
            } else {
                HIP_TRY(hipMemcpyAsync(z_out, dense.ptr, sizeof(double2) * n * n, hipMemcpyDeviceToHost,
                                       sys->stream));
            }
        }
        HIP_TRY(hipStreamSynchronize(sys->stream));
        return BDG_OK;
    };
    auto body = [&]() -> int {
        if (real_route) {
            if (int rc = dense_real.reserve((size_t)n * n)) return rc;
        } else if (int rc = dense.reserve((size_t)n * n)) {
            return rc;
        }
        if (int rc = eig.reserve((size_t)std::max<int64_t>(n, 2))) return rc;
        if (int rc = offdiag.reserve((size_t)std::max<int64_t>(n, 2))) return rc;
        if (int rc = info.reserve(2)) return rc;
        if (api->create_handle(&handle) != rocblas_status_success)
            return fail(BDG_ELIBRARY, "rocblas_create_handle failed");
        if (api->set_stream(handle, sys->stream) != rocblas_status_success)
            return fail(BDG_ELIBRARY, "rocblas_set_stream failed");
        const char* forced = getenv("BODGE_AMD_EIGH");
        bool nonfinite = false;
        // a forced driver returns whatever it produced (tests look at the defect itself)
        if (forced && *forced && std::string(forced) != "rocsolver") return attempt(forced, true, &nonfinite);
        // Divide & conquer is the fast driver.  On ROCm 7.2 / gfx950 zheevd was seen to return NaN
        // eigenvectors when it is handed a matrix with imag(H) = 0 and a degenerate spectrum
        // (profiles/r01_eigh_probe.log; eigenvalues unaffected) - a case the real route above never
        // sends it.  The device-side scan catches any such result before it is copied; only then is
        // the Jacobi driver run, as a safety net and at its own O(n^3) cost.
        if (int rc = attempt("evd", false, &nonfinite)) return rc;
        if (!nonfinite) return BDG_OK;
        if (int rc = attempt("evj", false, &nonfinite)) return rc;
        if (nonfinite) return fail(BDG_ELIBRARY, "eigensolver returned non-finite values");
        return BDG_OK;
    };
    int rc = body();
    if (handle) api->destroy_handle(handle);
    dense.release();
    dense_real.release();
    eig.release();
    offdiag.release();
    info.release();
    return rc;
}

int bdg_hermiticity_defect(bdg_system* sys, double* defect_out) {
    if (!sys || !defect_out) return fail(BDG_EINVAL, "null argument");
    if (sys->ncols != sys->nb) return fail(BDG_EINVAL, "bdg_hermiticity_defect needs a whole (square) matrix, not a slab");
    HIP_TRY(hipSetDevice(sys->device));
    const int grid = (int)std::min<int64_t>(2048, (sys->nb + 255) / 256);
    DeviceBuffer<double> partial;
    if (int rc = partial.reserve((size_t)grid)) return rc;
    std::vector<double> host((size_t)grid);
    auto body = [&]() -> int {
        bdg::hermiticity_defect<<<grid, 256, 0, sys->stream>>>(sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr,
                                                              (int)sys->nb, partial.ptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host.data(), partial.ptr, sizeof(double) * grid, hipMemcpyDeviceToHost, sys->stream));
        HIP_TRY(hipStreamSynchronize(sys->stream));
        return BDG_OK;
    };
    const int rc = body();
    partial.release();
    if (rc) return rc;
    double worst = 0.0;
    for (double v : host) worst = std::isnan(v) ? v : std::max(worst, v);
    *defect_out = worst;
    return BDG_OK;
}

// ---- host-side assembly helpers (CPU threads only; host_assembly.hpp)
int bdg_host_fill_terms(double* data, int64_t nnzb, const int64_t* ids, int64_t count, const double* values,
                        int per_term, int kind, uint8_t* touched) {
    if (count < 0 || nnzb < 0 || kind < 0 || kind > 2) return fail(BDG_EINVAL, "bad fill arguments (count=%lld kind=%d)", (long long)count, kind);
    if (count == 0) return BDG_OK;
    if (!data || !ids || !values) return fail(BDG_EINVAL, "null argument");
    for (int64_t n = 0; n < count; ++n)
        if (ids[n] < 0 || ids[n] >= nnzb) return fail(BDG_EINVAL, "term %lld names block %lld of %lld", (long long)n, (long long)ids[n], (long long)nnzb);
    bdg_host::fill_terms(data, nnzb, ids, count, values, per_term != 0, kind, touched);
    return BDG_OK;
}

static int check_indptr(const int32_t* indptr, int64_t nb) {
    if (!indptr || nb < 0) return fail(BDG_EINVAL, "null argument");
    if (indptr[0] != 0) return fail(BDG_EINVAL, "indptr does not start at 0");
    for (int64_t i = 0; i < nb; ++i)
        if (indptr[i + 1] < indptr[i]) return fail(BDG_EINVAL, "indptr is not monotone at row %lld", (long long)i);
    return BDG_OK;
}

int bdg_host_scan_blocks(const double* data, const int32_t* indptr, int64_t nb, uint8_t* nonzero, int64_t* n_nonzero,
                         double* ph_defect, double* row_sum_max, int32_t* all_real) {
    if (int rc = check_indptr(indptr, nb)) return rc;
    if (indptr[nb] > 0 && !data) return fail(BDG_EINVAL, "null argument");
    const bdg_host::BlockScan scan = bdg_host::scan_blocks(data, indptr, nb, nonzero);
    const double nan = std::nan("");
    if (n_nonzero) *n_nonzero = scan.n_nonzero;
    if (ph_defect) *ph_defect = scan.has_nan ? nan : scan.ph_defect;
    if (row_sum_max) *row_sum_max = scan.has_nan ? nan : scan.row_sum_max;
    if (all_real) *all_real = scan.all_real ? 1 : 0;
    return BDG_OK;
}

int bdg_host_compact_blocks(const double* data, const int32_t* indices, const int32_t* indptr, int64_t nb,
                            const uint8_t* keep, double* data_out, int32_t* indices_out, int32_t* indptr_out) {
    if (int rc = check_indptr(indptr, nb)) return rc;
    if (!indptr_out || (indptr[nb] > 0 && (!data || !indices || !keep))) return fail(BDG_EINVAL, "null argument");
    int64_t kept = 0;
    for (int64_t k = 0; k < indptr[nb]; ++k) kept += keep[k] ? 1 : 0;
    if (kept > 0 && (!data_out || !indices_out)) return fail(BDG_EINVAL, "null output");
    bdg_host::compact_blocks(data, indices, indptr, nb, keep, data_out, indices_out, indptr_out);
    return BDG_OK;
}

int bdg_dense_prefetch(void) {
    g_solver_prefetch.start();
    return BDG_OK;
}

int bdg_dense_prefetch_wait(double timeout_seconds, int32_t* ready) {
    if (!ready) return fail(BDG_EINVAL, "null ready pointer");
    *ready = g_solver_prefetch.wait(timeout_seconds) ? 1 : 0;
    return BDG_OK;
}

int bdg_rccl_prefetch(void) {
    g_rccl_prefetch.start();
    return BDG_OK;
}

int bdg_rccl_prefetch_wait(double timeout_seconds, int32_t* ready) {
    if (!ready) return fail(BDG_EINVAL, "null ready pointer");
    *ready = g_rccl_prefetch.wait(timeout_seconds) ? 1 : 0;
    return BDG_OK;
}

int bdg_lanczos_begin(bdg_system* sys, int32_t n_vectors, uint64_t seed, uint64_t first_vec_id,
                      int32_t vec_kind, int32_t max_iter) {
    if (!sys) return fail(BDG_EINVAL, "null system handle");
    if (vec_kind != BDG_VEC_RADEMACHER && vec_kind != BDG_VEC_Z4)
        return fail(BDG_EINVAL, "unknown start-vector kind %d", vec_kind);
    StartSpec start{StartKind::Random};
    start.seed = seed;
    start.first_id = first_vec_id;
    start.vec_kind = vec_kind;
    return lanczos_begin(sys, n_vectors, start, max_iter);
}

int bdg_lanczos_advance(bdg_system* sys, int32_t n_iter, double* alpha_out, double* beta_out) {
    if (!sys || !alpha_out || !beta_out) return fail(BDG_EINVAL, "null argument");
    return lanczos_advance(sys, n_iter, alpha_out, beta_out);
}

int bdg_lanczos_ritz_vectors(bdg_system* sys, int32_t n_iter, int32_t n_levels, const double* coef, double* y_out) {
    if (!sys || !coef || !y_out) return fail(BDG_EINVAL, "null argument");
    return lanczos_ritz_vectors(sys, n_iter, n_levels, coef, y_out);
}

int bdg_perf_query(bdg_system* sys, bdg_perf* out) {
    if (!sys || !out) return fail(BDG_EINVAL, "null argument");
    *out = sys->perf;
    return BDG_OK;
}

int bdg_comm_unique_id(uint8_t id_out[128]) {
    if (!id_out) return fail(BDG_EINVAL, "null id buffer");
    RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(api, api->get_unique_id(&id));
    memcpy(id_out, &id, 128);
    return BDG_OK;
}

int bdg_comm_init(int device, const uint8_t id[128], int32_t n_ranks, int32_t rank, bdg_comm** out) {
    if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(BDG_EINVAL, "bad communicator arguments");
    *out = nullptr;
    RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    HIP_TRY(hipSetDevice(device));
    bdg_comm* comm = new bdg_comm();
    comm->device = device;
    comm->n_ranks = n_ranks;
    comm->rank = rank;
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    if (hipStreamCreateWithFlags(&comm->stream, hipStreamNonBlocking) != hipSuccess) {
        delete comm;
        return fail(BDG_EDEVICE, "stream creation failed");
    }
    ncclResult_t res = api->comm_init_rank(&comm->comm, n_ranks, uid, rank);
    if (res != ncclSuccess) {
        (void)hipStreamDestroy(comm->stream);
        delete comm;
        return fail(BDG_ELIBRARY, "ncclCommInitRank failed: %s", api->error_string(res));
    }
    *out = comm;
    return BDG_OK;
}

int bdg_comm_info(bdg_comm* comm, int32_t* n_ranks, int32_t* rank, int32_t* device, char pci_bus_id[32]) {
    if (!comm || !n_ranks || !rank || !device || !pci_bus_id) return fail(BDG_EINVAL, "null argument");
    RcclApi* api = nullptr;
    if (int rc = load_rccl(&api)) return rc;
    int count = 0;
    NCCL_TRY(api, api->comm_count(comm->comm, &count));  // what RCCL itself says, not what was asked for
    *n_ranks = count;
    *rank = comm->rank;
    *device = comm->device;
    HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, 32, comm->device));
    return BDG_OK;
}

int bdg_comm_allreduce_sum(bdg_comm* comm, double* buf, int64_t count) {
    return comm_allreduce(comm, buf, count, ncclSum);
}

int bdg_comm_allreduce_max(bdg_comm* comm, double* buf, int64_t count) {
    return comm_allreduce(comm, buf, count, ncclMax);
}

int bdg_comm_destroy(bdg_comm* comm) {
    if (!comm) return BDG_OK;
    RcclApi* api = nullptr;
    (void)hipSetDevice(comm->device);
    if (load_rccl(&api) == BDG_OK && comm->comm) (void)api->comm_destroy(comm->comm);
    comm->scratch.release();
    if (comm->stream) (void)hipStreamDestroy(comm->stream);
    delete comm;
    return BDG_OK;
}

}  // extern "C"
