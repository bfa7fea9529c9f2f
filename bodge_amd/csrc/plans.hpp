// plans.hpp - kernel dispatch tables and launch plans of the recurrence kernels (one-step, sweeps, 3-D rolling)
// Part of the single translation unit bodge_hip.hip (included there, in this order:
// core, plans, libraries, recurrence, lanczos, dense); everything lives in its unnamed namespace.
#pragma once

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
// ... and when the on-site blocks are streamed (cheb_sweep3 OS): two workgroups per CU of four waves,
// each wave with 12 KB of hand-over rows and a 3.75 / 5.25 KB ring of on-site blocks, leave this much
constexpr size_t kStreamedTableLimit = 10 * 1024;

size_t table_limit(const bdg_system* sys) { return sys->onsite_streamed ? kStreamedTableLimit : kDictLdsLimit; }

// Dictionary kernel if the matrix has few enough distinct blocks for the table to sit in LDS.
StepKernel dict_kernel(const bdg_system* sys, const ModeInfo& mode, int rl) {
    const char* env = knob::raw("BODGE_AMD_DICT");
    if (env && env[0] == '0') return nullptr;
    if (sys->onsite_streamed) return nullptr;  // (the table lacks the diagonal blocks: only the three-step sweep can use it)
    if (sys->n_unique <= 0 || (size_t)sys->n_unique * mode.stride * sizeof(double2) > kDictLdsLimit)
        return nullptr;
    if (sys->max_row_blocks <= 3) return dict_for<3>(mode, rl);
    if (sys->max_row_blocks <= 5) return dict_for<5>(mode, rl);
    if (sys->max_row_blocks <= 7) return dict_for<7>(mode, rl);
    return nullptr;
}

StepKernel pipelined_kernel(const ModeInfo& mode, int rl, int max_row_blocks, int* maxb_out) {
    const char* env = knob::raw("BODGE_AMD_KERNEL");
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
    if (const char* cap = knob::raw("BODGE_AMD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(cap));
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

// Packed on-site records for the sweep that streams them (kernels.hpp pack_onsite; built once per arithmetic).
int ensure_onsite(bdg_system* sys, bool real, const double2** out) {
    DeviceBuffer<double2>& buf = sys->onsite[real ? 1 : 0];
    if (!buf.ptr) {
        if (int rc = buf.reserve((size_t)sys->nb * (real ? RealPHMode::kOnsiteSlots : ComplexPHMode::kOnsiteSlots))) return rc;
        bdg::pack_onsite<<<(unsigned)std::min<int64_t>(4096, (sys->nb + 255) / 256), 256, 0, sys->stream>>>(
            sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, (int)sys->nb, real ? 1 : 0, buf.ptr);
        HIP_TRY(hipGetLastError());
    }
    *out = buf.ptr;
    return BDG_OK;
}

// Per-site records (on-site + four bond blocks) of a matrix whose bond blocks are streamed too; needs the lattice shape.
int ensure_site_records(bdg_system* sys, int plane, bool real, const double2** out) {
    DeviceBuffer<double2>& buf = sys->site_records[real ? 1 : 0];
    int& built_for = sys->site_records_plane[real ? 1 : 0];
    if (!buf.ptr || built_for != plane) {
        if (int rc = buf.reserve((size_t)sys->nb * (real ? 8 : 14))) return rc;
        bdg::pack_site_records<<<(unsigned)std::min<int64_t>(4096, (sys->nb + 255) / 256), 256, 0, sys->stream>>>(
            sys->indptr.ptr, sys->indices.ptr, sys->blocks.ptr, (int)sys->nb, plane, real ? 1 : 0, buf.ptr);
        HIP_TRY(hipGetLastError());
        built_for = plane;
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
        args->dict_ell = sys->dict_ell.ptr;
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
    if (const char* env = knob::raw("BODGE_AMD_L2_BUDGET")) budget = atof(env);
    int64_t strip = (int64_t)(budget / (2.0 * row_bytes));
    strip = std::max<int64_t>(rows_per_tile, strip / rows_per_tile * rows_per_tile);
    if (strip >= plane || budget <= 0) return BDG_OK;
    *strip_out = (int)strip;
    for (const auto& cached : sys->tile_orders)
        if (cached->rows_per_tile == rows_per_tile && cached->strip_rows == strip && cached->ids.count >= (size_t)n_tiles) {
            *order_out = cached->ids.ptr;
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
    if (sys->tile_orders.size() >= 16) {  // (a handle driven through very many shapes: start over, once nothing reads the old ones)
        HIP_TRY(hipStreamSynchronize(sys->stream));
        for (auto& side : sys->side_sets) HIP_TRY(hipStreamSynchronize(side->stream));
        for (auto& cached : sys->tile_orders) cached->ids.release();
        sys->tile_orders.clear();
    }
    auto fresh = std::make_unique<bdg_system::TileOrder>();
    if (int rc = fresh->ids.reserve((size_t)n_tiles)) return rc;
    // (a new buffer: the blocking copy cannot disturb launches of earlier batches, which read other buffers)
    HIP_TRY(hipMemcpy(fresh->ids.ptr, order.data(), sizeof(int) * n_tiles, hipMemcpyHostToDevice));
    fresh->rows_per_tile = rows_per_tile;
    fresh->strip_rows = (int)strip;
    *order_out = fresh->ids.ptr;
    sys->tile_orders.push_back(std::move(fresh));
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
int choose_segments(int n_cols, int lx, int waves, int extra_planes, int min_planes, double keep = 0.97) {
    int best = 1;
    double best_cost = 0.0;
    for (int segs = 1; segs <= std::max(1, lx / min_planes); ++segs) {
        const int64_t units = (int64_t)n_cols * segs;
        const double rounds = (double)((units + waves - 1) / waves);
        const double cost = rounds * ((double)((lx + segs - 1) / segs) + extra_planes);
        if (segs == 1 || cost < keep * best_cost) {
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

// cheb_sweep3 with streamed on-site blocks (particle-hole modes); `bonds`: the bond blocks as well.  The forms built:
//   on-site records, real:     4 lanes x 4 waves (two workgroups per CU), 2 lanes x 4 waves (two per CU while the bond table
//                              is small) and 2 lanes x 7 waves (one per CU)
//   on-site records, complex:  4 lanes x 4 waves, 2 lanes x 7 waves
//   site records (bonds too):  real 4 lanes x 4 waves, complex 4 lanes x 7 waves
template <typename Mode, int RL, int OS, int WAVES>
SweepKernel sweep3_streamed_pick(bool reverse, bool gen) {
    return gen ? bdg::cheb_sweep3<Mode, RL, false, true, OS, WAVES>
               : reverse ? bdg::cheb_sweep3<Mode, RL, true, false, OS, WAVES> : bdg::cheb_sweep3<Mode, RL, false, false, OS, WAVES>;
}
SweepKernel sweep3_streamed_kernel(const ModeInfo& mode, int lanes, int waves, bool reverse, bool gen, bool bonds = false) {
    if (!mode.ph) return nullptr;
    if (bonds) {
        if (mode.real) return lanes == 4 && waves == 4 ? sweep3_streamed_pick<RealPHMode, 4, 2, 4>(reverse, gen) : nullptr;
        return lanes == 4 && waves == 7 ? sweep3_streamed_pick<ComplexPHMode, 4, 2, 7>(reverse, gen) : nullptr;
    }
    if (mode.real) {
        if (lanes == 4 && waves == 4) return sweep3_streamed_pick<RealPHMode, 4, 1, 4>(reverse, gen);
        if (lanes == 2 && waves == 4) return sweep3_streamed_pick<RealPHMode, 2, 1, 4>(reverse, gen);
        if (lanes == 2 && waves == 7) return sweep3_streamed_pick<RealPHMode, 2, 1, 7>(reverse, gen);
        return nullptr;
    }
    if (lanes == 4 && waves == 4) return sweep3_streamed_pick<ComplexPHMode, 4, 1, 4>(reverse, gen);
    if (lanes == 2 && waves == 7) return sweep3_streamed_pick<ComplexPHMode, 2, 1, 7>(reverse, gen);
    return nullptr;
}

// LDS of one workgroup of cheb_sweep3: table + compact diagonals + per wave three hand-over rows and (streamed forms)
// the ring of three planes of records.
size_t sweep3_lds_bytes(const bdg_system* sys, const ModeInfo& mode, int lanes, int waves) {
    // table + compact diagonals + compact "singlet" copies (RealPHMode::kSingletSlots = 3 slots per block; the other modes none)
    size_t bytes = (size_t)sys->n_unique * mode.stride * sizeof(double2) +
                   (size_t)sys->n_unique * (mode.id == 0 ? 4 : mode.id == 3 ? 1 : 2) * sizeof(double2) +
                   (size_t)sys->n_unique * (mode.id == 3 ? RealPHMode::kSingletSlots : 0) * sizeof(double2);
    size_t per_wave = (size_t)3 * bdg::kWave * 4 * sizeof(double2);
    if (sys->onsite_streamed) {
        const int stride = sys->bonds_streamed ? (mode.real ? bdg::sweep3_record_stride<RealPHMode, 2>() : bdg::sweep3_record_stride<ComplexPHMode, 2>())
                                               : (mode.real ? bdg::sweep3_record_stride<RealPHMode, 1>() : bdg::sweep3_record_stride<ComplexPHMode, 1>());
        per_wave += (size_t)3 * (bdg::kWave / lanes) * stride * sizeof(double2);
    }
    return bytes + (size_t)waves * per_wave;
}

// Waves per workgroup of the streamed form with `lanes` lanes per site (0 = that form does not exist or does not fit
// the LDS of this device): the shape that keeps the most waves on a CU.
int streamed_waves_for(const bdg_system* sys, const ModeInfo& mode, int lanes) {
    if (!sys->onsite_streamed || !mode.ph) return 0;
    int best = 0, best_resident = 0;
    for (int waves : {4, 7}) {
        if (!sweep3_streamed_kernel(mode, lanes, waves, false, false, sys->bonds_streamed)) continue;
        const size_t lds = sweep3_lds_bytes(sys, mode, lanes, waves);
        if (lds > sys->lds_per_cu) continue;
        const int resident = waves * (int)std::min<size_t>(waves > 4 ? 1 : 2, sys->lds_per_cu / lds);
        if (resident > best_resident) best = waves, best_resident = resident;
    }
    return best;
}

SweepKernel sweep3_kernel(const ModeInfo& mode, int lanes, bool reverse) {
    switch (mode.id) {
        case 1: return sweep3_kernel_for<RealMode>(lanes, reverse);
        case 2: return sweep3_kernel_for<ComplexPHMode>(lanes, reverse);
        case 3: return sweep3_kernel_for<RealPHMode>(lanes, reverse);
    }
    return sweep3_kernel_for<ComplexMode>(lanes, reverse);
}

// cheb_march3: the sweeps of a whole reduction chunk in one launch (sweep.hpp K7c)
using MarchKernel = void (*)(bdg::MarchArgs);
template <typename Mode>
MarchKernel march3_kernel_for(int lanes) {
    switch (lanes) {
        case 2: return bdg::cheb_march3<Mode, 2, 0>;
        case 4: return bdg::cheb_march3<Mode, 4, 0>;
    }
    return nullptr;
}
MarchKernel march3_kernel(const ModeInfo& mode, int lanes, bool streamed, bool bonds) {
    if (streamed) {
        if (!mode.ph || lanes != 4) return nullptr;
        if (bonds) return mode.real ? bdg::cheb_march3<RealPHMode, 4, 2> : nullptr;
        return mode.real ? bdg::cheb_march3<RealPHMode, 4, 1> : bdg::cheb_march3<ComplexPHMode, 4, 1>;
    }
    switch (mode.id) {
        case 1: return march3_kernel_for<RealMode>(lanes);
        case 2: return march3_kernel_for<ComplexPHMode>(lanes);
        case 3: return march3_kernel_for<RealPHMode>(lanes);
    }
    return march3_kernel_for<ComplexMode>(lanes);
}

struct SweepPlan {
    int lanes = bdg::kSweepLanes;
    int depth = 2;  // recurrence steps per sweep: 2 (cheb_sweep) or 3 (cheb_sweep3)
    SweepKernel kernel = nullptr, kernel_reverse = nullptr;
    SweepKernel kernel_gen = nullptr;  // depth 3: first sweep of a random-start run, t_0 made in registers
    MarchKernel march = nullptr;       // depth 3: all sweeps of a reduction chunk in one launch (nullptr = not available)
    int march_grid = 0;                // workgroups of such a launch
    int grid = 0;
    int wg_waves = bdg::kSweepWaves;   // waves per workgroup (the streamed forms with one workgroup per CU: 7)
    size_t lds_bytes = 0;
    bdg::SweepArgs args{};
    int waves = 0;  // resident waves of a launch (what a launch over a band of planes cuts its segments for)
    int whole_segs = 1;  // segments of a launch over the whole lattice
    // segments of a launch over planes [lo, hi): as for the whole lattice, for that many planes
    int segments_for(int planes) const {
        if (planes >= args.lx) return whole_segs;
        // (a band of few planes: short segments - a launch that leaves waves idle is as long as one wave's march)
        return std::max(1, std::min(choose_segments(args.n_cols, planes, waves, 2 * depth, 2), planes / 2));
    }
};

// Smallest lattices the stencil kernels are chosen for by default.  Below, the x-segments get so
// short that the planes each wave recomputes at their ends eat the saving, and the one-step
// kernels work from the Infinity Cache with wide batches.  Measured with 64-vector calls
// (profiles/r02_sweep_experiments.log): 300x300 one-step 725 k vector-steps/s vs 504 k, 400x400 407 k
// vs 458 k, 500x500 252 k vs 326 k, 700x700 130 k vs 193 k, 1000x1000 57 k vs 104 k.
// BODGE_AMD_SWEEP=0 never, =1 whenever the matrix qualifies.
constexpr int64_t kSweepMinSites = 150000;   // 2-D: multi-step sweeps (K7, K7b)
constexpr int64_t kSweepMinSitesWide = 90000;  // ... for calls of more than one lane group (two streams)
constexpr int64_t kRollMinSites = 600000;    // 3-D: rolling one-step kernel (K8)
constexpr int64_t kSweepTwoLaneSites = 250000;  // from here on 2 lanes per site beat 4 (two lane groups side by side)

// Stencil table of the matrix (built once per lattice shape).  *kind = 1: 5-point stencil whose
// planes are lines (2-D lattice: the two-steps-per-sweep kernel applies), 2: 7-point stencil of a
// 3-D lattice (one-step kernel with the x-neighbours in registers), 0: neither.
int ensure_stencil(bdg_system* sys, int* kind) {
    *kind = 0;
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    if (sys->stencil_state == 0) {
        sys->stencil_state = -1;
        sys->stencil_lo_base = sys->stencil_hi_base = -1;
        const bool shaped = plane >= 2 * bdg::kSweepOwned && sys->shape[0] >= 8 &&
                            (int64_t)sys->shape[0] * plane == sys->nb;
        const bool three_d = sys->shape[1] > 1 && sys->shape[2] > 1;
        // A row slab (ncols > nb) of a 3-D lattice qualifies if its halo columns are exactly the plane below
        // its first plane and / or the plane above its last one, referenced position by position.
        bool slab_ok = sys->ncols == sys->nb;
        if (!slab_ok && shaped && three_d) {
            slab_ok = true;
            const int64_t lx = sys->shape[0];
            int64_t lo = -1, hi = -1;
            for (const auto& ref : sys->halo_refs) {
                const int64_t x = ref.first / plane, p = ref.first % plane, base = (int64_t)ref.second - p;
                if (x == 0 && (lo < 0 || lo == base)) lo = base;
                else if (x == lx - 1 && (hi < 0 || hi == base)) hi = base;
                else slab_ok = false;
            }
            for (int64_t base : {lo, hi})
                if (base >= 0 && (base < sys->nb || base + plane > sys->ncols)) slab_ok = false;
            if (slab_ok) {
                sys->stencil_lo_base = (int)lo;
                sys->stencil_hi_base = (int)hi;
            }
        }
        if (shaped && slab_ok && sys->n_unique > 0 && sys->n_unique < (int)bdg::kNoBlock &&
            sys->max_row_blocks <= (three_d ? 7 : 5) && sys->nnzb > 0 && !(three_d && sys->onsite_streamed)) {
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
                                                                        sys->shape[2], sys->stencil_lo_base, sys->stencil_hi_base,
                                                                        sys->stencil.ptr, bad.ptr);
                else
                    bdg::build_stencil<<<grid, 256, 0, sys->stream>>>(sys->indptr.ptr, sys->dict_ids.ptr,
                                                                       sys->dict_diagonal.ptr, (int)sys->nb, (int)plane,
                                                                       sys->bonds_streamed ? 2 : sys->onsite_streamed ? 1 : 0,
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

inline int64_t plane_sites(const bdg_system* sys) { return (int64_t)sys->shape[1] * sys->shape[2]; }

// Should this batch run a stencil form, and which (see ensure_stencil)?  Whole square matrix (or a slab of a
// same-process group), no per-column scalars; unit start vectors when their band of planes gets wide
// (unit_run_wants_stencil).
// `wide_call`: the call has more vectors than one lane group (run_recurrence): its groups run side by side on two streams,
// and the multi-step sweeps then beat the one-step kernels from 300 x 300 sites on (16 / 64 vectors: 852 / 880 against
// 690 / 729 k vector-steps/s; 250 x 250: 919 against 994 k; a single group of 8 vectors wins from ~390^2 only - 350^2:
// 427 against 466 k, 400^2: 470 against 379 k; profiles/r03_sweep_threshold.log).
int sweep_wanted(bdg_system* sys, bool col_scalars, int* kind, bool wide_call = false) {
    *kind = 0;
    const char* env = knob::raw("BODGE_AMD_SWEEP");
    if ((env && env[0] == '0') || col_scalars) return BDG_OK;
    // Row slabs: members of a same-process group may run the 3-D rolling kernel, which reads the neighbouring
    // slabs' boundary planes where they are; slabs exchanging halos through RCCL keep the one-step kernels
    // (their exchange is overlapped with the rows that do not need it, Batch::step_overlapped).
    const bool slab = !sys->peers.empty() || sys->ncols != sys->nb;
    if (slab && (sys->slab_comm || sys->group_rows == 0)) return BDG_OK;
    const bool forced = env && env[0] == '1';
    const int64_t lattice_rows = std::max(sys->nb, sys->group_rows);  // (the size of the lattice, not of the slab, decides)
    const int64_t min_2d = wide_call ? kSweepMinSitesWide : kSweepMinSites;
    if (!forced && lattice_rows < std::min(min_2d, kRollMinSites)) return BDG_OK;
    const char* dict_env = knob::raw("BODGE_AMD_DICT");
    if (dict_env && dict_env[0] == '0') return BDG_OK;
    if (int rc = ensure_stencil(sys, kind)) return rc;
    if (slab && *kind != 2) *kind = 0;  // (2-D multi-step sweeps would need three-plane halos: not built)
    if (!forced && lattice_rows < (*kind == 2 ? kRollMinSites : min_2d)) *kind = 0;
    return BDG_OK;
}

// `share`: launches of this many batches are in flight together (run_recurrence runs the lane groups of a call side
// by side on as many streams): each gets that fraction of the wave slots, i.e. fewer, longer segments - 1000x1000, two
// streams: 26 segments of 38 planes instead of 52 of 19 (1014 units per launch, two launches fill the 2048 slots) is
// 116.7 against 111.8 k vector-steps/s and recomputes half as many planes (profiles/r03_segments.log).
int make_sweep_plan(bdg_system* sys, const ModeInfo& mode, int lanes, int depth, SweepPlan* plan, int share = 1) {
    plan->lanes = lanes;
    plan->depth = depth;
    const bool streamed = sys->onsite_streamed, bonds = sys->bonds_streamed;
    const int wg_waves = streamed ? streamed_waves_for(sys, mode, lanes) : bdg::kSweepWaves;
    if (streamed && (depth != 3 || wg_waves == 0))
        return fail(BDG_EINVAL, "streamed on-site blocks need the three-step sweep, particle-hole packed blocks and a lane count whose ring of records fits the LDS (%d lanes do not)", lanes);
    plan->wg_waves = wg_waves;
    plan->kernel = streamed ? sweep3_streamed_kernel(mode, lanes, wg_waves, false, false, bonds)
                            : depth == 3 ? sweep3_kernel(mode, lanes, false) : sweep_kernel(mode, lanes, false);
    plan->kernel_reverse = streamed ? sweep3_streamed_kernel(mode, lanes, wg_waves, true, false, bonds)
                                    : depth == 3 ? sweep3_kernel(mode, lanes, true) : sweep_kernel(mode, lanes, true);
    plan->kernel_gen = streamed ? sweep3_streamed_kernel(mode, lanes, wg_waves, false, true, bonds) : depth == 3 ? sweep3_gen_kernel(mode, lanes) : nullptr;
    if (!plan->kernel) return fail(BDG_EINVAL, "the sweep kernel has 1, 2 or 4 lanes per site, not %d", lanes);
    const size_t table = (size_t)sys->n_unique * mode.stride * sizeof(double2);
    if (table > table_limit(sys)) return fail(BDG_EINVAL, "block table too large for the sweep kernel");
    if (depth == 3) {
        plan->lds_bytes = sweep3_lds_bytes(sys, mode, lanes, wg_waves);
    } else {
        plan->lds_bytes = table + (size_t)bdg::kSweepWaves * depth * bdg::kWave * 4 * sizeof(double2);
    }
    if (plan->lds_bytes > 64 * 1024)
        for (SweepKernel k : {plan->kernel, plan->kernel_reverse, plan->kernel_gen})
            if (k) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)plan->lds_bytes));
    // one launch per chunk: the write-through stores address a buffer through a 32-bit byte offset
    plan->march = depth == 3 && wg_waves == bdg::kWavesPerBlock &&
                          (size_t)sys->ncols * lanes * 4 * sizeof(double2) < ((size_t)1 << 31) && sys->ncols == sys->nb
                      ? march3_kernel(mode, lanes, streamed, bonds) : nullptr;
    if (plan->march && plan->lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(plan->march), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)plan->lds_bytes));
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(plan->kernel),
                                                         wg_waves * bdg::kWave, plan->lds_bytes));
    per_cu = std::max(1, std::min(per_cu, wg_waves > 4 ? 1 : 2 * bdg::kWavesPerBlock / bdg::kSweepWaves));
    if (const char* cap = knob::raw("BODGE_AMD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(cap));
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    bdg::SweepArgs& a = plan->args;
    a = bdg::SweepArgs{};
    a.stencil = sys->stencil.ptr;
    if (int rc = ensure_dict_table(sys, mode, &a.dict_table)) return rc;
    if (bonds) {
        if (int rc = ensure_site_records(sys, (int)plane, mode.real, &a.onsite)) return rc;
    } else if (streamed) {
        if (int rc = ensure_onsite(sys, mode.real, &a.onsite)) return rc;
    }
    a.n_unique = sys->n_unique;
    a.nb = (int)sys->nb;
    a.plane = (int)plane;
    a.lx = sys->shape[0];
    const int owned = depth == 3 ? bdg::sweep3_owned(lanes) : bdg::sweep_owned(lanes);
    a.n_cols = (int)((plane + owned - 1) / owned);
    // one unit (segment x window) per resident wave, segments of at least 8 planes
    // (streamed forms: shared as the others since round 4 - real on-site records +2 %, 104.6 against 102.7 k vector-steps/s,
    // complex +1 %; the forms with one workgroup of seven waves per CU lose 3 % with half the segments and keep them all;
    // BODGE_AMD_STREAMED_SHARE=0: never)
    if (const char* env = knob::raw("BODGE_AMD_STREAMED_SHARE"); streamed && (wg_waves > 4 || (env && env[0] == '0'))) share = 1;
    const int waves = std::max(wg_waves, per_cu * sys->num_cus * wg_waves / std::max(1, share));
    // (launches side by side: the count that fills the share of the slots - 26 against 25 segments is the 2 % the model says)
    int n_segs = choose_segments(a.n_cols, a.lx, waves, 2 * depth, 8, share > 1 ? 0.985 : 0.97);
    if (const char* env = knob::raw("BODGE_AMD_SWEEP_SEGMENTS")) n_segs = atoi(env);
    a.n_segs = std::max(1, std::min(n_segs, a.lx / 8));
    a.x_lo = 0;
    a.x_hi = a.lx;
    plan->waves = waves;
    plan->whole_segs = a.n_segs;
    a.zigzag = 1;
    if (const char* env = knob::raw("BODGE_AMD_SWEEP_ZIGZAG")) a.zigzag = atoi(env) != 0;
    a.wrap_p = sys->stencil_wrap_p ? 1 : 0;
    a.wrap_x = sys->stencil_wrap_x ? 1 : 0;
    const int64_t units = (int64_t)a.n_cols * a.n_segs;
    const int grid = (int)std::min<int64_t>((int64_t)per_cu * sys->num_cus, (units + wg_waves - 1) / wg_waves);
    plan->grid = std::max(8, (grid + 7) / 8 * 8);
    // (a chunk in one launch: `share` lane groups are tasks of the same launch - every wave slot of the device)
    const int march_grid = (int)std::min<int64_t>((int64_t)per_cu * sys->num_cus,
                                                  (units * std::max(1, share) + wg_waves - 1) / wg_waves);
    plan->march_grid = std::max(8, (march_grid + 7) / 8 * 8);
    return BDG_OK;
}

// Algorithmic HBM bytes of one two-step sweep: one 8-byte stencil word per site, the block table
// once, and four passes over 4 x RL 16-byte payloads per site (read t_n, t_{n-1}; write t_{n+1},
// t_{n+2}).  The halo slots and segment-end planes the waves recompute are NOT counted.
// With streamed on-site blocks each site adds its packed record (64 B real / 96 B complex), read once.
double sweep_bytes(const bdg_system* sys, const ModeInfo& mode, int lanes) {
    const double onsite = !sys->onsite_streamed ? 0.0
                          : sys->bonds_streamed ? (mode.real ? 128.0 : 224.0)  // on-site + four bond blocks per site
                          : 16.0 * (mode.real ? RealPHMode::kOnsiteSlots : ComplexPHMode::kOnsiteSlots);
    return (8.0 + onsite) * (double)sys->nb + (mode.block_bytes - 4.0) * sys->n_unique +
           4.0 * (4.0 * lanes * sizeof(double2)) * (double)sys->nb;
}

// Lanes per site (vectors per launch) of the sweep kernel.  4 lanes move the fewest redundant
// bytes per vector-step; with 1 lane (2 real / 1 complex vector per launch) the four buffers of a
// run are a quarter the size, and when they then fit the 256 MB Infinity Cache together
// (4 x 64 B x sites + the stencil words <= ~252 MB: up to ~10^6 sites) every launch after the
// first streams from that cache instead of HBM.  BODGE_AMD_SWEEP_LANES overrides.
// Steps per sweep: 3 (cheb_sweep3, with 2 or 4 lanes per site) moves 4/9 of the one-step kernels'
// bytes against 2/3 for 2 (cheb_sweep, which also runs with 1 lane).  BODGE_AMD_SWEEP_STEPS=2|3 overrides.
int sweep_depth_for(const bdg_system* sys, int lanes) {
    if (sys->onsite_streamed) return 3;  // (the only form that streams on-site blocks)
    int depth = lanes >= 2 ? 3 : 2;
    if (const char* env = knob::raw("BODGE_AMD_SWEEP_STEPS")) {
        const int forced = atoi(env);
        if (forced == 2 || (forced == 3 && lanes >= 2)) depth = forced;
    }
    return depth;
}

// Default lanes per site: 2.  Fewer lanes mean wider windows (the 3-step kernel owns 26 of 32
// slots with 2 lanes, 10 of 16 with 4: less recomputed halo per useful site) at the price of
// shorter x-segments.  Measured on 1000x1000, 8 real vectors: 2 lanes 106.6 k vector-steps/s,
// 4 lanes 99.9 k (profiles/r02_sweep_experiments.log).
// With the lane groups of a call side by side on two streams (run_recurrence) 2 lanes win from 2.5e5 sites (500^2, 16
// vectors: 386 against 368 k vector-steps/s; 400^2: 468 against 537 k), and a call of ONE 4-lane group (5-8 real
// vectors) is better cut into two 2-lane groups from the smallest swept lattices on (400^2 / 500^2 / 600^2, 8 vectors:
// 466 / 388 / 271 k against 447 / 319 / 240 k; profiles/r03_lanes_midsize.log).  `whole_call`: n_active counts the
// call, which batch_width then cuts; a batch (begin) takes the lanes that hold it.
int sweep_lanes_for(const bdg_system* sys, int n_active, int per_lane, bool unit_start = false, bool whole_call = false) {
    if (sys->onsite_streamed) {
        // real arithmetic, on-site records: 2 lanes per site (32-slot windows, 19 % halo instead of 38 %; two lane groups
        // of 4 vectors side by side: 100.4 against 94.5 k vector-steps/s on 1000 x 1000) while the bond table leaves room
        // for two workgroups per CU; complex arithmetic keeps 4 lanes (the 2-lane form needs one workgroup of seven
        // waves per CU for its ring: 40.1 against 41.9 k), and so do the site records (bond blocks too)
        const ModeInfo mode = mode_info(per_lane == 2, true);
        int lanes = mode.real && !sys->bonds_streamed && streamed_waves_for(sys, mode, 2) == 4 ? 2 : 4;
        if (const char* env = knob::raw("BODGE_AMD_SWEEP_LANES")) {
            const int forced = atoi(env);
            if ((forced == 2 || forced == 4) && streamed_waves_for(sys, mode, forced) != 0) lanes = forced;
        }
        return lanes;
    }
    if (const char* env = knob::raw("BODGE_AMD_SWEEP_LANES")) {
        const int forced = atoi(env);
        if (forced == 1 || forced == 2 || forced == 4) return forced;
    }
    // unit start vectors run inside a band of planes for most of their steps: launches of few planes, bound by the
    // length of a wave's march and not by bytes - all the vectors of a lane group in one launch, then
    if (unit_start && n_active > 2 * per_lane) return 4;
    if (sys->nb >= kSweepTwoLaneSites) return 2;
    if (sys->nb >= kSweepMinSites && n_active <= (whole_call ? 4 : 2) * per_lane) return 2;
    return 4;  // small lattices: more work per launch matters more
}

// Unit start vectors (LDOS) spread by one plane per step.  Inside a narrow band of planes a stencil launch is as long
// as a wave's march over its segment (a few tens of microseconds), where a one-step launch over the band's rows
// takes a few: the stencil kernels pay when the band ends up wide.  Rule (scratch/r3_unit_band.py): the band of the
// last step covers at least 30 % of the planes.
bool unit_run_wants_stencil(const bdg_system* sys, const int64_t* rows, int count, int n_steps) {
    const char* env = knob::raw("BODGE_AMD_SWEEP");
    if (env && env[0] == '1') return true;
    if (sys->ncols != sys->nb || sys->group_rows != 0 || sys->shape[0] <= 0) return true;  // (slabs: no band either way)
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    int64_t lo = sys->shape[0], hi = 0;
    for (int r = 0; r < count; ++r) {
        lo = std::min<int64_t>(lo, (rows[r] >> 2) / plane);
        hi = std::max<int64_t>(hi, (rows[r] >> 2) / plane);
    }
    return (double)(hi - lo + 1 + 2 * (int64_t)n_steps) >= 0.3 * sys->shape[0];
}

// ---- 3-D: one step per launch with the x-neighbours in registers (cheb_roll3)
using RollKernel = void (*)(bdg::RollArgs);

template <typename Mode>
RollKernel roll_kernel_for(int lanes, bool nt) {
    if (lanes == 2) return nt ? bdg::cheb_roll3<Mode, 2, true> : bdg::cheb_roll3<Mode, 2, false>;
    if (lanes == 4) return nt ? bdg::cheb_roll3<Mode, 4, true> : bdg::cheb_roll3<Mode, 4, false>;
    return nullptr;
}

// `nt`: non-temporal t_{n-1} loads and t_{n+1} stores (vector pairs beyond the Infinity Cache)
RollKernel roll_kernel(const ModeInfo& mode, int lanes, bool nt = false) {
    switch (mode.id) {
        case 1: return roll_kernel_for<RealMode>(lanes, nt);
        case 2: return roll_kernel_for<ComplexPHMode>(lanes, nt);
        case 3: return roll_kernel_for<RealPHMode>(lanes, nt);
    }
    return roll_kernel_for<ComplexMode>(lanes, nt);
}

// Lanes per site of the rolling kernel: 4 (8 real vectors per launch, windows of 14 owned positions
// between two ghost slots), or 2 when the batch has no more vectors than two lanes carry (4 real /
// 2 complex: windows of 30 positions, and the launch no longer moves 8 vectors' worth of bytes
// for 4).  Measured on 100^3: 76.6 us per 4 vectors against 156.9 us per 8 - 2 lanes are 2 % faster
// per vector, but their L2-miss traffic is 1.27 x the algorithmic bytes against 1.08 x (shorter
// x-segments, more segment-end planes), so wide batches stay with 4.  BODGE_AMD_SWEEP_LANES overrides.
int roll_lanes_for(const bdg_system* sys, int n_vectors, int per_lane) {
    int lanes = n_vectors <= 2 * per_lane ? 2 : bdg::kSweepLanes;
    // (One lane group as two 2-lane batches side by side on two streams was tried for whole matrices: 54.2 against
    // 51.4 k vector-steps/s in a 63-step loop, nothing through bench.py, and 1.28 x instead of 1.08 x the algorithmic
    // traffic - twice the segment-end planes.  Not kept; calls of two lane groups and more run them side by side.)
    if (const char* env = knob::raw("BODGE_AMD_SWEEP_LANES")) {
        const int want = atoi(env);
        if (want == 2 || want == 4) lanes = want;
    }
    return lanes;
}

struct RollPlan {
    RollKernel kernel = nullptr;
    RollKernel kernel_nt = nullptr;  // the same with non-temporal t_{n-1} loads / t_{n+1} stores
    int lanes = bdg::kSweepLanes;
    int grid = 0;
    size_t lds_bytes = 0;
    bdg::RollArgs args{};
    int waves = 0;
    int whole_segs = 1;
    int share = 1;  // lane groups of the call in flight together (set by the batch)
    int segments_for(int planes) const {
        if (planes >= args.lx) return whole_segs;
        return std::max(1, std::min(choose_segments(args.n_cols, planes, waves, 2, 4), planes / 4));
    }
    // Plane-steps per unit when the units are cut for equal length (RollArgs.chunk), 0 = whole-column segments: used when
    // the segments leave more than a tenth of the wave slots empty or overfill them, and a run stays within two columns.
    int chunk_for(int planes) const {
        if (const char* env = knob::raw("BODGE_AMD_ROLL_CHUNKS"); env && env[0] == '0') return 0;
        // (`share` lane groups of the call run side by side on as many streams: each launch gets that part of the wave slots -
        // two launches cut for the whole device each were 3.5 % slower than whole-column segments)
        const int slots = std::max(1, waves / std::max(1, share));
        const int64_t total = (int64_t)args.n_cols * planes;
        const int64_t units = (int64_t)args.n_cols * segments_for(planes) * std::max(1, share);
        if (units >= (int64_t)(0.92 * waves) && units <= waves) return 0;
        const int chunk = (int)((total + slots - 1) / slots);
        return chunk >= 8 && chunk <= planes ? chunk : 0;
    }
};

int make_roll_plan(bdg_system* sys, const ModeInfo& mode, int lanes, RollPlan* plan) {
    plan->lanes = lanes;
    plan->kernel = roll_kernel(mode, lanes);
    plan->kernel_nt = roll_kernel(mode, lanes, true);
    if (!plan->kernel) return fail(BDG_EINVAL, "the rolling kernel has 2 or 4 lanes per site, not %d", lanes);
    const size_t table = (size_t)sys->n_unique * mode.stride * sizeof(double2);
    if (table > kDictLdsLimit) return fail(BDG_EINVAL, "block table too large for the rolling kernel");
    plan->lds_bytes = table + (size_t)bdg::kWavesPerBlock * bdg::kWave * 4 * sizeof(double2);
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(plan->kernel),
                                                         bdg::kBlockThreads, plan->lds_bytes));
    per_cu = std::max(1, std::min(per_cu, 2));
    if (const char* cap = knob::raw("BODGE_AMD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(cap));
    const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
    bdg::RollArgs& a = plan->args;
    a = bdg::RollArgs{};
    a.stencil = sys->stencil.ptr;
    if (int rc = ensure_dict_table(sys, mode, &a.dict_table)) return rc;
    a.n_unique = sys->n_unique;
    a.nb = (int)sys->nb;
    a.ld = (int)sys->ncols;
    a.plane = (int)plane;
    a.lz = sys->shape[2];
    a.lx = sys->shape[0];
    a.n_cols = (int)((plane + bdg::roll_owned(lanes) - 1) / bdg::roll_owned(lanes));
    const int waves = per_cu * sys->num_cus * bdg::kWavesPerBlock;
    int n_segs = choose_segments(a.n_cols, a.lx, waves, 2, 4);
    if (const char* env = knob::raw("BODGE_AMD_SWEEP_SEGMENTS")) n_segs = atoi(env);
    a.n_segs = std::max(1, std::min(n_segs, a.lx / 4));
    a.x_lo = 0;
    a.x_hi = a.lx;
    plan->waves = waves;
    plan->whole_segs = a.n_segs;
    const int64_t units = std::max<int64_t>((int64_t)a.n_cols * a.n_segs, plan->chunk_for(a.lx) > 0 ? waves : 0);
    const int grid = (int)std::min<int64_t>((int64_t)per_cu * sys->num_cus,
                                            (units + bdg::kWavesPerBlock - 1) / bdg::kWavesPerBlock);
    plan->grid = std::max(8, (grid + 7) / 8 * 8);
    return BDG_OK;
}

// Algorithmic bytes of one launch of the rolling kernel: stencil word + three passes per site.
double roll_bytes(const bdg_system* sys, const ModeInfo& mode, int lanes) {
    return 8.0 * (double)sys->nb + (mode.block_bytes - 4.0) * sys->n_unique +
           3.0 * (4.0 * lanes * sizeof(double2)) * (double)sys->nb;
}

enum class StartKind { Random, Unit };

struct StartSpec {
    StartKind kind;
    uint64_t seed = 0, first_id = 0;
    int vec_kind = 0;
    const int64_t* rows = nullptr;  // host
    int stencil = -1;  // unit starts: 1 / 0 = the call has decided for / against the stencil kernels, -1 = each batch decides
    bool wide_call = false;  // the call has more vectors than one lane group (its batches run side by side)
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

}  // namespace
