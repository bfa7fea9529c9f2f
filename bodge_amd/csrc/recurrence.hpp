// recurrence.hpp - one batch of start vectors through the recurrence (Batch), single handles and same-process slab groups
// Part of the single translation unit bodge_hip.hip (included there, in this order:
// core, plans, libraries, recurrence, lanczos, dense); everything lives in its unnamed namespace.
#pragma once

namespace {

// ------------------------------------------------------------------ recurrence
// One batch = up to 64 start vectors advanced together on one handle.  The three
// phases are separate so that a group of slabs can be driven in lock step:
//   begin()  choose kernel + mode, allocate, write t_0 (own rows) and zero t_{-1}
//   step(n)  one launch of K1 (after the caller has refreshed the halo of t_n)
//   finish() reduce partials (done per chunk inside step), copy dots to the host
constexpr int kMarchAborted = -77;  // (internal: a persistent launch gave up waiting - run_recurrence repeats the call sweep by sweep)

struct Batch {
    bdg_system* sys = nullptr;
    StreamSet* ss = nullptr;  // stream + vector buffers + dot partials of this batch: the handle itself, or one of its side sets
    int next = 0;             // next recurrence step to launch (advance)
    int stream_share = 1;     // batches of the call in flight together (set before begin: the sweep plan sizes its segments for it)
    StepPlan plan;
    bdg::StepArgs args{};
    bool real = false;
    bool alternate = false;  // dictionary kernel: sweep direction flips every launch
    // unit start vectors: block rows that can be non-zero after n steps are within
    // (n + 1) * bandwidth of [band_lo, band_hi]; -1 = no band (random vectors, slabs, strip order)
    int64_t band_lo = -1, band_hi = -1;
    // the same for the lattice-stencil kernels: planes of the start sites (-1 = none); a launch that ends with
    // t_m advances planes [plane_lo - m, plane_hi + m] only
    int plane_lo = -1, plane_hi = -1;
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
    // rolling kernel on a row slab: where t_n of the planes just outside the slab is read.  Default = the
    // halo rows of this handle's own buffer (filled by the exchange); a same-process group points these
    // at the neighbouring members' buffers instead and skips the exchange (run_group).
    const double2 *ext_lo = nullptr, *ext_hi = nullptr;
    int64_t ext_lo_site0 = 0, ext_lo_ld = 0, ext_hi_site0 = 0, ext_hi_ld = 0;
    int launch_grid = 0;   // workgroups whose dot partials one recurrence step leaves behind
    // The last launch of a run does not store its vectors: the calls built on Batch return dot products only,
    // and nothing reads t_{n_steps} (BODGE_AMD_KEEP_LAST=1 stores them all the same, for A/B runs).
    bool discard_last = true;
    double bytes_moved = 0.0;  // algorithmic bytes of the launches made so far (perf.bytes_moved)
    // cheb_march3: all the sweeps of a reduction chunk in one launch (march_run below).  Up to kMarchGroups batches of a
    // call are lane groups of the same launches: the first ("leader") owns the stream, the events and the launch count.
    bool march = false, march_follower = false;
    int march_levels = 0;           // sweeps per launch: 0 = a whole reduction chunk (persistent, flags between the sweeps), 1 = one
    bool march_fixed = false;       // persistent launches with a fixed unit per wave instead of tickets
    int n_sweeps = 0;               // sweeps made so far (the marching direction alternates with them)
    int64_t sweeps_counted = 0;     // ... for perf.sweeps
    hipStream_t work_stream = nullptr;  // where this batch's reductions and result copy are enqueued (default: its own stream)
    // one buffer of the batch: what a launch reads for t_n (or t_{n-1}) or writes for one new level
    double vector_bytes() const { return 4.0 * rl * sizeof(double2) * (double)sys->nb; }
    int n_launches = 0;

    int begin(bdg_system* system, double scale_in, int steps, int active, const StartSpec& start,
              int force_real /* -1 auto, 0 complex, 1 real */, bool col_scalars = false, int set_index = 0) {
        sys = system;
        scale = scale_in;
        n_steps = steps;
        n_active = active;
        next = 0;
        HIP_TRY(hipSetDevice(sys->device));
        ss = sys;
        if (set_index > 0) {
            while ((int)sys->side_sets.size() < set_index) {
                auto side = std::make_unique<StreamSet>();
                if (int rc = pooled_stream(sys->device, (int)sys->side_sets.size(), &side->stream)) return fail(rc, "stream creation failed");
                sys->side_sets.push_back(std::move(side));
            }
            ss = sys->side_sets[(size_t)set_index - 1].get();
        }
        // Real arithmetic applies when H has no imaginary part and the start vectors are real
        // (±1 or unit vectors): every t_n then stays real.  BODGE_AMD_REAL=0 forces complex.
        const bool start_is_real = start.kind == StartKind::Unit || start.vec_kind == BDG_VEC_RADEMACHER;
        const char* real_env = knob::raw("BODGE_AMD_REAL");
        const bool matrix_real = sys->slab_comm ? sys->slab_all_real : sys->is_real;  // slabs: agreed over all ranks
        real = matrix_real && start_is_real && !(real_env && real_env[0] == '0');
        if (force_real >= 0) real = force_real != 0;
        const char* ph_env = knob::raw("BODGE_AMD_PH");
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
            if (int rc = sweep_wanted(sys, col_scalars, &stencil_kind, start.wide_call)) return rc;
        // (the stencil kernels keep the whole block table in LDS; a matrix with many distinct blocks in a
        // wide arithmetic mode can exceed that: it takes the one-step kernels, which then stream its blocks)
        if ((size_t)sys->n_unique * mode.stride * sizeof(double2) > table_limit(sys)) stencil_kind = 0;
        if (sys->onsite_streamed && !mode.ph) stencil_kind = 0;  // (packed on-site records assume the Nambu form)
        if (stencil_kind != 0 && start.kind == StartKind::Unit &&
            !(start.stencil >= 0 ? start.stencil != 0 : unit_run_wants_stencil(sys, start.rows, n_active, n_steps)))
            stencil_kind = 0;
        if (stencil_kind == 1) {
            const int lanes = sweep_lanes_for(sys, n_active, per_lane, start.kind == StartKind::Unit);
            // (A/B builds in which the three-step sweep addresses the four components of a plane through one buffer descriptor: below 4 GB)
            const bool addressable = !BDG_SWEEP_ONE_DESCRIPTOR ||
                                     (size_t)3 * sys->ncols * lanes * sizeof(double2) + (size_t)plane_sites(sys) * lanes * sizeof(double2) < bdg::kSweepSpanLimit;
            if (n_active <= lanes * per_lane && addressable) {
                sweep = true;
                rl = lanes;
            }
        } else if (stencil_kind == 2) {
            const int lanes = roll_lanes_for(sys, n_active, per_lane);
            if (n_active <= lanes * per_lane) {
                roll = true;
                rl = lanes;
            }
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
            if (int rc = make_sweep_plan(sys, mode, rl, sweep_depth_for(sys, rl), &splan, stream_share)) return rc;
        } else if (roll) {
            // (as for the sweeps: no one-step plan; the table was seen to fit its LDS budget above)
            plan = StepPlan{};
            plan.rl = rl;
            plan.mode = mode;
            plan.dictionary = true;
            args = bdg::StepArgs{};
            if (int rc = make_roll_plan(sys, mode, rl, &rplan)) return rc;
            rplan.share = stream_share;
        } else {
            if (int rc = make_plan(sys, rl, mode, &plan, col_scalars)) return rc;
            if (int rc = matrix_args(sys, plan, &args)) return rc;
        }
        launch_grid = sweep ? splan.grid : roll ? rplan.grid : plan.grid;
        n_launches = 0;
        n_sweeps = 0;
        sweeps_counted = 0;
        bytes_moved = 0.0;
        work_stream = ss->stream;
        march_follower = false;
        // One launch per reduction chunk (cheb_march3) for random start vectors; unit start vectors, which run inside a
        // band of planes for most of their steps, keep one launch per sweep (their segments change from sweep to sweep).
        march = sweep && splan.depth == 3 && splan.march != nullptr && !sys->march_off && n_steps >= 4 &&
                start.kind == StartKind::Random;
        // BODGE_AMD_MARCH (default 0: measured no faster than one launch per sweep): 1 = a whole chunk per
        // launch, units claimed by ticket; 3 = the same with a fixed unit per wave (needs the grid resident; falls back after
        // the timeout otherwise); 2 = one launch per sweep that advances all the lane groups of a pair (no flags)
        march_levels = 0;
        march_fixed = false;
        {
            const char* env = knob::raw("BODGE_AMD_MARCH");
            const int kind = env ? atoi(env) : 0;
            march = march && kind >= 1 && kind <= 3;
            march_levels = kind == 2 ? 1 : 0;
            march_fixed = kind == 3;
        }
        if (march) launch_grid = splan.args.n_cols * splan.args.n_segs;  // dot partials per (step, unit)

        vec_count = (size_t)4 * sys->ncols * rl;  // 16-byte lane payloads
        // t_n and t_{n-1} together beyond the 256 MB Infinity Cache: the write of t_{n+1} and the
        // read of t_{n-1} are hinted non-temporal (+6 % at 10^6 sites x 8 vectors); smaller buffers
        // stay resident from one launch to the next and are faster with plain accesses
        // (profiles/r01_stream_probe.log)
        args.stream_vectors = 2 * vec_count * sizeof(double2) > kStreamVectorBytes ? 3 : 0;
        if (const char* env = knob::raw("BODGE_AMD_STREAM_VECTORS")) args.stream_vectors = std::atoi(env);
        alternate = true;
        if (const char* env = knob::raw("BODGE_AMD_ALTERNATE")) alternate = std::atoi(env) != 0;
        discard_last = true;
        if (const char* env = knob::raw("BODGE_AMD_KEEP_LAST")) discard_last = std::atoi(env) == 0;
        if (int rc = ss->vec_a.reserve(vec_count)) return rc;
        if (int rc = ss->vec_b.reserve(vec_count)) return rc;
        if (sweep) {
            if (int rc = ss->vec_c.reserve(vec_count)) return rc;
            if (int rc = ss->vec_d.reserve(vec_count)) return rc;
            spare1 = ss->vec_c.ptr;
            spare2 = ss->vec_d.ptr;
            splan.args.stream = 2 * vec_count * sizeof(double2) > kStreamVectorBytes ? 1 : 0;
            if (const char* env = knob::raw("BODGE_AMD_SWEEP_STREAM")) splan.args.stream = std::atoi(env);
        }
        width = (size_t)2 * rv;
        if (int rc = prepare_overlap()) return rc;
        // Dot partials are reduced every `chunk` launches.  Buffer sizes do not depend on
        // n_steps (up to 1024), so a short warm-up call leaves nothing to allocate later.
        per_step = (size_t)(overlapped ? grid_interior + grid_boundary : launch_grid) * width;
        constexpr int kChunk = 64;
        chunk = std::min(n_steps, sweep && splan.depth == 3 ? 63 : kChunk);  // a sweep must not straddle two chunks
        if (int rc = ss->partial.reserve((size_t)kChunk * per_step)) return rc;
        const size_t dots_count = (size_t)std::max(n_steps, 1024) * width;
        if (int rc = ss->dots.reserve(dots_count)) return rc;
        host_stride = dots_count;
        ev_base = slot * ((n_steps + chunk - 1) / chunk);
        if (sys->host_dots_count < dots_count * n_slots) {
            HIP_TRY(hipStreamSynchronize(sys->stream));  // (an earlier batch of this call may still be copying into it)
            for (auto& side : sys->side_sets) HIP_TRY(hipStreamSynchronize(side->stream));
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

        hipStream_t st = ss->stream;
        const int fill_grid = (int)std::min<size_t>(4096, (vec_count + 255) / 256);
        // Three-step sweeps make a random t_0 in registers during the first sweep (cheb_sweep3 GEN):
        // no fill kernel, and vec_a is only ever a spare buffer.  BODGE_AMD_SWEEP_GEN=0: fill and read.
        gen_start = sweep && splan.depth == 3 && splan.kernel_gen && start.kind == StartKind::Random &&
                    sys->row_offset == 0 && sys->ncols == sys->nb;
        if (const char* env = knob::raw("BODGE_AMD_SWEEP_GEN")) gen_start = gen_start && atoi(env) != 0;
        if (gen_start) {
            splan.args.gen_seed = start.seed;
            splan.args.gen_first_id = start.first_id;
            splan.args.gen_kind = real ? BDG_VEC_RADEMACHER : start.vec_kind;
            splan.args.gen_active = n_active;
        } else if (start.kind == StartKind::Random) {
            if (real)
                bdg::fill_random_real<<<fill_grid, 256, 0, st>>>(
                    reinterpret_cast<double*>(ss->vec_a.ptr), sys->nb, sys->ncols, rv, n_active,
                    start.seed, start.first_id, sys->row_offset);
            else
                bdg::fill_random<<<fill_grid, 256, 0, st>>>(ss->vec_a.ptr, sys->nb, sys->ncols, rv,
                                                            n_active, start.seed, start.first_id,
                                                            start.vec_kind, sys->row_offset);
        } else {
            if (int rc = ss->rows.reserve(64)) return rc;
            HIP_TRY(hipMemcpyAsync(ss->rows.ptr, start.rows, sizeof(int64_t) * n_active,
                                   hipMemcpyHostToDevice, st));
            if (sys->ncols == sys->nb && !knob::raw("BODGE_AMD_NO_BAND")) {
                band_lo = sys->nb;
                band_hi = 0;
                for (int r = 0; r < n_active; ++r) {
                    band_lo = std::min<int64_t>(band_lo, start.rows[r] >> 2);
                    band_hi = std::max<int64_t>(band_hi, start.rows[r] >> 2);
                }
            }
            bdg::fill_zero<<<fill_grid, 256, 0, st>>>(ss->vec_a.ptr, (int64_t)vec_count);
            if (sweep) {
                // the sweeps rotate four buffers and, inside the band of a unit start, write only the band's planes:
                // what lies outside must read as the zeros it is
                bdg::fill_zero<<<fill_grid, 256, 0, st>>>(ss->vec_b.ptr, (int64_t)vec_count);
                bdg::fill_zero<<<fill_grid, 256, 0, st>>>(ss->vec_c.ptr, (int64_t)vec_count);
                bdg::fill_zero<<<fill_grid, 256, 0, st>>>(ss->vec_d.ptr, (int64_t)vec_count);
            }
            if (real)
                bdg::set_unit_real<<<1, 64, 0, st>>>(reinterpret_cast<double*>(ss->vec_a.ptr), sys->nb,
                                                     sys->ncols, rv, n_active, ss->rows.ptr,
                                                     sys->row_offset);
            else
                bdg::set_unit<<<1, 64, 0, st>>>(ss->vec_a.ptr, sys->nb, sys->ncols, rv, n_active,
                                                ss->rows.ptr, sys->row_offset);
        }
        if (!sweep)  // (the sweep kernels are told that t_{-1} = 0 instead of reading 256 MB of zeros)
            bdg::fill_zero<<<fill_grid, 256, 0, st>>>(ss->vec_b.ptr, (int64_t)vec_count);
        HIP_TRY(hipGetLastError());

        // bytes of t_n per block row that neighbouring rows re-read: 4 entries per vector
        if (!sweep && !roll)
            if (int rc = prepare_tile_order(sys, plan.rows_per_tile, plan.n_tiles,
                                            (real ? 32.0 : 64.0) * rv, &args.tile_order, &strip_rows))
                return rc;
        if (args.tile_order) band_lo = band_hi = -1;  // the band is a range of naturally ordered tiles
        plane_lo = plane_hi = -1;
        if (sweep || roll) {
            strip_rows = 0;
            const bool ring = sweep ? splan.args.wrap_x != 0 : false;  // (a ring of planes has no band)
            if (band_lo >= 0 && band_lo <= band_hi && !ring && sys->ncols == sys->nb && sys->group_rows == 0) {
                const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
                plane_lo = (int)(band_lo / plane);
                plane_hi = (int)(band_hi / plane);
            }
            band_lo = band_hi = -1;
        }
        cur = ss->vec_a.ptr;
        prev = ss->vec_b.ptr;
        kernel_ms = 0.f;
        n_chunks = 0;
        return BDG_OK;
    }

    // Halo exchange, split so that a same-process group can interleave its members.
    int pack(hipStream_t st = nullptr) {
        if (sys->send_total == 0) return BDG_OK;
        if (!st) st = ss->stream;
        HIP_TRY(hipSetDevice(sys->device));
        const int64_t total = sys->send_total * 4 * rl;
        bdg::halo_pack<<<(unsigned)std::min<int64_t>(2048, (total + 255) / 256), 256, 0, st>>>(
            cur, sys->send_rows.ptr, sys->send_total, sys->ncols, rl, sys->send_buf.ptr);
        HIP_TRY(hipGetLastError());
        return BDG_OK;
    }
    int unpack(hipStream_t st = nullptr) {
        if (!st) st = ss->stream;
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
        const char* env = knob::raw("BODGE_AMD_OVERLAP");
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
        hipStream_t st = ss->stream, cs = sys->comm_stream;
        const int in_chunk = n % chunk;
        const int chunk_id = n / chunk;
        while ((int)ss->ev_pool.size() < 2 * (ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            ss->ev_pool.push_back(ev);
        }
        // exchange of t_n on the communication stream, after everything that produced t_n
        HIP_TRY(hipEventRecord(sys->ev_step_done, st));
        HIP_TRY(hipStreamWaitEvent(cs, sys->ev_step_done, 0));
        if (int rc = pack(cs)) return rc;
        if (int rc = rccl_transfer(cs)) return rc;
        if (int rc = unpack(cs)) return rc;
        HIP_TRY(hipEventRecord(sys->ev_halo_ready, cs));

        if (in_chunk == 0) HIP_TRY(hipEventRecord(ss->ev_pool[2 * (ev_base + chunk_id)], st));
        args.cur = cur;
        args.prev = prev;
        args.coef = (n == 0 ? 1.0 : 2.0) / scale;
        args.discard = discard_last && n == n_steps - 1;
        bytes_moved += algorithmic_bytes(sys, rv, mode, plan.dictionary) - (args.discard ? vector_bytes() : 0.0);
        double* slot = ss->partial.ptr + (size_t)in_chunk * per_step;
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
            HIP_TRY(hipEventRecord(ss->ev_pool[2 * (ev_base + chunk_id) + 1], st));
            bdg::reduce_partials<<<in_chunk + 1, 256, 0, st>>>(ss->partial.ptr, ss->dots.ptr + (size_t)s0 * width,
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
                                        ss->stream));
            if (peer.recv_count > 0)
                NCCL_TRY(api, api->recv(sys->recv_buf.ptr + (size_t)peer.recv_begin * 4 * rl,
                                        (size_t)peer.recv_count * unit, ncclDouble, peer.rank, comm->comm,
                                        ss->stream));
        }
        NCCL_TRY(api, api->group_end());
        return unpack();
    }

    int step(int n) {
        HIP_TRY(hipSetDevice(sys->device));
        hipStream_t st = ss->stream;
        const int in_chunk = n % chunk;
        const int chunk_id = n / chunk;
        // one event pair per chunk, read back in finish(): the host never waits inside the loop
        while ((int)ss->ev_pool.size() < 2 * (ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            ss->ev_pool.push_back(ev);
        }
        if (in_chunk == 0) HIP_TRY(hipEventRecord(ss->ev_pool[2 * (ev_base + chunk_id)], st));
        args.cur = cur;
        args.prev = prev;
        args.coef = (n == 0 ? 1.0 : 2.0) / scale;
        args.discard = discard_last && n == n_steps - 1;
        bytes_moved += (roll ? roll_bytes(sys, mode, rl) : algorithmic_bytes(sys, rv, mode, plan.dictionary)) -
                       (args.discard ? vector_bytes() : 0.0);
        args.partial = ss->partial.ptr + (size_t)in_chunk * per_step;
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
            bytes_moved -= algorithmic_bytes(sys, rv, mode, plan.dictionary) * (1.0 - (double)args.n_tiles / plan.n_tiles);
        }
        if (roll) {
            bdg::RollArgs& ra = rplan.args;
            ra.cur = cur;
            ra.prev = prev;
            ra.coef = args.coef;
            ra.partial = args.partial;
            ra.reverse = args.reverse;
            ra.stream = args.stream_vectors;
            ra.discard = args.discard;
            ra.x_lo = 0;
            ra.x_hi = ra.lx;
            ra.n_segs = rplan.segments_for(ra.lx);
            ra.chunk = rplan.chunk_for(ra.lx);
            if (plane_lo >= 0) {
                ra.x_lo = std::max(0, plane_lo - (n + 1));
                ra.x_hi = std::min(ra.lx, plane_hi + n + 2);
                ra.n_segs = rplan.segments_for(ra.x_hi - ra.x_lo);
                ra.chunk = rplan.chunk_for(ra.x_hi - ra.x_lo);
                bytes_moved -= roll_bytes(sys, mode, rl) * (1.0 - (double)(ra.x_hi - ra.x_lo) / ra.lx);
            }
            ra.lo_buf = ra.hi_buf = nullptr;
            if (sys->stencil_lo_base >= 0) {
                ra.lo_buf = ext_lo ? ext_lo : cur;
                ra.lo_site0 = (int)(ext_lo ? ext_lo_site0 : sys->stencil_lo_base);
                ra.lo_ld = (int)(ext_lo ? ext_lo_ld : sys->ncols);
            }
            if (sys->stencil_hi_base >= 0) {
                ra.hi_buf = ext_hi ? ext_hi : cur;
                ra.hi_site0 = (int)(ext_hi ? ext_hi_site0 : sys->stencil_hi_base);
                ra.hi_ld = (int)(ext_hi ? ext_hi_ld : sys->ncols);
            }
            // (stream bit 0 / 1: t_{n-1} loads / t_{n+1} stores non-temporal - one compiled form with both since round 4)
            (ra.stream ? rplan.kernel_nt : rplan.kernel)<<<rplan.grid, bdg::kBlockThreads, rplan.lds_bytes, st>>>(ra);
        } else {
            plan.kernel<<<plan.grid, bdg::kBlockThreads, plan.lds_bytes, st>>>(args);
        }
        std::swap(cur, prev);
        if (in_chunk == chunk - 1 || n == n_steps - 1) {
            const int s0 = n - in_chunk;
            HIP_TRY(hipEventRecord(ss->ev_pool[2 * (ev_base + chunk_id) + 1], st));
            bdg::reduce_partials<<<in_chunk + 1, 256, 0, st>>>(
                ss->partial.ptr, ss->dots.ptr + (size_t)s0 * width, launch_grid, (int)width);
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
        hipStream_t st = ss->stream;
        const int in_chunk = n % chunk;
        const int chunk_id = n / chunk;
        const int now = std::min({splan.depth, n_steps - n, chunk - in_chunk});
        *made = now;
        while ((int)ss->ev_pool.size() < 2 * (ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            ss->ev_pool.push_back(ev);
        }
        if (in_chunk == 0) HIP_TRY(hipEventRecord(ss->ev_pool[2 * (ev_base + chunk_id)], st));
        bdg::SweepArgs& a = splan.args;
        a.cur = cur;
        a.prev = n == 0 ? nullptr : prev;
        a.out1 = spare1;
        a.out2 = spare2;
        a.coef1 = (n == 0 ? 1.0 : 2.0) / scale;
        a.coef2 = 2.0 / scale;
        a.two = now >= 2 ? 1 : 0;
        a.steps = now;
        a.discard = discard_last && n + now == n_steps;
        a.partial1 = ss->partial.ptr + (size_t)in_chunk * per_step;
        a.partial2 = a.partial1 + per_step;
        a.partial3 = a.partial2 + per_step;
        const SweepKernel kernel = n == 0 && gen_start ? splan.kernel_gen
                                   : alternate && (n_launches & 1) ? splan.kernel_reverse : splan.kernel;
        if (n == 0 && gen_start) a.cur = nullptr;  // (never read)
        a.x_lo = 0;
        a.x_hi = a.lx;
        if (plane_lo >= 0) {
            a.x_lo = std::max(0, plane_lo - (n + now));
            a.x_hi = std::min(a.lx, plane_hi + n + now + 1);
        }
        a.n_segs = splan.segments_for(a.x_hi - a.x_lo);
        // sweep_bytes counts two buffers read and two written: t_{-1} = 0 is never read, a generated t_0 neither,
        // a lone step makes one new level, the last sweep of a run stores nothing; a band is its share of the planes
        bytes_moved += (sweep_bytes(sys, mode, rl) -
                        vector_bytes() * ((n == 0 ? 1 : 0) + (n == 0 && gen_start ? 1 : 0) + (a.discard ? 2 : now == 1 ? 1 : 0))) *
                       ((double)(a.x_hi - a.x_lo) / a.lx);
        kernel<<<splan.grid, splan.wg_waves * bdg::kWave, splan.lds_bytes, st>>>(a);
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
            HIP_TRY(hipEventRecord(ss->ev_pool[2 * (ev_base + chunk_id) + 1], st));
            bdg::reduce_partials<<<last_in_chunk + 1, 256, 0, st>>>(
                ss->partial.ptr, ss->dots.ptr + (size_t)s0 * width, launch_grid, (int)width);
            HIP_TRY(hipGetLastError());
            n_chunks = chunk_id + 1;
        }
        return BDG_OK;
    }

    // The next launch of this batch (single handle: whole matrix, or a slab exchanging its halo through RCCL).
    int advance() {
        if (sweep) {
            int made = 1;
            if (int rc = step_sweep(next, &made)) return rc;
            next += made;
            return BDG_OK;
        }
        if (overlapped) {
            if (int rc = step_overlapped(next)) return rc;
        } else {
            if (int rc = exchange_rccl()) return rc;
            if (int rc = step(next)) return rc;
        }
        ++next;
        return BDG_OK;
    }
    // A batch prepared for the persistent kernel that runs one launch per sweep after all: partials per workgroup again.
    void cancel_march() {
        if (!march) return;
        march = false;
        launch_grid = splan.grid;
        per_step = (size_t)launch_grid * width;
    }
    // first start / last stop event of the batch (after finish_enqueue): the ends of its launches on its stream
    hipEvent_t first_event() const { return ss->ev_pool[2 * (size_t)ev_base]; }
    hipEvent_t last_event() const { return ss->ev_pool[2 * (size_t)(ev_base + n_chunks - 1) + 1]; }

    // d/e of this handle's rows into columns [col0, col0 + n_active) of (n_steps x ld) arrays;
    // accumulate = true adds to what is there (summing the slabs of a group).
    int finish(double* d_out, double* e_out, int ld, int col0, bool accumulate, bool first_batch) {
        if (int rc = finish_enqueue()) return rc;
        HIP_TRY(hipStreamSynchronize(work_stream ? work_stream : ss->stream));
        return finish_collect(d_out, e_out, ld, col0, accumulate, first_batch);
    }
    // copy of the batch's dot products to the host, enqueued behind its last reduction
    int finish_enqueue() {
        HIP_TRY(hipSetDevice(sys->device));
        // pinned: a pageable target costs ~8 ms on its first use
        HIP_TRY(hipMemcpyAsync(sys->host_dots + (size_t)slot * host_stride, ss->dots.ptr,
                               (size_t)n_steps * width * sizeof(double), hipMemcpyDeviceToHost, work_stream ? work_stream : ss->stream));
        return BDG_OK;
    }
    // after the stream has been waited for
    int finish_collect(double* d_out, double* e_out, int ld, int col0, bool accumulate, bool first_batch) {
        HIP_TRY(hipSetDevice(sys->device));
        if (march && sys->march_seen && sys->march_seen[0] != 0) {
            // a wave of a persistent launch gave up waiting for its neighbours (a foreign kernel holding the GPU?):
            // the vectors are incomplete.  The handle goes back to one launch per sweep and the caller repeats the call.
            sys->march_seen[0] = 0;
            (void)hipMemsetAsync(sys->march_gave_up.ptr, 0, sizeof(unsigned), sys->stream);
            (void)hipStreamSynchronize(sys->stream);
            sys->march_off = true;
            return kMarchAborted;
        }
        const double* host = sys->host_dots + (size_t)slot * host_stride;
        if (march_follower) n_chunks = 0;  // (the leader of the lane groups holds the events of their launches)
        for (int c = 0; c < n_chunks; ++c) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ss->ev_pool[2 * (ev_base + c)], ss->ev_pool[2 * (ev_base + c) + 1]));
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
        p.bytes_moved += bytes_moved;
        p.window_ms = p.kernel_ms;  // (batches side by side on several streams: run_recurrence measures the window)
        p.streams = 1;
        p.vector_steps += (int64_t)n_steps * n_active;
        p.bytes_per_launch = sweep  ? sweep_bytes(sys, mode, rl)
                             : roll ? roll_bytes(sys, mode, rl)
                                    : algorithmic_bytes(sys, rv, mode, plan.dictionary);
        p.steps_per_launch = sweep ? splan.depth : 1;
        p.persistent = march ? 1 : 0;
        p.sweeps += march ? sweeps_counted : sweep ? n_launches : 0;
        p.rolling = roll ? 1 : 0;
        p.dict_skipped = sys->dict_skipped;
        p.onsite_streamed = sweep && sys->onsite_streamed ? (sys->bonds_streamed ? 2 : 1) : 0;
        p.lanes_per_row = rl;
        p.vectors_per_launch = rv;
        p.real_arithmetic = real ? 1 : 0;
        p.ph_packed = mode.ph ? 1 : 0;
        p.dict_blocks = plan.dictionary ? sys->n_unique : 0;
        p.strip_rows = strip_rows;
        p.grid = march ? splan.march_grid : launch_grid;
        p.lds_bytes = (int32_t)(sweep ? splan.lds_bytes : roll ? rplan.lds_bytes : plan.lds_footprint);
        p.pipelined = plan.pipelined ? 1 : 0;
        return BDG_OK;
    }
};

// The remaining sweeps of up to kMarchGroups batches of one call (same matrix, same plan, same step count, all at the
// same step), one cheb_march3 launch per reduction chunk on the first batch's stream.  The batches are lane groups of
// the same launches; every one keeps its own vector buffers, partials and dot products.
int march_run(Batch* const* list, int count) {
    Batch& lead = *list[0];
    bdg_system* sys = lead.sys;
    HIP_TRY(hipSetDevice(sys->device));
    hipStream_t st = lead.ss->stream;
    const bdg::SweepArgs& geo = lead.splan.args;
    const int units = geo.n_cols * geo.n_segs;
    for (int g = 0; g < count; ++g) {
        Batch& b = *list[g];
        if (!b.march || b.next != lead.next || b.n_steps != lead.n_steps || b.rl != lead.rl || b.mode.id != lead.mode.id ||
            b.splan.args.n_segs != geo.n_segs || b.chunk != lead.chunk)
            return fail(BDG_EINVAL, "lane groups of one persistent launch must share plan and progress");
        b.march_follower = g > 0;
        b.work_stream = st;
        if (g > 0) {  // whatever the group's begin() enqueued on its own stream comes first
            HIP_TRY(hipEventRecord(sys->ev_side, b.ss->stream));
            HIP_TRY(hipStreamWaitEvent(st, sys->ev_side, 0));
        }
    }
    const size_t sync_words = (bdg::march_sync_words(units, count) + 3) / 4 * 4;
    if (int rc = lead.ss->march_sync.reserve(sync_words)) return rc;
    if (sys->march_gave_up.count == 0) {
        if (int rc = sys->march_gave_up.reserve(4)) return rc;
        HIP_TRY(hipMemsetAsync(sys->march_gave_up.ptr, 0, 4 * sizeof(unsigned), st));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sys->march_seen), 4 * sizeof(unsigned), 0));
        sys->march_seen[0] = 0;
    }
    double timeout_ms = 2000.0;
    if (const char* env = knob::raw("BODGE_AMD_MARCH_TIMEOUT_MS")) timeout_ms = std::max(1.0, atof(env));

    // sweeps per launch: a whole reduction chunk (persistent, flags between the sweeps) or one (no flags, plain stores)
    const int max_levels = lead.march_levels > 0 ? std::min(lead.march_levels, bdg::kMarchMaxLevels) : bdg::kMarchMaxLevels;
    while (lead.next < lead.n_steps) {
        const int n = lead.next;
        const int chunk_id = n / lead.chunk, in_chunk = n % lead.chunk;
        const int chunk_end = std::min(lead.n_steps, (chunk_id + 1) * lead.chunk);
        const int n_end = std::min(chunk_end, n + 3 * max_levels);
        const int levels = (n_end - n + 2) / 3;
        if (in_chunk % 3 != 0 || levels > bdg::kMarchMaxLevels) return fail(BDG_EINVAL, "a persistent launch starts at a sweep boundary of its chunk");
        bdg::MarchArgs m{};
        m.base = geo;
        m.base.coef2 = 2.0 / lead.scale;
        m.base.coef1 = m.base.coef2;
        m.base.x_lo = 0;
        m.base.x_hi = geo.lx;
        if (levels == 1) m.base.stream |= 16;  // (nothing in the launch reads what it writes: plain stores)
        m.n_groups = count;
        m.n_levels = levels;
        m.last_steps = n_end - n - 3 * (levels - 1);
        m.discard_last = lead.discard_last && n_end == lead.n_steps;
        m.first_is_start = n == 0;
        m.gen = n == 0 && lead.gen_start;
        m.rev0 = lead.alternate ? (lead.n_sweeps & 1) : 0;
        m.units = units;
        m.per_step = lead.per_step;
        m.sync = lead.ss->march_sync.ptr;
        m.flags_at = (unsigned)(bdg::kMarchCounterWords * (1 + 8 * bdg::kMarchMaxLevels));
        m.timeout_ticks = (unsigned)std::min(4.0e9, timeout_ms * 1.0e5);
        m.gave_up = sys->march_gave_up.ptr;
        m.fixed = lead.march_fixed ? 1 : 0;
        m.poll_sleep = 64;
        if (const char* env = knob::raw("BODGE_AMD_MARCH_SLEEP")) m.poll_sleep = atoi(env);
        if (const char* env = knob::raw("BODGE_AMD_MARCH_DEBUG")) {
            m.debug = atoi(env);
            if (m.debug & 4) m.base.stream |= 16;
        }
        for (int g = 0; g < count; ++g) {
            Batch& b = *list[g];
            bdg::MarchGroup& grp = m.group[g];
            grp.buf[0] = b.cur;
            grp.buf[1] = b.prev;
            grp.buf[2] = b.spare1;
            grp.buf[3] = b.spare2;
            grp.partial = b.ss->partial.ptr + (size_t)in_chunk * b.per_step;
            grp.gen_first_id = b.splan.args.gen_first_id;
            grp.gen_active = b.splan.args.gen_active;
            if (b.gen_start != lead.gen_start || b.per_step != lead.per_step)
                return fail(BDG_EINVAL, "lane groups of one persistent launch must start the same way");
        }
        while ((int)lead.ss->ev_pool.size() < 2 * (lead.ev_base + chunk_id + 1)) {
            hipEvent_t ev = nullptr;
            HIP_TRY(hipEventCreate(&ev));
            lead.ss->ev_pool.push_back(ev);
        }
        if (in_chunk == 0) HIP_TRY(hipEventRecord(lead.ss->ev_pool[2 * (lead.ev_base + chunk_id)], st));
        if (levels > 1) HIP_TRY(hipMemsetAsync(m.sync, 0, sync_words * sizeof(unsigned), st));
        lead.splan.march<<<lead.splan.march_grid, bdg::kBlockThreads, lead.splan.lds_bytes, st>>>(m);
        HIP_TRY(hipGetLastError());
        ++lead.n_launches;
        for (int g = 0; g < count; ++g) {
            Batch& b = *list[g];
            for (int level = 0; level < levels; ++level) {  // what step_sweep does per launch
                const int at = n + 3 * level, now = std::min(3, n_end - at);
                const bool discard = b.discard_last && at + now == b.n_steps;
                b.bytes_moved += sweep_bytes(sys, b.mode, b.rl) -
                                 b.vector_bytes() * ((at == 0 ? 1 : 0) + (at == 0 && b.gen_start ? 1 : 0) + (discard ? 2 : now == 1 ? 1 : 0));
                double2* old_cur = b.cur;
                double2* old_prev = b.prev;
                if (now == 1) {
                    b.cur = b.spare2;
                    b.prev = old_cur;
                    b.spare2 = old_prev;
                } else {
                    b.cur = b.spare2;
                    b.prev = b.spare1;
                    b.spare1 = old_prev;
                    b.spare2 = old_cur;
                }
            }
            b.n_sweeps += levels;
            b.sweeps_counted += levels;
            b.next = n_end;
        }
        if (n_end == chunk_end) {
            HIP_TRY(hipEventRecord(lead.ss->ev_pool[2 * (lead.ev_base + chunk_id) + 1], st));
            lead.n_chunks = chunk_id + 1;
            const int s0 = chunk_id * lead.chunk;
            for (int g = 0; g < count; ++g) {
                Batch& b = *list[g];
                bdg::reduce_partials<<<chunk_end - s0, 256, 0, st>>>(b.ss->partial.ptr, b.ss->dots.ptr + (size_t)s0 * b.width, units,
                                                                    (int)b.width);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(sys->march_seen, sys->march_gave_up.ptr, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    if (const char* env = knob::raw("BODGE_AMD_MARCH_DEBUG"); env && (atoi(env) & 8)) {
        unsigned stats[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(stats, sys->march_gave_up.ptr, sizeof stats, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        fprintf(stderr, "[bdg] march: %u tasks, waiting %.2f us and claiming %.2f us per task\n", stats[3],
                stats[3] ? 0.16 * stats[1] / stats[3] : 0.0, stats[3] ? 0.16 * stats[2] / stats[3] : 0.0);
        HIP_TRY(hipMemsetAsync(sys->march_gave_up.ptr, 0, sizeof stats, st));
    }
    return BDG_OK;
}

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
int batch_width(bdg_system* sys, const StartSpec& start, int n_vectors, int n_steps) {
    if (const char* env = knob::raw("BODGE_AMD_BATCH")) return std::clamp(atoi(env), 1, 64);
    const bool start_is_real = start.kind == StartKind::Unit || start.vec_kind == BDG_VEC_RADEMACHER;
    const char* real_env = knob::raw("BODGE_AMD_REAL");
    const bool real = (sys->slab_comm ? sys->slab_all_real : sys->is_real) && start_is_real &&
                      !(real_env && real_env[0] == '0');
    int stencil_kind = 0;
    const bool unit = start.kind == StartKind::Unit;
    if (sys->lanes_override == 0 && sweep_wanted(sys, false, &stencil_kind, start.wide_call) == BDG_OK && stencil_kind != 0 &&
        !(unit && !(start.stencil >= 0 ? start.stencil != 0 : unit_run_wants_stencil(sys, start.rows, n_vectors, n_steps)))) {
        const int per_lane = real ? 2 : 1;
        const int lanes = stencil_kind == 1 ? sweep_lanes_for(sys, n_vectors, per_lane, unit, true) : roll_lanes_for(sys, n_vectors, per_lane);
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

constexpr double kSideBySideOneStepLimit = 1.6e6;  // site-vectors per launch up to which one-step batches run side by side

// Single handle (whole matrix, or one slab of a multi-process run with RCCL halos).
int run_recurrence_once(bdg_system* sys, double scale, int n_steps, int n_vectors, StartSpec start,
                        double* d_out, double* e_out) {
    if (int rc = check_recurrence_args(sys, scale, n_steps, n_vectors, d_out, e_out)) return rc;
    lanczos_free(sys);
    const bool trace = knob::raw("BODGE_AMD_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (start.kind == StartKind::Unit)  // one decision for all the batches of the call (their widths follow from it)
        start.stencil = unit_run_wants_stencil(sys, start.rows, n_vectors, n_steps) ? 1 : 0;
    start.wide_call = n_vectors > 8;
    const int width = batch_width(sys, start, n_vectors, n_steps);
    // Batches are enqueued back to back and waited for once (whole matrices; a slab's batches are
    // paced by its halo exchange anyway): the GPU does not idle while the host turns a batch around.
    const int n_batches = (n_vectors + width - 1) / width;
    const size_t staged = (size_t)n_batches * std::max(n_steps, 1024) * 2 * (size_t)width;  // doubles of pinned memory
    const bool pipelined = n_batches > 1 && n_batches <= 64 && staged <= ((size_t)4 << 20) && sys->ncols == sys->nb &&
                           !knob::raw("BODGE_AMD_NO_BATCH_PIPELINE");
    // ... and side by side: the batches of a call are independent, and a launch of the marching kernels leaves the
    // memory system idle while its waves load their first planes and again while the last ones drain.  Batches
    // alternate between the handle's stream and a side stream with its own vector buffers, their launches enqueued
    // in turn: the gaps of one fill with the other's work (1000x1000, two lane groups of 4 vectors: 100.4 -> 112 k
    // vector-steps/s, scratch/r3_two_streams.py).  BODGE_AMD_STREAMS=1..4 (default 2).
    // The marching kernels gain (K7b with 2 lanes per site +13-15 %, with 4 lanes +7 %, streamed on-site blocks +6 %,
    // K8 on 100^3 +11 %; a third stream adds nothing); the one-step kernels lose 6-9 % on large launches (no idle
    // ends, two of them only share the caches: profiles/r03_streams.log) and gain on small ones, which leave most of
    // the GPU empty: 64-vector batches on 32^2 / 64^2 / 100^2 / 140^2 sites +80 / +68 / +32 / +20 %, on 200^2 -9 %
    // (profiles/r03_small_lattice_streams.log; four streams are worse than two).  They take two streams up to
    // 1.6 M site-vectors per launch.
    int n_streams = 1;
    const char* streams_env = knob::raw("BODGE_AMD_STREAMS");
    if (pipelined) {
        n_streams = 2;
        if (streams_env) n_streams = std::clamp(atoi(streams_env), 1, 4);
        n_streams = std::min(n_streams, n_batches);
    }
    if (!pipelined) {
        for (int col = 0; col < n_vectors; col += width) {
            Batch batch;
            const auto t0 = now();
            if (int rc = batch.begin(sys, scale, n_steps, std::min(width, n_vectors - col), batch_start(start, col), -1))
                return rc;
            const auto t1 = now();
            if (batch.march) {
                Batch* one = &batch;
                if (int rc = march_run(&one, 1)) return rc;
            }
            while (batch.next < n_steps)
                if (int rc = batch.advance()) return rc;
            const auto t2 = now();
            if (int rc = batch.finish(d_out, e_out, n_vectors, col, false, col == 0)) return rc;
            if (trace)
                fprintf(stderr, "[bdg] begin %.3f ms, steps %.3f ms (kernels %.3f), finish %.3f ms\n", ms(t0, t1),
                        ms(t1, t2), batch.kernel_ms, ms(t2, now()));
        }
        sys->perf.window_ms = sys->perf.kernel_ms;
        sys->perf.streams = 1;
        return BDG_OK;
    }

    HIP_TRY(hipSetDevice(sys->device));
    if (n_streams > 1 && !streams_env) {
        // a side set is four more vector buffers (up to 256 B per site each): only where the device has the room
        size_t free_bytes = 0, total_bytes = 0;
        HIP_TRY(hipMemGetInfo(&free_bytes, &total_bytes));
        const bool have = (int)sys->side_sets.size() >= n_streams - 1 && sys->side_sets[0]->vec_a.count > 0;
        if (!have && (double)free_bytes < 1.25 * (n_streams - 1) * 4.0 * 256.0 * (double)sys->ncols + (64 << 20)) n_streams = 1;
    }
    while ((int)sys->side_sets.size() < n_streams - 1) {
        auto side = std::make_unique<StreamSet>();
        if (int rc = pooled_stream(sys->device, (int)sys->side_sets.size(), &side->stream)) return fail(rc, "stream creation failed");
        sys->side_sets.push_back(std::move(side));
    }
    if (!sys->ev_side) HIP_TRY(hipEventCreateWithFlags(&sys->ev_side, hipEventDisableTiming));
    std::vector<Batch> queued((size_t)n_batches);
    size_t stride0 = 0;
    bool marched = false;
    // (an error leaves nothing in flight: later calls reuse the streams' buffers)
    auto drained = [&](int rc) {
        (void)hipStreamSynchronize(sys->stream);
        for (auto& side : sys->side_sets) (void)hipStreamSynchronize(side->stream);
        return rc;
    };
    for (int first = 0, last = 0; first < n_batches; first = last) {
        const auto t0 = now();
        for (last = first; last < std::min(n_batches, first + n_streams); ++last) {  // (n_streams may drop to 1 below)
            const int index = last;
            Batch& batch = queued[(size_t)index];
            batch.slot = index;
            batch.n_slots = n_batches;
            batch.stream_share = std::min(n_streams, n_batches - first);  // (a last batch on its own has the GPU to itself)
            const int col = index * width;
            if (int rc = batch.begin(sys, scale, n_steps, std::min(width, n_vectors - col), batch_start(start, col), -1, false,
                                     index - first)) {
                if (rc == BDG_ENOMEM && index > first) {
                    // no room for a side set's buffers after all: give them back and run the rest of the call on one stream
                    (void)drained(0);
                    release_side_sets(sys);
                    n_streams = 1;
                    break;  // (`last` = index: this batch begins again in the next round, on the handle's own set)
                }
                return drained(rc);
            }
            if (index == 0) {  // one spacing of the result pieces for the whole call: the first batch is the widest
                stride0 = batch.host_stride;
                // one-step kernels: two streams only while one launch leaves the GPU part empty (sites x vectors of a batch)
                if (!streams_env && !batch.sweep && !batch.roll && (double)sys->ncols * batch.rv > kSideBySideOneStepLimit) n_streams = 1;
                // the first begin() has built whatever tables the kernels share (on the handle's stream)
                HIP_TRY(hipEventRecord(sys->ev_side, sys->stream));
                for (int side = 0; side < n_streams - 1; ++side)
                    HIP_TRY(hipStreamWaitEvent(sys->side_sets[(size_t)side]->stream, sys->ev_side, 0));
            }
            batch.host_stride = stride0;
        }
        const auto t1 = now();
        // lane groups that take the persistent kernel are tasks of the same launches, on the handle's stream
        bool all_march = last - first <= bdg::kMarchGroups;
        for (int index = first; index < last; ++index) all_march = all_march && queued[(size_t)index].march;
        if (all_march) {
            Batch* group[bdg::kMarchGroups];
            for (int index = first; index < last; ++index) group[index - first] = &queued[(size_t)index];
            if (int rc = march_run(group, last - first)) return drained(rc);
            marched = true;
        } else {
            for (int index = first; index < last; ++index) queued[(size_t)index].cancel_march();
        }
        for (bool more = true; more;) {
            more = false;
            for (int index = first; index < last; ++index) {
                Batch& batch = queued[(size_t)index];
                if (batch.next >= n_steps) continue;
                if (int rc = batch.advance()) return drained(rc);
                more = true;
            }
        }
        for (int index = first; index < last; ++index)
            if (int rc = queued[(size_t)index].finish_enqueue()) return drained(rc);
        if (trace) fprintf(stderr, "[bdg] batches %d..%d: begin %.3f ms, enqueue %.3f ms\n", first, last - 1, ms(t0, t1), ms(t1, now()));
    }
    HIP_TRY(hipStreamSynchronize(sys->stream));
    for (auto& side : sys->side_sets) HIP_TRY(hipStreamSynchronize(side->stream));
    for (int index = 0; index < n_batches; ++index)
        if (int rc = queued[(size_t)index].finish_collect(d_out, e_out, n_vectors, index * width, false, index == 0))
            return rc;
    // the window all launches of the call fell into: first start event to the latest stop event of any stream
    float window = 0.f;
    for (int index = 0; index < n_batches; ++index) {
        float t = 0.f;
        if (queued[(size_t)index].n_chunks == 0) continue;  // (a follower of a persistent launch: its leader has the events)
        HIP_TRY(hipEventElapsedTime(&t, queued[0].first_event(), queued[(size_t)index].last_event()));
        window = std::max(window, t);
    }
    sys->perf.window_ms = window;
    sys->perf.streams = marched ? 1 : n_streams;
    sys->perf.groups_per_launch = marched ? std::min(n_streams, n_batches) : 1;
    return BDG_OK;
}

int run_recurrence(bdg_system* sys, double scale, int n_steps, int n_vectors, StartSpec start,
                   double* d_out, double* e_out) {
    int rc = run_recurrence_once(sys, scale, n_steps, n_vectors, start, d_out, e_out);
    if (rc == kMarchAborted) {
        // a persistent launch gave up waiting (finish_collect has switched the handle to one launch per sweep)
        (void)hipStreamSynchronize(sys->stream);
        for (auto& side : sys->side_sets) (void)hipStreamSynchronize(side->stream);
        if (knob::raw("BODGE_AMD_TRACE")) fprintf(stderr, "[bdg] persistent sweep gave up waiting: the call is repeated one launch per sweep\n");
        rc = run_recurrence_once(sys, scale, n_steps, n_vectors, start, d_out, e_out);
    }
    if (rc == kMarchAborted) return fail(BDG_EDEVICE, "persistent sweep gave up waiting twice");
    return rc;
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
    const char* real_env = knob::raw("BODGE_AMD_REAL");
    const int force_real = (all_real && start_is_real && !(real_env && real_env[0] == '0')) ? 1 : 0;

    // Where a member that runs the rolling stencil kernel finds the plane below / above its slab in the
    // member that owns it: the rows that member would send are one whole plane, in order.
    auto neighbour_plane = [&](size_t m, int base, bdg_system::NeighbourPlane* out) -> bool {
        bdg_system* sys = group->members[m];
        const int64_t plane = (int64_t)sys->shape[1] * sys->shape[2];
        for (const ExchangePeer& peer : sys->peers) {
            if (base < peer.recv_col || base + plane > peer.recv_col + peer.recv_count) continue;
            bdg_system* src = group->members[peer.rank];
            for (const ExchangePeer& back : src->peers) {
                if (back.rank != (int)m) continue;
                const int64_t first = back.send_begin + (base - peer.recv_col);
                if (first + plane > (int64_t)src->send_rows_host.size()) return false;
                for (int64_t j = 1; j < plane; ++j)
                    if (src->send_rows_host[(size_t)(first + j)] != src->send_rows_host[(size_t)first] + j) return false;
                out->owner = src;
                out->site0 = src->send_rows_host[(size_t)first];
                return true;
            }
        }
        return false;
    };

    const int width = batch_width(group->members[0], start, n_vectors, n_steps);
    for (int col = 0; col < n_vectors; col += width) {
        std::vector<Batch> batch(n_members);
        for (size_t m = 0; m < n_members; ++m)
            if (int rc = batch[m].begin(group->members[m], scale, n_steps, std::min(width, n_vectors - col),
                                        batch_start(start, col), force_real))
                return rc;
        for (size_t m = 1; m < n_members; ++m)
            if (batch[m].rl != batch[0].rl)
                return fail(BDG_EINVAL, "group members chose different kernel configurations");
        // Zero-copy: every member runs the rolling stencil kernel and can address its neighbours' buffers.
        // Then no halo row is packed, copied or unpacked: step n of a member waits for step n-1 of its
        // neighbours (their t_n is complete, and they have finished reading the buffer step n overwrites).
        bool zero_copy = true;
        for (size_t m = 0; m < n_members; ++m) {
            bdg_system* sys = group->members[m];
            zero_copy = zero_copy && batch[m].roll && sys->group_peer_access;
            if (!zero_copy) break;
            sys->group_lo = sys->group_hi = bdg_system::NeighbourPlane{};
            if (sys->stencil_lo_base >= 0) zero_copy = zero_copy && neighbour_plane(m, sys->stencil_lo_base, &sys->group_lo);
            if (sys->stencil_hi_base >= 0) zero_copy = zero_copy && neighbour_plane(m, sys->stencil_hi_base, &sys->group_hi);
        }
        if (zero_copy) {
            auto index_of = [&](bdg_system* sys) {
                for (size_t m = 0; m < n_members; ++m)
                    if (group->members[m] == sys) return m;
                return (size_t)0;
            };
            for (size_t m = 0; m < n_members; ++m) {  // t_0 written everywhere before anyone reads a neighbour's
                HIP_TRY(hipSetDevice(group->members[m]->device));
                HIP_TRY(hipEventRecord(group->stepped[1][m], group->members[m]->stream));
            }
            for (int n = 0; n < n_steps; ++n) {
                for (size_t m = 0; m < n_members; ++m) {
                    bdg_system* sys = group->members[m];
                    HIP_TRY(hipSetDevice(sys->device));
                    Batch& b = batch[m];
                    b.ext_lo = b.ext_hi = nullptr;
                    for (const bdg_system::NeighbourPlane* side : {&sys->group_lo, &sys->group_hi}) {
                        if (!side->owner) continue;
                        const size_t o = index_of(side->owner);
                        HIP_TRY(hipStreamWaitEvent(sys->stream, group->stepped[(n + 1) & 1][o], 0));
                        (side == &sys->group_lo ? b.ext_lo : b.ext_hi) = batch[o].cur;
                        (side == &sys->group_lo ? b.ext_lo_site0 : b.ext_hi_site0) = side->site0;
                        (side == &sys->group_lo ? b.ext_lo_ld : b.ext_hi_ld) = side->owner->ncols;
                    }
                    if (int rc = b.step(n)) return rc;  // (swaps b.cur / b.prev: the neighbours later in this loop must see t_n)
                    std::swap(b.cur, b.prev);
                }
                for (size_t m = 0; m < n_members; ++m) {
                    bdg_system* sys = group->members[m];
                    HIP_TRY(hipSetDevice(sys->device));
                    std::swap(batch[m].cur, batch[m].prev);
                    HIP_TRY(hipEventRecord(group->stepped[n & 1][m], sys->stream));
                }
            }
            for (size_t m = 0; m < n_members; ++m)
                if (int rc = batch[m].finish(d_out, e_out, n_vectors, col, m > 0, col == 0)) return rc;
            continue;
        }
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

}  // namespace
