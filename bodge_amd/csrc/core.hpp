// core.hpp - error reporting, device buffers, the handle structures (bdg_system, bdg_comm, bdg_group)
// Part of the single translation unit bodge_hip.hip (included there, in this order:
// core, plans, libraries, recurrence, lanczos, dense); everything lives in its unnamed namespace.
#pragma once

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

// What a batch of vectors in flight needs for itself: a stream, the rotating vector buffers, the dot partials.
// The handle is the first such set; a call with several batches runs them side by side on further ones
// (`side_sets`, created on first use) - the start and the end of a launch leave the memory system idle, and
// launches of independent batches fill each other's gaps (recurrence.hpp: run_recurrence).
// Streams are per device, not per handle.  ROCm serves the streams of a process from four hardware queues; the fifth
// stream shares one.  With a stream pair per handle, the second handle of a process (bench.py's comparison matrices, a
// user holding two Hamiltonians) could get both its streams on ONE queue: its two lane groups then ran one after the
// other, each on half the wave slots - 0.36 instead of 0.22 ms per launch of the same call (profiles/r04_stream_pool.log).
// So every handle of a device uses the same main stream and the same side streams (created on first use, kept for the
// life of the process): null stream + main + side (+ the RCCL stream of slabs) stay within four queues however many
// handles exist.  Calls on two handles of one device from two host threads are therefore ordered on the GPU like calls on
// one handle - correct (every call waits for its own work; buffers are per handle), not concurrent.
struct DeviceStreams {
    hipStream_t main = nullptr;
    std::vector<hipStream_t> side;
};
inline int pooled_stream(int device, int side_index /* -1 = the main stream */, hipStream_t* out) {
    static std::mutex mutex;
    static std::map<int, DeviceStreams> pool;
    std::lock_guard<std::mutex> lock(mutex);
    DeviceStreams& d = pool[device];
    hipStream_t* slot = &d.main;
    if (side_index >= 0) {
        if ((int)d.side.size() <= side_index) d.side.resize((size_t)side_index + 1, nullptr);
        slot = &d.side[(size_t)side_index];
    }
    if (!*slot && hipStreamCreateWithFlags(slot, hipStreamNonBlocking) != hipSuccess) {
        *slot = nullptr;
        (void)hipGetLastError();
        return BDG_EDEVICE;
    }
    *out = *slot;
    return BDG_OK;
}

struct StreamSet {
    hipStream_t stream = nullptr;  // (from the device's pool: never destroyed here)
    std::vector<hipEvent_t> ev_pool;  // (start, stop) pairs: one per reduction chunk of a call
    DeviceBuffer<double2> vec_a, vec_b;
    DeviceBuffer<double2> vec_c, vec_d;  // multi-step sweeps: the new levels are written out of place
    DeviceBuffer<double> partial, dots;
    DeviceBuffer<int64_t> rows;
    DeviceBuffer<unsigned> march_sync;  // cheb_march3: abort word, ticket counters, unit flags (zeroed before every launch)
    void release_set() {
        march_sync.release();
        vec_a.release();
        vec_b.release();
        vec_c.release();
        vec_d.release();
        partial.release();
        dots.release();
        rows.release();
        for (hipEvent_t ev : ev_pool) (void)hipEventDestroy(ev);
        ev_pool.clear();
        stream = nullptr;
    }
};

struct bdg_system : StreamSet {
    int device = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<std::unique_ptr<StreamSet>> side_sets;
    hipEvent_t ev_side = nullptr;  // recorded on `stream` once a call's shared tables exist: the side streams wait for it
    // cheb_march3 (one launch per reduction chunk): `march_gave_up` is a device word the launches of a call OR their abort
    // words into, read back (pinned `march_seen`) with the dot products; once a launch has given up waiting the handle
    // keeps to one launch per sweep (`march_off`).
    DeviceBuffer<unsigned> march_gave_up;
    unsigned* march_seen = nullptr;
    bool march_off = false;
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
    std::vector<std::pair<int32_t, int32_t>> halo_refs;  // host: every (block row, halo column) pair of the matrix
    std::vector<int64_t> send_rows_host;                 // host copy of send_rows (per peer, concatenated)
    // Same-process slab groups (bdg_group_create): lattice-stencil slabs read the neighbouring slab's t_n in
    // place.  stencil_lo_base / hi_base: first halo column of the plane below plane 0 / above plane lx-1
    // (-1: none); group_rows: block rows of the whole group (kernel choice goes by the size of the lattice,
    // not of the slab); lo / hi: which member holds that plane and where (filled by the group).
    int stencil_lo_base = -1, stencil_hi_base = -1;
    int64_t group_rows = 0;
    struct NeighbourPlane {
        bdg_system* owner = nullptr;
        int64_t site0 = 0;
    } group_lo, group_hi;
    bool group_peer_access = true;  // every peer of this member sits on the same GPU or one it can address
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_step_done = nullptr, ev_halo_ready = nullptr;
    DeviceBuffer<int> tiles_interior, tiles_boundary;
    int split_rows_per_tile = 0, n_interior = 0, n_boundary = 0;
    void* lanczos = nullptr;  // LanczosState of a run in progress (defined with the driver)
    int max_row_blocks = 0;
    int64_t bandwidth = 0;  // max |column - row| over the stored blocks (square matrices)
    int num_cus = 0;
    size_t lds_per_cu = 0;  // LDS bytes of a CU (and the most one workgroup may take)
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
    // Position-dependent on-site terms: the diagonal blocks are all different, the bond blocks few.  Then
    // the dictionary above holds the distinct OFF-diagonal blocks only, the words of the diagonal blocks
    // carry kStreamedId, and the three-step sweep streams a packed per-site record instead
    // (cheb_sweep3<..., OS>); every other kernel family streams all blocks of such a matrix.
    bool onsite_streamed = false;
    DeviceBuffer<double2> onsite[2];  // [0] complex (6 x 16 B per site), [1] real (4 x 16 B); built on first use
    // ... and when the bond blocks differ from bond to bond as well (the reference's ssd() profile, bond disorder,
    // Peierls phases of a position-dependent gauge) but are all diagonal as 4x4 matrices of the Nambu form
    // diag(a, b, -conj a, -conj b) (spin-diagonal hopping, no bond pairing): no table at all, every site carries its
    // four bond blocks next to its on-site block in one record (`site_records`, built once the lattice shape is known):
    // [1] real arithmetic (real matrices), 128 B per site; [0] complex arithmetic, 224 B per site.
    bool bonds_streamed = false;
    DeviceBuffer<double2> site_records[2];
    int site_records_plane[2] = {0, 0};  // lattice plane size the records were built for
    DeviceBuffer<int> dict_ids;
    // the same words in rows of fixed width (4 words for up to 3 blocks per row, 8 for up to 7; 0xFFFFFFFF = no block): the
    // one-step dictionary kernel reads a row's words with one or two 16-byte loads instead of indptr -> words (a dependent
    // round trip less per tile: small lattices are bound by that chain)
    DeviceBuffer<unsigned> dict_ell;
    int dict_ell_words = 0;
    DeviceBuffer<int> dict_diagonal;      // per distinct block: 1 = diagonal as a 4x4 matrix, 2 = "singlet" form (stencil kernels)
    DeviceBuffer<double2> dict_full;      // n_unique x 16 complex entries
    DeviceBuffer<double2> dict_table[4];  // packed per storage mode, built on first use
    // lattice-stencil form of the matrix (sweep.hpp): 0 = not examined, 1 = 5-point table built (planes
    // are lines), 2 = 7-point table built (3-D), -1 = not a stencil
    DeviceBuffer<uint2> stencil;
    int stencil_state = 0;
    bool stencil_wrap_p = false, stencil_wrap_x = false;  // periodic edge blocks: planes / stack of planes are rings
    double* host_dots = nullptr;  // pinned staging for the dot products (sized like `dots`)
    size_t host_dots_count = 0;
    // lattice geometry hint (rows = z + lz*(y + ly*x)) and the cached strip-major tile order
    int shape[3] = {0, 0, 0};
    // One cached permutation per (rows per tile, strip width): the batches of a call may differ in both (a narrower last
    // batch), and an order is never rewritten while launches that read it can still be in flight.
    struct TileOrder {
        int rows_per_tile = 0, strip_rows = 0;
        DeviceBuffer<int> ids;
    };
    std::vector<std::unique_ptr<TileOrder>> tile_orders;
    bdg_perf perf{};
};

namespace {
// The side sets of a handle (vector buffers, partials, streams of the batches that ran beside the first) are kept from call
// to call; whoever needs the memory for something else - a Lanczos run, a dense solve - gives them back first.
inline void release_side_sets(bdg_system* sys) {
    for (auto& side : sys->side_sets) {
        if (side->stream) (void)hipStreamSynchronize(side->stream);
        side->release_set();
    }
    sys->side_sets.clear();
}
}  // namespace

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
    std::vector<hipEvent_t> stepped[2];      // per member, alternating by step parity (zero-copy stencil slabs)
};
