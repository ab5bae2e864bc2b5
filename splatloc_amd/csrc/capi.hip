// capi.hip — the C ABI declared in include/splatraster.h: buffer layouts, argument
// validation and stage sequencing.  No torch types; everything is raw device pointers.
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "common.h"

namespace sr {

static thread_local std::string g_last_error;
static int g_deterministic = 0;   // splatraster_debug_set_deterministic
static int g_split_max_waves = SPLIT_MAX_WAVES;
void set_split_max_waves(int waves) { g_split_max_waves = waves < 0 ? SPLIT_MAX_WAVES : (waves > SPLIT_MAX_WAVES ? SPLIT_MAX_WAVES : waves); }
int split_max_waves() { return g_split_max_waves; }

void set_hip_error(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}

void set_error_text(const char* text) { g_last_error = text; }

// ---- stage timing ---------------------------------------------------------------------
struct StageRec { int stage; hipEvent_t a, b; };
static uint32_t g_timing_mask = 0;  // bit s: stage s is bracketed by an event pair
static std::mutex g_timing_mu;
static std::vector<StageRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct StageTimer {
    hipStream_t stream;
    StageRec rec{};
    bool on;
    StageTimer(int stage, hipStream_t s) : stream(s), on((g_timing_mask >> stage) & 1u)
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_timing_mu);
        rec.stage = stage; rec.a = get_event(); rec.b = get_event();
        if (!rec.a || !rec.b) { on = false; return; }
        (void)hipEventRecord(rec.a, stream);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(rec.b, stream);
        std::lock_guard<std::mutex> lk(g_timing_mu);
        g_recs.push_back(rec);
    }
};

// ---- host landing slot of the early instance count ---------------------------------------
// One pinned, device-mapped buffer + one event per host thread (and device): preprocess_kernel writes its per-(view, block)
// sums straight into it (round 5: a 4-us copy operation used to sit between preprocess and the next kernel of every frame), the
// event is recorded behind preprocess, the depth sort and scan are enqueued behind it, and the host waits on the EVENT only —
// the GPU keeps working while the caller sizes and allocates the binning buffer.
struct HostSlot {
    uint32_t* p = nullptr;    // host address
    uint32_t* dp = nullptr;   // the same memory as the device addresses it
    size_t cap = 0;  // elements
    hipEvent_t ev = nullptr;
    int device = -1;
};
static thread_local HostSlot t_slot;

static int host_slot(size_t n, HostSlot** out)
{
    HostSlot& s = t_slot;
    int dev = 0;
    SR_HIP_CHECK(hipGetDevice(&dev));
    if (s.cap < n) {
        if (s.p) (void)hipHostFree(s.p);
        s.p = nullptr;
        s.dp = nullptr;
        s.cap = 0;
        const size_t cap = n < 4096 ? 4096 : n + n / 2;
        SR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.p), sizeof(uint32_t) * cap, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent));
        SR_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.dp), s.p, 0));
        s.cap = cap;
    }
    if (!s.ev || s.device != dev) {
        if (s.ev) (void)hipEventDestroy(s.ev);
        s.ev = nullptr;
        SR_HIP_CHECK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
        s.device = dev;
    }
    *out = &s;
    return SPLATRASTER_OK;
}

static inline int tile_bits(int tiles)
{
    int b = 1;
    while ((1 << b) < tiles) ++b;
    return b;
}

struct GeomLayout {
    size_t rec0, rec1, tiles_touched, depth_order, offsets, rgb, clamped, sort_keys, keys_alt, vals_alt,
        sort_tmp, scan_tmp, total, block_tiles, span_owner, bytes;
};
// n = V * P rows (V views of a window; V = 1 for the plain call)
static GeomLayout geom_layout(int32_t P, int32_t V)
{
    const size_t p1 = (size_t)(P > 0 ? P : 1);
    const size_t n = p1 * (size_t)(V > 0 ? V : 1);
    GeomLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.rec0 = take(32 * n);
    L.rec1 = L.rec0 + 16;
    L.tiles_touched = take(4 * n);
    L.depth_order = take(4 * n);
    L.offsets = take(4 * n);
    L.rgb = take(12 * p1);       // SH colours: single-view calls only
    L.clamped = take(3 * p1);
    L.sort_keys = take(4 * n);
    L.keys_alt = take(4 * n);
    L.vals_alt = take(4 * n);
    L.sort_tmp = take(sort_tmp_bytes((int64_t)n));
    L.scan_tmp = take(scan_tmp_bytes((int64_t)n));
    L.total = take(16);
    L.block_tiles = take(4 * (size_t)preprocess_blocks((int32_t)p1) * (size_t)(V > 0 ? V : 1));
    L.span_owner = take(4 * (size_t)SPAN_OWNER_CAP);
    L.bytes = o;
    return L;
}

GeomView geom_view(void* base, int32_t P, int32_t V)
{
    const GeomLayout L = geom_layout(P, V);
    char* b = reinterpret_cast<char*>(base);
    GeomView g;
    g.rec = reinterpret_cast<float4*>(b + L.rec0);
    g.tiles_touched = reinterpret_cast<uint32_t*>(b + L.tiles_touched);
    g.depth_order = reinterpret_cast<uint32_t*>(b + L.depth_order);
    g.offsets = reinterpret_cast<uint32_t*>(b + L.offsets);
    g.rgb = reinterpret_cast<float*>(b + L.rgb);
    g.clamped = reinterpret_cast<uint8_t*>(b + L.clamped);
    g.sort_keys = reinterpret_cast<uint32_t*>(b + L.sort_keys);
    g.sort_tmp = reinterpret_cast<uint32_t*>(b + L.sort_tmp);
    g.total = reinterpret_cast<uint32_t*>(b + L.total);
    g.block_tiles = reinterpret_cast<uint32_t*>(b + L.block_tiles);
    g.span_owner = reinterpret_cast<uint32_t*>(b + L.span_owner);
    return g;
}

struct BinLayout {
    size_t keysA, valsA, keysB, valsB, ranges, sort_tmp, irec, ipack, featp, gacc, pose_acc, ckpt, tile_order, nparts, bytes;
};
static BinLayout bin_layout(int32_t P, int32_t V, int64_t R, int32_t W, int32_t H, int32_t C)
{
    const size_t n = (size_t)(R > 0 ? R : 1);
    const size_t nv = (size_t)(V > 0 ? V : 1);
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * (size_t)((H + TILE - 1) / TILE) * nv;
    BinLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    L.valsA = take(4 * n);
    L.keysA = take(4 * n);
    L.keysB = take(4 * n);
    L.valsB = take(4 * n);
    L.ranges = take(8 * (tiles > 0 ? tiles : 1));
    L.sort_tmp = take(sort_tmp_bytes((int64_t)n));
    L.irec = take(32 * n);
    L.ipack = take(4 * n);
    // padded feature table only when the rows are not already 16-byte aligned (shared by the views)
    L.featp = take((C % 4) ? (size_t)(P > 0 ? P : 1) * padded_channels(C) * sizeof(float) : 16);
    L.gacc = take((size_t)(P > 0 ? P : 1) * nv * gacc_row_floats(C) * sizeof(float));
    L.pose_acc = take(POSE_ACC_BYTES);      // (directly behind gacc: the backward zeroes both with one fill)
    // (the deterministic debug mode's 64-bit accumulator is NOT part of this buffer: it is a stream-ordered
    //  allocation made by the backward only while that mode is on)
    // mid-list checkpoints of the forward for split launches (small frames, narrow layouts): the maximum is reserved
    // whenever the shape qualifies, whatever the run-time knob says
    const bool ck = C <= 4 && 4 * (size_t)tiles <= (size_t)SPLIT_MAX_WAVES;
    L.ckpt = take(ck ? nv * (size_t)SPLIT_PARTS_MAX * (size_t)(C + 2) * (size_t)W * (size_t)H * sizeof(float) : 16);
    L.tile_order = take(4 * (tiles > 0 ? tiles : 1));
    L.nparts = take(4 * (tiles > 0 ? tiles : 1));
    L.bytes = o;
    return L;
}

BinView bin_view(void* base, int32_t P, int32_t V, int64_t R, int32_t W, int32_t H, int32_t C)
{
    const BinLayout L = bin_layout(P, V, R, W, H, C);
    char* b = reinterpret_cast<char*>(base);
    BinView v;
    v.point_list = reinterpret_cast<uint32_t*>(b + L.valsA);
    v.tile_list = reinterpret_cast<uint32_t*>(b + L.keysA);
    v.keys_tmp = reinterpret_cast<uint32_t*>(b + L.keysB);
    v.vals_tmp = reinterpret_cast<uint32_t*>(b + L.valsB);
    v.ranges = reinterpret_cast<uint32_t*>(b + L.ranges);
    v.sort_tmp = b + L.sort_tmp;
    v.irec = reinterpret_cast<float4*>(b + L.irec);
    v.ipack = reinterpret_cast<uint32_t*>(b + L.ipack);
    v.featp = reinterpret_cast<float*>(b + L.featp);
    v.gacc = reinterpret_cast<float*>(b + L.gacc);
    v.pose_acc = reinterpret_cast<float*>(b + L.pose_acc);
    v.ckpt = reinterpret_cast<float*>(b + L.ckpt);
    v.tile_order = reinterpret_cast<uint32_t*>(b + L.tile_order);
    v.nparts = reinterpret_cast<uint32_t*>(b + L.nparts);
    return v;
}

static size_t img_plane_bytes(int32_t W, int32_t H, int32_t V)
{
    return align_up((size_t)W * H * 4 * (size_t)(V > 0 ? V : 1), 256);
}

ImgView img_view(void* base, int32_t W, int32_t H, int32_t V)
{
    char* b = reinterpret_cast<char*>(base);
    ImgView v;
    v.final_T = reinterpret_cast<float*>(b);
    v.n_contrib = reinterpret_cast<uint32_t*>(b + img_plane_bytes(W, H, V));
    return v;
}

// The compositing kernels index feature / accumulator rows with 24-bit x 24-bit multiplies (one
// full-rate instruction instead of a 64-bit multiply-add pair per address): Gaussian ids must fit 24
// bits and a row's float offset 32 bits.  16.7 M Gaussians per scene — SplatLoc maps hold < 1 M.
static int check_row_index_range(int32_t P, int32_t V, int32_t C)
{
    const uint64_t row = (uint64_t)(gacc_row_floats(C) > padded_channels(C) ? gacc_row_floats(C) : padded_channels(C));
    const uint64_t n = (uint64_t)P * (uint64_t)V;   // rows of the window
    if (n > (1ull << 24) || n * row >= (1ull << 32)) return SPLATRASTER_ERR_UNSUPPORTED;
    return SPLATRASTER_OK;
}

static int check_settings(const splatraster_settings* s)
{
    if (!s) return SPLATRASTER_ERR_BAD_ARG;
    if (s->image_width <= 0 || s->image_height <= 0 || s->channels <= 0) return SPLATRASTER_ERR_BAD_ARG;
    if (s->bg_channels < 0) return SPLATRASTER_ERR_BAD_ARG;
    return SPLATRASTER_OK;
}

}  // namespace sr

using namespace sr;

extern "C" {

int splatraster_abi_version(void) { return SPLATRASTER_ABI_VERSION; }

const char* splatraster_error_string(int status)
{
    switch (status) {
        case SPLATRASTER_OK: return "ok";
        case SPLATRASTER_ERR_BAD_ARG: return "bad argument";
        case SPLATRASTER_ERR_HIP: return "HIP runtime error";
        case SPLATRASTER_ERR_UNSUPPORTED: return "unsupported configuration";
        case SPLATRASTER_ERR_OVERFLOW: return "tile instance count overflow";
        case SPLATRASTER_WARN_LOOKBACK_STALL: return "look-back stall (results late, never wrong)";
        default: return "unknown status";
    }
}

const char* splatraster_last_hip_error(void) { return g_last_error.c_str(); }

size_t splatraster_geometry_bytes(int32_t P) { return geom_layout(P, 1).bytes; }
size_t splatraster_binning_bytes(int32_t P, int64_t R, int32_t width, int32_t height, int32_t channels)
{
    return bin_layout(P, 1, R, width, height, channels).bytes;
}
size_t splatraster_image_bytes(int32_t width, int32_t height) { return 2 * img_plane_bytes(width, height, 1); }

size_t splatraster_window_geometry_bytes(int32_t P, int32_t n_views) { return geom_layout(P, n_views).bytes; }
size_t splatraster_window_binning_bytes(int32_t P, int32_t n_views, int64_t R_total, int32_t width, int32_t height,
                                        int32_t channels)
{
    return bin_layout(P, n_views, R_total, width, height, channels).bytes;
}
size_t splatraster_window_image_bytes(int32_t width, int32_t height, int32_t n_views)
{
    return 2 * img_plane_bytes(width, height, n_views);
}

int splatraster_get_geometry_layout(int32_t P, splatraster_geometry_layout* out)
{
    return splatraster_get_window_geometry_layout(P, 1, out);
}
int splatraster_get_window_geometry_layout(int32_t P, int32_t n_views, splatraster_geometry_layout* out)
{
    if (!out || n_views < 1 || n_views > MAX_VIEWS) return SPLATRASTER_ERR_BAD_ARG;
    const GeomLayout L = geom_layout(P, n_views);
    out->rec0 = L.rec0; out->rec1 = L.rec1; out->tiles_touched = L.tiles_touched;
    out->depth_order = L.depth_order; out->offsets = L.offsets; out->rgb = L.rgb;
    out->clamped = L.clamped; out->total = L.bytes;
    return SPLATRASTER_OK;
}
int splatraster_get_binning_layout(int32_t P, int64_t R, int32_t width, int32_t height, int32_t channels,
                                   splatraster_binning_layout* out)
{
    return splatraster_get_window_binning_layout(P, 1, R, width, height, channels, out);
}
int splatraster_get_window_binning_layout(int32_t P, int32_t n_views, int64_t R_total, int32_t width, int32_t height,
                                          int32_t channels, splatraster_binning_layout* out)
{
    if (!out || n_views < 1 || n_views > MAX_VIEWS) return SPLATRASTER_ERR_BAD_ARG;
    const BinLayout L = bin_layout(P, n_views, R_total, width, height, channels);
    out->point_list = L.valsA; out->tile_list = L.keysA; out->ranges = L.ranges; out->total = L.bytes;
    return SPLATRASTER_OK;
}
int splatraster_get_image_layout(int32_t width, int32_t height, splatraster_image_layout* out)
{
    return splatraster_get_window_image_layout(width, height, 1, out);
}
int splatraster_get_window_image_layout(int32_t width, int32_t height, int32_t n_views, splatraster_image_layout* out)
{
    if (!out || n_views < 1 || n_views > MAX_VIEWS) return SPLATRASTER_ERR_BAD_ARG;
    out->final_T = 0;
    out->n_contrib = img_plane_bytes(width, height, n_views);
    out->total = 2 * img_plane_bytes(width, height, n_views);
    return SPLATRASTER_OK;
}

}  // extern "C"

namespace sr {

static int check_window(const splatraster_settings* s, int32_t V, const splatraster_window_view* views)
{
    int st = check_settings(s);
    if (st) return st;
    if (V < 1 || V > MAX_VIEWS || !views) return SPLATRASTER_ERR_BAD_ARG;
    return SPLATRASTER_OK;
}

static WinCams make_cams(int32_t V, const splatraster_window_view* views)
{
    WinCams c{};
    for (int v = 0; v < V; ++v) {
        c.view[v] = views[v].viewmatrix;
        c.proj[v] = views[v].projmatrix;
        c.campos[v] = views[v].campos;
        c.tanfovx[v] = views[v].tanfovx;
        c.tanfovy[v] = views[v].tanfovy;
        c.radii[v] = views[v].radii;
    }
    return c;
}

// The binned front end (binsort.hip) keeps its (tile, chunk) table and the state of its scan where the radix front end
// keeps the depth-sort buffers (sort_keys ... scan_tmp): the geometry buffer's size does not depend on the path.
struct BinScratch { bool on; uint32_t* table; void* scan_tmp; int64_t entries; };

// Which front end the GEOMETRY stage chose for a geometry buffer: the render stage is a separate public call and must follow
// that choice, not re-derive it from the process-wide debug switch (splatraster_debug_set_front_end between the two stages
// would otherwise make the render read a (tile, chunk) table that was never built).  Host-side, keyed by the buffer's address;
// an address the geometry stage has not seen (or that was dropped when the map was trimmed) falls back to use_bins().
static std::mutex g_front_end_mu;
static std::unordered_map<const void*, bool> g_front_end;
static void front_end_record(const void* geometry, bool on)
{
    std::lock_guard<std::mutex> lk(g_front_end_mu);
    if (g_front_end.size() > (1u << 14)) g_front_end.clear();
    g_front_end[geometry] = on;
}
static int front_end_recorded(const void* geometry)   // -1 unknown, 0 radix, 1 binned
{
    std::lock_guard<std::mutex> lk(g_front_end_mu);
    auto it = g_front_end.find(geometry);
    return it == g_front_end.end() ? -1 : (it->second ? 1 : 0);
}

static BinScratch bin_scratch(const splatraster_settings& s, int32_t P, int32_t V, void* geometry, bool geometry_stage)
{
    const int gx = (s.image_width + TILE - 1) / TILE, gy = (s.image_height + TILE - 1) / TILE, tiles = gx * gy;
    const GeomLayout L = geom_layout(P, V);
    BinScratch b{};
    if (!geometry || P <= 0) return b;
    const int recorded = geometry_stage ? -1 : front_end_recorded(geometry);
    b.on = recorded >= 0 ? recorded == 1 : use_bins(P, V, gx, gy, L.total - L.sort_keys);
    if (geometry_stage) front_end_record(geometry, b.on);
    if (!b.on) return b;
    char* base = reinterpret_cast<char*>(geometry);
    b.entries = (int64_t)bin_table_entries(P, V, tiles);
    b.table = reinterpret_cast<uint32_t*>(base + L.sort_keys);
    b.scan_tmp = base + L.sort_keys + align_up((size_t)b.entries * sizeof(uint32_t), 256);
    return b;
}

// Stage 1 of the forward over a window of V views (V = 1: the plain call): one preprocess, ONE depth sort of the
// V * P rows, one scan; the per-view instance counts are read back under the sort.
static int window_geometry(const splatraster_settings* s, int32_t V, const splatraster_window_view* views, int32_t P,
                           const float* means3D, const float* shs, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, void* geometry, int64_t* num_rendered,
                           hipStream_t stream, const RawFwd* raw = nullptr)
{
    int st = check_window(s, V, views);
    if (st) return st;
    if (P < 0 || !num_rendered) return SPLATRASTER_ERR_BAD_ARG;
    for (int v = 0; v < V; ++v) num_rendered[v] = 0;
    if (P == 0) return SPLATRASTER_OK;
    if (raw) {      // the raw tensors stand in for opacities / scales / rotations (which are this call's OUTPUTS: raw->scales ...)
        if (shs || cov3D_precomp || !raw->scaling || !raw->rotation || !raw->opacity || !raw->f_dc || !raw->scales || !raw->rotations ||
            !raw->opacities || !raw->colors || raw->E != s->channels - 3 || (raw->E > 0 && !raw->extra))
            return SPLATRASTER_ERR_BAD_ARG;
        opacities = raw->opacities; scales = raw->scales; rotations = raw->rotations;
    }
    if (!means3D || !opacities || !geometry) return SPLATRASTER_ERR_BAD_ARG;
    for (int v = 0; v < V; ++v)
        if (!views[v].viewmatrix || !views[v].projmatrix || !views[v].radii) return SPLATRASTER_ERR_BAD_ARG;
    const bool have_sr = scales && rotations;
    if (have_sr == (cov3D_precomp != nullptr)) return SPLATRASTER_ERR_BAD_ARG;
    if ((scales == nullptr) != (rotations == nullptr)) return SPLATRASTER_ERR_BAD_ARG;
    if (shs) {
        if (V != 1) return SPLATRASTER_ERR_UNSUPPORTED;   // view-dependent colours: one table per view
        if (s->channels != 3 || !views[0].campos) return SPLATRASTER_ERR_BAD_ARG;
        if (s->sh_degree < 0 || s->sh_degree > 3) return SPLATRASTER_ERR_UNSUPPORTED;
        if (s->sh_coeffs < (s->sh_degree + 1) * (s->sh_degree + 1)) return SPLATRASTER_ERR_BAD_ARG;
    }
    if ((int64_t)P * V >= ((int64_t)1 << 31)) return SPLATRASTER_ERR_OVERFLOW;
    st = check_row_index_range(P, V, s->channels);   // (the depth-order words keep the row in 24 bits)
    if (st) return st;
    st = lookback_error_init();
    if (st) return st;
    const int32_t n = P * V;
    const GeomLayout L = geom_layout(P, V);
    GeomView g = geom_view(geometry, P, V);
    char* base = reinterpret_cast<char*>(geometry);
    const WinCams cams = make_cams(V, views);
    const BinScratch bins = bin_scratch(*s, P, V, geometry, true);
    HostSlot* slot = nullptr;
    const size_t nblk = (size_t)preprocess_blocks(P);
    st = host_slot(nblk * (size_t)V, &slot);
    if (st) return st;
    g.block_tiles = slot->dp;     // the per-(view, block) instance sums land in host memory
    // preprocess_kernel stores into the thread's pinned slot: an error return between its launch and the wait below must not
    // leave the kernel writing a slot the next call on this thread may free, reallocate or read (state 1: launched, wait for
    // the stream; 2: the event behind the kernel is recorded, wait for that; 0: nothing in flight)
    struct SlotGuard {
        HostSlot* slot; hipStream_t stream; int state;
        ~SlotGuard() { if (state == 2) (void)hipEventSynchronize(slot->ev); else if (state == 1) (void)hipStreamSynchronize(stream); }
    } guard{slot, stream, 1};
    {
        StageTimer t(SPLATRASTER_STAGE_PREPROCESS, stream);
        // the look-back state of the depth sort and of the scan is cleared by preprocess_kernel
        if (bins.on)
            st = launch_preprocess(*s, P, V, cams, means3D, shs, opacities, scales, rotations, cov3D_precomp, g, nullptr, 0u,
                                   reinterpret_cast<uint32_t*>(bins.scan_tmp), (uint32_t)(scan_state_bytes(bins.entries) / 4),
                                   stream, false, raw);
        else
            st = launch_preprocess(*s, P, V, cams, means3D, shs, opacities, scales, rotations, cov3D_precomp, g,
                                   g.sort_tmp, (uint32_t)(sort_zero_bytes(n, 32) / 4),
                                   reinterpret_cast<uint32_t*>(base + L.scan_tmp), (uint32_t)(scan_state_bytes(n) / 4), stream, true, raw);
    }
    if (st) return st;
    SR_HIP_CHECK(hipEventRecord(slot->ev, stream));
    guard.state = 2;
    if (bins.on) {
        // binned front end (binsort.hip): per-(tile, chunk) counts + their scan instead of the depth sort + offsets scan
        StageTimer t(SPLATRASTER_STAGE_DEPTH_SORT, stream);
        st = launch_bin_count(*s, P, V, g, bins.table, bins.scan_tmp, stream);
        if (st) return st;
    } else {
    {
        StageTimer t(SPLATRASTER_STAGE_DEPTH_SORT, stream);
        bool in_alt = false;
        uint32_t* keys_alt = reinterpret_cast<uint32_t*>(base + L.keys_alt);
        uint32_t* vals_alt = reinterpret_cast<uint32_t*>(base + L.vals_alt);
        st = sort_pairs_u32(n, g.sort_keys, g.depth_order, keys_alt, vals_alt, 32, g.sort_tmp, stream, &in_alt, true);
        if (st) return st;
        if (in_alt) return SPLATRASTER_ERR_UNSUPPORTED;  // 4 passes: never
    }
    {
        StageTimer t(SPLATRASTER_STAGE_SCAN, stream);
        st = inclusive_scan_u32(n, g.tiles_touched, g.depth_order, g.offsets, g.total, base + L.scan_tmp, stream, true,
                                g.span_owner, (uint32_t)EMIT_SPAN, SPAN_OWNER_CAP);
    }
    if (st) return st;
    }
    SR_HIP_CHECK(hipEventSynchronize(slot->ev));  // preprocess only: sort and scan may still be running
    guard.state = 0;
    uint64_t total = 0;
    for (int v = 0; v < V; ++v) {
        uint64_t tv = 0;
        for (size_t k = 0; k < nblk; ++k) tv += slot->p[(size_t)v * nblk + k];
        num_rendered[v] = (int64_t)tv;
        total += tv;
    }
    if (total >= ((uint64_t)1 << 31)) return SPLATRASTER_ERR_OVERFLOW;
    return SPLATRASTER_OK;
}

// Stage 2: ONE emission, ONE tile sort keyed by (view, tile), one payload pass, one compositing grid over the V views.
static int window_render(const splatraster_settings* s, int32_t V, const splatraster_window_view* views, int32_t P,
                         int64_t R, const float* bg, const float* colors_precomp, void* geometry, void* binning,
                         void* image, hipStream_t stream)
{
    int st = check_window(s, V, views);
    if (st) return st;
    if (P < 0 || R < 0 || !image) return SPLATRASTER_ERR_BAD_ARG;
    for (int v = 0; v < V; ++v)
        if (!views[v].out_color || !views[v].out_depth || !views[v].out_alpha) return SPLATRASTER_ERR_BAD_ARG;
    if (s->bg_channels > 0 && !bg) return SPLATRASTER_ERR_BAD_ARG;
    if (P > 0 && (!geometry || !binning)) return SPLATRASTER_ERR_BAD_ARG;
    if (V > 1 && !colors_precomp && P > 0) return SPLATRASTER_ERR_UNSUPPORTED;
    st = check_row_index_range(P, V, s->channels);
    if (st) return st;
    if (!binning) return SPLATRASTER_ERR_BAD_ARG;
    const int W = s->image_width, H = s->image_height;
    const int tiles = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    const int64_t gtiles = (int64_t)tiles * V;
    if (gtiles >= ((int64_t)1 << 31)) return SPLATRASTER_ERR_OVERFLOW;
    ImgView im = img_view(image, W, H, V);
    GeomView g{};
    if (geometry) g = geom_view(geometry, P, V);
    BinView b = bin_view(binning, P, V, R, W, H, s->channels);
    const float* feat = colors_precomp ? colors_precomp : g.rgb;
    if (R > 0 && !feat) return SPLATRASTER_ERR_BAD_ARG;
    const int bits = tile_bits((int)gtiles);
    const int passes = (bits + 7) / 8;
    // emit into the buffer pair from which `passes` ping-pongs end in (tile_list, point_list)
    uint32_t* k0 = (passes & 1) ? b.keys_tmp : b.tile_list;
    uint32_t* v0 = (passes & 1) ? b.vals_tmp : b.point_list;
    uint32_t* k1 = (passes & 1) ? b.tile_list : b.keys_tmp;
    uint32_t* v1 = (passes & 1) ? b.point_list : b.vals_tmp;
    const BinScratch bins = bin_scratch(*s, P, V, geometry, false);
    if (R > 0 && bins.on) {
        // binned front end: scatter the 64-bit keys into their (tile, chunk) pieces, sort every tile's list in LDS and write
        // the payload + lists + ranges (binsort.hip); the keys live where the radix path keeps its unsorted pairs
        StageTimer t(SPLATRASTER_STAGE_TILE_SORT, stream);
        st = launch_bin_scatter_sort(*s, P, V, R, g, bins.table, b, reinterpret_cast<uint64_t*>(b.keys_tmp), stream);
        if (st) return st;
    }
    if (R > 0 && !bins.on) {
        {
            StageTimer t(SPLATRASTER_STAGE_EMIT, stream);
            st = launch_emit(*s, P, V, R, g, k0, v0, b.ranges, 2u * (uint32_t)gtiles, stream);  // also clears the range table
        }
        if (st) return st;
        bool in_alt = false;
        {
            StageTimer t(SPLATRASTER_STAGE_TILE_SORT, stream);
            st = sort_pairs_u32(R, k0, v0, k1, v1, bits, b.sort_tmp, stream, &in_alt);
        }
        if (st) return st;
    }
    if (R == 0) {   // nothing was emitted: the table is cleared here instead
        StageTimer t(SPLATRASTER_STAGE_RANGES, stream);
        st = launch_ranges_clear((int32_t)gtiles, b.ranges, stream);
    }
    if (st) return st;
    const float* featp = feat;  // 16-byte aligned rows for the compositing kernels
    if (R > 0) {
        StageTimer t(SPLATRASTER_STAGE_PAYLOAD, stream);
        if (!bins.on) st = launch_payload(*s, V, R, g, b, stream);
        if (st) return st;
        if (s->channels % 4) {
            st = launch_pad_features(P, s->channels, feat, b.featp, stream);
            featp = b.featp;
        }
    }
    if (st) return st;
    if (!(R > 0 && bins.on)) {   // launch order of the compositing grids (the range table is final here, also when nothing was
                                 // emitted); the binned front end's last launch has computed it (binsort.hip)
        StageTimer t(SPLATRASTER_STAGE_RANGES, stream);
        st = launch_tile_order(*s, V, b, stream);
    }
    if (st) return st;
    WinOut outs{};
    for (int v = 0; v < V; ++v) {
        outs.color[v] = views[v].out_color;
        outs.depth[v] = views[v].out_depth;
        outs.alpha[v] = views[v].out_alpha;
    }
    StageTimer t(SPLATRASTER_STAGE_COMPOSITE_FWD, stream);
    return launch_composite_fwd(*s, P, V, R, g, b, im, featp, bg, outs, stream);
}

// Backward of the window: one compositing grid over the V views into per-(view, Gaussian) accumulator rows, then
// ONE per-Gaussian pass that sums the views into a single set of parameter gradients.
static int window_backward(const splatraster_settings* s, int32_t V, const splatraster_window_view* views, int32_t P,
                           int64_t R, const float* bg, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* scales, const float* rotations, const float* cov3D_precomp, void* geometry,
                           const void* binning, const void* image, float* dL_dmeans3D, float* dL_dcolors,
                           float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                           float* dL_dshs, float* dL_dviewmatrix, float* dL_dprojmatrix, float* dL_dcampos,
                           hipStream_t stream, const RawBwd* raw = nullptr)
{
    int st = check_window(s, V, views);
    if (st) return st;
    if (P < 0 || R < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (V > 1 && (shs || dL_dviewmatrix || dL_dprojmatrix || dL_dcampos)) return SPLATRASTER_ERR_UNSUPPORTED;
    if (raw && (shs || cov3D_precomp || dL_dviewmatrix || dL_dprojmatrix || dL_dcampos)) return SPLATRASTER_ERR_UNSUPPORTED;
    if (P == 0) {   // nothing to differentiate: the camera gradients are still defined (zero)
        if (dL_dviewmatrix) SR_HIP_CHECK(hipMemsetAsync(dL_dviewmatrix, 0, 16 * sizeof(float), stream));
        if (dL_dprojmatrix) SR_HIP_CHECK(hipMemsetAsync(dL_dprojmatrix, 0, 16 * sizeof(float), stream));
        if (dL_dcampos) SR_HIP_CHECK(hipMemsetAsync(dL_dcampos, 0, 3 * sizeof(float), stream));
        return SPLATRASTER_OK;
    }
    if (!means3D || !geometry || !binning || !image || !dL_dmeans3D || (!dL_dopacities && !raw)) return SPLATRASTER_ERR_BAD_ARG;
    if (raw && (P > 0) && (!raw->scaling || !raw->rotation || !raw->opacity || !raw->f_dc || !raw->d_scaling || !raw->d_rotation ||
                           !raw->d_opacity || !raw->d_f_dc || raw->E != s->channels - 3 || (raw->E > 0 && !raw->d_extra)))
        return SPLATRASTER_ERR_BAD_ARG;
    for (int v = 0; v < V; ++v)
        if (!views[v].viewmatrix || !views[v].projmatrix || !views[v].radii || !views[v].out_color || !views[v].out_depth ||
            !views[v].dL_dout_color || !views[v].dL_dmeans2D)
            return SPLATRASTER_ERR_BAD_ARG;
    if (shs && (!dL_dshs || !views[0].campos)) return SPLATRASTER_ERR_BAD_ARG;
    if (!shs && (!colors_precomp || (!dL_dcolors && !raw))) return SPLATRASTER_ERR_BAD_ARG;
    if (cov3D_precomp ? !dL_dcov3D : (!scales || !rotations || ((!dL_dscales || !dL_drotations) && !raw)))
        return SPLATRASTER_ERR_BAD_ARG;
    st = check_row_index_range(P, V, s->channels);
    if (st) return st;
    const int W = s->image_width, H = s->image_height;
    GeomView g = geom_view(geometry, P, V);
    BinView b = bin_view(const_cast<void*>(binning), P, V, R, W, H, s->channels);
    ImgView im = img_view(const_cast<void*>(image), W, H, V);
    const int C = s->channels;
    const float* feat = shs ? g.rgb : colors_precomp;
    const WinCams cams = make_cams(V, views);
    WinGrad grads{};
    grads.gc = C;
    for (int v = 0; v < V; ++v) {
        const int gcv = views[v].color_grad_channels;
        if (gcv < 0 || gcv > C) return SPLATRASTER_ERR_BAD_ARG;
        if (gcv != 0 && gcv < C) grads.gc = gcv;
    }
    for (int v = 0; v < V; ++v) {   // one convention per launch: all views split the last channel off, or none does
        const int gcv = views[v].color_grad_channels ? views[v].color_grad_channels : C;
        if (gcv != grads.gc) return SPLATRASTER_ERR_BAD_ARG;
        grads.dL_dlast[v] = grads.gc < C ? views[v].dL_dout_last : nullptr;
    }
    for (int v = 0; v < V; ++v) {
        grads.out_color[v] = views[v].out_color;
        grads.out_depth[v] = views[v].out_depth;
        grads.dL_dcolor[v] = views[v].dL_dout_color;
        grads.dL_ddepth[v] = views[v].dL_dout_depth;
        grads.dL_dalpha[v] = views[v].dL_dout_alpha;
        grads.dL_dmeans2D[v] = views[v].dL_dmeans2D;
    }
    // zero the accumulator rows (outside the stage bracket: the stage is the kernel alone, so its
    // figure can be held against the per-kernel rocprofv3 average)
    const size_t gacc_n = (size_t)gacc_row_floats(C) * (size_t)P * (size_t)V;
    const bool det = g_deterministic != 0;
    long long* gacc64 = nullptr;   // debug mode only: stream-ordered scratch, freed below (never part of `binning`)
    if (det) {
        SR_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&gacc64), sizeof(long long) * gacc_n, stream));
        SR_HIP_CHECK(hipMemsetAsync(gacc64, 0, sizeof(long long) * gacc_n, stream));
    }
    const bool pose = dL_dviewmatrix && dL_dprojmatrix;
    // (one fill: the camera-gradient accumulator sets lie directly behind the rows)
    const size_t fill = pose ? (size_t)(reinterpret_cast<char*>(b.pose_acc) - reinterpret_cast<char*>(b.gacc)) + POSE_ACC_BYTES
                             : sizeof(float) * gacc_n;
    SR_HIP_CHECK(hipMemsetAsync(b.gacc, 0, fill, stream));
    grads.bg = bg;
    grads.bg_channels = bg ? s->bg_channels : 0;
    {
        StageTimer t(SPLATRASTER_STAGE_COMPOSITE_BWD, stream);
        // deterministic mode: the kernel runs twice — per-element max of |partial| (into the zeroed float rows), then the
        // fixed-point sums scaled by that maximum (composite_bwd.hip acc_add)
        if (det) st = launch_composite_bwd(*s, P, V, R, g, b, im, (C % 4) ? b.featp : feat, C, grads, b.gacc, gacc64, 0, stream);
        if (!st) st = launch_composite_bwd(*s, P, V, R, g, b, im, (C % 4) ? b.featp : feat, C, grads, b.gacc, gacc64, 1, stream);
    }
    if (det) {
        if (!st) st = launch_fixed_to_float((int64_t)gacc_n, gacc64, b.gacc, stream);
        (void)hipFreeAsync(gacc64, stream);
    }
    if (st) return st;
    StageTimer t(SPLATRASTER_STAGE_PREPROCESS_BWD, stream);
    return launch_preprocess_bwd(*s, P, V, cams, grads, means3D, shs, scales, rotations, cov3D_precomp, g.clamped, g.rec,
                                 b.gacc, C, shs ? nullptr : dL_dcolors, dL_dmeans3D, dL_dopacities,
                                 cov3D_precomp ? nullptr : dL_dscales, cov3D_precomp ? nullptr : dL_drotations,
                                 cov3D_precomp ? dL_dcov3D : nullptr, dL_dshs, dL_dviewmatrix, dL_dprojmatrix, dL_dcampos,
                                 b.pose_acc, stream, raw);
}

}  // namespace sr

extern "C" {

int splatraster_forward_geometry(const splatraster_settings* s, int32_t P, const float* means3D,
                                 const float* shs, const float* opacities, const float* scales,
                                 const float* rotations, const float* cov3D_precomp,
                                 const float* viewmatrix, const float* projmatrix, const float* campos,
                                 void* geometry, int32_t* radii, int64_t* num_rendered, void* stream_)
{
    if (!s) return SPLATRASTER_ERR_BAD_ARG;
    splatraster_window_view w{};
    w.viewmatrix = viewmatrix; w.projmatrix = projmatrix; w.campos = campos;
    w.tanfovx = s->tanfovx; w.tanfovy = s->tanfovy; w.radii = radii;
    if (num_rendered) *num_rendered = 0;
    if (P > 0 && !radii) return SPLATRASTER_ERR_BAD_ARG;
    return window_geometry(s, 1, &w, P, means3D, shs, opacities, scales, rotations, cov3D_precomp, geometry, num_rendered,
                           reinterpret_cast<hipStream_t>(stream_));
}

int splatraster_forward_render(const splatraster_settings* s, int32_t P, int64_t R, const float* bg,
                               const float* colors_precomp, void* geometry, void* binning, void* image,
                               float* out_color, float* out_depth, float* out_alpha, void* stream_)
{
    if (!s) return SPLATRASTER_ERR_BAD_ARG;
    splatraster_window_view w{};
    w.tanfovx = s->tanfovx; w.tanfovy = s->tanfovy;
    w.out_color = out_color; w.out_depth = out_depth; w.out_alpha = out_alpha;
    return window_render(s, 1, &w, P, R, bg, colors_precomp, geometry, binning, image, reinterpret_cast<hipStream_t>(stream_));
}

int splatraster_backward(const splatraster_settings* s, int32_t P, int64_t R, const float* bg,
                         const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                         const float* campos, const int32_t* radii, void* geometry,
                         const void* binning, const void* image, const float* out_color,
                         const float* out_depth, const float* out_alpha, const float* dL_dout_color,
                         const float* dL_dout_depth, const float* dL_dout_alpha, float* dL_dmeans3D,
                         float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacities, float* dL_dscales,
                         float* dL_drotations, float* dL_dcov3D, float* dL_dshs, float* dL_dviewmatrix,
                         float* dL_dprojmatrix, float* dL_dcampos, void* stream_)
{
    (void)opacities; (void)out_alpha;
    if (!s) return SPLATRASTER_ERR_BAD_ARG;
    splatraster_window_view w{};
    w.viewmatrix = viewmatrix; w.projmatrix = projmatrix; w.campos = campos;
    w.tanfovx = s->tanfovx; w.tanfovy = s->tanfovy; w.radii = const_cast<int32_t*>(radii);
    w.out_color = const_cast<float*>(out_color); w.out_depth = const_cast<float*>(out_depth);
    w.dL_dout_color = dL_dout_color; w.dL_dout_depth = dL_dout_depth; w.dL_dout_alpha = dL_dout_alpha;
    w.dL_dmeans2D = dL_dmeans2D;
    return window_backward(s, 1, &w, P, R, bg, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, geometry, binning,
                           image, dL_dmeans3D, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D, dL_dshs,
                           dL_dviewmatrix, dL_dprojmatrix, dL_dcampos, reinterpret_cast<hipStream_t>(stream_));
}

int splatraster_forward_window_geometry(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                        int32_t P, const float* means3D, const float* opacities, const float* scales,
                                        const float* rotations, const float* cov3D_precomp, void* geometry,
                                        int64_t* num_rendered, void* stream)
{
    return window_geometry(s, n_views, views, P, means3D, nullptr, opacities, scales, rotations, cov3D_precomp, geometry,
                           num_rendered, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_forward_window_render(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                      int32_t P, const int64_t* num_rendered, const float* bg, const float* colors_precomp,
                                      void* geometry, void* binning, void* image, void* stream)
{
    if (!num_rendered || n_views < 1 || n_views > MAX_VIEWS) return SPLATRASTER_ERR_BAD_ARG;
    int64_t R = 0;
    for (int v = 0; v < n_views; ++v) {
        if (num_rendered[v] < 0) return SPLATRASTER_ERR_BAD_ARG;
        R += num_rendered[v];
    }
    return window_render(s, n_views, views, P, R, bg, colors_precomp, geometry, binning, image,
                         reinterpret_cast<hipStream_t>(stream));
}

int splatraster_backward_window(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                int32_t P, const int64_t* num_rendered, const float* bg, const float* means3D, const float* colors_precomp,
                                const float* scales, const float* rotations, const float* cov3D_precomp, void* geometry,
                                const void* binning, const void* image, float* dL_dmeans3D, float* dL_dcolors,
                                float* dL_dopacities, float* dL_dscales, float* dL_drotations, float* dL_dcov3D, void* stream)
{
    if (!num_rendered || n_views < 1 || n_views > MAX_VIEWS) return SPLATRASTER_ERR_BAD_ARG;
    int64_t R = 0;
    for (int v = 0; v < n_views; ++v) {
        if (num_rendered[v] < 0) return SPLATRASTER_ERR_BAD_ARG;
        R += num_rendered[v];
    }
    return window_backward(s, n_views, views, P, R, bg, means3D, nullptr, colors_precomp, scales, rotations, cov3D_precomp,
                           geometry, binning, image, dL_dmeans3D, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations,
                           dL_dcov3D, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_forward_window_geometry_raw(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                            int32_t P, const float* means3D, const splatraster_raw_forward* rf, void* geometry,
                                            int64_t* num_rendered, void* stream)
{
    if (!rf) return SPLATRASTER_ERR_BAD_ARG;
    const RawFwd raw{rf->scaling, rf->rotation, rf->opacity, rf->f_dc, rf->extra, rf->extra_channels, rf->scales, rf->rotations,
                     rf->opacities, rf->colors};
    return window_geometry(s, n_views, views, P, means3D, nullptr, nullptr, nullptr, nullptr, nullptr, geometry, num_rendered,
                           reinterpret_cast<hipStream_t>(stream), &raw);
}

int splatraster_backward_window_raw(const splatraster_settings* s, int32_t n_views, const splatraster_window_view* views,
                                    int32_t P, const int64_t* num_rendered, const float* bg, const float* means3D,
                                    const float* colors_precomp, const float* scales, const float* rotations, void* geometry,
                                    const void* binning, const void* image, const splatraster_raw_params* rp, float* dL_dmeans3D,
                                    void* stream)
{
    if (!num_rendered || n_views < 1 || n_views > MAX_VIEWS || !rp) return SPLATRASTER_ERR_BAD_ARG;
    int64_t R = 0;
    for (int v = 0; v < n_views; ++v) {
        if (num_rendered[v] < 0) return SPLATRASTER_ERR_BAD_ARG;
        R += num_rendered[v];
    }
    if (rp->reg_row_grad && !rp->reg_out) return SPLATRASTER_ERR_BAD_ARG;
    const RawBwd raw{rp->scaling, rp->rotation, rp->opacity, rp->f_dc, rp->extra_channels, rp->dL_dscaling, rp->dL_drotation,
                     rp->dL_dopacity, rp->dL_df_dc, rp->dL_dextra, rp->reg_row_grad, rp->reg_out, rp->reg_weight};
    return window_backward(s, n_views, views, P, R, bg, means3D, nullptr, colors_precomp, scales, rotations, nullptr,
                           geometry, binning, image, dL_dmeans3D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, reinterpret_cast<hipStream_t>(stream), &raw);
}

int splatraster_debug_set_small_panel_max_waves(int waves)
{
    set_small_panel_max_waves(waves);
    return SPLATRASTER_OK;
}

int splatraster_debug_set_payload_stream_min(int64_t instances)
{
    sr::set_payload_stream_min(instances);
    return SPLATRASTER_OK;
}

int splatraster_debug_set_front_end(int mode)
{
    set_bin_mode(mode);
    return SPLATRASTER_OK;
}

int splatraster_debug_set_sort_fork(int mode)
{
    set_bin_fork(mode);
    return SPLATRASTER_OK;
}

int splatraster_debug_set_tile_sort_cap(int keys)
{
    set_bin_tile_cap(keys);
    return SPLATRASTER_OK;
}

int splatraster_debug_set_split_max_waves(int waves)
{
    set_split_max_waves(waves);
    return SPLATRASTER_OK;
}

int splatraster_debug_set_fwd_team(int mode)
{
    set_fwd_team(mode < 0 ? -1 : (mode > 2 ? 2 : mode));
    return SPLATRASTER_OK;
}

int splatraster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix,
                             const float* projmatrix, uint8_t* present, void* stream)
{
    (void)projmatrix;
    if (P < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (P == 0) return SPLATRASTER_OK;
    if (!means3D || !viewmatrix || !present) return SPLATRASTER_ERR_BAD_ARG;
    return launch_mark_visible(P, means3D, viewmatrix, present, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_poll_errors(void) { return lookback_error_poll(); }

int splatraster_debug_set_deterministic(int on)
{
    g_deterministic = on ? 1 : 0;
    return SPLATRASTER_OK;
}

int splatraster_debug_set_spin_limit(uint32_t limit)
{
    int st = lookback_error_init();
    if (st) return st;
    return lookback_set_spin_limit(limit);
}

int splatraster_debug_exp2(int64_t n, const float* x, float* y, void* stream)
{
    if (n < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (n == 0) return SPLATRASTER_OK;
    if (!x || !y) return SPLATRASTER_ERR_BAD_ARG;
    return launch_debug_exp2(n, x, y, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_debug_poison_lds(uint32_t pattern, void* stream)
{
    return launch_poison_lds(pattern, reinterpret_cast<hipStream_t>(stream));
}

size_t splatraster_sort_tmp_bytes(int64_t n)
{
    const size_t m = (size_t)(n > 0 ? n : 1);
    return align_up(4 * m, 256) * 2 + sort_tmp_bytes(n);
}

int splatraster_sort_pairs_u32(int64_t n, uint32_t* keys, uint32_t* vals, int32_t key_bits, void* tmp,
                               void* stream_)
{
    if (n < 0 || key_bits < 0 || key_bits > 32) return SPLATRASTER_ERR_BAD_ARG;
    if (n == 0 || key_bits == 0) return SPLATRASTER_OK;
    if (!keys || !vals || !tmp) return SPLATRASTER_ERR_BAD_ARG;
    {
        int st0 = lookback_error_init();
        if (st0) return st0;
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    char* t = reinterpret_cast<char*>(tmp);
    const size_t stride = align_up(4 * (size_t)n, 256);
    uint32_t* ka = reinterpret_cast<uint32_t*>(t);
    uint32_t* va = reinterpret_cast<uint32_t*>(t + stride);
    bool in_alt = false;
    int st = sort_pairs_u32(n, keys, vals, ka, va, key_bits, t + 2 * stride, stream, &in_alt);
    if (st) return st;
    if (in_alt) {
        SR_HIP_CHECK(hipMemcpyAsync(keys, ka, 4 * (size_t)n, hipMemcpyDeviceToDevice, stream));
        SR_HIP_CHECK(hipMemcpyAsync(vals, va, 4 * (size_t)n, hipMemcpyDeviceToDevice, stream));
    }
    return SPLATRASTER_OK;
}

int splatraster_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_timing_mu);
    g_timing_mask = on ? 0xffffffffu : 0u;
    return SPLATRASTER_OK;
}

int splatraster_timing_select(uint32_t stage_mask)
{
    std::lock_guard<std::mutex> lk(g_timing_mu);
    g_timing_mask = stage_mask;
    return SPLATRASTER_OK;
}

int splatraster_timing_collect(double* ms, int64_t* counts)
{
    if (!ms || !counts) return SPLATRASTER_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_timing_mu);
    for (const StageRec& r : g_recs) {
        SR_HIP_CHECK(hipEventSynchronize(r.b));
        float t = 0.f;
        SR_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
        if (r.stage >= 0 && r.stage < SPLATRASTER_STAGE_COUNT) { ms[r.stage] += (double)t; counts[r.stage] += 1; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return SPLATRASTER_OK;
}

size_t splatknn_workspace_bytes(int32_t N) { return knn_workspace_bytes(N); }

static int check_activate(int32_t P, int32_t K, int32_t deg, int32_t SC, int32_t E, const void* f_rest,
                          const void* extra)
{
    if (P < 0 || K < 1 || deg < 0 || E < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (deg > 3) return SPLATRASTER_ERR_UNSUPPORTED;
    if ((deg + 1) * (deg + 1) > K) return SPLATRASTER_ERR_BAD_ARG;
    if (SC != 1 && SC != 3) return SPLATRASTER_ERR_BAD_ARG;
    if (P > 0 && K > 1 && !f_rest) return SPLATRASTER_ERR_BAD_ARG;
    if (P > 0 && E > 0 && !extra) return SPLATRASTER_ERR_BAD_ARG;
    return SPLATRASTER_OK;
}

int splatraster_activate_forward(int32_t P, int32_t sh_coeffs, int32_t active_sh_degree, int32_t scaling_cols,
                                 int32_t extras, const float* xyz, const float* f_dc, const float* f_rest,
                                 const float* scaling, const float* rotation, const float* opacity,
                                 const float* extra, const float* campos, float* scales, float* rotations,
                                 float* opacities, float* colors, void* stream)
{
    int st = check_activate(P, sh_coeffs, active_sh_degree, scaling_cols, extras, f_rest, extra);
    if (st) return st;
    if (P == 0) return SPLATRASTER_OK;
    if (!xyz || !f_dc || !scaling || !rotation || !opacity || !scales || !rotations || !opacities || !colors)
        return SPLATRASTER_ERR_BAD_ARG;
    if (active_sh_degree > 0 && !campos) return SPLATRASTER_ERR_BAD_ARG;
    return launch_activate_fwd(P, sh_coeffs, active_sh_degree, scaling_cols, extras, xyz, f_dc, f_rest, scaling,
                               rotation, opacity, extra, campos, scales, rotations, opacities, colors,
                               reinterpret_cast<hipStream_t>(stream));
}

int splatraster_activate_backward(int32_t P, int32_t sh_coeffs, int32_t active_sh_degree, int32_t scaling_cols,
                                  int32_t extras, const float* xyz, const float* f_dc, const float* f_rest,
                                  const float* scaling, const float* rotation, const float* opacity,
                                  const float* campos, const float* dL_dscales, const float* dL_drotations,
                                  const float* dL_dopacities, const float* dL_dcolors, float* dL_dxyz,
                                  float* dL_df_dc, float* dL_df_rest, float* dL_dscaling, float* dL_drotation,
                                  float* dL_dopacity, float* dL_dextra, void* stream)
{
    int st = check_activate(P, sh_coeffs, active_sh_degree, scaling_cols, extras, f_rest, dL_dextra);
    if (st) return st;
    if (P == 0) return SPLATRASTER_OK;
    if (!xyz || !f_dc || !scaling || !rotation || !opacity || !dL_dscales || !dL_drotations || !dL_dopacities ||
        !dL_dcolors || !dL_df_dc || !dL_dscaling || !dL_drotation || !dL_dopacity)
        return SPLATRASTER_ERR_BAD_ARG;
    if (sh_coeffs > 1 && !dL_df_rest) return SPLATRASTER_ERR_BAD_ARG;
    if (active_sh_degree > 0 && !campos) return SPLATRASTER_ERR_BAD_ARG;
    return launch_activate_bwd(P, sh_coeffs, active_sh_degree, scaling_cols, extras, xyz, f_dc, f_rest, scaling,
                               rotation, opacity, campos, dL_dscales, dL_drotations, dL_dopacities, dL_dcolors,
                               dL_dxyz, dL_df_dc, dL_df_rest, dL_dscaling, dL_drotation, dL_dopacity, dL_dextra,
                               reinterpret_cast<hipStream_t>(stream));
}

int splatraster_densification_stats(int32_t P, const float* viewspace_grad, const int32_t* radii,
                                    float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream)
{
    const float* g[1] = {viewspace_grad};
    const int32_t* r[1] = {radii};
    return splatraster_densification_stats_window(P, 1, g, r, xyz_gradient_accum, denom, max_radii2D, stream);
}

int splatraster_densification_stats_window(int32_t P, int32_t n_views, const float* const* viewspace_grads,
                                           const int32_t* const* radii, float* xyz_gradient_accum, float* denom,
                                           float* max_radii2D, void* stream)
{
    if (P < 0 || n_views < 0 || n_views > MAX_VIEWS) return SPLATRASTER_ERR_BAD_ARG;
    if (P == 0 || n_views == 0) return SPLATRASTER_OK;
    if (!radii || !max_radii2D || (xyz_gradient_accum == nullptr) != (denom == nullptr)) return SPLATRASTER_ERR_BAD_ARG;
    if (xyz_gradient_accum && !viewspace_grads) return SPLATRASTER_ERR_BAD_ARG;
    StatsViews sv{};
    for (int v = 0; v < n_views; ++v) {
        if (!radii[v] || (xyz_gradient_accum && !viewspace_grads[v])) return SPLATRASTER_ERR_BAD_ARG;
        sv.radii[v] = radii[v];
        sv.vs_grad[v] = viewspace_grads ? viewspace_grads[v] : nullptr;
    }
    return launch_densification_stats(P, n_views, sv, xyz_gradient_accum, denom, max_radii2D,
                                      reinterpret_cast<hipStream_t>(stream));
}

size_t splatraster_mapping_loss_workspace_bytes(int32_t pixels) { return mapping_loss_workspace_bytes(pixels); }

int splatraster_mapping_loss(int32_t pixels, const float* image, const float* depth, const float* marker,
                             const float* gt_image, const float* gt_depth, const float* kp,
                             float rgb_boundary_threshold, const float* exposure, float* g_image, float* g_depth,
                             float* g_marker, float* out, void* workspace, void* stream)
{
    if (pixels <= 0) return SPLATRASTER_ERR_BAD_ARG;
    if (!image || !depth || !marker || !gt_image || !gt_depth || !kp || !g_image || !g_depth || !g_marker || !out ||
        !workspace)
        return SPLATRASTER_ERR_BAD_ARG;
    return launch_mapping_loss(pixels, image, depth, marker, gt_image, gt_depth, kp, rgb_boundary_threshold, exposure,
                               g_image, g_depth, g_marker, out, workspace, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_mapping_loss_window(int32_t n_views, int32_t pixels, const splatraster_loss_view* views,
                                    float rgb_boundary_threshold, float* out, void* workspace, void* stream)
{
    if (pixels <= 0 || n_views < 1 || n_views > SPLATRASTER_MAX_WINDOW_VIEWS || !views || !out || !workspace)
        return SPLATRASTER_ERR_BAD_ARG;
    for (int v = 0; v < n_views; ++v) {
        const splatraster_loss_view& w = views[v];
        if (!w.image || !w.depth || !w.marker || !w.gt_image || !w.gt_depth || !w.kp || !w.g_image || !w.g_depth || !w.g_marker)
            return SPLATRASTER_ERR_BAD_ARG;
    }
    return launch_mapping_loss_window(n_views, pixels, views, rgb_boundary_threshold, out, workspace,
                                      reinterpret_cast<hipStream_t>(stream));
}

size_t splatraster_refinement_loss_workspace_bytes(int32_t channels, int32_t height, int32_t width)
{
    if (channels <= 0 || height <= 0 || width <= 0) return 0;
    return refinement_loss_workspace_bytes(channels, height, width);
}

int splatraster_refinement_loss(int32_t channels, int32_t height, int32_t width, float lambda_dssim,
                                const float* image, const float* gt, float* g_image, float* out, void* workspace,
                                void* stream)
{
    if (channels <= 0 || height <= 0 || width <= 0) return SPLATRASTER_ERR_BAD_ARG;
    if (!image || !gt || !g_image || !out || !workspace) return SPLATRASTER_ERR_BAD_ARG;
    return launch_refinement_loss(channels, height, width, lambda_dssim, image, gt, g_image, out, workspace,
                                  reinterpret_cast<hipStream_t>(stream));
}

size_t splatraster_eval_metrics_workspace_bytes(int32_t channels, int32_t height, int32_t width)
{
    if (channels <= 0 || height <= 0 || width <= 0) return 0;
    return eval_metrics_workspace_bytes(channels, height, width);
}

int splatraster_eval_metrics(int32_t channels, int32_t height, int32_t width, const float* render, const float* gt, float* out,
                             void* workspace, void* stream)
{
    if (channels <= 0 || height <= 0 || width <= 0) return SPLATRASTER_ERR_BAD_ARG;
    if (!render || !gt || !out || !workspace) return SPLATRASTER_ERR_BAD_ARG;
    return launch_eval_metrics(channels, height, width, render, gt, out, workspace, reinterpret_cast<hipStream_t>(stream));
}

int splatraster_l1_rgbd_loss(int64_t n_color, const float* color, const float* target_color, int64_t n_depth, const float* depth,
                             const float* target_depth, float depth_weight, float* g_color, float* g_depth, float* loss_out,
                             void* stream)
{
    if (n_color <= 0 || n_depth < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (!color || !target_color || !g_color || !loss_out) return SPLATRASTER_ERR_BAD_ARG;
    if (n_depth > 0 && target_depth && !depth) return SPLATRASTER_ERR_BAD_ARG;
    return launch_l1_rgbd_loss(n_color, color, target_color, n_depth, depth, target_depth, depth_weight, g_color, g_depth, loss_out,
                               reinterpret_cast<hipStream_t>(stream));
}

int splatraster_pose_step(const float* dL_dviewmatrix, const float* dL_dprojmatrix, const float* dL_dcampos, const float* W2C_init,
                          const float* projection_matrix, float lr_rot, float lr_trans, float beta1, float beta2, float eps,
                          int advance, float* state, float* viewmatrix, float* projmatrix, float* campos, void* stream)
{
    if (!W2C_init || !projection_matrix || !state || !viewmatrix || !projmatrix) return SPLATRASTER_ERR_BAD_ARG;
    if (advance && (!dL_dviewmatrix || !dL_dprojmatrix)) return SPLATRASTER_ERR_BAD_ARG;
    return launch_pose_step(dL_dviewmatrix, dL_dprojmatrix, dL_dcampos, W2C_init, projection_matrix, lr_rot, lr_trans, beta1, beta2,
                            eps, advance, state, viewmatrix, projmatrix, campos, reinterpret_cast<hipStream_t>(stream));
}

int splatknn_debug_set_grid_min(int32_t n)
{
    knn_set_grid_min(n);
    return SPLATRASTER_OK;
}

int splatknn_dist2(int32_t N, const float* points, float* out, void* workspace, void* stream)
{
    if (N < 0) return SPLATRASTER_ERR_BAD_ARG;
    if (N == 0) return SPLATRASTER_OK;
    if (!points || !out || !workspace) return SPLATRASTER_ERR_BAD_ARG;
    return knn_dist2(N, points, out, workspace, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
