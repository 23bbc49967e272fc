// api_index.hip — C-ABI of the flat cosine index (the chromadb Collection's numeric half) and the two
// stateless glue kernels (blend, shard merge).
#include "common.h"
#include <queue>
#include "retrieval_kernels.h"
#include "gemm_bf16_256.h"
#include "gemm_bf16_p256.h"
#include <algorithm>

struct mmiss_index {
    int dim = 0, dtype = MMISS_F32, device = 0, elt = 4;
    int qelt() const { return dtype == MMISS_F32 ? 4 : 2; }   // bytes per element of the scan's query operand (f16 for f16 and fp8 rows)
    hipStream_t own_stream = nullptr, user_stream = nullptr;
    bool has_user_stream = false;
    // a call that returns with work still queued on a caller's stream (device outputs) marks its end with this event;
    // switching streams waits for exactly that work, not for whatever else the caller has put on the stream since
    hipEvent_t done_ev = nullptr;
    bool async_pending = false;
    std::mutex mu;
    int64_t count = 0, capacity = 0;
    DevBuf rows, labels_d;
    DevBuf inv;   // MMISS_F8 only: [capacity] f32, inverse norm of every row's values (f8_row_inv_kernel); zero behind count
    std::vector<int64_t> labels_h;
    // scratch
    DevBuf stage, qn, qs, qstage, lists_s, lists_r, lists2_s, lists2_r, cand, cur_s, cur_r, out_c, map, gmax;
    // exactness guard + widen pass (see "exactness contract" above mmiss_index_query)
    DevBuf qmap, qmap64, qs2, cand2;
    DevBuf eps_q, thr, thr2, swp_cnt, swp_list;   // per-query error bound, sweep thresholds (all / flagged queries), appended rows
    DevBuf seed_s, seed_r, fcnt, fbuf_s, fbuf_g;  // threshold-filtered selection (Q > 128)
    // Pinned, device-visible host block a query's LAST kernel (the rerank) writes straight into: the guard's per-query flags
    // always, and labels / distances / counts when the caller's outputs are host buffers. The call then ends with ONE stream
    // synchronisation and no copy-engine operation (round 3: four small device-to-host copies used to sit there, ~25 us).
    char* pin = nullptr;
    size_t pin_bytes = 0;
    int32_t* swp_pin = nullptr;   // pinned: the widen pass's list lengths, written by the re-rank kernel itself (RerankArgs::cnt_host)
    size_t swp_pin_n = 0;
    int ensure_swp_pin(size_t n) {
        if (n <= swp_pin_n) return MMISS_OK;
        if (swp_pin) { MM_HIP(hipStreamSynchronize(stream())); MM_HIP(hipHostFree(swp_pin)); swp_pin = nullptr; swp_pin_n = 0; }
        const size_t cap = n < 1024 ? 1024 : n * 2;
        MM_HIP(hipHostMalloc(reinterpret_cast<void**>(&swp_pin), cap * 4, hipHostMallocDefault));
        swp_pin_n = cap;
        return MMISS_OK;
    }
    int ensure_pin(size_t need) {
        if (need <= pin_bytes) return MMISS_OK;
        size_t cap = pin_bytes ? pin_bytes : 4096;
        while (cap < need) cap *= 2;
        if (pin) { MM_HIP(hipStreamSynchronize(stream())); MM_HIP(hipHostFree(pin)); pin = nullptr; pin_bytes = 0; }
        MM_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin), cap, hipHostMallocDefault));
        pin_bytes = cap;
        return MMISS_OK;
    }
    int64_t stat_queries = 0, stat_flagged = 0, stat_rounds = 0, stat_pages = 0, stat_exhaustive = 0, stat_swept_rows = 0;
    // a query between mmiss_index_query_begin and mmiss_index_query_end: its first pass is queued, `done_ev` marks its end
    struct Pending {
        bool active = false, out_dev = false, guard = false;
        int Q = 0, k = 0;
        int64_t* d_lab = nullptr; float* d_dist = nullptr; int32_t* d_cnt = nullptr; int32_t* h_flags = nullptr;
        int64_t* out_labels = nullptr; float* out_dist = nullptr; int32_t* out_count = nullptr;
    } pend;
    DevBuf dist_all;  // exhaustive fallback: one canonical distance per row
    float* dist_pin = nullptr;   // ... and the pinned host block they come back through
    int64_t dist_pin_rows = 0;
    hipStream_t stream() const { return has_user_stream ? user_stream : own_stream; }
};

// every other entry point refuses to run between query_begin and query_end (the scratch buffers belong to that query)
#define MM_NO_PENDING(ix, who) \
    do { if ((ix)->pend.active) MM_FAIL(MMISS_ERR_STATE, "%s: a query begun with mmiss_index_query_begin is still open", who); } while (0)

int mmiss_index_build_flags(void) {
#if defined(P256_NO_LATE_WAIT) || defined(P256_SPLIT_STAGE) || defined(P256_STAGE_FIRST) || defined(P256_A_POLICY) || defined(P256_W_POLICY) || defined(P256_PRIO) || defined(MMISS_SCAN_NT)
    return 2;
#else
    return 0;
#endif
}

namespace {

int index_reserve(mmiss_index* ix, int64_t need, hipStream_t st) {
    if (need <= ix->capacity) return MMISS_OK;
    int64_t cap = ix->capacity > 0 ? ix->capacity : 1024;
    while (cap < need) cap *= 2;
    DevBuf nr, nl, ni;
    MM_TRY(nr.alloc((size_t)cap * ix->dim * ix->elt));
    MM_TRY(nl.alloc((size_t)cap * 8));
    if (ix->dtype == MMISS_F8) {
        // zero behind the rows: the score GEMM reads the inverse norms of a whole 256-row tile, the scan of a 16-row one
        MM_TRY(ni.alloc((size_t)cap * 4));
        MM_HIP(hipMemsetAsync(ni.p, 0, (size_t)cap * 4, st));
    }
    if (ix->count > 0) {
        MM_HIP(hipMemcpyAsync(nr.p, ix->rows.p, (size_t)ix->count * ix->dim * ix->elt, hipMemcpyDeviceToDevice, st));
        MM_HIP(hipMemcpyAsync(nl.p, ix->labels_d.p, (size_t)ix->count * 8, hipMemcpyDeviceToDevice, st));
        if (ni.p) MM_HIP(hipMemcpyAsync(ni.p, ix->inv.p, (size_t)ix->count * 4, hipMemcpyDeviceToDevice, st));
    }
    MM_HIP(hipStreamSynchronize(st));
    std::swap(ix->rows.p, nr.p); std::swap(ix->rows.bytes, nr.bytes);
    std::swap(ix->labels_d.p, nl.p); std::swap(ix->labels_d.bytes, nl.bytes);
    std::swap(ix->inv.p, ni.p); std::swap(ix->inv.bytes, ni.bytes);
    ix->capacity = cap;
    return MMISS_OK;
}

// the inverse norms of fp8 rows [row0, row0 + n) from their codes (after every write of rows: add, update, load)
int launch_f8_inv(mmiss_index* ix, int64_t row0, int64_t n, hipStream_t st) {
    if (n <= 0 || ix->dtype != MMISS_F8) return MMISS_OK;
    hipLaunchKernelGGL(f8_row_inv_kernel, dim3((int)((n + 3) / 4)), dim3(256), 0, st, ix->rows.as<uint8_t>() + (size_t)row0 * ix->dim, n,
                       ix->dim, ix->inv.as<float>() + row0);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// `row0`: the index row dst stands for (fp8 rows: where the inverse norms go)
int launch_normalize(mmiss_index* ix, const float* src_dev, void* dst, int64_t n, hipStream_t st, int64_t row0) {
    if (n <= 0) return MMISS_OK;
    MM_PROF("normalize_rows", st, 4.0 * n * ix->dim, (double)n * ix->dim * (4 + ix->elt));
    const int grid = (int)((n + 3) / 4);
    if (ix->dtype == MMISS_F16)
        hipLaunchKernelGGL(normalize_rows_kernel<_Float16>, dim3(grid), dim3(256), 0, st, src_dev, (_Float16*)dst, n, ix->dim);
    else if (ix->dtype == MMISS_F8) {
        hipLaunchKernelGGL(normalize_rows_kernel<F8>, dim3(grid), dim3(256), 0, st, src_dev, (F8*)dst, n, ix->dim);
        MM_HIP(hipGetLastError());
        return launch_f8_inv(ix, row0, n, st);
    } else
        hipLaunchKernelGGL(normalize_rows_kernel<float>, dim3(grid), dim3(256), 0, st, src_dev, (float*)dst, n, ix->dim);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// host copy of a (host or device) int64 array
int fetch_i64(const int64_t* p, int64_t n, std::vector<int64_t>& out) {
    out.resize((size_t)n);
    if (n == 0) return MMISS_OK;
    if (mmiss_is_device_ptr(p))
        MM_HIP(hipMemcpy(out.data(), p, (size_t)n * 8, hipMemcpyDeviceToHost));
    else
        memcpy(out.data(), p, (size_t)n * 8);
    return MMISS_OK;
}

int64_t find_row(const mmiss_index* ix, int64_t label) {
    auto it = std::lower_bound(ix->labels_h.begin(), ix->labels_h.end(), label);
    if (it == ix->labels_h.end() || *it != label) return -1;
    return it - ix->labels_h.begin();
}

template <typename T, int NQT, int CAP, int GS = 8>
int launch_scan_t(hipStream_t st, const ScanArgs& a, int slabs, int qtiles, int lds) {
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&scan_topk_kernel<T, NQT, CAP, GS>), lds));
    hipLaunchKernelGGL((scan_topk_kernel<T, NQT, CAP, GS>), dim3(slabs, qtiles), dim3(256), lds, st, a);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <typename T, int NQT>
int launch_scan_thr_t(hipStream_t st, const ScanArgs& a, int slabs, int qtiles, int lds) {
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&scan_topk_kernel<T, NQT, 32, 8, true>), lds));
    hipLaunchKernelGGL((scan_topk_kernel<T, NQT, 32, 8, true>), dim3(slabs, qtiles), dim3(256), lds, st, a);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

struct ScanPlan {
    int nqt, cap, lds, slabs, qtiles, tiles_per_block;
};

constexpr int SCAN_LDS_LIMIT = 160 * 1024;
// dynamic LDS of the streaming scan for a block of 16 * nqt queries: the staged query rows (+ 16 B of padding each) and,
// in list mode, four waves' (score, row) lists of `cap` entries per query with their counts and thresholds
int scan_lds_bytes(int D, int qelt, int nqt, int cap) {
    const int NQ = 16 * nqt;
    return (((NQ * (D * qelt + 16)) + 15) & ~15) + 2 * 4 * NQ * cap * 4 + 2 * 4 * NQ * 4;
}
// ... in threshold mode (the widen pass): the query rows only
int sweep_scan_lds(int D, int qelt, int nqt) { return ((16 * nqt * (D * qelt + 16)) + 15) & ~15; }

ScanPlan plan_scan(int D, int elt, int Q, int kp, int64_t N) {
    ScanPlan p{};
    const int limit = SCAN_LDS_LIMIT;
    p.nqt = 1; p.cap = 64;
    if (kp <= 16 && Q > 16) {
        for (int nqt : {4, 2}) {
            const int NQ = 16 * nqt;
            const int lds = (((NQ * (D * elt + 16)) + 15) & ~15) + 2 * 4 * NQ * 32 * 4 + 2 * 4 * NQ * 4;
            if (lds <= limit && (Q > 16 * nqt / 2)) { p.nqt = nqt; p.cap = 32; break; }
        }
    }
    const int NQ = 16 * p.nqt;
    p.lds = (((NQ * (D * elt + 16)) + 15) & ~15) + 2 * 4 * NQ * p.cap * 4 + 2 * 4 * NQ * 4;
    p.qtiles = (Q + NQ - 1) / NQ;
    const int64_t ntiles = (N + 15) / 16;
    const int blocks_per_cu = std::max(1, std::min(8, limit / p.lds));
    int64_t target = (int64_t)256 * blocks_per_cu * mmiss_option("scan_rounds", 1) / p.qtiles;
    if (target < 1) target = 1;
    if (target > mmiss_option("scan_max_slabs", 1024)) target = mmiss_option("scan_max_slabs", 1024);
    int64_t tpb = (ntiles + target - 1) / target;
    tpb = (tpb + 3) / 4 * 4;  // every wave of a block gets the same number of tiles
    if (tpb < 4) tpb = 4;
    p.tiles_per_block = (int)tpb;
    p.slabs = (int)((ntiles + tpb - 1) / tpb);
    return p;
}

int launch_scan(mmiss_index* ix, hipStream_t st, const ScanArgs& a, const ScanPlan& p) {
    const double flops = 2.0 * a.Q * (double)a.N * a.D;
    const double bytes = (double)a.N * a.D * ix->elt;
    MM_PROF(ix->dtype == MMISS_F16 ? "scan_topk_f16" : ix->dtype == MMISS_F8 ? "scan_topk_f8" : "scan_topk_f32", st, flops, bytes);
    if (ix->dtype == MMISS_F8) {
        if (p.nqt == 1) return launch_scan_t<F8, 1, 64>(st, a, p.slabs, p.qtiles, p.lds);
        if (p.nqt == 2) return launch_scan_t<F8, 2, 32>(st, a, p.slabs, p.qtiles, p.lds);
        return launch_scan_t<F8, 4, 32>(st, a, p.slabs, p.qtiles, p.lds);
    }
    if (ix->dtype == MMISS_F16) {
        if (p.nqt == 1) {
            if (mmiss_option("scan_group", 8) == 16) return launch_scan_t<_Float16, 1, 64, 16>(st, a, p.slabs, p.qtiles, p.lds);
            return launch_scan_t<_Float16, 1, 64>(st, a, p.slabs, p.qtiles, p.lds);
        }
        if (p.nqt == 2) return launch_scan_t<_Float16, 2, 32>(st, a, p.slabs, p.qtiles, p.lds);
        return launch_scan_t<_Float16, 4, 32>(st, a, p.slabs, p.qtiles, p.lds);
    }
    if (p.nqt == 1) return launch_scan_t<float, 1, 64>(st, a, p.slabs, p.qtiles, p.lds);
    if (p.nqt == 2) return launch_scan_t<float, 2, 32>(st, a, p.slabs, p.qtiles, p.lds);
    return launch_scan_t<float, 4, 32>(st, a, p.slabs, p.qtiles, p.lds);
}

// merge L per-slab lists of every query into its candidate page; above 64 lists in two levels so that a single
// query (the API path) is not merged by one lone workgroup
int launch_merge(mmiss_index* ix, hipStream_t st, MergeArgs m) {
    if (m.L > mmiss_option("merge_two_level_min", 64)) {
        const int per = 32;
        const int S = (m.L + per - 1) / per;
        MM_TRY(ix->lists2_s.ensure((size_t)S * m.Q * m.kp * 4));
        MM_TRY(ix->lists2_r.ensure((size_t)S * m.Q * m.kp * 4));
        MergeArgs m1 = m;
        m1.lists_per_block = per; m1.out_s = ix->lists2_s.as<float>(); m1.out_r = ix->lists2_r.as<int32_t>();
        m1.cur_s = nullptr; m1.cur_r = nullptr;
        {
            MM_PROF("merge_lists", st, 0.0, (double)m.L * m.Q * m.kp * 8);
            hipLaunchKernelGGL(merge_lists_kernel, dim3(m.Q, S), dim3(256), 0, st, m1);
            MM_HIP(hipGetLastError());
        }
        m.in_s = ix->lists2_s.as<float>(); m.in_r = ix->lists2_r.as<int32_t>(); m.L = S;
    }
    m.lists_per_block = 0; m.out_s = nullptr; m.out_r = nullptr;
    MM_PROF("merge_lists", st, 0.0, (double)m.L * m.Q * m.kp * 8);
    hipLaunchKernelGGL(merge_lists_kernel, dim3(m.Q), dim3(256), 0, st, m);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

}  // namespace

// ================================================================================================ lifecycle
extern "C" int mmiss_index_create(int32_t dim, int32_t storage_dtype, int device, int64_t capacity_hint,
                                  mmiss_index** out) {
    if (!out) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_create: null out");
    if (dim <= 0 || dim % 128 || dim > 4096) MM_FAIL(MMISS_ERR_UNSUPPORTED, "index dim %d: need a multiple of 128, <= 4096", dim);
    if (storage_dtype != MMISS_F32 && storage_dtype != MMISS_F16 && storage_dtype != MMISS_F8)
        MM_FAIL(MMISS_ERR_ARG, "unknown storage dtype %d", storage_dtype);
    // the streaming scan stages 16 queries in the storage's query dtype (f32 rows: f32 queries) beside its lists: f32 rows
    // fit up to dim 1920, f16 / fp8 rows up to 3968 (a create-time answer instead of a failed launch at the first query)
    if (scan_lds_bytes(dim, storage_dtype == MMISS_F32 ? 4 : 2, 1, 64) > SCAN_LDS_LIMIT)
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "index dim %d with storage dtype %d: a 16-query block of the scan needs %d bytes of LDS (limit %d); "
                "f32 rows go up to dim 1920, f16 / fp8 rows up to 3968", dim, storage_dtype,
                scan_lds_bytes(dim, storage_dtype == MMISS_F32 ? 4 : 2, 1, 64), SCAN_LDS_LIMIT);
    MM_TRY(mmiss_use_device(device));
    mmiss_index* ix = new (std::nothrow) mmiss_index();
    if (!ix) MM_FAIL(MMISS_ERR_NOMEM, "out of host memory");
    ix->dim = dim; ix->dtype = storage_dtype; ix->device = device; ix->elt = storage_dtype == MMISS_F16 ? 2 : storage_dtype == MMISS_F8 ? 1 : 4;
    if (hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ix;
        MM_FAIL(MMISS_ERR_HIP, "hipStreamCreate failed");
    }
    if (hipEventCreateWithFlags(&ix->done_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(ix->own_stream);
        delete ix;
        MM_FAIL(MMISS_ERR_HIP, "hipEventCreate failed");
    }
    if (capacity_hint > 0) {
        int rc = index_reserve(ix, capacity_hint, ix->own_stream);
        if (rc != MMISS_OK) {
            (void)hipStreamDestroy(ix->own_stream); (void)hipEventDestroy(ix->done_ev);
            delete ix;
            return rc;
        }
    }
    *out = ix;
    return MMISS_OK;
}

extern "C" int mmiss_index_destroy(mmiss_index* ix) {
    if (!ix) return MMISS_OK;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
    if (ix->pin) (void)hipHostFree(ix->pin);
    if (ix->swp_pin) (void)hipHostFree(ix->swp_pin);
    if (ix->dist_pin) (void)hipHostFree(ix->dist_pin);
    if (ix->done_ev) (void)hipEventDestroy(ix->done_ev);
    delete ix;
    return MMISS_OK;
}

extern "C" int mmiss_index_set_stream(mmiss_index* ix, void* hip_stream, int32_t use_own) {
    if (!ix) MM_FAIL(MMISS_ERR_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_set_stream");
    hipStream_t next = reinterpret_cast<hipStream_t>(hip_stream);
    const bool next_user = use_own == 0;
    if (next_user != ix->has_user_stream || (next_user && next != ix->user_stream)) {
        // the handle's workspaces are shared by consecutive calls: wait for what the last call left on the stream being
        // left (calls on the handle's own stream, and calls with host outputs, return drained)
        MM_TRY(mmiss_use_device(ix->device));
        if (ix->async_pending) MM_HIP(hipEventSynchronize(ix->done_ev));
        ix->async_pending = false;
    }
    ix->user_stream = next;
    ix->has_user_stream = next_user;
    return MMISS_OK;
}

extern "C" int mmiss_index_count(mmiss_index* ix, int64_t* count) {
    if (!ix || !count) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_count: null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    *count = ix->count;
    return MMISS_OK;
}

extern "C" int mmiss_index_clear(mmiss_index* ix) {
    if (!ix) MM_FAIL(MMISS_ERR_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_clear");
    if (ix->dtype == MMISS_F8 && ix->capacity > 0) {   // (inverse norms are zero behind the rows)
        MM_TRY(mmiss_use_device(ix->device));
        MM_HIP(hipMemsetAsync(ix->inv.p, 0, (size_t)ix->capacity * 4, ix->stream()));
        MM_HIP(hipStreamSynchronize(ix->stream()));
    }
    ix->count = 0;
    ix->labels_h.clear();
    return MMISS_OK;
}

// ================================================================================================ mutation
extern "C" int mmiss_index_add(mmiss_index* ix, const float* vecs, const int64_t* labels, int64_t n) {
    if (!ix || (n > 0 && (!vecs || !labels))) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_add: null argument");
    if (n < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_add: n = %lld", (long long)n);
    if (n == 0) return MMISS_OK;
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_add");
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    std::vector<int64_t> lab;
    MM_TRY(fetch_i64(labels, n, lab));
    int64_t prev = ix->labels_h.empty() ? INT64_MIN : ix->labels_h.back();
    for (int64_t i = 0; i < n; ++i) {
        if (lab[i] < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_add: negative label %lld", (long long)lab[i]);
        if (lab[i] <= prev)
            MM_FAIL(MMISS_ERR_ARG, "mmiss_index_add: labels must be strictly increasing (label %lld after %lld)",
                    (long long)lab[i], (long long)prev);
        prev = lab[i];
    }
    if (ix->count + n >= (int64_t)INT32_MAX - 64) MM_FAIL(MMISS_ERR_UNSUPPORTED, "index shard limited to 2^31 rows");
    MM_TRY(index_reserve(ix, ix->count + n, st));
    const int D = ix->dim;
    const bool in_dev = mmiss_is_device_ptr(vecs);
    const int64_t chunk = 1 << 16;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t nr = std::min(chunk, n - r0);
        const float* src = vecs + r0 * D;
        if (!in_dev) {
            MM_TRY(ix->stage.ensure((size_t)chunk * D * 4));
            MM_HIP(hipMemcpyAsync(ix->stage.p, src, (size_t)nr * D * 4, hipMemcpyHostToDevice, st));
            src = ix->stage.as<float>();
        }
        MM_TRY(launch_normalize(ix, src, ix->rows.as<char>() + (size_t)(ix->count + r0) * D * ix->elt, nr, st, ix->count + r0));
        if (!in_dev) MM_HIP(hipStreamSynchronize(st));
    }
    MM_HIP(hipMemcpyAsync(ix->labels_d.as<int64_t>() + ix->count, lab.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    MM_HIP(hipStreamSynchronize(st));
    ix->labels_h.insert(ix->labels_h.end(), lab.begin(), lab.end());
    ix->count += n;
    return MMISS_OK;
}

extern "C" int mmiss_index_update(mmiss_index* ix, const int64_t* labels, const float* vecs, int64_t n) {
    if (!ix || (n > 0 && (!vecs || !labels))) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_update: null argument");
    if (n <= 0) return n == 0 ? MMISS_OK : MMISS_ERR_ARG;
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_update");
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    std::vector<int64_t> lab;
    MM_TRY(fetch_i64(labels, n, lab));
    std::vector<int64_t> rows((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        rows[i] = find_row(ix, lab[i]);
        if (rows[i] < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_update: label %lld not in the index", (long long)lab[i]);
    }
    const int D = ix->dim;
    const float* src = vecs;
    if (!mmiss_is_device_ptr(vecs)) {
        MM_TRY(ix->stage.ensure((size_t)n * D * 4));
        MM_HIP(hipMemcpyAsync(ix->stage.p, vecs, (size_t)n * D * 4, hipMemcpyHostToDevice, st));
        src = ix->stage.as<float>();
    }
    for (int64_t i = 0; i < n; ++i)
        MM_TRY(launch_normalize(ix, src + i * D, ix->rows.as<char>() + (size_t)rows[i] * D * ix->elt, 1, st, rows[i]));
    MM_HIP(hipStreamSynchronize(st));
    return MMISS_OK;
}

extern "C" int mmiss_index_remove(mmiss_index* ix, const int64_t* labels, int64_t n, int64_t* removed) {
    if (!ix || (n > 0 && !labels)) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_remove: null argument");
    if (removed) *removed = 0;
    if (n <= 0) return n == 0 ? MMISS_OK : MMISS_ERR_ARG;
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_remove");
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    std::vector<int64_t> lab;
    MM_TRY(fetch_i64(labels, n, lab));
    std::vector<char> drop((size_t)ix->count, 0);
    int64_t ndrop = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = find_row(ix, lab[i]);
        if (r >= 0 && !drop[r]) { drop[r] = 1; ++ndrop; }
    }
    if (removed) *removed = ndrop;
    if (ndrop == 0) return MMISS_OK;
    const int64_t keep = ix->count - ndrop;
    std::vector<int64_t> map((size_t)keep), nl((size_t)keep);
    for (int64_t r = 0, o = 0; r < ix->count; ++r)
        if (!drop[r]) { map[o] = r; nl[o] = ix->labels_h[r]; ++o; }
    if (keep > 0) {
        // stable compaction into fresh buffers (row order == label order is the tie-break contract)
        DevBuf nr, nlab, ninv;
        MM_TRY(nr.alloc((size_t)ix->capacity * ix->dim * ix->elt));
        MM_TRY(nlab.alloc((size_t)ix->capacity * 8));
        if (ix->dtype == MMISS_F8) {
            MM_TRY(ninv.alloc((size_t)ix->capacity * 4));
            MM_HIP(hipMemsetAsync(ninv.p, 0, (size_t)ix->capacity * 4, st));
        }
        MM_TRY(ix->map.ensure((size_t)keep * 8));
        MM_HIP(hipMemcpyAsync(ix->map.p, map.data(), (size_t)keep * 8, hipMemcpyHostToDevice, st));
        const int grid = (int)std::min<int64_t>(4096, (keep * ix->dim + 255) / 256);
        if (ix->dtype == MMISS_F16)
            hipLaunchKernelGGL(gather_rows_kernel<_Float16>, dim3(grid), dim3(256), 0, st, ix->rows.as<_Float16>(),
                               ix->map.as<int64_t>(), nr.as<_Float16>(), keep, ix->dim);
        else if (ix->dtype == MMISS_F8) {
            hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, ix->rows.as<uint8_t>(),
                               ix->map.as<int64_t>(), nr.as<uint8_t>(), keep, ix->dim);
            hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((int)std::min<int64_t>(4096, (keep + 255) / 256)), dim3(256), 0, st,
                               ix->inv.as<float>(), ix->map.as<int64_t>(), ninv.as<float>(), keep, 1);
        } else
            hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid), dim3(256), 0, st, ix->rows.as<float>(),
                               ix->map.as<int64_t>(), nr.as<float>(), keep, ix->dim);
        MM_HIP(hipGetLastError());
        MM_HIP(hipMemcpyAsync(nlab.p, nl.data(), (size_t)keep * 8, hipMemcpyHostToDevice, st));
        MM_HIP(hipStreamSynchronize(st));
        std::swap(ix->rows.p, nr.p); std::swap(ix->rows.bytes, nr.bytes);
        std::swap(ix->labels_d.p, nlab.p); std::swap(ix->labels_d.bytes, nlab.bytes);
        std::swap(ix->inv.p, ninv.p); std::swap(ix->inv.bytes, ninv.bytes);
    }
    else if (ix->dtype == MMISS_F8 && ix->capacity > 0) {   // every row removed: the inverse norms stay zero behind `count` (mmiss_index_clear's invariant)
        MM_HIP(hipMemsetAsync(ix->inv.p, 0, (size_t)ix->capacity * 4, st));
        MM_HIP(hipStreamSynchronize(st));
    }
    ix->labels_h.swap(nl);
    ix->count = keep;
    return MMISS_OK;
}

extern "C" int mmiss_index_get(mmiss_index* ix, const int64_t* labels, int64_t n, float* out) {
    if (!ix || (n > 0 && (!labels || !out))) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_get: null argument");
    if (n <= 0) return n == 0 ? MMISS_OK : MMISS_ERR_ARG;
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_get");
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    std::vector<int64_t> lab;
    MM_TRY(fetch_i64(labels, n, lab));
    std::vector<int64_t> map((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        map[i] = find_row(ix, lab[i]);
        if (map[i] < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_get: label %lld not in the index", (long long)lab[i]);
    }
    MM_TRY(ix->map.ensure((size_t)n * 8));
    MM_HIP(hipMemcpyAsync(ix->map.p, map.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    const bool out_dev = mmiss_is_device_ptr(out);
    float* dst = out;
    if (!out_dev) { MM_TRY(ix->stage.ensure((size_t)n * ix->dim * 4)); dst = ix->stage.as<float>(); }
    const int grid = (int)std::min<int64_t>(4096, (n * ix->dim + 255) / 256);
    if (ix->dtype == MMISS_F16)
        hipLaunchKernelGGL(gather_rows_f32_kernel<_Float16>, dim3(grid), dim3(256), 0, st, ix->rows.as<_Float16>(),
                           ix->map.as<int64_t>(), dst, n, ix->dim);
    else if (ix->dtype == MMISS_F8)   // the vectors the rows represent: values x inverse norm
        hipLaunchKernelGGL(gather_rows_f8_f32_kernel, dim3(grid), dim3(256), 0, st, ix->rows.as<F8>(), ix->inv.as<float>(),
                           ix->map.as<int64_t>(), dst, n, ix->dim);
    else
        hipLaunchKernelGGL(gather_rows_f32_kernel<float>, dim3(grid), dim3(256), 0, st, ix->rows.as<float>(),
                           ix->map.as<int64_t>(), dst, n, ix->dim);
    MM_HIP(hipGetLastError());
    if (!out_dev) MM_HIP(hipMemcpyAsync(out, dst, (size_t)n * ix->dim * 4, hipMemcpyDeviceToHost, st));
    MM_HIP(hipStreamSynchronize(st));
    return MMISS_OK;
}

extern "C" int mmiss_index_labels(mmiss_index* ix, int64_t* out, int64_t cap) {
    if (!ix || (!out && cap > 0)) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_labels: null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    const int64_t n = std::min<int64_t>(cap, ix->count);
    if (n > 0) memcpy(out, ix->labels_h.data(), (size_t)n * 8);
    return MMISS_OK;
}

// ================================================================================================ exactness contract
// Stage 1 orders rows by APPROXIMATE scores (matrix cores, f32 accumulation, f16 query operand for f16 / fp8 rows) and keeps
// k' > k of them; stage 2 re-scores the survivors canonically (fp64, fixed order) and sorts. That is exact by construction
// only if no row left out by stage 1 can belong to the true top-k, so stage 2 PROVES it per query:
//   every left-out row r has approx(r) <= tau (tau = the k'-th approximate score / group maximum),
//   |approx(r) - canonical(r)| <= eps_q for every row (bound below), so canonical(r) <= tau + eps_q;
//   if the k-th canonical score c_k of the candidates satisfies c_k - tau > eps_q, nothing left out reaches c_k.
// A query that fails the test is WIDENED by ONE threshold pass (round 4; it used to page through the index 32 rows per
// full scan): c_k is a lower bound of the true k-th canonical score, so every row of the true top-k has
// approx >= c_k - eps_q =: thr_q. One more pass over the index — the score GEMM for many flagged queries of an f16 index, the
// streaming scan otherwise — appends EVERY row with approx >= thr_q to the query's list; the canonical re-rank of that list
// is the exact answer (rows not on it have canonical <= approx + eps_q < c_k <= true c_k: strictly behind rank k, whatever
// the tie-break). A list that overflows its capacity (a plateau: more than SWEEP_CAP rows within eps of the k-th score,
// e.g. 10^5 uploads of one placeholder image) sends the query to the exhaustive canonical pass. Exact for any data; on
// random data the test fails for about one query in 10^4, on the encoder's own embeddings (pairwise cosine 0.99) for all.
namespace {

constexpr int SWEEP_CAP = 8192;   // rows a widened query may collect (= the re-rank kernel's sort capacity)

// the guard's bound, split into what is the same for every query and the factor of the query operand's ACTUAL rounding error
// |qs - qn|_2 (prep_queries_kernel computes eps_q = fixed + cnorm * |qs - qn|_2 per query):
//   f32 accumulation of D exact products in hardware order: <= 4 D 2^-24 sum|q_d c_d| (factor 4: margin for the MFMA's
//   internal alignment / truncation) <= 4 D 2^-24 |q||c|, |q||c| <= (1 + 2^-10)^2 (f16 / f32 rows);
//   f16 / fp8 rows: the scan's query operand is f16(qn); <f16(qn) - qn, c> <= |f16(qn) - qn|_2 |c|_2 (+ D 2^-25: products
//   of subnormal-range operands); a stored fp8 row's norm is within 1.07 of 1 (three mantissa bits per component);
//   the float rounding of the canonical distance (<= 2) is 2^-23 at most.
void guard_terms(const mmiss_index* ix, double* fixed, double* cnorm) {
    const double D = ix->dim;
    // (fp8 rows: |values| x inv = 1 up to inv's own rounding; 1.07 is kept from the un-renormalised form of round 4 — it only
    // widens the bound — and the scan's multiplication by inv adds one more f32 rounding of a score <= 1.08)
    const double cn = ix->dtype == MMISS_F8 ? 1.07 : 1.0 + ldexp(1.0, -9);
    double e = 4.0 * D * ldexp(1.0, -24) * cn * (1.0 + ldexp(1.0, -9));
    if (ix->dtype != MMISS_F32) e += D * ldexp(1.0, -25) * cn;
    if (ix->dtype == MMISS_F8) e += ldexp(1.0, -23);
    *fixed = e * 1.003 + ldexp(1.0, -21);
    *cnorm = ix->dtype == MMISS_F32 ? 0.0 : cn * 1.003;
}

int launch_rerank(mmiss_index* ix, hipStream_t st, const RerankArgs& r, int blocks) {
    int npow = 1;
    while (npow < r.ncand) npow <<= 1;
    const int lds = npow * 8 + 16;
    MM_PROF("rerank", st, 2.0 * blocks * r.ncand * r.D, (double)blocks * r.ncand * r.D * ix->elt);
    const int threads = r.ncand >= 16 ? 1024 : 256;  // one wave per candidate row: more waves hide the gather latency
    if (ix->dtype == MMISS_F16) {
        if (lds > 32768) MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&rerank_kernel<_Float16>), lds));
        hipLaunchKernelGGL(rerank_kernel<_Float16>, dim3(blocks), dim3(threads), lds, st, r);
    } else if (ix->dtype == MMISS_F8) {
        if (lds > 32768) MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&rerank_kernel<F8>), lds));
        hipLaunchKernelGGL(rerank_kernel<F8>, dim3(blocks), dim3(threads), lds, st, r);
    } else {
        if (lds > 32768) MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&rerank_kernel<float>), lds));
        hipLaunchKernelGGL(rerank_kernel<float>, dim3(blocks), dim3(threads), lds, st, r);
    }
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// Exhaustive canonical pass for the queries listed in `which`: canonical distance of every row (canonical_scan_kernel), the
// k best by (distance, row) picked on the host (a bounded heap: O(N log k)), final ordering / labels / counts by the ordinary
// rerank kernel with "nothing was left out" (tau = -inf).
int exhaustive_queries(mmiss_index* ix, hipStream_t st, const std::vector<int32_t>& which, int k, int64_t* d_lab, float* d_dist,
                       int32_t* d_cnt) {
    const int64_t N = ix->count;
    const int D = ix->dim, KP = 32;
    const int kk = (int)std::min<int64_t>(k, N);
    const int kkpad = (int)round_up(std::max(kk, 1), KP);
    const int nq = (int)which.size();
    if (nq == 0 || N <= 0) return MMISS_OK;
    MM_TRY(ix->dist_all.ensure((size_t)N * 4));
    // the distances come back through a PINNED block (a pageable destination makes the copy synchronous and slow: ADVICE r3)
    if (ix->dist_pin_rows < N) {
        if (ix->dist_pin) { MM_HIP(hipHostFree(ix->dist_pin)); ix->dist_pin = nullptr; ix->dist_pin_rows = 0; }
        MM_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->dist_pin), (size_t)N * 4, hipHostMallocDefault));
        ix->dist_pin_rows = N;
    }
    const float* dist = ix->dist_pin;
    std::vector<int32_t> cand((size_t)nq * kkpad, -1);
    typedef std::pair<float, int32_t> Ent;  // (distance, row); NaN distances never enter
    for (int i = 0; i < nq; ++i) {
        const int32_t q = which[i];
        {
            MM_PROF("canonical_scan", st, 2.0 * N * D, (double)N * D * ix->elt);
            const int grid = (int)std::min<int64_t>(8192, (N + 15) / 16);
            if (ix->dtype == MMISS_F16)
                hipLaunchKernelGGL(canonical_scan_kernel<_Float16>, dim3(grid), dim3(256), 0, st, ix->rows.p, N, D,
                                   ix->qn.as<float>() + (size_t)q * D, ix->dist_all.as<float>(), (const float*)nullptr);
            else if (ix->dtype == MMISS_F8)
                hipLaunchKernelGGL(canonical_scan_kernel<F8>, dim3(grid), dim3(256), 0, st, ix->rows.p, N, D,
                                   ix->qn.as<float>() + (size_t)q * D, ix->dist_all.as<float>(), (const float*)ix->inv.as<float>());
            else
                hipLaunchKernelGGL(canonical_scan_kernel<float>, dim3(grid), dim3(256), 0, st, ix->rows.p, N, D,
                                   ix->qn.as<float>() + (size_t)q * D, ix->dist_all.as<float>(), (const float*)nullptr);
            MM_HIP(hipGetLastError());
        }
        MM_HIP(hipMemcpyAsync(ix->dist_pin, ix->dist_all.p, (size_t)N * 4, hipMemcpyDeviceToHost, st));
        MM_HIP(hipStreamSynchronize(st));
        std::priority_queue<Ent> heap;  // max-heap: top = the worst of the best kk so far, by (distance, row)
        for (int64_t r = 0; r < N; ++r) {
            const float d = dist[(size_t)r];
            if (d != d) continue;
            if ((int)heap.size() < kk) heap.push(Ent(d, (int32_t)r));
            else if (Ent(d, (int32_t)r) < heap.top()) { heap.pop(); heap.push(Ent(d, (int32_t)r)); }
        }
        int32_t* c = cand.data() + (size_t)i * kkpad;
        for (size_t j = 0; !heap.empty(); ++j) { c[j] = heap.top().second; heap.pop(); }
        ix->stat_exhaustive += 1;
    }
    // ONE re-rank launch for all of them (final ordering, labels, counts): block b = query which[b], its kkpad selected rows
    MM_TRY(ix->cand2.ensure((size_t)nq * kkpad * 4));
    MM_TRY(ix->qmap.ensure((size_t)nq * 4));
    MM_HIP(hipMemcpyAsync(ix->cand2.p, cand.data(), (size_t)nq * kkpad * 4, hipMemcpyHostToDevice, st));
    MM_HIP(hipMemcpyAsync(ix->qmap.p, which.data(), (size_t)nq * 4, hipMemcpyHostToDevice, st));
    RerankArgs r{};
    r.rows = ix->rows.p; r.D = D; r.inv = ix->inv.as<float>(); r.qn = ix->qn.as<float>(); r.cand = ix->cand2.as<int32_t>();
    r.cand_stride = kkpad; r.ncand = kkpad; r.group_mode = 0; r.nrows = N;
    r.labels = ix->labels_d.as<int64_t>(); r.k = k;
    r.out_labels = d_lab; r.out_dist = d_dist; r.out_count = d_cnt;
    r.qmap = ix->qmap.as<int32_t>();
    MM_TRY(launch_rerank(ix, st, r, nq));
    MM_HIP(hipStreamSynchronize(st));  // (`cand` and `which` leave scope with the caller)
    return MMISS_OK;
}

// consecutive index tiles per workgroup of the strip score GEMM (Mq queries padded to 256, ncol column tiles of 256 rows):
// long enough to amortise the pipeline fill, short enough to leave >= 8 workgroups per CU for balance; among the strip
// lengths down to half that, the one whose rounds x length (+ a quarter tile of pipeline fill per workgroup and round) is
// smallest (10M rows x 1024 queries: strip 32 gave 4884 workgroups = 19.08 rounds of 256 -> 5 % of the launch spent with 18
// workgroups on the chip).
int strip_length(int Mq, int64_t nbn, int64_t ncol) {
    const int64_t tiles = nbn * (Mq / 256);
    int strip = (int)std::min<int64_t>(32, std::max<int64_t>(1, tiles / 2048));
    if (strip >= 8 && mmiss_option("score_strip_fit", 1) != 0) {
        double best = 1e300;
        int best_s = strip;
        for (int sl = strip; sl >= strip / 2; --sl) {
            const int64_t wgs = (Mq / 256) * ((ncol + sl - 1) / sl);
            const int64_t rounds = (wgs + 255) / 256;
            const double cost = (double)rounds * (sl + 0.25);
            if (cost < best - 1e-9) { best = cost; best_s = sl; }
        }
        strip = best_s;
    }
    const int forced = mmiss_option("score_strip", 0);
    if (forced > 0) strip = forced;
    return strip;
}

// The widen pass for the queries listed in `which` (original indices): ONE threshold pass over the index for all of them
// (thr_q = c_k - eps_q, left in ix->thr by the first pass's re-rank), the canonical re-rank of the rows it collected, and the
// exhaustive pass for the lists that overflowed. Results overwrite the queries' rows of d_lab / d_dist / d_cnt. Returns with
// the stream drained.
int sweep_queries(mmiss_index* ix, hipStream_t st, const std::vector<int32_t>& which, int Q_all, int k, int64_t* d_lab,
                  float* d_dist, int32_t* d_cnt) {
    const int Qf = (int)which.size(), D = ix->dim;
    const int64_t N = ix->count;
    if (Qf == 0 || N <= 0) return MMISS_OK;
    // the score GEMM for an f16 index once the flagged queries fill a quarter of a 256-query tile (it costs about what 1.2
    // streaming scans cost, and a scan serves 64 queries), the scan otherwise
    const bool gemm8 = ix->dtype == MMISS_F8 && mmiss_option("score_f8_gemm", 1) != 0;
    const bool gemm = (ix->dtype == MMISS_F16 || gemm8) && (D % 128) == 0 && D >= 256 && Qf > mmiss_option("sweep_gemm_min_q", 64) &&
                      mmiss_option("score_strip_v3", 1) != 0;
    const int Qfp = (int)round_up(Qf, 256);
    MM_TRY(ix->swp_cnt.ensure((size_t)Qfp * 4));
    MM_TRY(ix->swp_list.ensure((size_t)Qf * SWEEP_CAP * 4));
    MM_HIP(hipMemsetAsync(ix->swp_cnt.p, 0, (size_t)Qfp * 4, st));
    // Every query of the call flagged (the encoder's own embeddings as the index: all of them, every call): the first pass's
    // operands are the sweep's operands — no gather, no map, nothing crosses PCIe. Otherwise the flagged queries are compacted.
    const bool all = Qf == Q_all;
    const void* qs_sw = ix->qs.p;          // [>= Qfp rows][D] scan operand (rows >= Qf are zero or masked by thr = +inf)
    const float* thr_sw = ix->thr.as<float>();
    const int32_t* qmap_sw = nullptr;
    if (!all) {
        std::vector<int64_t> map64(which.begin(), which.end());
        MM_TRY(ix->qmap.ensure((size_t)Qf * 4));
        MM_TRY(ix->qmap64.ensure((size_t)Qf * 8));
        MM_TRY(ix->qs2.ensure((size_t)Qfp * D * ix->qelt()));
        MM_TRY(ix->thr2.ensure((size_t)Qfp * 4));
        MM_HIP(hipMemcpyAsync(ix->qmap.p, which.data(), (size_t)Qf * 4, hipMemcpyHostToDevice, st));
        MM_HIP(hipMemcpyAsync(ix->qmap64.p, map64.data(), (size_t)Qf * 8, hipMemcpyHostToDevice, st));
        MM_HIP(hipStreamSynchronize(st));   // (pageable sources: `map64` goes out of scope)
        if (Qfp > Qf) MM_HIP(hipMemsetAsync(ix->qs2.as<char>() + (size_t)Qf * D * ix->qelt(), 0, (size_t)(Qfp - Qf) * D * ix->qelt(), st));
        const int grid = (int)std::min<int64_t>(4096, ((int64_t)Qf * D + 255) / 256);
        if (ix->dtype != MMISS_F32)   // (f16 query operand for f16 and fp8 rows)
            hipLaunchKernelGGL(gather_rows_kernel<_Float16>, dim3(grid), dim3(256), 0, st, ix->qs.as<_Float16>(),
                               ix->qmap64.as<int64_t>(), ix->qs2.as<_Float16>(), (int64_t)Qf, D);
        else
            hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid), dim3(256), 0, st, ix->qs.as<float>(),
                               ix->qmap64.as<int64_t>(), ix->qs2.as<float>(), (int64_t)Qf, D);
        MM_HIP(hipGetLastError());
        hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((Qf + 255) / 256), dim3(256), 0, st, ix->thr.as<float>(),
                           ix->qmap64.as<int64_t>(), ix->thr2.as<float>(), (int64_t)Qf, 1);
        MM_HIP(hipGetLastError());
        qs_sw = ix->qs2.p; thr_sw = ix->thr2.as<float>(); qmap_sw = ix->qmap.as<int32_t>();
    }
    if (gemm) {
        const int64_t Npad = round_up(N, 256), nbn = Npad / 256;
        GemmEpi ep{};
        ep.m_valid = Qf; ep.p0 = (int)N; ep.m_fast = 1;
        ep.aux = ix->inv.as<float>();   // (fp8 rows: inverse norms; null otherwise)
        StripFilter flt{};
        flt.tau = thr_sw; flt.tau_stride = 1; flt.cnt = ix->swp_cnt.as<int32_t>();
        flt.buf_s = nullptr; flt.buf_g = ix->swp_list.as<int32_t>(); flt.cap = SWEEP_CAP; flt.bn_begin = 0;
        const int strip = strip_length(Qfp, nbn, nbn);
        MM_PROF(gemm8 ? "sweep_gemm_f8" : "sweep_gemm_f16", st, 2.0 * Qf * (double)N * D, (double)N * D * ix->elt);
        if (gemm8) MM_TRY((launch_gemm256s<_Float16, true>(st, qs_sw, ix->rows.p, ep, Qfp, (int)Npad, D, strip, &flt)));
        else MM_TRY((launch_gemm256s<_Float16>(st, qs_sw, ix->rows.p, ep, Qfp, (int)Npad, D, strip, &flt)));
        ix->stat_pages += 1;
    } else {
        ScanArgs a{};
        a.rows = ix->rows.p; a.N = N; a.D = D; a.inv = ix->inv.as<float>(); a.qs = qs_sw; a.Q = Qf;
        a.thr = thr_sw; a.gcnt = ix->swp_cnt.as<int32_t>(); a.glist = ix->swp_list.as<int32_t>(); a.gcap = SWEEP_CAP;
        // the widest query tile the flagged queries fill AND whose staged query block fits the CU's LDS (f32 rows at the
        // reference's D = 768: 64 queries x 3088 B = 193 KB do not, 32 do). nqt = 1 always fits: mmiss_index_create
        // admits only dims whose 16-query block does.
        int nqt = Qf > 32 ? 4 : Qf > 16 ? 2 : 1;
        while (nqt > 1 && sweep_scan_lds(D, ix->qelt(), nqt) > SCAN_LDS_LIMIT) nqt >>= 1;
        const int NQ = 16 * nqt;
        const int lds = sweep_scan_lds(D, ix->qelt(), nqt);
        if (lds > SCAN_LDS_LIMIT)
            MM_FAIL(MMISS_ERR_UNSUPPORTED, "widen pass: a 16-query block of dim %d does not fit the LDS (%d bytes)", D, lds);
        const int qtiles = (Qf + NQ - 1) / NQ;
        const int64_t ntiles = (N + 15) / 16;
        const int per_cu = std::max(1, std::min(8, SCAN_LDS_LIMIT / lds));
        int64_t target = std::max<int64_t>(1, (int64_t)256 * per_cu / qtiles);
        int64_t tpb = std::max<int64_t>(4, (((ntiles + target - 1) / target) + 3) / 4 * 4);
        a.tiles_per_block = (int)tpb;
        const int slabs = (int)((ntiles + tpb - 1) / tpb);
        MM_PROF(ix->dtype == MMISS_F16 ? "sweep_scan_f16" : ix->dtype == MMISS_F8 ? "sweep_scan_f8" : "sweep_scan_f32", st,
                2.0 * Qf * (double)N * D, (double)qtiles * N * D * ix->elt);
        if (ix->dtype == MMISS_F16) {
            if (nqt == 4) MM_TRY((launch_scan_thr_t<_Float16, 4>(st, a, slabs, qtiles, lds)));
            else if (nqt == 2) MM_TRY((launch_scan_thr_t<_Float16, 2>(st, a, slabs, qtiles, lds)));
            else MM_TRY((launch_scan_thr_t<_Float16, 1>(st, a, slabs, qtiles, lds)));
        } else if (ix->dtype == MMISS_F8) {
            if (nqt == 4) MM_TRY((launch_scan_thr_t<F8, 4>(st, a, slabs, qtiles, lds)));
            else if (nqt == 2) MM_TRY((launch_scan_thr_t<F8, 2>(st, a, slabs, qtiles, lds)));
            else MM_TRY((launch_scan_thr_t<F8, 1>(st, a, slabs, qtiles, lds)));
        } else {
            if (nqt == 4) MM_TRY((launch_scan_thr_t<float, 4>(st, a, slabs, qtiles, lds)));
            else if (nqt == 2) MM_TRY((launch_scan_thr_t<float, 2>(st, a, slabs, qtiles, lds)));
            else MM_TRY((launch_scan_thr_t<float, 1>(st, a, slabs, qtiles, lds)));
        }
        ix->stat_pages += qtiles;
    }
    // Re-rank what was collected. The typical list is short (tens to hundreds of rows), so the first launch sorts at most
    // SWEEP_FAST rows per query WITHOUT waiting to learn the counts; the counts come back with it (one small copy behind the
    // kernels, one wait), and only the queries that collected more get a second, larger launch — or, past SWEEP_CAP, the
    // exhaustive pass.
    const int SWEEP_FAST = 1024;
    RerankArgs r{};
    r.rows = ix->rows.p; r.D = D; r.inv = ix->inv.as<float>(); r.qn = ix->qn.as<float>(); r.cand = ix->swp_list.as<int32_t>();
    r.cand_stride = SWEEP_CAP; r.ncand = SWEEP_FAST; r.cand_cnt = ix->swp_cnt.as<int32_t>();
    r.group_mode = 0; r.nrows = N;
    r.labels = ix->labels_d.as<int64_t>(); r.k = k;
    r.out_labels = d_lab; r.out_dist = d_dist; r.out_count = d_cnt;
    r.qmap = qmap_sw;
    // the list lengths come back with the re-rank itself: every block stores its list's raw length into a pinned block (a copy
    // into pageable memory behind the kernel was a second operation and a staged host copy in front of the next query's first
    // launch: part of the ~87 us the GPU idled per pipelined step, round 6)
    MM_TRY(ix->ensure_swp_pin((size_t)Qf));
    r.cnt_host = ix->swp_pin;
    MM_TRY(launch_rerank(ix, st, r, Qf));
    r.cnt_host = nullptr;
    MM_HIP(hipStreamSynchronize(st));
    const int32_t* cnt = ix->swp_pin;
    ix->stat_rounds += 1;
    std::vector<int32_t> big_b, rest;   // blocks that need the large re-rank / original indices for the exhaustive pass
    int maxc = 1;
    for (int b = 0; b < Qf; ++b) {
        if (cnt[b] > SWEEP_CAP) rest.push_back(which[b]);
        else {
            ix->stat_swept_rows += cnt[b];
            if (cnt[b] > SWEEP_FAST) { big_b.push_back(b); maxc = std::max(maxc, cnt[b]); }
        }
    }
    if (!big_b.empty()) {
        // block j of this launch = block big_b[j] of the sweep: its list and count by bmap, its query by qmap
        const int nb = (int)big_b.size();
        std::vector<int32_t> qm((size_t)nb);
        for (int j = 0; j < nb; ++j) qm[j] = which[big_b[j]];
        MM_TRY(ix->cand2.ensure((size_t)nb * 8));
        MM_HIP(hipMemcpyAsync(ix->cand2.p, big_b.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
        MM_HIP(hipMemcpyAsync(ix->cand2.as<int32_t>() + nb, qm.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
        MM_HIP(hipStreamSynchronize(st));
        r.ncand = (int)round_up(maxc, 32);
        r.bmap = ix->cand2.as<int32_t>();
        r.qmap = ix->cand2.as<int32_t>() + nb;
        MM_TRY(launch_rerank(ix, st, r, nb));
    }
    MM_TRY(exhaustive_queries(ix, st, rest, k, d_lab, d_dist, d_cnt));
    MM_HIP(hipStreamSynchronize(st));
    return MMISS_OK;
}

}  // namespace

// ================================================================================================ query
namespace {

// first pass of a query: everything up to and including the rerank is QUEUED on the stream; ix->pend describes it
int query_begin_locked(mmiss_index* ix, const float* queries, int32_t Q, int32_t k, int64_t* out_labels, float* out_dist,
                       int32_t* out_count) {
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    const int D = ix->dim;
    const int64_t N = ix->count;
    const bool out_dev = mmiss_is_device_ptr(out_labels);
    if (out_dev != mmiss_is_device_ptr(out_dist) || (out_count && out_dev != mmiss_is_device_ptr(out_count)))
        MM_FAIL(MMISS_ERR_ARG, "mmiss_index_query: output pointers must be all host or all device");

    // outputs: the caller's device buffers, or the pinned host block (copied to the caller's host buffers at the end)
    int64_t* d_lab = out_labels; float* d_dist = out_dist; int32_t* d_cnt = out_count;
    const size_t pin_lab = 0, pin_dist = (size_t)Q * k * 8, pin_cnt = pin_dist + (size_t)Q * k * 4, pin_flags = pin_cnt + (size_t)Q * 4;
    MM_TRY(ix->ensure_pin(pin_flags + (size_t)Q * 4));
    if (!out_dev) {
        d_lab = reinterpret_cast<int64_t*>(ix->pin + pin_lab); d_dist = reinterpret_cast<float*>(ix->pin + pin_dist);
        d_cnt = reinterpret_cast<int32_t*>(ix->pin + pin_cnt);
    } else if (!out_count) {
        MM_TRY(ix->out_c.ensure((size_t)Q * 4));
        d_cnt = ix->out_c.as<int32_t>();
    }
    int32_t* const h_flags = reinterpret_cast<int32_t*>(ix->pin + pin_flags);

    // effective k: never more than the rows there are
    int kp, pages;
    const int64_t want = std::min<int64_t>(k, N);
    if (want <= 10) { kp = 16; pages = 1; }
    else if (want <= 24) { kp = 32; pages = 1; }
    else {
        kp = 32;
        pages = (int)((std::min<int64_t>(want + 8, N) + 31) / 32);
    }
    const int ncand = kp * pages;
    if (ncand > 2048) MM_FAIL(MMISS_ERR_UNSUPPORTED, "mmiss_index_query: k = %d too large (max 2040)", k);

    // queries -> device, canonical normalisation
    const int Qpad = (int)round_up(Q, 256);
    const float* qsrc = queries;
    if (!mmiss_is_device_ptr(queries)) {
        MM_TRY(ix->qstage.ensure((size_t)Q * D * 4));
        MM_HIP(hipMemcpyAsync(ix->qstage.p, queries, (size_t)Q * D * 4, hipMemcpyHostToDevice, st));
        qsrc = ix->qstage.as<float>();
    }
    MM_TRY(ix->qn.ensure((size_t)Qpad * D * 4));
    MM_TRY(ix->qs.ensure((size_t)Qpad * D * ix->qelt()));
    MM_TRY(ix->eps_q.ensure((size_t)Qpad * 4));
    MM_TRY(ix->thr.ensure((size_t)Qpad * 4));
    {
        MM_PROF("prep_queries", st, 4.0 * Q * D, (double)Q * D * (8 + ix->elt));
        const int grid = (Qpad + 3) / 4;
        double eps_fixed, cnorm;
        guard_terms(ix, &eps_fixed, &cnorm);
        if (ix->dtype != MMISS_F32)   // (f16 query operand for f16 and fp8 rows)
            hipLaunchKernelGGL(prep_queries_kernel<_Float16>, dim3(grid), dim3(256), 0, st, qsrc, ix->qn.as<float>(),
                               ix->qs.as<_Float16>(), Q, Qpad, D, ix->eps_q.as<float>(), eps_fixed, cnorm);
        else
            hipLaunchKernelGGL(prep_queries_kernel<float>, dim3(grid), dim3(256), 0, st, qsrc, ix->qn.as<float>(),
                               ix->qs.as<float>(), Q, Qpad, D, ix->eps_q.as<float>(), eps_fixed, cnorm);
        MM_HIP(hipGetLastError());
    }

    // batched path: f16 rows, more than one MFMA tile of queries, single page
    // ... and fp8 rows from two 128-query tiles on (round 4: the strip kernel widens the codes to f16 in its operand load —
    // gemm256s_kernel<_Float16, .., W8>; below that the streaming scan, which reads half the bytes per row, is the faster)
    const bool dense8 = (N > 0) && ix->dtype == MMISS_F8 && pages == 1 && Q >= mmiss_option("score_big_min_q", 129) &&
                        (D % 128) == 0 && D >= 256 && mmiss_option("score_strip_v3", 1) != 0 && mmiss_option("score_f8_gemm", 1) != 0;
    const bool dense = ((N > 0) && ix->dtype == MMISS_F16 && Q > 16 && pages == 1) || dense8;
    MM_TRY(ix->cand.ensure((size_t)Q * ncand * 4));
    // (score, row) of the last candidate stage 1 kept per query, -inf when it kept every row: the paging cursor, and the
    // guard's bound on what was left out
    MM_TRY(ix->cur_s.ensure((size_t)Q * 4));
    MM_TRY(ix->cur_r.ensure((size_t)Q * 4));
    const bool guard = N > 0 && mmiss_option("exact_guard", 1) != 0;
    const int32_t* ovf_cnt = nullptr;  // threshold-filtered selection: per-query append counts and their capacity
    int ovf_cap = 0;
    if (N == 0) {
        MM_HIP(hipMemsetAsync(ix->cand.p, 0xff, (size_t)Q * ncand * 4, st));  // all -1
    } else if (dense) {
        // the pad rows up to the tile multiple are readable (capacity is a multiple of 1024) and masked in the epilogue
        // 256x256 phase-pipelined tile once there are two 128-query tiles to share a row panel
        const bool big = Q >= mmiss_option("score_big_min_q", 129);
        const int64_t Npad = round_up(N, big ? 256 : 128);
        const int Mq = (int)round_up(Q, big ? 256 : 128);
        // Threshold-filtered selection (Q > 128, index of at least 128 tiles): the first 1/32 of the rows (the SAMPLE) goes
        // through the dense path — group maxima to HBM, select, merge — which yields each query's k' best sample groups and
        // their k'-th score tau_q. tau_q is a lower bound of the final k'-th best group maximum, so over the other 31/32 of
        // the rows the strip kernel only APPENDS the groups reaching tau_q (~31 k' per query) to a per-query candidate
        // list; one merge over {sample list} + {appended} gives exactly the k' groups the dense path selects. The Q x N/16
        // group-maximum matrix (2.56 GB at Q = 1024, N = 10M) is neither written nor read back. A query whose list overflows
        // (a tau_q far below the final one: sorted or clustered data) is handed to the widen pass by the guard.
        // It costs four extra small launches (~35 us), so it is taken when the matrix it avoids is worth more: from
        // Q x N = 10^9 scores (256 MB of group maxima written and read back). Option score_filter: 0 never, 2 whenever possible.
        const int64_t nbn = Npad / 256;
        const int64_t ns_tiles = std::max<int64_t>(16, nbn / std::max(1, mmiss_option("score_sample_div", 32)));
        const int filter_opt = mmiss_option("score_filter", 1);
        const bool filtered = big && nbn >= 128 && filter_opt != 0 && (filter_opt == 2 || (double)Mq * (double)Npad >= 1e9);
        const int64_t Ndense = filtered ? ns_tiles * 256 : Npad;   // rows whose group maxima are materialised
        const int ng = (int)(Ndense / 16);
        MM_TRY(ix->gmax.ensure((size_t)Mq * ng * 4));
        GemmEpi ep{};
        ep.out = ix->gmax.p; ep.ldo = ng; ep.m_valid = Q; ep.p0 = (int)N; ep.m_fast = 1;
        ep.aux = ix->inv.as<float>();   // (fp8 rows: inverse norms; null otherwise)
        int strip = 1;
        // the strip kernel on the staggered loop of the persistent encoder GEMM (gemm_bf16_p256.h; round 3): D % 128 == 0
        const bool strip_v3 = mmiss_option("score_strip_v3", 1) != 0 && (D % 128) == 0 && D >= 256;
        if (big) strip = strip_length(Mq, nbn, filtered ? nbn - ns_tiles : nbn);
        {
            MM_PROF(dense8 ? (filtered ? "score_gemm_f8_sample" : "score_gemm_f8") : (filtered ? "score_gemm_f16_sample" : "score_gemm_f16"), st,
                    2.0 * Q * (double)std::min<int64_t>(N, Ndense) * D, (double)std::min<int64_t>(N, Ndense) * D * ix->elt);
            if (dense8)
                MM_TRY((launch_gemm256s<_Float16, true>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Ndense, D, strip)));
            else if (big && strip_v3)
                MM_TRY((launch_gemm256s<_Float16>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Ndense, D, strip)));
            else if (big)
                MM_TRY((launch_gemm256_strip<_Float16>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Ndense, D, strip)));
            else
                MM_TRY((launch_gemm_inst<_Float16, 128, MMISS_EPI_GROUPMAX_F32>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Npad, D)));
        }
        const int units = (ng + 1023) / 1024;
        int splits = (1024 + Q - 1) / Q;
        if (splits > units) splits = units;
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
        MM_TRY(ix->lists_s.ensure((size_t)splits * Q * kp * 4));
        MM_TRY(ix->lists_r.ensure((size_t)splits * Q * kp * 4));
        SelectArgs sa{};
        sa.G = ix->gmax.as<float>(); sa.ldg = ng; sa.ng = ng; sa.kp = kp; sa.Q = Q;
        sa.out_s = ix->lists_s.as<float>(); sa.out_r = ix->lists_r.as<int32_t>();
        {
            MM_PROF("select_topk", st, 0.0, (double)Q * ng * 4);
            hipLaunchKernelGGL(select_topk_kernel, dim3(Q, splits), dim3(256), 0, st, sa);
            MM_HIP(hipGetLastError());
        }
        MergeArgs m{};
        m.in_s = ix->lists_s.as<float>(); m.in_r = ix->lists_r.as<int32_t>();
        m.L = splits; m.Q = Q; m.kp = kp;
        if (filtered) {
            const int cap = mmiss_option("score_filter_cap", 2048);
            MM_TRY(ix->seed_s.ensure((size_t)Mq * kp * 4));
            MM_TRY(ix->seed_r.ensure((size_t)Mq * kp * 4));
            MM_TRY(ix->fcnt.ensure((size_t)Mq * 4));
            MM_TRY(ix->fbuf_s.ensure((size_t)Mq * cap * 4));
            MM_TRY(ix->fbuf_g.ensure((size_t)Mq * cap * 4));
            {   // the sample's k' best groups per query WITH their scores: seed of the final merge, source of tau_q
                MergeArgs m1 = m;
                m1.lists_per_block = splits; m1.out_s = ix->seed_s.as<float>(); m1.out_r = ix->seed_r.as<int32_t>();
                MM_PROF("merge_lists", st, 0.0, (double)splits * Q * kp * 8);
                hipLaunchKernelGGL(merge_lists_kernel, dim3(Q, 1), dim3(256), 0, st, m1);
                MM_HIP(hipGetLastError());
            }
            MM_HIP(hipMemsetAsync(ix->fcnt.p, 0, (size_t)Mq * 4, st));
            StripFilter flt{};
            flt.tau = ix->seed_s.as<float>() + (kp - 1); flt.tau_stride = kp;
            flt.cnt = ix->fcnt.as<int32_t>(); flt.buf_s = ix->fbuf_s.as<float>(); flt.buf_g = ix->fbuf_g.as<int32_t>();
            flt.cap = cap; flt.bn_begin = (int)ns_tiles;
            {
                MM_PROF(dense8 ? "score_gemm_f8" : "score_gemm_f16", st, 2.0 * Q * (double)(N - Ndense) * D, (double)(N - Ndense) * D * ix->elt);
                if (dense8) MM_TRY((launch_gemm256s<_Float16, true>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Npad, D, strip, &flt)));
                else if (strip_v3) MM_TRY((launch_gemm256s<_Float16>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Npad, D, strip, &flt)));
                else MM_TRY((launch_gemm256_strip<_Float16>(st, ix->qs.p, ix->rows.p, ep, Mq, (int)Npad, D, strip, &flt)));
            }
            m.in_s = ix->seed_s.as<float>(); m.in_r = ix->seed_r.as<int32_t>(); m.L = 1;
            m.flat_s = ix->fbuf_s.as<float>(); m.flat_r = ix->fbuf_g.as<int32_t>(); m.flat_cnt = ix->fcnt.as<int32_t>();
            m.flat_cap = cap;
            ovf_cnt = ix->fcnt.as<int32_t>();
            ovf_cap = cap;
        }
        m.cand = ix->cand.as<int32_t>(); m.cand_stride = ncand; m.page_off = 0;
        m.cur_s = ix->cur_s.as<float>(); m.cur_r = ix->cur_r.as<int32_t>();  // k'-th group maximum
        MM_TRY(launch_merge(ix, st, m));
    } else {
        const ScanPlan p = plan_scan(D, ix->qelt(), Q, kp, N);
        MM_TRY(ix->lists_s.ensure((size_t)p.slabs * Q * kp * 4));
        MM_TRY(ix->lists_r.ensure((size_t)p.slabs * Q * kp * 4));
        const bool paging = pages > 1;
        if (paging) {
            std::vector<float> inf((size_t)Q, INFINITY);
            std::vector<int32_t> neg((size_t)Q, -1);
            MM_HIP(hipMemcpyAsync(ix->cur_s.p, inf.data(), (size_t)Q * 4, hipMemcpyHostToDevice, st));
            MM_HIP(hipMemcpyAsync(ix->cur_r.p, neg.data(), (size_t)Q * 4, hipMemcpyHostToDevice, st));
            MM_HIP(hipStreamSynchronize(st));  // the host vectors go out of scope
        }
        for (int page = 0; page < pages; ++page) {
            ScanArgs a{};
            a.rows = ix->rows.p; a.N = N; a.D = D; a.inv = ix->inv.as<float>(); a.qs = ix->qs.p; a.Q = Q;
            a.cur_s = paging ? ix->cur_s.as<float>() : nullptr;
            a.cur_r = paging ? ix->cur_r.as<int32_t>() : nullptr;
            a.kp = kp; a.tiles_per_block = p.tiles_per_block;
            a.out_s = ix->lists_s.as<float>(); a.out_r = ix->lists_r.as<int32_t>();
            MM_TRY(launch_scan(ix, st, a, p));
            MergeArgs m{};
            m.in_s = ix->lists_s.as<float>(); m.in_r = ix->lists_r.as<int32_t>();
            m.L = p.slabs; m.Q = Q; m.kp = kp;
            m.cand = ix->cand.as<int32_t>(); m.cand_stride = ncand; m.page_off = page * kp;
            m.cur_s = ix->cur_s.as<float>(); m.cur_r = ix->cur_r.as<int32_t>();  // the scan reads them only when paging
            MM_TRY(launch_merge(ix, st, m));
        }
    }
    {
        RerankArgs r{};
        r.rows = ix->rows.p; r.D = D; r.inv = ix->inv.as<float>(); r.qn = ix->qn.as<float>(); r.cand = ix->cand.as<int32_t>();
        r.cand_stride = ncand; r.ncand = dense ? ncand * 16 : ncand; r.group_mode = dense ? 1 : 0; r.nrows = N;
        r.labels = ix->labels_d.as<int64_t>(); r.k = k;
        r.out_labels = d_lab; r.out_dist = d_dist; r.out_count = d_cnt;
        if (guard) {
            r.tau = ix->cur_s.as<float>(); r.eps_q = ix->eps_q.as<float>();
            r.thr_out = ix->thr.as<float>();   // what the widen pass starts from
            r.flags = h_flags; r.nflag = nullptr;  // one plain store per query into the pinned block; counted on the host
            r.force_flag = mmiss_option("guard_force", 0);
            r.ovf_cnt = ovf_cnt; r.ovf_cap = ovf_cap;
        }
        MM_TRY(launch_rerank(ix, st, r, Q));
    }
    ix->stat_queries += Q;
    MM_HIP(hipEventRecord(ix->done_ev, st));
    mmiss_index::Pending& pd = ix->pend;
    pd.active = true; pd.out_dev = out_dev; pd.guard = guard; pd.Q = Q; pd.k = k;
    pd.d_lab = d_lab; pd.d_dist = d_dist; pd.d_cnt = d_cnt; pd.h_flags = h_flags;
    pd.out_labels = out_labels; pd.out_dist = out_dist; pd.out_count = out_count;
    return MMISS_OK;
}

// second half: wait for the first pass (its event, NOT the stream: the caller may have queued the next batch's encode behind
// it), widen what the guard could not prove, hand host outputs over
int query_end_locked(mmiss_index* ix) {
    mmiss_index::Pending pd = ix->pend;
    ix->pend.active = false;
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    if (!pd.guard && pd.out_dev && ix->has_user_stream) {  // device outputs on the caller's stream, nothing to decide: stays queued
        ix->async_pending = true;
        return MMISS_OK;
    }
    MM_HIP(hipEventSynchronize(ix->done_ev));
    if (pd.guard) {
        std::vector<int32_t> which;
        for (int q = 0; q < pd.Q; ++q)
            if (pd.h_flags[q]) which.push_back(q);
        if (!which.empty()) {
            ix->stat_flagged += (int64_t)which.size();
            MM_TRY(sweep_queries(ix, st, which, pd.Q, pd.k, pd.d_lab, pd.d_dist, pd.d_cnt));  // returns with the stream drained
        }
    }
    if (!pd.out_dev) {
        memcpy(pd.out_labels, pd.d_lab, (size_t)pd.Q * pd.k * 8);
        memcpy(pd.out_dist, pd.d_dist, (size_t)pd.Q * pd.k * 4);
        if (pd.out_count) memcpy(pd.out_count, pd.d_cnt, (size_t)pd.Q * 4);
    }
    return MMISS_OK;
}

int query_check_args(const mmiss_index* ix, const float* queries, int32_t Q, int32_t k, const int64_t* out_labels,
                     const float* out_dist, const char* who) {
    if (!ix || !queries || !out_labels || !out_dist) MM_FAIL(MMISS_ERR_ARG, "%s: null argument", who);
    if (Q < 0 || k <= 0) MM_FAIL(MMISS_ERR_ARG, "%s: Q=%d k=%d", who, Q, k);
    return MMISS_OK;
}

}  // namespace

extern "C" int mmiss_index_query(mmiss_index* ix, const float* queries, int32_t Q, int32_t k, int64_t* out_labels,
                                 float* out_dist, int32_t* out_count) {
    MM_TRY(query_check_args(ix, queries, Q, k, out_labels, out_dist, "mmiss_index_query"));
    if (Q == 0) return MMISS_OK;
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_query");
    MM_TRY(query_begin_locked(ix, queries, Q, k, out_labels, out_dist, out_count));
    return query_end_locked(ix);
}

extern "C" int mmiss_index_query_begin(mmiss_index* ix, const float* queries, int32_t Q, int32_t k, int64_t* out_labels,
                                       float* out_dist, int32_t* out_count) {
    MM_TRY(query_check_args(ix, queries, Q, k, out_labels, out_dist, "mmiss_index_query_begin"));
    if (Q == 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_query_begin: Q = 0");
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_query_begin");
    return query_begin_locked(ix, queries, Q, k, out_labels, out_dist, out_count);
}

extern "C" int mmiss_index_query_end(mmiss_index* ix) {
    if (!ix) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_query_end: null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!ix->pend.active) MM_FAIL(MMISS_ERR_STATE, "mmiss_index_query_end: no query was begun");
    return query_end_locked(ix);
}

// Drop a query opened with mmiss_index_query_begin without taking its results: waits for the first pass that is queued (its
// kernels write the caller's output buffers and the handle's scratch), then frees the handle for other calls. A no-op when
// no query is open.
extern "C" int mmiss_index_query_abort(mmiss_index* ix) {
    if (!ix) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_query_abort: null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!ix->pend.active) return MMISS_OK;
    ix->pend.active = false;
    MM_TRY(mmiss_use_device(ix->device));
    MM_HIP(hipEventSynchronize(ix->done_ev));
    return MMISS_OK;
}

extern "C" int mmiss_index_guard_stats_ex(mmiss_index* ix, int64_t out[8]) {
    if (!ix || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_guard_stats_ex: null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = ix->stat_queries; out[1] = ix->stat_flagged; out[2] = ix->stat_rounds; out[3] = ix->stat_pages;
    out[4] = ix->stat_exhaustive; out[5] = ix->stat_swept_rows;
    return MMISS_OK;
}

extern "C" int mmiss_index_guard_stats(mmiss_index* ix, int64_t out[4]) {
    if (!ix || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_guard_stats: null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    out[0] = ix->stat_queries; out[1] = ix->stat_flagged; out[2] = ix->stat_rounds; out[3] = ix->stat_pages;
    return MMISS_OK;
}

// ================================================================================================ persistence
namespace {
struct IdxHeader {
    char magic[8];
    int32_t version, dim, dtype, reserved;
    int64_t count;
};
}  // namespace

extern "C" int mmiss_index_save(mmiss_index* ix, const char* path) {
    if (!ix || !path) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_save: null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_save");
    MM_TRY(mmiss_use_device(ix->device));
    MM_HIP(hipStreamSynchronize(ix->stream()));
    FILE* f = fopen(path, "wb");
    if (!f) MM_FAIL(MMISS_ERR_IO, "cannot open %s for writing", path);
    IdxHeader h{};
    memcpy(h.magic, "MMISSIDX", 8);
    h.version = 1; h.dim = ix->dim; h.dtype = ix->dtype; h.count = ix->count;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    if (ok && ix->count) ok = fwrite(ix->labels_h.data(), 8, (size_t)ix->count, f) == (size_t)ix->count;
    const size_t row_bytes = (size_t)ix->dim * ix->elt;
    const int64_t chunk = std::max<int64_t>(1, (64 << 20) / (int64_t)row_bytes);
    std::vector<char> buf((size_t)std::min<int64_t>(chunk, std::max<int64_t>(ix->count, 1)) * row_bytes);
    for (int64_t r0 = 0; ok && r0 < ix->count; r0 += chunk) {
        const int64_t nr = std::min(chunk, ix->count - r0);
        if (hipMemcpy(buf.data(), ix->rows.as<char>() + (size_t)r0 * row_bytes, (size_t)nr * row_bytes,
                      hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = fwrite(buf.data(), row_bytes, (size_t)nr, f) == (size_t)nr;
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) MM_FAIL(MMISS_ERR_IO, "write to %s failed", path);
    return MMISS_OK;
}

extern "C" int mmiss_index_load(mmiss_index* ix, const char* path) {
    if (!ix || !path) MM_FAIL(MMISS_ERR_ARG, "mmiss_index_load: null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    MM_NO_PENDING(ix, "mmiss_index_load");
    MM_TRY(mmiss_use_device(ix->device));
    hipStream_t st = ix->stream();
    FILE* f = fopen(path, "rb");
    if (!f) MM_FAIL(MMISS_ERR_IO, "cannot open %s", path);
    IdxHeader h{};
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "MMISSIDX", 8) != 0 || h.version != 1) {
        fclose(f);
        MM_FAIL(MMISS_ERR_IO, "%s is not an mmiss index file", path);
    }
    if (h.dim != ix->dim || h.dtype != ix->dtype || h.count < 0) {
        fclose(f);
        MM_FAIL(MMISS_ERR_ARG, "%s holds dim %d dtype %d; this index is dim %d dtype %d", path, h.dim, h.dtype, ix->dim,
                ix->dtype);
    }
    const size_t row_bytes = (size_t)ix->dim * ix->elt;
    {   // a truncated file is refused BEFORE anything of the index is overwritten: the header's count against the file's length
        const long at = ftell(f);
        long end = -1;
        if (at >= 0 && fseek(f, 0, SEEK_END) == 0) end = ftell(f);
        if (at < 0 || end < 0 || fseek(f, at, SEEK_SET) != 0) { fclose(f); MM_FAIL(MMISS_ERR_IO, "cannot seek in %s", path); }
        if ((uint64_t)h.count > (uint64_t)(end - at) / (8 + row_bytes)) {
            fclose(f);
            MM_FAIL(MMISS_ERR_IO, "%s is truncated or corrupt (%ld bytes behind the header, %lld rows need %llu); the index is unchanged",
                    path, end - at, (long long)h.count, (unsigned long long)((uint64_t)h.count * (8 + row_bytes)));
        }
    }
    std::vector<int64_t> lab((size_t)h.count);
    bool ok = h.count == 0 || fread(lab.data(), 8, (size_t)h.count, f) == (size_t)h.count;
    for (int64_t i = 1; ok && i < h.count; ++i) ok = lab[i] > lab[i - 1];
    if (!ok) { fclose(f); MM_FAIL(MMISS_ERR_IO, "%s: labels unreadable or not strictly increasing; the index is unchanged", path); }
    int rc = index_reserve(ix, h.count, st);
    if (rc != MMISS_OK) { fclose(f); return rc; }
    const int64_t chunk = std::max<int64_t>(1, (64 << 20) / (int64_t)row_bytes);
    std::vector<char> buf((size_t)std::min<int64_t>(chunk, std::max<int64_t>(h.count, 1)) * row_bytes);
    for (int64_t r0 = 0; ok && rc == MMISS_OK && r0 < h.count; r0 += chunk) {
        const int64_t nr = std::min(chunk, h.count - r0);
        ok = fread(buf.data(), row_bytes, (size_t)nr, f) == (size_t)nr;
        if (ok && hipMemcpy(ix->rows.as<char>() + (size_t)r0 * row_bytes, buf.data(), (size_t)nr * row_bytes,
                            hipMemcpyHostToDevice) != hipSuccess) ok = false;
    }
    fclose(f);
    if (!ok) {
        // a read or copy error with part of the rows already overwritten: old labels / inverse norms no longer describe what the
        // buffer holds, so the index is left EMPTY rather than inconsistent
        if (ix->dtype == MMISS_F8 && ix->capacity > 0) { (void)hipMemsetAsync(ix->inv.p, 0, (size_t)ix->capacity * 4, st); (void)hipStreamSynchronize(st); }
        ix->count = 0;
        ix->labels_h.clear();
        MM_FAIL(MMISS_ERR_IO, "%s could not be read to its end; the index was cleared", path);
    }
    if (h.count) MM_HIP(hipMemcpy(ix->labels_d.p, lab.data(), (size_t)h.count * 8, hipMemcpyHostToDevice));
    // fp8 rows: the file holds the codes only; their inverse norms are a function of the codes (files of round 4 load as they are)
    if (ix->dtype == MMISS_F8 && ix->capacity > 0) {
        MM_HIP(hipMemsetAsync(ix->inv.p, 0, (size_t)ix->capacity * 4, st));
        MM_TRY(launch_f8_inv(ix, 0, h.count, st));
        MM_HIP(hipStreamSynchronize(st));
    }
    ix->labels_h.swap(lab);
    ix->count = h.count;
    return MMISS_OK;
}

// ================================================================================================ glue kernels
extern "C" int mmiss_blend(int device, void* hip_stream, const float* img, const float* txt, double w, int32_t Q,
                           int32_t dim, float* out) {
    if (!img || !txt || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_blend: null argument");
    if (Q < 0 || dim <= 0 || dim % 64) MM_FAIL(MMISS_ERR_ARG, "mmiss_blend: Q=%d dim=%d (dim must be a multiple of 64)", Q, dim);
    if (Q == 0) return MMISS_OK;
    MM_TRY(mmiss_use_device(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const size_t bytes = (size_t)Q * dim * 4;
    DevBuf bi, bt, bo;
    const float *di = img, *dt = txt;
    float* dout = out;
    if (!mmiss_is_device_ptr(img)) { MM_TRY(bi.alloc(bytes)); MM_HIP(hipMemcpyAsync(bi.p, img, bytes, hipMemcpyHostToDevice, st)); di = bi.as<float>(); }
    if (!mmiss_is_device_ptr(txt)) { MM_TRY(bt.alloc(bytes)); MM_HIP(hipMemcpyAsync(bt.p, txt, bytes, hipMemcpyHostToDevice, st)); dt = bt.as<float>(); }
    const bool out_dev = mmiss_is_device_ptr(out);
    if (!out_dev) { MM_TRY(bo.alloc(bytes)); dout = bo.as<float>(); }
    {
        MM_PROF("blend", st, 12.0 * Q * dim, 3.0 * bytes);
        hipLaunchKernelGGL(blend_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, di, dt, (float)w, (float)(1.0 - w), Q, dim, dout);
        MM_HIP(hipGetLastError());
    }
    if (!out_dev) MM_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
    if (!out_dev || bi.p || bt.p) MM_HIP(hipStreamSynchronize(st));
    return MMISS_OK;
}

// one merge level: S lists of k per query -> one list of k (S * k <= 8192 entries sorted in LDS)
static int launch_shard_merge(hipStream_t st, const float* dd, const int64_t* dl, int S, int Q, int k, float* dod,
                              int64_t* dol, int32_t* doc) {
    int npow = 1;
    while (npow < S * k) npow <<= 1;
    const int lds = npow * 12 + 16;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&shard_merge_kernel), lds));
    MM_PROF("shard_merge", st, 0.0, (double)S * Q * k * 12);
    hipLaunchKernelGGL(shard_merge_kernel, dim3(Q), dim3(256), lds, st, dd, dl, S, Q, k, dod, dol, doc);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

extern "C" int mmiss_merge_topk(int device, void* hip_stream, const float* dist, const int64_t* labels, int32_t S,
                                int32_t Q, int32_t k, float* out_dist, int64_t* out_labels, int32_t* out_count) {
    if (!dist || !labels || !out_dist || !out_labels) MM_FAIL(MMISS_ERR_ARG, "mmiss_merge_topk: null argument");
    if (S <= 0 || Q < 0 || k <= 0 || k > 4096)
        MM_FAIL(MMISS_ERR_ARG, "mmiss_merge_topk: S=%d Q=%d k=%d (need S > 0, 0 < k <= 4096)", S, Q, k);
    if (Q == 0) return MMISS_OK;
    MM_TRY(mmiss_use_device(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    const size_t n_in = (size_t)S * Q * k, n_out = (size_t)Q * k;
    DevBuf bd, bl, bod, bol, boc, td[2], tl[2];
    const float* dd = dist; const int64_t* dl = labels;
    float* dod = out_dist; int64_t* dol = out_labels; int32_t* doc = out_count;
    if (!mmiss_is_device_ptr(dist)) { MM_TRY(bd.alloc(n_in * 4)); MM_HIP(hipMemcpyAsync(bd.p, dist, n_in * 4, hipMemcpyHostToDevice, st)); dd = bd.as<float>(); }
    if (!mmiss_is_device_ptr(labels)) { MM_TRY(bl.alloc(n_in * 8)); MM_HIP(hipMemcpyAsync(bl.p, labels, n_in * 8, hipMemcpyHostToDevice, st)); dl = bl.as<int64_t>(); }
    const bool out_dev = mmiss_is_device_ptr(out_dist);
    if (!out_dev) {
        MM_TRY(bod.alloc(n_out * 4)); MM_TRY(bol.alloc(n_out * 8)); MM_TRY(boc.alloc((size_t)Q * 4));
        dod = bod.as<float>(); dol = bol.as<int64_t>(); doc = boc.as<int32_t>();
    }
    // The reference's "All" search asks for 1000 hits (main.py:757): 8 shards x 1000 = 8000 entries sort in one 96 KB LDS
    // pass. Beyond 8192 entries the lists are merged in groups of G = 8192 / k lists per level (a top-k of top-k's is
    // the top-k), each level S -> ceil(S / G) lists, until one is left.
    const int G = 8192 / k;  // >= 2 because k <= 4096
    int cur_S = S, flip = 0;
    while (cur_S > G) {
        const int nxt = (cur_S + G - 1) / G;
        MM_TRY(td[flip].ensure((size_t)nxt * n_out * 4));
        MM_TRY(tl[flip].ensure((size_t)nxt * n_out * 8));
        for (int g = 0; g < nxt; ++g) {
            const int s0 = g * G, sn = std::min(G, cur_S - s0);
            MM_TRY(launch_shard_merge(st, dd + (size_t)s0 * n_out, dl + (size_t)s0 * n_out, sn, Q, k,
                                      td[flip].as<float>() + (size_t)g * n_out, tl[flip].as<int64_t>() + (size_t)g * n_out, nullptr));
        }
        dd = td[flip].as<float>(); dl = tl[flip].as<int64_t>();
        cur_S = nxt;
        flip ^= 1;
    }
    MM_TRY(launch_shard_merge(st, dd, dl, cur_S, Q, k, dod, dol, doc));
    if (!out_dev) {
        MM_HIP(hipMemcpyAsync(out_dist, dod, n_out * 4, hipMemcpyDeviceToHost, st));
        MM_HIP(hipMemcpyAsync(out_labels, dol, n_out * 8, hipMemcpyDeviceToHost, st));
        if (out_count) MM_HIP(hipMemcpyAsync(out_count, doc, (size_t)Q * 4, hipMemcpyDeviceToHost, st));
    }
    if (!out_dev || bd.p || bl.p || td[0].p || td[1].p) MM_HIP(hipStreamSynchronize(st));
    return MMISS_OK;
}
