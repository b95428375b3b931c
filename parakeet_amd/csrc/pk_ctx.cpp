// pk_ctx.cpp -- context, error reporting, per-kernel profiler, parameter store.
#include <cmath>

#include "pk_common.h"

static thread_local char g_err[1024] = "";

void pk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pk_last_error(void) { return g_err; }
// PK_SOURCE_HASH = sha256 over csrc/* and include/pk_synth.h at build time (parakeet_amd/build.py passes it with
// -D).  build() compares the hash inside an existing libpk_synth.so with the sources on disk and rebuilds on
// any mismatch -- file times do not decide (the .so is git-ignored but shipped, so a stale binary can be
// newer than an edited source).
#ifndef PK_SOURCE_HASH
#define PK_SOURCE_HASH "unknown"
#endif
#define PK_STR2(x) #x
#define PK_STR(x) PK_STR2(x)
extern "C" const char* pk_version(void) {
    return "parakeet_amd 0.4 (gfx950) PK_SOURCE_HASH=" PK_SOURCE_HASH "; PK_PROFILE_BUILD=" PK_STR(PK_PROFILE_BUILD) ";";
}

extern "C" int pk_ctx_create(int device_id, pk_ctx** out) {
    if (!out) PK_FAIL(PK_EINVAL, "pk_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        PK_FAIL(PK_EHIP, "pk_ctx_create: no HIP device visible (%s)",
                e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        PK_FAIL(PK_EINVAL, "pk_ctx_create: device %d out of range [0,%d)", device_id, n);
    PK_DEVICE(device_id);
    pk_ctx* c = new pk_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        PK_FAIL(PK_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
    *out = c;
    return PK_OK;
}

extern "C" int pk_ctx_set_stream(pk_ctx* ctx, void* hip_stream) {
    if (!ctx) PK_FAIL(PK_EINVAL, "pk_ctx_set_stream: ctx is NULL");
    if (ctx->own_stream && ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return PK_OK;
}

extern "C" int pk_sync(pk_ctx* ctx) {
    if (!ctx) PK_FAIL(PK_EINVAL, "pk_sync: ctx is NULL");
    PK_DEVICE(ctx->device);
    PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}

extern "C" void pk_ctx_destroy(pk_ctx* ctx) {
    if (!ctx) return;
    pk_device_guard _dg(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->prof_recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    for (auto& kv : ctx->scratch) {
        kv.second->row_amax.release();
        kv.second->row_amax2.release();
        kv.second->attn_amax.release();
        delete kv.second;
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

pk_ctx_scratch* pk_ctx_get_scratch(pk_ctx* ctx) {
    pk_ctx_scratch*& sc = ctx->scratch[ctx->stream];   // the stream the caller's launches go to
    if (!sc) sc = new pk_ctx_scratch();
    return sc;
}

// ------------------------------------------------------------------ profiler
static hipEvent_t take_event(pk_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t ev = c->event_pool.back();
        c->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

int pk_ctx::prof_begin(const char* name) {
    auto it = prof_ids.find(name);
    int id;
    if (it == prof_ids.end()) {
        id = (int)prof_names.size();
        prof_names.push_back(name);
        prof_ids[name] = id;
    } else {
        id = it->second;
    }
    pk_prof_rec r;
    r.name_id = id;
    r.start = take_event(this);
    r.stop = take_event(this);
    (void)hipEventRecord(r.start, stream);
    prof_recs.push_back(r);
    return (int)prof_recs.size() - 1;
}

void pk_ctx::prof_end(int rec) { (void)hipEventRecord(prof_recs[rec].stop, stream); }

extern "C" int pk_prof_enable(pk_ctx* ctx, int on) {
    if (!ctx) PK_FAIL(PK_EINVAL, "pk_prof_enable: ctx is NULL");
    ctx->prof_on = on != 0;
    return PK_OK;
}

extern "C" int pk_prof_reset(pk_ctx* ctx) {
    if (!ctx) PK_FAIL(PK_EINVAL, "pk_prof_reset: ctx is NULL");
    PK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->prof_recs) {
        ctx->event_pool.push_back(r.start);
        ctx->event_pool.push_back(r.stop);
    }
    ctx->prof_recs.clear();
    return PK_OK;
}

static int prof_collect(pk_ctx* ctx, std::vector<int64_t>& cnt, std::vector<double>& ms) {
    PK_HIP(hipStreamSynchronize(ctx->stream));
    cnt.assign(ctx->prof_names.size(), 0);
    ms.assign(ctx->prof_names.size(), 0.0);
    for (auto& r : ctx->prof_recs) {
        float t = 0.f;
        PK_HIP(hipEventElapsedTime(&t, r.start, r.stop));
        cnt[r.name_id] += 1;
        ms[r.name_id] += t;
    }
    return PK_OK;
}

extern "C" int pk_prof_read(pk_ctx* ctx, const char* name, int64_t* launches, double* total_ms) {
    if (!ctx || !name) PK_FAIL(PK_EINVAL, "pk_prof_read: NULL argument");
    std::vector<int64_t> cnt;
    std::vector<double> ms;
    PK_TRY(prof_collect(ctx, cnt, ms));
    auto it = ctx->prof_ids.find(name);
    int64_t n = 0;
    double t = 0;
    if (it != ctx->prof_ids.end()) {
        n = cnt[it->second];
        t = ms[it->second];
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = t;
    return PK_OK;
}

extern "C" int pk_prof_dump(pk_ctx* ctx, char* buf, int64_t buflen) {
    if (!ctx || !buf || buflen <= 0) PK_FAIL(PK_EINVAL, "pk_prof_dump: bad argument");
    std::vector<int64_t> cnt;
    std::vector<double> ms;
    PK_TRY(prof_collect(ctx, cnt, ms));
    std::string s;
    char line[256];
    for (size_t i = 0; i < ctx->prof_names.size(); ++i) {
        snprintf(line, sizeof(line), "%s %lld %.6f\n", ctx->prof_names[i].c_str(), (long long)cnt[i],
                 ms[i]);
        s += line;
    }
    if ((int64_t)s.size() + 1 > buflen) PK_FAIL(PK_EINVAL, "pk_prof_dump: buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return PK_OK;
}

// ------------------------------------------------------------ parameter store
int pk_store_param(pk_param_map& m, const char* name, const float* data, const int64_t* shape,
                   int32_t ndim) {
    if (!name || !data || (ndim > 0 && !shape)) PK_FAIL(PK_EINVAL, "set_param: NULL argument");
    if (ndim < 0 || ndim > 8) PK_FAIL(PK_EINVAL, "set_param(%s): ndim %d out of range", name, ndim);
    pk_param p;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0) PK_FAIL(PK_ESHAPE, "set_param(%s): non-positive dim", name);
        p.shape.push_back(shape[i]);
        n *= shape[i];
    }
    p.data.assign(data, data + n);
    m[name] = std::move(p);
    return PK_OK;
}

static bool shape_matches(const std::vector<int64_t>& have, const std::vector<int64_t>& want) {
    int64_t a = 1, b = 1;
    for (auto s : have) a *= s;
    for (auto s : want) b *= s;
    if (a != b) return false;
    // accept exact match, or the same dims with size-1 axes squeezed/unsqueezed
    std::vector<int64_t> x, y;
    for (auto s : have) if (s != 1) x.push_back(s);
    for (auto s : want) if (s != 1) y.push_back(s);
    return x == y;
}

static std::string shape_str(const std::vector<int64_t>& s) {
    std::string r = "[";
    for (size_t i = 0; i < s.size(); ++i) r += (i ? "," : "") + std::to_string(s[i]);
    return r + "]";
}

int pk_get_weight(const pk_param_map& m, const std::string& base, const std::vector<int64_t>& shape,
                  std::vector<float>& out) {
    auto w = m.find(base + ".weight");
    if (w != m.end()) {
        if (!shape_matches(w->second.shape, shape))
            PK_FAIL(PK_ESHAPE, "parameter %s.weight has shape %s, expected %s", base.c_str(),
                    shape_str(w->second.shape).c_str(), shape_str(shape).c_str());
        out = w->second.data;
        return PK_OK;
    }
    auto g = m.find(base + ".weight_g");
    auto v = m.find(base + ".weight_v");
    if (g == m.end() || v == m.end())
        PK_FAIL(PK_ESTATE, "parameter %s.weight (or weight_g/weight_v) was never set", base.c_str());
    if (!shape_matches(v->second.shape, shape))
        PK_FAIL(PK_ESHAPE, "parameter %s.weight_v has shape %s, expected %s", base.c_str(),
                shape_str(v->second.shape).c_str(), shape_str(shape).c_str());
    int64_t c0 = shape[0];
    if (g->second.numel() != c0)
        PK_FAIL(PK_ESHAPE, "parameter %s.weight_g has %lld elements, expected %lld", base.c_str(),
                (long long)g->second.numel(), (long long)c0);
    int64_t per = v->second.numel() / c0;
    out.resize(v->second.numel());
    for (int64_t o = 0; o < c0; ++o) {
        double s = 0;
        for (int64_t i = 0; i < per; ++i) {
            double x = v->second.data[o * per + i];
            s += x * x;
        }
        double scale = (double)g->second.data[o] / std::sqrt(s);
        for (int64_t i = 0; i < per; ++i)
            out[o * per + i] = (float)((double)v->second.data[o * per + i] * scale);
    }
    return PK_OK;
}

int pk_get_vector(const pk_param_map& m, const std::string& name, int64_t n, std::vector<float>& out) {
    auto it = m.find(name);
    if (it == m.end()) PK_FAIL(PK_ESTATE, "parameter %s was never set", name.c_str());
    if (it->second.numel() != n)
        PK_FAIL(PK_ESHAPE, "parameter %s has %lld elements, expected %lld", name.c_str(),
                (long long)it->second.numel(), (long long)n);
    out = it->second.data;
    return PK_OK;
}

int pk_upload(pk_ctx* ctx, pk_dbuf& buf, const void* host, size_t bytes) {
    PK_TRY(buf.reserve(bytes));
    PK_HIP(hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    PK_HIP(hipStreamSynchronize(ctx->stream));
    return PK_OK;
}
