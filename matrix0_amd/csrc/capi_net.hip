// extern "C" entry points of the network seam (include/m0_engine.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/m0_engine.h"
#include "net.h"
#include "capi_common.h"

thread_local std::string g_m0_last_error;

void m0_set_error(const std::string& s) { g_m0_last_error = s; }

struct m0_net {
    Net* net = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // staging buffers for the host-facing infer call
    float* planes_dev = nullptr;
    float* logits_dev = nullptr;
    float* value_dev = nullptr;
    float* ssl_dev = nullptr;
    int cap = 0;
};

Net* m0_net_impl(m0_net* n) { return n ? n->net : nullptr; }
hipStream_t m0_net_stream(m0_net* n) { return n ? n->stream : nullptr; }
int m0_net_device(m0_net* n) { return n ? n->device : 0; }
void m0_net_lock(m0_net* n) { if (n) n->mu.lock(); }
void m0_net_unlock(m0_net* n) { if (n) n->mu.unlock(); }

extern "C" {

const char* m0_last_error(void) { return g_m0_last_error.c_str(); }
const char* m0_version(void) { return "m0engine 0.2 (gfx950)"; }
int m0_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

m0_net* m0_net_create(const m0_net_cfg* cfg, int hip_device) {
    if (!cfg) { m0_set_error("cfg is null"); return nullptr; }
    const char* why = Net::check_supported(*cfg);
    if (why) { m0_set_error(std::string("unsupported network config: ") + why); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        m0_set_error("no HIP device available (the engine has no CPU fallback)");
        return nullptr;
    }
    if (hip_device < 0 || hip_device >= ndev) { m0_set_error("hip_device out of range"); return nullptr; }
    if (hipSetDevice(hip_device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return nullptr; }
    m0_net* h = new m0_net();
    h->device = hip_device;
    // M0_NET_CU_MASK=w0,w1,...,w7 (hex words, bit b = CU b in the driver's numbering): the network's stream runs on those CUs
    // only (hipExtStreamCreateWithCUMask).  Measurement switch of tools/exp_cumask.py (two networks on complementary halves
    // of the chip); read at creation, so two networks of one process can get different masks.
    hipError_t se;
    if (const char* mk = getenv("M0_NET_CU_MASK"); mk && *mk) {
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nw = 0;
        for (const char* q = mk; *q && nw < 8; ++nw) {
            char* end = nullptr;
            words[nw] = (uint32_t)strtoul(q, &end, 16);
            if (end == q) break;
            q = (*end == ',') ? end + 1 : end;
        }
        se = hipExtStreamCreateWithCUMask(&h->stream, 8, words);
    } else {
        se = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) {
        m0_set_error("hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    h->net = new Net(*cfg, hip_device, h->stream);
    return h;
}

void m0_net_destroy(m0_net* n) {
    if (!n) return;
    (void)hipSetDevice(n->device);
    if (n->stream) { (void)hipStreamSynchronize(n->stream); }
    if (n->planes_dev) (void)hipFree(n->planes_dev);
    if (n->logits_dev) (void)hipFree(n->logits_dev);
    if (n->value_dev) (void)hipFree(n->value_dev);
    if (n->ssl_dev) (void)hipFree(n->ssl_dev);
    delete n->net;
    if (n->stream) (void)hipStreamDestroy(n->stream);
    delete n;
}

int m0_net_load_weight(m0_net* n, const char* name, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!n) { m0_set_error("net is null"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    std::string err;
    int rc = n->net->load(name, data, dtype, shape, ndim, err);
    if (rc != M0_OK) m0_set_error(err);
    return rc;
}

int m0_net_finalize(m0_net* n) {
    if (!n) { m0_set_error("net is null"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    std::string err;
    int rc = n->net->finalize(err);
    if (rc != M0_OK) m0_set_error(err);
    return rc;
}

static int ensure_io(m0_net* n, int B) {
    if (B <= n->cap) return M0_OK;
    if (n->planes_dev) (void)hipFree(n->planes_dev);
    if (n->logits_dev) (void)hipFree(n->logits_dev);
    if (n->value_dev) (void)hipFree(n->value_dev);
    if (n->ssl_dev) (void)hipFree(n->ssl_dev);
    n->planes_dev = n->logits_dev = n->value_dev = n->ssl_dev = nullptr;
    n->cap = 0;
    int cap = B < 64 ? 64 : B;
    const int P = n->net->cfg().planes;
    int sslc = n->net->ssl_channels_total();
    if (hipMalloc((void**)&n->planes_dev, (size_t)cap * P * 64 * 4) != hipSuccess ||
        hipMalloc((void**)&n->logits_dev, (size_t)cap * M0_POLICY_SIZE * 4) != hipSuccess ||
        hipMalloc((void**)&n->value_dev, (size_t)cap * 4) != hipSuccess ||
        hipMalloc((void**)&n->ssl_dev, (size_t)cap * (sslc > 0 ? sslc : 1) * 64 * 4) != hipSuccess) {
        m0_set_error("hipMalloc failed for I/O staging");
        return M0_ERR_HIP;
    }
    n->cap = cap;
    return M0_OK;
}

int m0_net_infer(m0_net* n, const float* planes, int B, float* policy, float* value, float* ssl) {
    if (!n || !planes || !policy || !value) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    if (B <= 0) { m0_set_error("batch must be positive"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    if (hipSetDevice(n->device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return M0_ERR_HIP; }
    int rc = ensure_io(n, B);
    if (rc != M0_OK) return rc;
    const int P = n->net->cfg().planes;
    const int sslc = n->net->ssl_channels_total();
    if (ssl && sslc == 0) { m0_set_error("ssl output requested but the net has no SSL heads"); return M0_ERR_INVALID; }
    std::string err;
    hipError_t e = hipMemcpyAsync(n->planes_dev, planes, (size_t)B * P * 64 * 4, hipMemcpyHostToDevice, n->stream);
    if (e != hipSuccess) { m0_set_error(std::string("H2D copy: ") + hipGetErrorString(e)); return M0_ERR_HIP; }
    rc = n->net->forward(n->planes_dev, nullptr, B, n->logits_dev, n->value_dev, ssl ? n->ssl_dev : nullptr, n->stream, err);
    if (rc != M0_OK) { m0_set_error(err); return rc; }
    (void)hipMemcpyAsync(policy, n->logits_dev, (size_t)B * M0_POLICY_SIZE * 4, hipMemcpyDeviceToHost, n->stream);
    (void)hipMemcpyAsync(value, n->value_dev, (size_t)B * 4, hipMemcpyDeviceToHost, n->stream);
    if (ssl) (void)hipMemcpyAsync(ssl, n->ssl_dev, (size_t)B * sslc * 64 * 4, hipMemcpyDeviceToHost, n->stream);
    e = hipStreamSynchronize(n->stream);
    if (e != hipSuccess) { m0_set_error(std::string("forward failed: ") + hipGetErrorString(e)); return M0_ERR_HIP; }
    // NaN/Inf must surface as errors (mcts.py:1165-1177, tests/test_error_handling.py:23-53)
    for (size_t i = 0, nn = (size_t)B * M0_POLICY_SIZE; i < nn; ++i)
        if (!isfinite(policy[i])) { m0_set_error("network produced non-finite policy logits"); return M0_ERR_NONFINITE; }
    for (int i = 0; i < B; ++i)
        if (!isfinite(value[i])) { m0_set_error("network produced non-finite value"); return M0_ERR_NONFINITE; }
    return M0_OK;
}

int m0_net_ssl_channels(const m0_net* n) { return n ? n->net->ssl_channels_total() : 0; }
int64_t m0_net_param_count(const m0_net* n) { return n ? (int64_t)n->net->param_count() : 0; }
double m0_net_flops_per_position(const m0_net* n, int with_ssl) { return n ? n->net->flops_per_position(with_ssl != 0) : 0.0; }

int m0_net_profile_enable(m0_net* n, int on) {
    if (!n) { m0_set_error("net is null"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    n->net->set_profile(on != 0);
    return M0_OK;
}

int m0_net_profile_get(m0_net* n, double* conv_ms, double* conv_flop, int64_t* launches, int reset) {
    if (!n) { m0_set_error("net is null"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    if (conv_ms) *conv_ms = n->net->prof_conv_ms();
    if (conv_flop) *conv_flop = n->net->prof_conv_flop();
    if (launches) *launches = n->net->prof_conv_launches();
    if (reset) n->net->reset_profile();
    return M0_OK;
}

int m0_net_profile_get_tail(m0_net* n, double* tail_ms, int64_t* tail_launches) {
    if (!n) { m0_set_error("net is null"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    if (tail_ms) *tail_ms = n->net->prof_tail_ms();
    if (tail_launches) *tail_launches = n->net->prof_tail_launches();
    return M0_OK;
}

int m0_net_bench_forward(m0_net* n, int B, int iters, int with_ssl, float* ms_per_forward) {
    if (!n || !ms_per_forward || B <= 0 || iters <= 0) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(n->mu);
    if (hipSetDevice(n->device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return M0_ERR_HIP; }
    int rc = ensure_io(n, B);
    if (rc != M0_OK) return rc;
    std::string err;
    // synthetic resident input: sparse 0/1 piece planes + constant planes (with_ssl bit 1: keep the input of the previous call
    // of the same batch size -- tools/exp_cumask.py times two networks side by side without the host-side generation)
    if (!(with_ssl & 2)) {
        const int P = n->net->cfg().planes;
        std::vector<float> h((size_t)B * P * 64, 0.f);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < h.size(); ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            int p = (int)((i / 64) % P);
            if (p < 12) h[i] = ((s >> 20) % 100) < 8 ? 1.f : 0.f;
            else h[i] = (float)((((i / 64 / P) * 7 + p) % 10) / 10.0);
        }
        (void)hipMemcpy(n->planes_dev, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    float* ssl = ((with_ssl & 1) && n->net->ssl_channels_total() > 0) ? n->ssl_dev : nullptr;
    rc = n->net->forward(n->planes_dev, nullptr, B, n->logits_dev, n->value_dev, ssl, n->stream, err);   // warm-up
    if (rc != M0_OK) { m0_set_error(err); return rc; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, n->stream);
    for (int i = 0; i < iters; ++i) {
        rc = n->net->forward(n->planes_dev, nullptr, B, n->logits_dev, n->value_dev, ssl, n->stream, err);
        if (rc != M0_OK) { m0_set_error(err); return rc; }
    }
    (void)hipEventRecord(e1, n->stream);
    hipError_t e = hipEventSynchronize(e1);
    if (e != hipSuccess) { m0_set_error(std::string("bench forward failed: ") + hipGetErrorString(e)); return M0_ERR_HIP; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_forward = ms / iters;
    n->net->harvest_profile();
    return M0_OK;
}

}  // extern "C"
