// C ABI of libqcqp_mi.so (declared in include/qcqp_mi.h): context, problem upload, population
// management, launches of the kernels in kernels.hip, HIP-event timing, lazy RCCL.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/qcqp_mi.h"
#include "kernels.hip"
#include "admm.h"
#include "admm_fused.h"
#include "cd_queue.h"
#include "cd_life.h"
#include "gemm_pk.h"
#include "cd_general.h"
#include "cd_dense.h"
#include "sdr_solve.h"

using namespace qcqpmi;

namespace {

thread_local std::string g_create_error;

struct HostQuad {
    bool set = false;
    std::vector<int> ci, cj;  // COO
    std::vector<double> cv;
    std::vector<double> q;
    double r = 0.0;
    int relop = 0;
    // generated on the device (qcqpmi_set_quad_generated)
    bool gen = false;
    uint64_t gseed = 0;
    double gscale = 0.0, gqscale = 0.0, gdiag = 0.0;
    // packed on the device at qcqpmi_set_quad time (problems whose constraint matrices exceed the row-major budget)
    bool streamed = false;
};

struct Timer {
    hipEvent_t beg = nullptr, end = nullptr;
    bool valid = false;
};

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names) {
            api.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.h) break;
        }
        if (api.h) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
            api.Broadcast = (decltype(api.Broadcast))dlsym(api.h, "ncclBroadcast");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
        }
    }
    return (api.h && api.GetUniqueId && api.CommInitRank && api.AllGather && api.Broadcast &&
            api.AllReduce) ? &api : nullptr;
}

}  // namespace

// Busy-wait for the stream.  hipStreamSynchronize and the implicit wait of a pageable copy go to sleep in slices once a kernel
// has run for a few milliseconds and wake up 10-30 ms late in about one call of four on this stack (measured round 6 with the 34 ms
// lifecycle launch: tools/timed_region_probe.py, profiles/r06_timed_region.md); the long launches of the coordinate-descent paths
// are therefore awaited by polling hipStreamQuery -- one host core spins for the length of the launch.
hipError_t spin_sync(hipStream_t st) {
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
}

struct qcqpmi_ctx {
    int device = 0;
    int64_t n = 0, n16 = 0, m = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<HostQuad> quads;  // m+1
    bool finalized = false;
    bool sep = false;
    int maxc = 0;
    int K = 0;
    int objclass = 0;  // 1: every P0[i,i] > 0, 2: every P0[i,i] == 0, 0: mixed
    bool symcls = false;  // one constraint class of the form p x_i^2 + r == 0 (feasible sets mirrored about 0)
    DevProblem dp{};
    std::vector<void *> prob_allocs;
    // population
    int64_t R = 0, Rpad = 0, Rcap = 0;
    double *X = nullptr, *Xi = nullptr;
    double *d_f0 = nullptr, *d_mv = nullptr, *d_F = nullptr;
    int64_t *d_visits = nullptr, *d_acc = nullptr, *d_sweeps = nullptr, *d_sweeps1 = nullptr;
    int *d_status = nullptr, *d_status1 = nullptr;
    uint8_t *d_flag = nullptr;
    int64_t *d_best_idx = nullptr;
    double *d_best_key = nullptr;
    double *d_stage = nullptr;  // host-layout staging (n x Rcap)
    int64_t F_cap = 0;
    bool evaluated = false;
    // SDR factor
    double *d_Fpack = nullptr, *d_Frow = nullptr, *d_mu = nullptr;
    Timer timers[5];
    // ADMM: the stacked bases B = [B_1 ... B_m] packed for the two products (gemm_pk.h), eigenvalues, B^T q, brackets
    // unit bases (every basis vector is +-e_i: separable constraints): ZQ = W^T Z is a gather, S = W D a scatter (no GEMM)
    int *ad_uidx = nullptr, *ad_uptr = nullptr, *ad_ulist = nullptr;       // [Mh16] coordinate of basis row h (-1: padding); CSR coordinate -> rows
    double *ad_usgn = nullptr;                                             // [Mh16] its sign
    bool ad_unit = false, ad_unit_off = false;
    double *ad_WTpk = nullptr, *ad_Wpk = nullptr, *ad_lam = nullptr, *ad_qhat = nullptr, *ad_rk = nullptr, *ad_slo = nullptr, *ad_ehi = nullptr;
    double *ad_Minvpk = nullptr;
    int *ad_relop = nullptr;
    int64_t ad_rows = 0, ad_Mh16 = 0;   // hat rows per constraint (n, or rp of a reduced basis); m * rows padded to 16
    int ad_lowrank = 0;
    bool ad_qzero = false;              // every B_k^T q_k is zero
    void *rb_handle = nullptr;          // rocBLAS handle: only the rocSOLVER setup path needs one
    bool p0_diag = false;               // P0 has no off-diagonal entries (z-update and f0 need no product then)
    std::vector<double> p0_diag_host;
    // outputs of a run packed into one device buffer, copied with ONE transfer into pinned host memory
    char *d_out = nullptr, *h_out = nullptr;
    char *h_pin = nullptr;                // pinned bounce buffer of the relaxation solver's downloads (pin_reserve)
    size_t h_pin_cap = 0;
    int64_t out_cap = 0;
    int eval_zs = 1;   // planes of the last dense evaluation
    double *d_wS = nullptr, *d_wY = nullptr, *d_ww = nullptr, *d_wz = nullptr;   // qcqpmi_pop_weighted_product work buffers
    int64_t wY_cap = 0;
    double *d_planes = nullptr;   // partial planes of x'P0x from the GEMM evaluation
    int64_t planes_cap = 0;
    double *d_gP = nullptr;   // dense constraint matrices [m][n][n] (problems whose constraints couple coordinates)
    // dense-constraint path (cd_dense.h): all matrices in block-major fragment order + work buffers
    const double *dn_Gpack = nullptr, *dn_q = nullptr, *dn_qT = nullptr, *dn_r = nullptr;
    const int *dn_relop = nullptr;
    double *dn_G = nullptr, *dn_Dg = nullptr, *dn_Ft = nullptr;
    int dense_chain_mode = 0;             // 0: four waves per restart where it applies, 1: one wave per restart (cross-check)
    long long *dn_prof = nullptr;         // 32 tick sums of the dense chain kernel (debug profile)
    int64_t dn_G_cap = 0, dn_state_cap = 0;
    void *dn_state = nullptr;
    hipStream_t stream2 = nullptr;   // second stream of the dense path: products of block b+1 while the chain walks b
    hipEvent_t dn_ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t dn_evn[2] = {nullptr, nullptr};   // "live count of sweep t copied" (dense path: the host runs one sweep ahead)
    int *dn_hn = nullptr;                        // pinned: the two live counts in flight
    bool dn_force = false;    // generated functions: the dense path is the only one that holds them
    // streaming upload (coupled constraints beyond `stream_limit` bytes of row-major storage): every function is packed for the
    // matrix cores as it arrives; neither the host nor the device ever holds a second copy
    double stream_limit = 16e9;
    double *dn_gp_pre = nullptr, *dn_tmp = nullptr;
    // comm
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *d_comm = nullptr, *d_comm_big = nullptr;
    int64_t comm_big_cap = 0;
    long long *d_prof = nullptr;
    int cd_stage = 0;                   // qcqpmi_cd_run_stage: last stage completed of a split run
    int64_t Xi_cap = 0;                 // doubles reserved for the standard normals of pop_sdr_sample
    bool sdr_factor_resident = false;   // d_Fpack / d_mu hold the pair of the last pop_sdr_sample
    double ad_Minv_rho = 0.0;           // rho of the z-solver matrix formed by qcqpmi_admm_zsolver_device
    bool ad_Minv_device = false;
    const char *last_cd2_kernel = "";     // name of the phase-2 kernel of the most recent cd run (bench / profiles)
    bool profile = false;
    int dbg = 0;
    // fused persistent ADMM kernel (admm_fused.h): one work buffer kept across runs, grown on demand
    char *af_work = nullptr;
    size_t af_work_cap = 0;
    bool ad_fused = true;                 // qcqpmi_admm_fused: use the fused kernel where it applies
    int ad_fused_nt = 512;                // threads per workgroup of the fused kernel (256: qcqpmi_admm_fused(ctx, 2))
    const char *last_admm_kernel = "";    // "admm_fused_kernel" / "admm_multi_launch"
    int last_admm_C = 0;                  // workgroups per tile of the last fused run
    long long af_prof[16] = {0};          // stage cycle counters of the last fused run (when qcqpmi_debug_profile enabled them)
    int cd_queue = 2;                     // qcqpmi_cd_queue: 0 off, 1 restart-level scheduling (cd_phase2_qs_kernel) wherever it applies,
                                          // 2 auto: when there are more tiles than CUs
    int *d_qnext = nullptr;               // [0] queue head of the slot-queue kernel
    bool q_prepared = false;              // the queue of the resident population has been reset
    CdLife *d_life = nullptr;    // qcqpmi_cd_stream_run: parameters of the lifecycle launch
    // second-generation lifecycle kernel (cd_life.h): the workgroups' X tiles, the staged diagonal blocks, the watchdog word
    int Kreal = 0;               // constraint classes among the REAL coordinates (the padded ones of n16 carry no constraint)
    double fbound = 0.0;         // sum |P0| + sum |q0| + |r0|
    double *l2_scratch = nullptr; size_t l2_scratch_cap = 0;
    double *l2_D = nullptr, *l2_S = nullptr;
    int *l2_abort = nullptr;
    int *l2_cuslot = nullptr;    // [4096] arrival counters per compute unit of cd_life_kernel (zeroed before every launch)
    // factored objective P0 = L L^T (qcqpmi_cd_set_objective_factor): L (n16 x 16 lr_RB, zero-padded) and its fragment packs
    double *lr_L = nullptr, *lr_G = nullptr, *lr_U = nullptr; int lr_RB = 0;
    int life_version = 0;        // qcqpmi_cd_life_version: 0 = the faster one for the shape, 2 = cd_life_kernel wherever it applies, 1 = cd_phase2_qs_kernel<lifecycle> only
    long long *d_life_prof = nullptr;
    int64_t *d_bestK_idx = nullptr; double *d_bestK_key = nullptr, *d_bestK_x = nullptr; int64_t bestK_cap = 0;
    bool cd_ref_order = false;   // qcqpmi_cd_reference_order: coupled constraints in the reference's summation order
    bool force_generic = false;  // debug/tests: run the general phase-2 kernel even when the pipelined one applies
    std::vector<int> last_st1, last_st2;   // per-restart status codes of the last coordinate-descent run (qcqpmi_cd_status)
};

namespace {

void admm_free(qcqpmi_ctx *c, bool keep_zsolver);

int fail(qcqpmi_ctx *c, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(c, QCQPMI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

template <typename T>
int dev_alloc(qcqpmi_ctx *c, T **p, size_t count, bool zero = true) {
    void *v = nullptr;
    HIPCHK(c, hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)));
    if (zero) HIPCHK(c, hipMemsetAsync(v, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
    *p = (T *)v;
    return 0;
}

template <typename T>
int prob_upload(qcqpmi_ctx *c, const T **dst, const std::vector<T> &src) {
    T *p = nullptr;
    int rc = dev_alloc(c, &p, src.size(), false);
    if (rc) return rc;
    c->prob_allocs.push_back(p);
    if (!src.empty())
        HIPCHK(c, hipMemcpyAsync(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dst = p;
    return 0;
}

void free_population(qcqpmi_ctx *c) {
    void *ptrs[] = {c->X, c->Xi, c->d_f0, c->d_mv, c->d_F, c->d_visits, c->d_acc, c->d_sweeps,
                    c->d_sweeps1, c->d_status, c->d_status1, c->d_flag, c->d_stage};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    c->X = c->Xi = c->d_f0 = c->d_mv = c->d_F = c->d_stage = nullptr;
    c->d_visits = c->d_acc = c->d_sweeps = c->d_sweeps1 = nullptr;
    c->d_status = nullptr; c->d_status1 = nullptr; c->d_flag = nullptr;
    c->Rcap = 0; c->F_cap = 0;
}

int pop_reserve(qcqpmi_ctx *c, int64_t R) {
    if (R <= 0) return fail(c, QCQPMI_EINVAL, "population size must be positive");
    int64_t Rpad = (R + 15) / 16 * 16;
    if (Rpad > c->Rcap) {
        HIPCHK(c, spin_sync(c->stream));
        free_population(c);
        int rc = 0;
        size_t xe = (size_t)Rpad * (size_t)c->n16;
        if ((rc = dev_alloc(c, &c->X, xe))) return rc;
        if ((rc = dev_alloc(c, &c->d_stage, (size_t)Rpad * (size_t)c->n))) return rc;
        if ((rc = dev_alloc(c, &c->d_f0, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_mv, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_visits, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_acc, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_sweeps, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_sweeps1, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_status, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_status1, Rpad))) return rc;
        if ((rc = dev_alloc(c, &c->d_flag, Rpad))) return rc;
        c->Rcap = Rpad;
    }
    c->R = R;
    c->Rpad = Rpad;
    c->evaluated = false;
    return 0;
}

void tic(qcqpmi_ctx *c, int which) { (void)hipEventRecord(c->timers[which].beg, c->stream); }
void toc(qcqpmi_ctx *c, int which) {
    (void)hipEventRecord(c->timers[which].end, c->stream);
    c->timers[which].valid = true;
}

bool dense_on(const qcqpmi_ctx *c);
int launch_eval_dense(qcqpmi_ctx *c, bool with_Ft);
int eval_parts_dense(qcqpmi_ctx *c);

// per-restart results of a coordinate-descent run -> host: one pack kernel, one copy into pinned memory,
// one synchronisation (nine pageable copies cost ~0.3 ms of staging kernels per call)
__global__ void pack_cd_outputs_kernel(char *out, int64_t R, const int64_t *s1, const int64_t *s2, const int64_t *vis,
                                       const int64_t *acc, const double *f0, const double *mv, const int *st,
                                       const int *st1, const uint8_t *flag) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int64_t *o64 = (int64_t *)out;
    o64[r] = s1[r]; o64[R + r] = s2[r]; o64[2 * R + r] = vis[r]; o64[3 * R + r] = acc[r];
    double *od = (double *)(out + 32 * R);
    od[r] = f0[r]; od[R + r] = mv[r];
    int *oi = (int *)(out + 48 * R);
    oi[r] = st[r]; oi[R + r] = st1[r];
    ((uint8_t *)(out + 56 * R))[r] = flag[r];
}

// pinned host memory for downloads that are repeated thousands of times (a 2-D copy into pageable memory is staged row by
// row by the runtime: 1025 rows of 47 doubles cost 15 ms)
int pin_reserve(qcqpmi_ctx *c, size_t bytes) {
    if (bytes <= c->h_pin_cap) return 0;
    HIPCHK(c, spin_sync(c->stream));
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr; c->h_pin_cap = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->h_pin, bytes, hipHostMallocDefault));
    c->h_pin_cap = bytes;
    return 0;
}

// staging buffers (device + pinned host) of fetch_cd_outputs for R restarts
int cd_outputs_reserve(qcqpmi_ctx *c, int64_t R) {
    const int64_t bytes = 57 * R;
    if (bytes > c->out_cap) {
        HIPCHK(c, spin_sync(c->stream));
        if (c->d_out) (void)hipFree(c->d_out);
        if (c->h_out) (void)hipHostFree(c->h_out);
        c->d_out = c->h_out = nullptr;
        HIPCHK(c, hipMalloc((void **)&c->d_out, (size_t)bytes));
        HIPCHK(c, hipHostMalloc((void **)&c->h_out, (size_t)bytes, hipHostMallocDefault));
        c->out_cap = bytes;
    }
    return 0;
}

// device buffers of the per-population winners of a streamed run (K populations)
int cd_bestK_reserve(qcqpmi_ctx *c, int64_t K) {
    if (K > c->bestK_cap) {
        if (c->d_bestK_idx) (void)hipFree(c->d_bestK_idx);
        if (c->d_bestK_key) (void)hipFree(c->d_bestK_key);
        if (c->d_bestK_x) (void)hipFree(c->d_bestK_x);
        c->d_bestK_idx = nullptr; c->d_bestK_key = nullptr; c->d_bestK_x = nullptr;
        HIPCHK(c, hipMalloc((void **)&c->d_bestK_idx, (size_t)K * 2 * sizeof(int64_t)));
        HIPCHK(c, hipMalloc((void **)&c->d_bestK_key, (size_t)K * 2 * sizeof(double)));
        HIPCHK(c, hipMalloc((void **)&c->d_bestK_x, (size_t)K * c->n * sizeof(double)));
        c->bestK_cap = K;
    }
    return 0;
}

int fetch_cd_outputs(qcqpmi_ctx *c, int64_t *sweeps1, int64_t *sweeps2, int64_t *visits2, int64_t *accepted2,
                     uint8_t *ran_phase2, double *f0, double *maxviol, std::vector<int> &st, std::vector<int> &st1) {
    const int64_t R = c->R, bytes = 57 * R;
    int rco = cd_outputs_reserve(c, R);
    if (rco) return rco;
    hipLaunchKernelGGL(pack_cd_outputs_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, c->stream, c->d_out, R,
                       (const int64_t *)c->d_sweeps1, (const int64_t *)c->d_sweeps, (const int64_t *)c->d_visits,
                       (const int64_t *)c->d_acc, (const double *)c->d_f0, (const double *)c->d_mv,
                       (const int *)c->d_status, (const int *)c->d_status1, (const uint8_t *)c->d_flag);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    const char *h = c->h_out;
    const size_t n8 = (size_t)R * 8;
    if (sweeps1) memcpy(sweeps1, h, n8);
    if (sweeps2) memcpy(sweeps2, h + n8, n8);
    if (visits2) memcpy(visits2, h + 2 * n8, n8);
    if (accepted2) memcpy(accepted2, h + 3 * n8, n8);
    if (f0) memcpy(f0, h + 4 * n8, n8);
    if (maxviol) memcpy(maxviol, h + 5 * n8, n8);
    st.resize((size_t)R); st1.resize((size_t)R);
    memcpy(st.data(), h + 6 * n8, (size_t)R * 4);
    memcpy(st1.data(), h + 6 * n8 + (size_t)R * 4, (size_t)R * 4);
    if (ran_phase2) memcpy(ran_phase2, h + 7 * n8, (size_t)R);
    return 0;
}

// Per-restart status policy of a coordinate-descent run.  A restart on which the reference would raise
// (code != 0) is a FAILED restart, not a failed population: its objective / max violation become +inf (host
// outputs and the device copies select_best reads), the codes stay readable through qcqpmi_cd_status.  The call
// itself fails -- with the reference's message -- only when every restart failed (in particular R == 1, the
// reference's single-point behaviour).
int cd_apply_status(qcqpmi_ctx *c, const std::vector<int> &st, const std::vector<int> &st1, double *f0, double *maxviol,
                    int seg_cap) {
    const int64_t R = c->R;
    c->last_st1 = st1; c->last_st2 = st;
    int64_t nfail = 0, first = -1;
    for (int64_t r = 0; r < R; r++)
        if (st1[(size_t)r] || st[(size_t)r]) { if (first < 0) first = r; nfail++; }
    if (nfail == 0) return 0;
    if (nfail < R) {
        const double inf = INFINITY;
        for (int64_t r = 0; r < R; r++) {
            if (!(st1[(size_t)r] || st[(size_t)r])) continue;
            if (f0) f0[r] = inf;
            if (maxviol) maxviol[r] = inf;
            HIPCHK(c, hipMemcpyAsync(c->d_f0 + r, &inf, sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_mv + r, &inf, sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(c, spin_sync(c->stream));
        return 0;
    }
    const int64_t r = first;
    const int s1 = st1[(size_t)r], s2 = st[(size_t)r];
    if (s1 == -3) return fail(c, QCQPMI_EREFERENCE, "phase 1: a variable appears in no constraint (reference: ValueError: max() arg is an empty sequence, qcqp.py:117); restart %lld", (long long)r);
    if (s1 == -4 || s2 == -4) return fail(c, QCQPMI_EUNSUPPORTED, "feasible set with more than %d segments; restart %lld", seg_cap, (long long)r);
    if (s1) return fail(c, QCQPMI_EREFERENCE, "phase 1: unbounded feasible interval with zero objective (reference: OverflowError in np.random.uniform, utilities.py:267); restart %lld", (long long)r);
    return fail(c, QCQPMI_EREFERENCE, "phase 2: the reference raises on restart %lld (code %d: unbounded interval with zero objective / NameError in OneVarQuadraticFunction.eval)", (long long)r, s2);
}

int launch_eval(qcqpmi_ctx *c, bool want_F) {
    if (dense_on(c)) return launch_eval_dense(c, false);   // coupled constraints: every function on the matrix cores
    if (want_F) {
        int64_t need = (c->m + 1) * c->Rpad;
        if (need > c->F_cap) {
            if (c->d_F) { HIPCHK(c, spin_sync(c->stream)); (void)hipFree(c->d_F); c->d_F = nullptr; }
            int rc = dev_alloc(c, &c->d_F, need);
            if (rc) return rc;
            c->F_cap = need;
        }
    }
    EvalArgs a;
    a.P = c->dp; a.X = c->X; a.R = c->R; a.f0 = c->d_f0; a.maxviol = c->d_mv;
    a.F = want_F ? c->d_F : nullptr; a.Rpad = c->Rpad;
    a.planes = nullptr; a.nplanes = 0;
    tic(c, 0);
    const int NB = (int)(c->n16 / 16), ntiles = (int)(c->Rpad / 16);
    if (NB >= 8 && ntiles >= 8) {
        // x'P0x through the LDS-tiled GEMM: 8 row blocks x 8 tiles per workgroup share their operands
        // (the per-tile MFMA loop of eval_kernel re-reads all of P0 from L2 for every tile)
        const int groups = (NB + DP_FG - 1) / DP_FG, nplanes = 2 * groups;
        if ((int64_t)nplanes * c->Rpad > c->planes_cap) {
            if (c->d_planes) { HIPCHK(c, spin_sync(c->stream)); (void)hipFree(c->d_planes); c->d_planes = nullptr; }
            int rc = dev_alloc(c, &c->d_planes, (size_t)nplanes * c->Rpad);
            if (rc) return rc;
            c->planes_cap = (int64_t)nplanes * c->Rpad;
        }
        DenseProdArgs pa;
        pa.D.Gpack = c->dp.Apack; pa.D.q = nullptr; pa.D.qT = nullptr; pa.D.r = nullptr; pa.D.relop = nullptr;
        pa.D.n = c->n; pa.D.n16 = c->n16; pa.D.NB = NB; pa.D.KS = (int)(c->n16 / 4); pa.D.m1 = 1; pa.D.m1p = 64;
        pa.X = c->X; pa.ntiles = ntiles; pa.b = 0; pa.zs = 1; pa.G = nullptr; pa.F = c->d_planes; pa.Rpad = c->Rpad;
        pa.tile_on = nullptr; pa.hole = -1; pa.ch_only = -1; pa.zplane = 0;
        auto kg = dense_products_kernel<2>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_BYTES));
        hipLaunchKernelGGL(kg, dim3((unsigned)groups, (unsigned)((ntiles + DP_TG - 1) / DP_TG), 1), dim3(256), DP_LDS_BYTES, c->stream, pa);
        a.planes = c->d_planes; a.nplanes = nplanes;
    }
    hipLaunchKernelGGL(eval_kernel, dim3((unsigned)(c->Rpad / 16)), dim3(256), 0, c->stream, a);
    toc(c, 0);
    HIPCHK(c, hipGetLastError());
    c->evaluated = true;
    return 0;
}

// does phase 2 of the resident population go through the slot-queue kernel?
bool cd_queue_eligible(qcqpmi_ctx *c, bool profiling) {      // the problem's shape and the context's switches allow the kernel
    if (profiling || c->force_generic || (c->dbg & 64) || !c->sep || c->maxc > 1 || c->cd_queue == 0) return false;
    if (!(c->K == 1 && c->objclass == 1 && c->symcls && c->n % 16 == 0)) return false;
    return cd_queue_lds_bytes(c->dp) != 0 && c->R < (1LL << 30);
}

bool cd_queue_applies(qcqpmi_ctx *c, bool profiling) {
    if (!cd_queue_eligible(c, profiling)) return false;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) return false;
    if (c->cd_queue == 1) return true;
    if (c->cd_queue == 2) return c->Rpad / 16 > cus;
    return false;
}

// reset the queue of the resident population and the per-restart outputs the kernel only writes for the restarts it runs,
// then publish the population (generation number) to whoever may pull from it
int cd_queue_prepare(qcqpmi_ctx *c) {
    if (!c->d_qnext) { int rcq = dev_alloc(c, &c->d_qnext, 16); if (rcq) return rcq; }
    HIPCHK(c, hipMemsetAsync(c->d_visits, 0, (size_t)c->Rpad * sizeof(int64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_acc, 0, (size_t)c->Rpad * sizeof(int64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_sweeps, 0, (size_t)c->Rpad * sizeof(int64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_status, 0, (size_t)c->Rpad * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_qnext, 0, 16 * sizeof(int), c->stream));
    c->q_prepared = true;
    return 0;
}

void cd_queue_fill_batch(qcqpmi_ctx *c, CdBatch &B, uint64_t seed, uint64_t first_index) {
    B.X = c->X; B.f0cur = c->d_f0; B.slack = c->d_mv; B.flag = c->d_flag; B.visits = c->d_visits; B.accepted = c->d_acc;
    B.sweeps = c->d_sweeps; B.status = c->d_status; B.f0out = c->d_f0; B.mvout = c->d_mv; B.R = c->R; B.seed = seed;
    B.first_index = first_index; B.next = c->d_qnext;
}

template <int MAXC>
int launch_cd(qcqpmi_ctx *c, const CdArgs &a1, bool phase1, bool &used_lds, bool *used_rs = nullptr) {
    if (used_rs) *used_rs = false;
    if (!phase1) c->last_cd2_kernel = "cd_phase2_kernel";
    dim3 grid((unsigned)(c->Rpad / 16)), block(256);
    if (phase1) {
        tic(c, 1);
        hipLaunchKernelGGL(cd_phase1_sep_kernel<MAXC>, grid, dim3(P1_THREADS), 0, c->stream, a1);
        toc(c, 1);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    const DevProblem &dp = c->dp;
    // second-generation role-split kernel (cd_phase2_q.h): one mirrored equality class, positive diagonal, n a
    // multiple of 16, the tile resident in LDS -- the Boolean least squares family of the headline benchmark
    if (MAXC == 1 && c->K == 1 && c->objclass == 1 && c->symcls && c->n % 16 == 0 && !c->force_generic && !(c->dbg & 64)) {
        // (at least RQ_LDS_MIN: the product loop's look-ahead loads address a full-size tile)
        size_t q_lds = ((size_t)RQ_LDS_COMMON + (size_t)c->n16 * 16) * sizeof(double);
        if (q_lds < RQ_LDS_MIN) q_lds = RQ_LDS_MIN;
        const int NBq = (int)(c->n16 / 16);
        int cs = (c->dbg & 128) ? ((c->dbg >> 8) & 7) : 4;     // debug knob: blocks of the contraction the chain wave multiplies
        cs = cs > RQ_CSMAX ? RQ_CSMAX : cs;
        if (cs >= NBq) cs = 0;
        cs &= ~1;
        if (cd_queue_applies(c, a1.prof != nullptr) && NBq - cs <= RQ_NSIMD * RQ_MAXU && NBq >= 3) {
            // restart-level scheduling (cd_queue.h): 16 slots per workgroup, refilled from a device-side queue
            used_lds = true;
            int rcq;
            if (!c->q_prepared && (rcq = cd_queue_prepare(c))) return rcq;
            c->q_prepared = false;
            CdQueueArgs qa;
            qa.P = dp; qa.num_iters = a1.num_iters; qa.tol = a1.tol; qa.life = nullptr; qa.life_on = 0;
            cd_queue_fill_batch(c, qa.b, a1.seed, a1.first_index);
            int cus = 0;
            HIPCHK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            (void)hipEventRecord(c->timers[2].beg, c->stream);
            hipError_t qe = (hipError_t)cd_queue_launch(qa, cs, cus, c->stream);
            (void)hipEventRecord(c->timers[2].end, c->stream);
            c->timers[2].valid = true;
            if (qe != hipSuccess) return fail(c, QCQPMI_EHIP, "cd_queue_launch: %s", hipGetErrorString(qe));
            c->last_cd2_kernel = "cd_phase2_qs_kernel";
            if (used_rs) *used_rs = true;
            return 0;
        }
        if (q_lds <= 160 * 1024 && NBq - cs <= RQ_NSIMD * RQ_MAXU && NBq >= 3) {
            used_lds = true;
            const bool prof_ = a1.prof != nullptr;
            auto k = cd_phase2_q_kernel<4, false>;
            if (prof_) k = cs == 0 ? cd_phase2_q_kernel<0, true> : cs == 2 ? cd_phase2_q_kernel<2, true> : cs == 4 ? cd_phase2_q_kernel<4, true> : cd_phase2_q_kernel<6, true>;
            else k = cs == 0 ? cd_phase2_q_kernel<0, false> : cs == 2 ? cd_phase2_q_kernel<2, false> : cs == 4 ? cd_phase2_q_kernel<4, false> : cd_phase2_q_kernel<6, false>;
            HIPCHK(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q_lds));
            tic(c, 2);
            hipLaunchKernelGGL(k, grid, dim3(512), q_lds, c->stream, a1, dp.Apack, dp.Apack2, dp.P0, dp.q0, dp.rcp2d);
            toc(c, 2);
            c->last_cd2_kernel = "cd_phase2_q_kernel";
            HIPCHK(c, hipGetLastError());
            if (used_rs) *used_rs = true;
            return 0;
        }
    }
    // role-split pipelined kernel for the single-class Boolean / box families
    if (MAXC == 1 && c->K == 1 && (c->objclass == 1 || c->objclass == 2) && !c->force_generic) {
        const size_t rs_common = (size_t)(2 * 6 * 256 + 256 + 2 * 256 + 2 * 16 + 2 * 16 + 16 + 4 * 16 + 8 + 8 + 8) * sizeof(double);
        const size_t rs_with_x = rs_common + (size_t)c->n16 * 16 * sizeof(double);
        used_lds = rs_with_x <= 160 * 1024;
        const size_t rs_lds = used_lds ? rs_with_x : rs_common;
        dim3 block512(512);
#define QM_RS(XL, FA)                                                                               \
    do {                                                                                            \
        auto k = cd_phase2_rs_kernel<XL, FA, false, false, false>;                                  \
        const bool full_ = (c->n % 16 == 0), sym_ = c->symcls, prof_ = a1.prof != nullptr;          \
        if (prof_) k = full_ ? (sym_ ? cd_phase2_rs_kernel<XL, FA, true, true, true> : cd_phase2_rs_kernel<XL, FA, true, false, true>) \
                             : (sym_ ? cd_phase2_rs_kernel<XL, FA, false, true, true> : cd_phase2_rs_kernel<XL, FA, false, false, true>); \
        else k = full_ ? (sym_ ? cd_phase2_rs_kernel<XL, FA, true, true, false> : cd_phase2_rs_kernel<XL, FA, true, false, false>) \
                       : (sym_ ? cd_phase2_rs_kernel<XL, FA, false, true, false> : cd_phase2_rs_kernel<XL, FA, false, false, false>); \
        HIPCHK(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_lds)); \
        tic(c, 2);                                                                                  \
        hipLaunchKernelGGL(k, grid, block512, rs_lds, c->stream, a1, dp.Apack, dp.Apack2, dp.P0, dp.q0, dp.rcp2d); \
        toc(c, 2);                                                                                  \
    } while (0)
        if (used_lds && c->objclass == 1) QM_RS(true, 1);
        else if (used_lds) QM_RS(true, 2);
        else if (c->objclass == 1) QM_RS(false, 1);
        else QM_RS(false, 2);
#undef QM_RS
        c->last_cd2_kernel = "cd_phase2_rs_kernel";
        HIPCHK(c, hipGetLastError());
        if (used_rs) *used_rs = true;
        return 0;
    }
    // dynamic LDS: [X tile] + partial tiles + G + diagonal block + slack + feasible-set table
    const bool use_cls = c->K <= 16;
    const size_t slots = use_cls ? (size_t)c->K * 16 : 0;
    size_t common = (size_t)(4 * 256 + 256 + 256 + 3 * 16 + 512) * sizeof(double) +              // part, G, Dblk, slk/q0b/rcpb, midb/thrb
                    (size_t)(2 * (MAXC + 1) * 256 + 256) * sizeof(double) +                      // block table
                    ((size_t)(2 * (MAXC + 1)) * slots + ((slots + 1) / 2) * 2) * sizeof(double) + // class table
                    64;
    size_t with_x = common + (size_t)c->n16 * 16 * sizeof(double);
    used_lds = with_x <= 160 * 1024;
    const size_t lds = used_lds ? with_x : common;
    const int fast = (MAXC == 1) ? c->objclass : 0;
    const bool uni = use_cls && c->K == 1;
#define QM_LAUNCH2(XL, CL, FA, UN)                                                                     \
    do {                                                                                            \
        auto k = cd_phase2_kernel<MAXC, XL, CL, FA, UN>;                                                \
        HIPCHK(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        tic(c, 2);                                                                                  \
        hipLaunchKernelGGL(k, grid, block, lds, c->stream, a1, dp.Apack, dp.P0, dp.q0, dp.rcp2d, dp.cls);      \
        toc(c, 2);                                                                                  \
    } while (0)
#define QM_LAUNCH(XL, CL)                                                 \
    do {                                                                  \
        if (fast == 1 && uni && CL) QM_LAUNCH2(XL, CL, 1, CL);            \
        else if (fast == 2 && uni && CL) QM_LAUNCH2(XL, CL, 2, CL);       \
        else if (fast == 1) QM_LAUNCH2(XL, CL, 1, false);                 \
        else if (fast == 2) QM_LAUNCH2(XL, CL, 2, false);                 \
        else QM_LAUNCH2(XL, CL, 0, false);                                \
    } while (0)
    if (used_lds && use_cls) QM_LAUNCH(true, true);
    else if (used_lds) QM_LAUNCH(true, false);
    else if (use_cls) QM_LAUNCH(false, true);
    else QM_LAUNCH(false, false);
#undef QM_LAUNCH
#undef QM_LAUNCH2
    HIPCHK(c, hipGetLastError());
    return 0;
}

int check_ready(qcqpmi_ctx *c, bool need_pop) {
    if (!c) return QCQPMI_EINVAL;
    if (!c->finalized) return fail(c, QCQPMI_ESTATE, "qcqpmi_finalize has not been called");
    if (need_pop && c->R <= 0) return fail(c, QCQPMI_ESTATE, "no resident population (upload / randn / sdr_sample first)");
    return 0;
}

}  // namespace

#include "cd_dense_mw.h"
// which chain kernel the dense path dispatches to (four waves per restart unless a thread would hold more than 8 slots)
static bool dense_chain_mw(const qcqpmi_ctx *c) { return c->dense_chain_mode != 1 && mw_geometry((int)c->m + 1).SL <= 8; }
static const char *dense_chain_name(const qcqpmi_ctx *c) { return dense_chain_mw(c) ? "dense_chain_mw_kernel" : "dense_chain_kernel"; }
#include "capi_dense.inc"

// coordinate descent for constraints that couple coordinates (cd_general.h)
int cd_run_general(qcqpmi_ctx *c, int phase1, int64_t num_iters, double viol_tol, double tol, uint64_t seed,
                   uint64_t first_index, int64_t *sweeps1, int64_t *sweeps2, int64_t *visits2, int64_t *accepted2,
                   uint8_t *ran_phase2, double *f0, double *maxviol) {
    if (!c->d_gP) return fail(c, QCQPMI_EUNSUPPORTED, "coupled constraints: dense storage m*n*n exceeds 16 GB");
    if (num_iters < 0 || !(tol > 0.0)) return fail(c, QCQPMI_EINVAL, "cd_run: bad num_iters / tol");
    const int64_t m = c->m;
    const size_t lds = ((size_t)(m + 1) * 16 * 4 + 4 * 16 * GEN_CAP + 16 * GEN_CAP + 16) * sizeof(double);
    if (lds > 160 * 1024) return fail(c, QCQPMI_EUNSUPPORTED, "coupled constraints: m = %lld too large for the LDS-resident coefficient table", (long long)m);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    CdGenArgs g;
    CdArgs &a = g.b;
    a.P = c->dp; a.X = c->X; a.R = c->R; a.f0cur = c->d_f0; a.slack = c->d_mv;
    a.num_iters = num_iters; a.viol_tol = viol_tol; a.tol = tol; a.seed = seed; a.first_index = first_index;
    a.visits = c->d_visits; a.accepted = c->d_acc; a.sweeps = c->d_sweeps; a.status = c->d_status;
    a.flag = c->d_flag; a.prof = nullptr; a.dbg = 0; a.f0out = nullptr; a.mvout = nullptr;
    g.gP = c->d_gP; g.Rpad = c->Rpad;
    g.exact_t0 = ((c->n <= 64 || c->cd_ref_order) && !(c->dbg & 16)) ? 1 : 0;   // small problems (or on request): the reference's own arithmetic for t0
    dim3 grid((unsigned)(c->Rpad / 16)), block(256);
    auto k1 = cd_general_kernel<1>;
    auto k2 = cd_general_kernel<2>;
    HIPCHK(c, hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipMemsetAsync(c->d_sweeps1, 0, (size_t)c->Rpad * sizeof(int64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_status1, 0, (size_t)c->Rpad * sizeof(int), c->stream));
    if (phase1) {
        if ((rc = launch_eval(c, true))) return rc;
        g.F = c->d_F;
        CdGenArgs g1 = g;
        g1.b.sweeps = c->d_sweeps1;
        g1.b.status = c->d_status1;
        tic(c, 1);
        hipLaunchKernelGGL(k1, grid, block, lds, c->stream, g1);
        toc(c, 1);
        HIPCHK(c, hipGetLastError());
    }
    if ((rc = launch_eval(c, true))) return rc;
    g.F = c->d_F;
    hipLaunchKernelGGL(gate_kernel, dim3((unsigned)((c->Rpad + 255) / 256)), dim3(256), 0, c->stream, c->d_mv,
                       c->d_status1, c->d_flag, c->R, c->Rpad, viol_tol);
    HIPCHK(c, hipGetLastError());
    if (ran_phase2) HIPCHK(c, hipMemcpyAsync(ran_phase2, c->d_flag, (size_t)c->R, hipMemcpyDeviceToHost, c->stream));
    tic(c, 2);
    hipLaunchKernelGGL(k2, grid, block, lds, c->stream, g);
    toc(c, 2);
    HIPCHK(c, hipGetLastError());
    if ((rc = launch_eval(c, false))) return rc;
    std::vector<int> st, st1;
    if ((rc = fetch_cd_outputs(c, sweeps1, sweeps2, visits2, accepted2, nullptr, f0, maxviol, st, st1))) return rc;
    if ((rc = cd_apply_status(c, st, st1, f0, maxviol, GEN_CAP))) return rc;
    return 0;
}

// ============================================================================================

extern "C" {

int qcqpmi_abi_version(void) { return QCQPMI_ABI_VERSION; }

int qcqpmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *qcqpmi_last_error(const qcqpmi_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int qcqpmi_ctx_create(qcqpmi_ctx **out, int64_t n, int64_t m, int device) {
    if (!out || n <= 0 || m < 0) return fail(nullptr, QCQPMI_EINVAL, "ctx_create: bad arguments");
    int ndev = qcqpmi_device_count();
    if (ndev <= 0)
        return fail(nullptr, QCQPMI_EHIP, "no HIP device visible: the qcqp_amd engine requires an MI355X (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, QCQPMI_EINVAL, "device %d out of range (%d visible)", device, ndev);
    qcqpmi_ctx *c = new qcqpmi_ctx();
    c->device = device; c->n = n; c->m = m; c->n16 = (n + 15) / 16 * 16;
    c->quads.resize((size_t)m + 1);
    if (const char *sl = getenv("QCQPMI_STREAM_LIMIT")) { const double v = atof(sl); if (v > 0.0) c->stream_limit = v; }   // tests: exercise the streaming upload at small sizes
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < 5 && e == hipSuccess; i++) {
        e = hipEventCreate(&c->timers[i].beg);
        if (e == hipSuccess) e = hipEventCreate(&c->timers[i].end);
    }
    if (e != hipSuccess) {
        int rc = fail(nullptr, QCQPMI_EHIP, "ctx_create: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

void qcqpmi_ctx_destroy(qcqpmi_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)spin_sync(c->stream);
    if (c->comm && rccl() && rccl()->CommDestroy) rccl()->CommDestroy(c->comm);
    free_population(c);
    admm_free(c, false);
    for (void *p : c->prob_allocs) (void)hipFree(p);
    void *ptrs[] = {c->d_Fpack, c->d_Frow, c->d_mu, c->d_best_idx, c->d_best_key, c->d_comm, c->d_comm_big,   // d_gP is in prob_allocs
                    c->dn_G, c->dn_Dg, c->dn_Ft, c->dn_prof, c->dn_state, c->d_planes, c->d_out, c->d_wS, c->d_wY, c->d_ww, c->d_wz, c->af_work, c->d_qnext, c->d_life, c->d_life_prof, c->d_bestK_idx, c->d_bestK_key, c->d_bestK_x, c->l2_scratch, c->l2_D, c->l2_S, c->l2_abort, c->l2_cuslot, c->lr_L, c->lr_G, c->lr_U};
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &t : c->timers) { if (t.beg) (void)hipEventDestroy(t.beg); if (t.end) (void)hipEventDestroy(t.end); }
    for (auto &e : c->dn_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->dn_evn) if (e) (void)hipEventDestroy(e);
    if (c->dn_hn) (void)hipHostFree(c->dn_hn);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int qcqpmi_set_quad(qcqpmi_ctx *c, int64_t k, int format, const double *vals, const int64_t *idx,
                    const int64_t *ptr, int64_t nnz, const double *q, double r, int relop) {
    if (!c) return QCQPMI_EINVAL;
    if (c->finalized) return fail(c, QCQPMI_ESTATE, "set_quad after finalize");
    if (k < 0 || k > c->m || !q) return fail(c, QCQPMI_EINVAL, "set_quad: bad k or q");
    if ((k == 0) != (relop == 0) || relop < 0 || relop > 2)
        return fail(c, QCQPMI_EINVAL, "set_quad: relop %d invalid for function %lld", relop, (long long)k);
    HostQuad &h = c->quads[(size_t)k];
    h = HostQuad();
    const int64_t n = c->n;
    if (format == QCQPMI_FMT_DENSE) {
        if (!vals) return fail(c, QCQPMI_EINVAL, "set_quad: dense vals missing");
        for (int64_t i = 0; i < n; i++)
            for (int64_t j = 0; j < n; j++) {
                double v = vals[i * n + j];
                if (v != 0.0) { h.ci.push_back((int)i); h.cj.push_back((int)j); h.cv.push_back(v); }
            }
    } else if (format == QCQPMI_FMT_CSR) {
        if (!ptr || (nnz > 0 && (!idx || !vals))) return fail(c, QCQPMI_EINVAL, "set_quad: CSR arrays missing");
        for (int64_t i = 0; i < n; i++)
            for (int64_t e = ptr[i]; e < ptr[i + 1]; e++) {
                if (idx[e] < 0 || idx[e] >= n || e >= nnz) return fail(c, QCQPMI_EINVAL, "set_quad: CSR index out of range");
                if (vals[e] != 0.0) { h.ci.push_back((int)i); h.cj.push_back((int)idx[e]); h.cv.push_back(vals[e]); }
            }
    } else {
        return fail(c, QCQPMI_EINVAL, "set_quad: unknown format %d", format);
    }
    // P must be symmetric (include/qcqp_mi.h): get_onevar_func (utilities.py:99-105) and the in-block
    // Gauss-Seidel fold read P[k, j] where the mathematics has P[j, k].  The reference symmetrises in
    // get_qcqp_form (utilities.py:333, 345); raw arrays are checked here instead of being trusted.
    if (format == QCQPMI_FMT_DENSE) {
        for (int64_t i = 0; i < n; i++)
            for (int64_t j = i + 1; j < n; j++)
                if (vals[i * n + j] != vals[j * n + i])
                    return fail(c, QCQPMI_EINVAL, "set_quad: P of function %lld is not symmetric (entry %lld,%lld); pass (P + P^T)/2",
                                (long long)k, (long long)i, (long long)j);
    } else {
        std::vector<std::pair<int64_t, double>> ent(h.cv.size());
        for (size_t e = 0; e < h.cv.size(); e++) ent[e] = {(int64_t)h.ci[e] * n + h.cj[e], h.cv[e]};
        std::sort(ent.begin(), ent.end(), [](const std::pair<int64_t, double> &a, const std::pair<int64_t, double> &b) { return a.first < b.first; });
        size_t w = 0;
        for (size_t e = 0; e < ent.size(); e++) {   // duplicates add up (COO semantics)
            if (w > 0 && ent[w - 1].first == ent[e].first) ent[w - 1].second += ent[e].second;
            else ent[w++] = ent[e];
        }
        ent.resize(w);
        for (const auto &en : ent) {
            const int64_t i = en.first / n, j = en.first % n;
            if (i == j) continue;
            const int64_t tk = j * n + i;
            auto it = std::lower_bound(ent.begin(), ent.end(), tk, [](const std::pair<int64_t, double> &a, int64_t key) { return a.first < key; });
            const double tv = (it != ent.end() && it->first == tk) ? it->second : 0.0;
            if (tv != en.second)
                return fail(c, QCQPMI_EINVAL, "set_quad: P of function %lld is not symmetric (entry %lld,%lld); pass (P + P^T)/2",
                            (long long)k, (long long)i, (long long)j);
        }
    }
    bool coupled = false;      // does the function touch more than one coordinate?  (separable constraints stay per-coordinate lists)
    {
        int64_t coord = -1;
        for (size_t e = 0; e < h.cv.size() && !coupled; e++) {
            if (h.ci[e] != h.cj[e] || (coord >= 0 && coord != h.ci[e])) coupled = true;
            coord = h.ci[e];
        }
        for (int64_t j = 0; j < n && !coupled; j++)
            if (q[j] != 0.0) { if (coord >= 0 && coord != j) coupled = true; coord = j; }
    }
    if (k >= 1 && coupled && (double)c->m * (double)n * (double)n * 8.0 > c->stream_limit) {
        // ---- streaming upload: the row-major copy of all constraint matrices would not fit the budget (16 GB; BASELINE.json
        // configs[4] is 137 GB).  The function goes to the device now, straight into the block-major fragment layout of the
        // dense path (cd_dense.h); its entries are not kept on the host.  Such a problem runs on the dense path only.
        const int64_t n16 = c->n16, m1 = c->m + 1;
        if ((double)m1 * (double)n16 * (double)n16 * 8.0 > 200e9)
            return fail(c, QCQPMI_EUNSUPPORTED, "set_quad: %lld matrices of %lld x %lld exceed the 200 GB kept for packed matrices", (long long)m1, (long long)n, (long long)n);
        HIPCHK(c, hipSetDevice(c->device));
        int rcs;
        if (!c->dn_gp_pre && (rcs = dev_alloc(c, &c->dn_gp_pre, (size_t)m1 * n16 * n16, false))) return rcs;
        if (!c->dn_tmp && (rcs = dev_alloc(c, &c->dn_tmp, (size_t)n * n, false))) return rcs;
        std::vector<double> dense;
        const double *src = vals;
        if (format != QCQPMI_FMT_DENSE) {
            dense.assign((size_t)n * n, 0.0);
            for (size_t e = 0; e < h.cv.size(); e++) dense[(size_t)h.ci[e] * n + h.cj[e]] += h.cv[e];
            src = dense.data();
        }
        HIPCHK(c, hipMemcpyAsync(c->dn_tmp, src, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        const int64_t total = n16 * n16;
        // (dense_pack_kernel addresses function k at gP + (k - 1) n n: hand it a base that puts the staging buffer there)
        hipLaunchKernelGGL(dense_pack_kernel, dim3((unsigned)((total + 255) / 256), 1), dim3(256), 0, c->stream, (const double *)nullptr,
                           (const double *)(c->dn_tmp - (k - 1) * n * n), c->dn_gp_pre, n, n16, (int)m1, (int)k);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, spin_sync(c->stream));      // the staging buffer and the caller's array are free again
        std::vector<int>().swap(h.ci); std::vector<int>().swap(h.cj); std::vector<double>().swap(h.cv);
        h.streamed = true;
    }
    h.q.assign(q, q + n);
    h.r = r; h.relop = relop; h.set = true;
    return 0;
}

int qcqpmi_set_quad_generated(qcqpmi_ctx *c, int64_t k, uint64_t seed, double scale, double qscale,
                              double diag_add, double r, int relop) {
    if (!c) return QCQPMI_EINVAL;
    if (c->finalized) return fail(c, QCQPMI_ESTATE, "set_quad_generated after finalize");
    if (k < 0 || k > c->m) return fail(c, QCQPMI_EINVAL, "set_quad_generated: bad k");
    if ((k == 0) != (relop == 0) || relop < 0 || relop > 2)
        return fail(c, QCQPMI_EINVAL, "set_quad_generated: relop %d invalid for function %lld", relop, (long long)k);
    HostQuad &h = c->quads[(size_t)k];
    h = HostQuad();
    h.gen = true; h.gseed = seed; h.gscale = scale; h.gqscale = qscale; h.gdiag = diag_add;
    h.r = r; h.relop = relop; h.set = true;
    // the linear term comes back to the host (n values): the generic finalize code reads it there
    HIPCHK(c, hipSetDevice(c->device));
    double *dq = nullptr;
    HIPCHK(c, hipMalloc((void **)&dq, (size_t)c->n * sizeof(double)));
    hipLaunchKernelGGL(dense_gen_q_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, dq, c->n, (int)k, seed, qscale);
    h.q.resize((size_t)c->n);
    hipError_t e = hipMemcpyAsync(h.q.data(), dq, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = spin_sync(c->stream);
    (void)hipFree(dq);
    if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "set_quad_generated: %s", hipGetErrorString(e));
    return 0;
}

int qcqpmi_finalize(qcqpmi_ctx *c) {
    if (!c) return QCQPMI_EINVAL;
    if (c->finalized) return 0;
    for (size_t k = 0; k < c->quads.size(); k++)
        if (!c->quads[k].set) return fail(c, QCQPMI_ESTATE, "finalize: function %zu was never set", k);
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = c->n, n16 = c->n16, m = c->m;
    DevProblem &dp = c->dp;
    dp.n = n; dp.n16 = n16; dp.NB = n16 / 16; dp.KS = n16 / 4; dp.m = m;
    // ---- objective: dense padded + MFMA-packed
    {
        const HostQuad &h = c->quads[0];
        std::vector<double> P((size_t)n16 * n16, 0.0), q((size_t)n16, 0.0);
        for (size_t e = 0; e < h.cv.size(); e++) P[(size_t)h.ci[e] * n16 + h.cj[e]] += h.cv[e];
        for (int64_t j = 0; j < n; j++) q[j] = h.q[j];
        int rc;
        if (h.gen) {   // generated on the device, mirrored to the host for the diagonal classification below
            double *tmp = nullptr;
            if ((rc = dev_alloc(c, &tmp, (size_t)n16 * n16, false))) return rc;
            hipLaunchKernelGGL(dense_gen_rowmajor_kernel, dim3((unsigned)((n16 * n16 + 255) / 256)), dim3(256), 0, c->stream,
                               tmp, n, n16, 0, h.gseed, h.gscale, h.gdiag);
            hipError_t e = hipMemcpyAsync(P.data(), tmp, P.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = spin_sync(c->stream);
            (void)hipFree(tmp);
            if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "finalize: generated objective: %s", hipGetErrorString(e));
        }
        {
            bool dg = true;
            for (size_t e = 0; e < h.cv.size() && dg; e++) dg = h.ci[e] == h.cj[e];
            c->p0_diag = dg && !h.gen;
            c->p0_diag_host.assign((size_t)n, 0.0);
            for (int64_t i = 0; i < n; i++) c->p0_diag_host[(size_t)i] = P[(size_t)i * n16 + i];
        }
        if ((rc = prob_upload(c, &dp.P0, P))) return rc;
        if ((rc = prob_upload(c, &dp.q0, q))) return rc;
        std::vector<double> rcp2d((size_t)n16, 0.0);
        for (int64_t i = 0; i < n; i++) {
            const double d = P[(size_t)i * n16 + i];
            if (d != 0.0) rcp2d[i] = 1.0 / (2.0 * d);
        }
        if ((rc = prob_upload(c, &dp.rcp2d, rcp2d))) return rc;
        {
            bool allpos = true, allzero = true;
            for (int64_t i = 0; i < n; i++) {
                const double d = P[(size_t)i * n16 + i];
                allpos = allpos && d > 0.0;
                allzero = allzero && d == 0.0;
            }
            c->objclass = allpos ? 1 : (allzero ? 2 : 0);
            double fb = fabs(h.r);
            for (int64_t i = 0; i < n; i++) {
                fb += fabs(q[(size_t)i]);
                for (int64_t j = 0; j < n; j++) fb += fabs(P[(size_t)i * n16 + j]);
            }
            c->fbound = fb;
        }
        double *ap = nullptr;
        if ((rc = dev_alloc(c, &ap, (size_t)n16 * n16, false))) return rc;
        c->prob_allocs.push_back(ap);
        int64_t total = n16 * n16;
        hipLaunchKernelGGL(pack_A_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                           dp.P0, ap, n16, dp.KS);
        HIPCHK(c, hipGetLastError());
        double *ap2 = nullptr;
        if ((rc = dev_alloc(c, &ap2, (size_t)n16 * n16, false))) return rc;
        c->prob_allocs.push_back(ap2);
        hipLaunchKernelGGL(pack_A2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                           dp.P0, ap2, n16, dp.KS);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, spin_sync(c->stream));  // P, q vectors go out of scope
        dp.Apack = ap;
        dp.Apack2 = ap2;
        dp.r0 = h.r;
    }
    // ---- constraints: separable iff one diagonal entry / one linear entry on the same coordinate
    bool sep = true;
    std::vector<int> coord_of((size_t)m, -1);
    for (int64_t k = 1; k <= m && sep; k++) {
        const HostQuad &h = c->quads[(size_t)k];
        int coord = -1;
        if (h.gen || h.streamed) { sep = false; break; }
        for (size_t e = 0; e < h.cv.size(); e++) {
            if (h.ci[e] != h.cj[e]) { sep = false; break; }
            if (coord >= 0 && coord != h.ci[e]) { sep = false; break; }
            coord = h.ci[e];
        }
        for (int64_t j = 0; j < n && sep; j++)
            if (h.q[j] != 0.0) { if (coord >= 0 && coord != (int)j) sep = false; coord = (int)j; }
        if (coord < 0) sep = false;  // constant constraint
        coord_of[(size_t)k - 1] = coord;
    }
    if (sep) {
        // the per-lane one-variable solver holds at most 4 constraints per coordinate (onevar.h); problems with more go
        // through the paths for general constraints below, which take any number (utilities.py:241-255 takes any m)
        std::vector<int> per((size_t)n, 0);
        for (int64_t k = 0; k < m && sep; k++) sep = ++per[(size_t)coord_of[k]] <= 4;
    }
    c->sep = sep;
    dp.sep = sep ? 1 : 0;
    int rc;
    if (sep) {
        std::vector<int> cptr((size_t)n16 + 1, 0);
        for (int64_t k = 0; k < m; k++) cptr[(size_t)coord_of[k] + 1]++;
        int maxc = 0;
        for (int64_t i = 0; i < n16; i++) { maxc = std::max(maxc, cptr[i + 1]); cptr[i + 1] += cptr[i]; }
        std::vector<double> cp((size_t)m), cq((size_t)m), cr((size_t)m);
        std::vector<int> crel((size_t)m), cidx((size_t)m), fill(cptr.begin(), cptr.end() - 1);
        for (int64_t k = 0; k < m; k++) {  // constraint order preserved inside a coordinate's list
            const HostQuad &h = c->quads[(size_t)k + 1];
            int i = coord_of[k], e = fill[i]++;
            double p = 0.0;
            for (size_t t = 0; t < h.cv.size(); t++) p += h.cv[t];
            cp[e] = p; cq[e] = h.q[i]; cr[e] = h.r; crel[e] = h.relop; cidx[e] = (int)k + 1;
        }
        c->maxc = maxc; dp.maxc = maxc;
        // constraint classes: coordinates whose (p, q, r, relop) lists are bit-identical
        {
            std::vector<int> cls((size_t)n16, 0), krep;
            for (int64_t i = 0; i < n16; i++) {
                int found = -1;
                const int len = cptr[i + 1] - cptr[i];
                for (size_t k = 0; k < krep.size() && found < 0; k++) {
                    const int j = krep[k];
                    if (cptr[j + 1] - cptr[j] != len) continue;
                    bool same = true;
                    for (int e = 0; e < len && same; e++) {
                        const int a = cptr[i] + e, b = cptr[j] + e;
                        same = memcmp(&cp[a], &cp[b], 8) == 0 && memcmp(&cq[a], &cq[b], 8) == 0 &&
                               memcmp(&cr[a], &cr[b], 8) == 0 && crel[a] == crel[b];
                    }
                    if (same) found = (int)k;
                }
                if (found < 0) {
                    krep.push_back((int)i);
                    found = (int)krep.size() - 1;
                }
                cls[i] = found;
            }
            dp.K = (int)krep.size();
            c->K = dp.K;
            {
                std::vector<char> seen(krep.size(), 0);
                int kr = 0;
                for (int64_t i = 0; i < n; i++) if (!seen[(size_t)cls[i]]) { seen[(size_t)cls[i]] = 1; kr++; }
                c->Kreal = kr;
            }
            c->symcls = false;
            if (c->Kreal == 1 && maxc == 1) {        // (Kreal: the padded coordinates of n16 form a class of their own)
                const int e0 = cptr[krep[0]];
                c->symcls = cq[e0] == 0.0 && crel[e0] == RELOP_EQ && cp[e0] != 0.0;
            }
            if ((rc = prob_upload(c, &dp.krep, krep))) return rc;
            if ((rc = prob_upload(c, &dp.cls, cls))) return rc;
        }
        if ((rc = prob_upload(c, &dp.cptr, cptr))) return rc;
        if ((rc = prob_upload(c, &dp.cp, cp))) return rc;
        if ((rc = prob_upload(c, &dp.cq, cq))) return rc;
        if ((rc = prob_upload(c, &dp.cr, cr))) return rc;
        if ((rc = prob_upload(c, &dp.crel, crel))) return rc;
        if ((rc = prob_upload(c, &dp.cidx, cidx))) return rc;
    } else {
        std::vector<int64_t> gptr((size_t)m + 1, 0);
        std::vector<int> gi, gj, grel((size_t)m);
        std::vector<double> gv, gq((size_t)m * n16, 0.0), gr((size_t)m);
        for (int64_t k = 0; k < m; k++) {
            const HostQuad &h = c->quads[(size_t)k + 1];
            gi.insert(gi.end(), h.ci.begin(), h.ci.end());
            gj.insert(gj.end(), h.cj.begin(), h.cj.end());
            gv.insert(gv.end(), h.cv.begin(), h.cv.end());
            gptr[(size_t)k + 1] = (int64_t)gv.size();
            for (int64_t j = 0; j < n; j++) gq[(size_t)k * n16 + j] = h.q[j];
            gr[k] = h.r; grel[k] = h.relop;
        }
        bool all_gen = true;
        for (int64_t k = 1; k <= m; k++) all_gen = all_gen && c->quads[(size_t)k].gen;
        c->dn_force = false;
        bool any_streamed = false;
        for (int64_t k = 0; k <= m; k++) { c->dn_force = c->dn_force || c->quads[(size_t)k].gen || c->quads[(size_t)k].streamed; any_streamed = any_streamed || c->quads[(size_t)k].streamed; }
        if (!all_gen && !any_streamed && (double)m * (double)n * (double)n * 8.0 <= 16e9) {
            std::vector<double> gP((size_t)m * n * n, 0.0);
            for (int64_t k = 0; k < m; k++) {
                const HostQuad &h = c->quads[(size_t)k + 1];
                for (size_t e = 0; e < h.cv.size(); e++) gP[((size_t)k * n + h.ci[e]) * n + h.cj[e]] += h.cv[e];
            }
            const double *tmp = nullptr;
            if ((rc = prob_upload(c, &tmp, gP))) return rc;
            c->d_gP = const_cast<double *>(tmp);
            HIPCHK(c, spin_sync(c->stream));
        }
        if ((rc = prob_upload(c, &dp.gptr, gptr))) return rc;
        if ((rc = prob_upload(c, &dp.gi, gi))) return rc;
        if ((rc = prob_upload(c, &dp.gj, gj))) return rc;
        if ((rc = prob_upload(c, &dp.gv, gv))) return rc;
        if ((rc = prob_upload(c, &dp.gq, gq))) return rc;
        if ((rc = prob_upload(c, &dp.gr, gr))) return rc;
        if ((rc = prob_upload(c, &dp.grel, grel))) return rc;
        if ((c->d_gP || all_gen || any_streamed) && (double)(m + 1) * (double)n16 * (double)n16 * 8.0 <= 200e9 && (rc = dense_build(c, gq, gr, grel))) return rc;
        if (c->dn_force && !c->dn_Gpack) return fail(c, QCQPMI_EUNSUPPORTED, "generated functions need the dense path (matrices exceed 200 GB, or generated and uploaded coupled constraints are mixed beyond 16 GB)");
    }
    if ((rc = dev_alloc(c, &c->d_best_idx, 2))) return rc;
    if ((rc = dev_alloc(c, &c->d_best_key, 2))) return rc;
    HIPCHK(c, spin_sync(c->stream));
    // host copies are no longer needed
    for (auto &h : c->quads) { std::vector<int>().swap(h.ci); std::vector<int>().swap(h.cj); std::vector<double>().swap(h.cv); }
    c->finalized = true;
    return 0;
}

int qcqpmi_sdr_solve_unitdiag(qcqpmi_ctx *c, const double *C, int64_t N, double *V, int max_sweeps, double tol,
                              double *hist, int *sweeps_done) {
    if (!c) return QCQPMI_EINVAL;
    if (!C || !V || !hist || !sweeps_done || N < 2 || max_sweeps < 0 || !(tol >= 0.0))
        return fail(c, QCQPMI_EINVAL, "sdr_solve_unitdiag: bad arguments");
    if (N > SDR_NMAX) return fail(c, QCQPMI_EUNSUPPORTED, "sdr_solve_unitdiag: N = %lld exceeds %d", (long long)N, SDR_NMAX);
    HIPCHK(c, hipSetDevice(c->device));
    double *dC = nullptr, *dV = nullptr, *dh = nullptr;
    int *ds = nullptr;
    SdrWork *dw = nullptr;
    int rc = 0;
    if ((rc = dev_alloc(c, &dC, (size_t)N * N, false))) return rc;
    if (!rc) rc = dev_alloc(c, &dV, (size_t)N * SDR_K, false);
    if (!rc) rc = dev_alloc(c, &dh, (size_t)max_sweeps + 2);
    if (!rc) rc = dev_alloc(c, &ds, 1);
    if (!rc) rc = dev_alloc(c, &dw, 1);      // zeroed: arrival counter, abort flag
    hipError_t e = hipSuccess;
    int sw = 0;
    if (!rc) {
        e = hipMemcpyAsync(dC, C, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dV, V, (size_t)N * SDR_K * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            // 64 single-wave workgroups that synchronise through memory: cooperative launch = co-residency guaranteed
            const double *aC = dC;
            int aN = (int)N;
            void *args[] = {(void *)&aC, (void *)&dV, (void *)&aN, (void *)&max_sweeps, (void *)&tol, (void *)&dh, (void *)&ds, (void *)&dw};
            e = hipLaunchCooperativeKernel((const void *)sdr_mixing_kernel, dim3(SDR_K), dim3(64), args,
                                           (unsigned)(2 * (size_t)N * sizeof(double)), c->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(V, dV, (size_t)N * SDR_K * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hist, dh, ((size_t)max_sweeps + 2) * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&sw, ds, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = spin_sync(c->stream);
    }
    void *ptrs[] = {dC, dV, dh, ds, dw};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "sdr_solve_unitdiag: %s", hipGetErrorString(e));
    if (sw < 0) return fail(c, QCQPMI_EHIP, "sdr_solve_unitdiag: the workgroups lost step (spin limit reached)");
    *sweeps_done = sw;
    return 0;
}

int qcqpmi_is_separable(const qcqpmi_ctx *c) { return (c && c->finalized && c->sep) ? 1 : 0; }

// -------------------------------------------------------------------------------- population

int qcqpmi_pop_upload(qcqpmi_ctx *c, const double *X, int64_t R) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!X) return fail(c, QCQPMI_EINVAL, "pop_upload: X is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = pop_reserve(c, R))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_stage, X, (size_t)R * c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int64_t total = c->Rpad * c->n16;
    hipLaunchKernelGGL(to_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_stage, c->X, c->n, c->n16, R, c->Rpad);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

int qcqpmi_pop_download(qcqpmi_ctx *c, double *X, int64_t R) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!X || R <= 0 || R > c->R) return fail(c, QCQPMI_EINVAL, "pop_download: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int64_t total = R * c->n;
    hipLaunchKernelGGL(from_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       c->X, c->d_stage, c->n, c->n16, R);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(X, c->d_stage, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

int64_t qcqpmi_pop_size(const qcqpmi_ctx *c) { return c ? c->R : 0; }

int qcqpmi_pop_randn(qcqpmi_ctx *c, int64_t R, uint64_t seed, uint64_t first_index) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = pop_reserve(c, R))) return rc;
    int64_t total = c->Rpad * c->n16;
    hipLaunchKernelGGL(randn_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       c->X, c->n, c->n16, R, c->Rpad, seed, first_index);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int qcqpmi_pop_sdr_sample(qcqpmi_ctx *c, const double *mu, const double *F, int64_t S, uint64_t seed,
                          uint64_t first_index, const double *Xi) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    // mu == NULL && F == NULL: the factor and mean of the previous call are still resident (packed for the matrix cores)
    const bool reuse = !mu && !F;
    if (!reuse && (!mu || !F)) return fail(c, QCQPMI_EINVAL, "pop_sdr_sample: mu / F missing (both NULL = reuse the previous pair)");
    if (reuse && !c->sdr_factor_resident) return fail(c, QCQPMI_ESTATE, "pop_sdr_sample: no resident factor to reuse");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n = c->n, n16 = c->n16;
    if (!c->d_Fpack) {
        if ((rc = dev_alloc(c, &c->d_Fpack, (size_t)n16 * n16))) return rc;
        if ((rc = dev_alloc(c, &c->d_Frow, (size_t)n16 * n16))) return rc;
        if ((rc = dev_alloc(c, &c->d_mu, (size_t)n16))) return rc;
    }
    if (!reuse) {
        HIPCHK(c, hipMemsetAsync(c->d_Frow, 0, (size_t)n16 * n16 * sizeof(double), c->stream));
        HIPCHK(c, hipMemcpy2DAsync(c->d_Frow, (size_t)n16 * sizeof(double), F, (size_t)n * sizeof(double),
                                   (size_t)n * sizeof(double), (size_t)n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_mu, mu, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        int64_t total = n16 * n16;
        hipLaunchKernelGGL(pack_A_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_Frow, c->d_Fpack, n16, c->dp.KS);
        c->sdr_factor_resident = true;
        // the 2D copy may pin the caller's pages and return with the DMA in flight; the caller (Engine.sdr_sample passes
        // temporaries) may free F / mu as soon as this call returns.  The reuse path (mu = F = NULL) stays asynchronous.
        HIPCHK(c, spin_sync(c->stream));
    }
    if ((rc = pop_reserve(c, S))) return rc;
    // the standard normals: caller-provided (host layout) or device Philox; the buffer is kept across calls
    if ((int64_t)c->Rpad * n16 > c->Xi_cap) {
        if (c->Xi) { HIPCHK(c, spin_sync(c->stream)); (void)hipFree(c->Xi); c->Xi = nullptr; c->Xi_cap = 0; }
        if ((rc = dev_alloc(c, &c->Xi, (size_t)c->Rpad * n16, false))) return rc;
        c->Xi_cap = (int64_t)c->Rpad * n16;
    }
    int64_t tot2 = c->Rpad * n16;
    if (Xi) {
        HIPCHK(c, hipMemcpyAsync(c->d_stage, Xi, (size_t)S * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(to_tiles_kernel, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_stage, c->Xi, n, n16, S, c->Rpad);
    } else {
        hipLaunchKernelGGL(randn_tiles_kernel, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, c->stream,
                           c->Xi, n, n16, S, c->Rpad, seed, first_index);
    }
    tic(c, 3);
    if (c->dp.NB >= 8 && c->Rpad / 16 >= 8) {
        // x = mu + F xi as one LDS-tiled MFMA GEMM (8 row blocks of F x 8 tiles of samples per workgroup)
        DenseProdArgs pa;
        pa.D.Gpack = c->d_Fpack; pa.D.q = c->d_mu; pa.D.qT = nullptr; pa.D.r = nullptr; pa.D.relop = nullptr;
        pa.D.n = n; pa.D.n16 = n16; pa.D.NB = (int)c->dp.NB; pa.D.KS = (int)c->dp.KS; pa.D.m1 = 1; pa.D.m1p = 64;
        pa.X = c->Xi; pa.ntiles = (int)(c->Rpad / 16); pa.b = 0; pa.zs = 1; pa.G = c->X; pa.F = nullptr; pa.Rpad = c->Rpad;
        pa.tile_on = nullptr; pa.hole = -1; pa.ch_only = -1; pa.zplane = 0;
        auto kg = dense_products_kernel<3>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_BYTES));
        hipLaunchKernelGGL(kg, dim3((unsigned)((pa.D.NB + DP_FG - 1) / DP_FG), (unsigned)((pa.ntiles + DP_TG - 1) / DP_TG), 1),
                           dim3(256), DP_LDS_BYTES, c->stream, pa);
    } else {
        hipLaunchKernelGGL(affine_tiles_kernel, dim3((unsigned)(c->Rpad / 16)), dim3(256), 0, c->stream,
                           c->d_Fpack, c->d_mu, c->Xi, c->X, n, n16, c->dp.NB, c->dp.KS);
    }
    toc(c, 3);
    HIPCHK(c, hipGetLastError());
    // F / mu were synchronised above; a caller-provided Xi went through a pageable 1D copy (host-synchronous staging).  The
    // samples are in stream order for whatever comes next (evaluation, download); no synchronisation here.
    return 0;
}

// --------------------------------------------------------------------------------- evaluation

int qcqpmi_pop_eval(qcqpmi_ctx *c, double *f0, double *maxviol, double *F) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = launch_eval(c, F != nullptr))) return rc;
    if (f0) HIPCHK(c, hipMemcpyAsync(f0, c->d_f0, (size_t)c->R * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (maxviol) HIPCHK(c, hipMemcpyAsync(maxviol, c->d_mv, (size_t)c->R * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (F)
        HIPCHK(c, hipMemcpy2DAsync(F, (size_t)c->R * sizeof(double), c->d_F, (size_t)c->Rpad * sizeof(double),
                                   (size_t)c->R * sizeof(double), (size_t)(c->m + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

// ---- suggest(SDR) for S samples in one call (qcqp.py:396-401: x = multivariate_normal(mu, Sigma), f0.eval(x), max(violations(x));
// SURVEY section 8b's qcqpmi_sdr_sample_eval).  The samples are drawn (x = mu + F xi, normals by keyed Philox: sample first_index + s
// is the same point whatever S and the chunking) and evaluated chunk by chunk on the stream: the normals and the points of a chunk
// live in two buffers every chunk reuses (<= 32 MB each, sized to fit the 256 MB Infinity Cache between the sampler's GEMM and the
// evaluation's GEMM; not measured separately: at configs[2]'s size the round trip of X is 1 % of the two GEMMs either way); a
// population of S points is never laid out in HBM.  X_opt == NULL: the points are not kept at all -- the caller
// re-draws the winner from its index (S = 1, the same first_index + s).  Afterwards the resident population is the LAST chunk.
int qcqpmi_sdr_sample_eval(qcqpmi_ctx *c, const double *mu, const double *F, int64_t S, uint64_t seed, uint64_t first_index,
                           double *X_opt, double *f0, double *maxviol) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (S < 1 || !f0 || !maxviol) return fail(c, QCQPMI_EINVAL, "sdr_sample_eval: S < 1 or f0 / maxviol missing");
    int64_t chunk = ((int64_t)32 << 20) / (c->n16 * (int64_t)sizeof(double));
    chunk = chunk / 128 * 128;                       // whole groups of 8 tiles of 16 samples (the LDS-tiled GEMMs' unit)
    if (chunk < 128) chunk = 128;
    if (const char *ev = getenv("QCQPMI_SDR_CHUNK")) { const int64_t v = atoll(ev); if (v >= 16) chunk = v / 16 * 16; }     // tests: several chunks at small sizes
    for (int64_t off = 0; off < S; off += chunk) {
        const int64_t cnt = S - off < chunk ? S - off : chunk;
        if ((rc = qcqpmi_pop_sdr_sample(c, off == 0 ? mu : nullptr, off == 0 ? F : nullptr, cnt, seed, first_index + (uint64_t)off, nullptr))) return rc;
        if ((rc = launch_eval(c, false))) return rc;
        HIPCHK(c, hipMemcpyAsync(f0 + off, c->d_f0, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(maxviol + off, c->d_mv, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (X_opt) {
            if ((rc = qcqpmi_pop_download(c, X_opt + off * c->n, cnt))) return rc;
        }
    }
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

int qcqpmi_pop_weighted_product(qcqpmi_ctx *c, const double *w, double *Y) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!w || !Y) return fail(c, QCQPMI_EINVAL, "pop_weighted_product: w / Y missing");
    HIPCHK(c, hipSetDevice(c->device));
    DenseProblem D = dense_problem(c);
    if (!c->dn_Gpack) {
        // separable constraints: only the objective has a matrix (the packed P0 of the separable path: the same fragment
        // layout with one function); the weights of the constraints are the caller's business (elementwise operators)
        D.Gpack = c->dp.Apack; D.m1 = 1; D.m1p = 64; D.q = nullptr; D.qT = nullptr; D.r = nullptr; D.relop = nullptr;
    }
    const int64_t n16 = c->n16, ntiles = c->Rpad / 16;
    if (!c->d_wS) {
        if ((rc = dev_alloc(c, &c->d_ww, (size_t)D.m1, false))) return rc;
        if ((rc = dev_alloc(c, &c->d_wS, (size_t)n16 * n16, false))) return rc;
        if ((rc = dev_alloc(c, &c->d_wz, (size_t)n16))) return rc;   // zero offset vector of the affine map
    }
    if (c->Rpad * n16 > c->wY_cap) {
        HIPCHK(c, spin_sync(c->stream));
        if (c->d_wY) (void)hipFree(c->d_wY);
        c->d_wY = nullptr;
        if ((rc = dev_alloc(c, &c->d_wY, (size_t)c->Rpad * n16, false))) return rc;
        c->wY_cap = c->Rpad * n16;
    }
    double *dw = c->d_ww, *dS = c->d_wS, *dY = c->d_wY, *dz = c->d_wz;
    if ((rc = pin_reserve(c, (size_t)(c->R * c->n) * sizeof(double) + (size_t)D.m1 * sizeof(double)))) return rc;
    char *hw = c->h_pin + (size_t)(c->R * c->n) * sizeof(double);       // the weights go up through pinned memory too
    memcpy(hw, w, (size_t)D.m1 * sizeof(double));
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(dw, hw, (size_t)D.m1 * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(dense_wsum_pack_kernel, dim3((unsigned)((n16 * n16 + 255) / 256)), dim3(256), 0, c->stream, D, (const double *)dw, dS);
            DenseProdArgs pa;
            pa.D = D; pa.D.Gpack = dS; pa.D.q = dz; pa.D.m1 = 1; pa.D.m1p = 64;
            pa.X = c->X; pa.ntiles = (int)ntiles; pa.b = 0; pa.zs = 1; pa.G = dY; pa.F = nullptr; pa.Rpad = c->Rpad;
            pa.tile_on = nullptr; pa.hole = -1; pa.ch_only = -1; pa.zplane = 0;
            auto kg = dense_products_kernel<3>;
            e = hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_BYTES);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(kg, dim3((unsigned)((D.NB + DP_FG - 1) / DP_FG), (unsigned)((ntiles + DP_TG - 1) / DP_TG), 1),
                                   dim3(256), DP_LDS_BYTES, c->stream, pa);
                const int64_t total = c->R * c->n;
                hipLaunchKernelGGL(from_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                                   (const double *)dY, c->d_stage, c->n, c->n16, c->R);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(c->h_pin, c->d_stage, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            }
        }
        if (e == hipSuccess) e = spin_sync(c->stream);
        if (e == hipSuccess) memcpy(Y, c->h_pin, (size_t)(c->R * c->n) * sizeof(double));
    }
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "pop_weighted_product: %s", hipGetErrorString(e));
    return 0;
}

int qcqpmi_get_linear(qcqpmi_ctx *c, int64_t k, double *q, double *r, int *relop) {
    if (!c) return QCQPMI_EINVAL;
    if (k < 0 || k > c->m || !c->quads[(size_t)k].set) return fail(c, QCQPMI_EINVAL, "get_linear: bad k");
    const HostQuad &h = c->quads[(size_t)k];
    if (q) memcpy(q, h.q.data(), (size_t)c->n * sizeof(double));
    if (r) *r = h.r;
    if (relop) *relop = h.relop;
    return 0;
}

int qcqpmi_weighted_matrix(qcqpmi_ctx *c, const double *w, double *S) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!w || !S) return fail(c, QCQPMI_EINVAL, "weighted_matrix: w / S missing");
    if (!c->dn_Gpack) return fail(c, QCQPMI_EUNSUPPORTED, "weighted_matrix needs the packed dense matrices");
    HIPCHK(c, hipSetDevice(c->device));
    const DenseProblem D = dense_problem(c);
    const int64_t n = c->n, n16 = c->n16;
    double *dw = nullptr, *dS = nullptr, *dU = nullptr;
    rc = dev_alloc(c, &dw, (size_t)D.m1, false);
    if (!rc) rc = dev_alloc(c, &dS, (size_t)n16 * n16, false);
    if (!rc) rc = dev_alloc(c, &dU, (size_t)n * n, false);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(dw, w, (size_t)D.m1 * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(dense_wsum_pack_kernel, dim3((unsigned)((n16 * n16 + 255) / 256)), dim3(256), 0, c->stream, D, (const double *)dw, dS);
            hipLaunchKernelGGL(dense_unpack_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, c->stream, (const double *)dS, dU, n, (int)(n16 / 4));
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(S, dU, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = spin_sync(c->stream);
    }
    void *ptrs[] = {dw, dS, dU};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "weighted_matrix: %s", hipGetErrorString(e));
    return 0;
}

int qcqpmi_pop_eval_parts(qcqpmi_ctx *c, double *quad, double *lin) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!quad || !lin) return fail(c, QCQPMI_EINVAL, "pop_eval_parts: outputs missing");
    if (!c->dn_Gpack) return fail(c, QCQPMI_EUNSUPPORTED, "pop_eval_parts needs the packed dense matrices (constraints that couple coordinates)");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = eval_parts_dense(c))) return rc;
    const int64_t m1 = c->m + 1;
    const double *dlin = c->d_F + (int64_t)c->eval_zs * m1 * c->Rpad;
    // the two planes ([m1][Rpad], contiguous) through pinned memory in one copy each, the padding columns dropped on the host
    const size_t plane = (size_t)m1 * c->Rpad * sizeof(double);
    if ((rc = pin_reserve(c, 2 * plane))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_F, plane, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_pin + plane, dlin, plane, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    const double *hq = (const double *)c->h_pin, *hl = (const double *)(c->h_pin + plane);
    for (int64_t k = 0; k < m1; k++) {
        memcpy(quad + k * c->R, hq + k * c->Rpad, (size_t)c->R * sizeof(double));
        memcpy(lin + k * c->R, hl + k * c->Rpad, (size_t)c->R * sizeof(double));
    }
    c->evaluated = false;    // plane 0 no longer holds the full function values
    return 0;
}

int qcqpmi_eval_batch(qcqpmi_ctx *c, const double *X, int64_t S, double *f0, double *maxviol, double *F) {
    int rc = qcqpmi_pop_upload(c, X, S);
    if (rc) return rc;
    return qcqpmi_pop_eval(c, f0, maxviol, F);
}

// ------------------------------------------------------------------------- coordinate descent

// stage 0: the whole run.  Stages 1-3 split it for callers that overlap the preparation of the NEXT population with the
// phase-2 kernel of the current one (two contexts = two streams): 1 = phase 1 + evaluation + gate (asynchronous),
// 2 = launch of phase 2 (asynchronous), 3 = results (blocking).  Paths without a pipelined phase-2 kernel run everything
// in stage 3.
int qcqpmi_cd_run_stage(qcqpmi_ctx *c, int stage, int phase1, int64_t num_iters, double viol_tol, double tol,
                        uint64_t seed, uint64_t first_index, int64_t *sweeps1, int64_t *sweeps2,
                        int64_t *visits2, int64_t *accepted2, uint8_t *ran_phase2, double *f0,
                        double *maxviol) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (stage < 0 || stage > 3) return fail(c, QCQPMI_EINVAL, "cd_run_stage: stage %d", stage);
    if (stage >= 2 && c->cd_stage != stage - 1) return fail(c, QCQPMI_ESTATE, "cd_run_stage: stage %d without stage %d", stage, stage - 1);
    const bool whole = dense_on(c) || !c->sep;      // these paths are not split
    if (whole) {
        if (stage == 1 || stage == 2) { c->cd_stage = stage; return 0; }
        c->cd_stage = 0;
        c->last_cd2_kernel = (dense_on(c) && !c->cd_ref_order) ? dense_chain_name(c) : "cd_general_kernel";
        if (c->cd_ref_order && !c->d_gP)
            return fail(c, QCQPMI_EUNSUPPORTED, "reference-order coordinate descent needs the row-major constraint matrices "
                        "(uploaded functions, m n^2 <= 2e9 entries); device-generated functions only exist packed");
        if (dense_on(c) && !c->cd_ref_order) return cd_run_dense(c, phase1, num_iters, viol_tol, tol, seed, first_index, sweeps1, sweeps2, visits2,
                                             accepted2, ran_phase2, f0, maxviol);
        return cd_run_general(c, phase1, num_iters, viol_tol, tol, seed, first_index, sweeps1, sweeps2, visits2,
                              accepted2, ran_phase2, f0, maxviol);
    }
    if (num_iters < 0 || !(tol > 0.0)) return fail(c, QCQPMI_EINVAL, "cd_run: bad num_iters / tol");
    HIPCHK(c, hipSetDevice(c->device));
    CdArgs a;
    a.P = c->dp; a.X = c->X; a.R = c->R; a.f0cur = c->d_f0; a.slack = c->d_mv;
    a.num_iters = num_iters; a.viol_tol = viol_tol; a.tol = tol; a.seed = seed; a.first_index = first_index;
    a.visits = c->d_visits; a.accepted = c->d_acc; a.sweeps = c->d_sweeps; a.status = c->d_status;
    a.flag = c->d_flag;
    a.prof = nullptr; a.f0out = nullptr; a.mvout = nullptr;
    a.dbg = c->dbg;
    bool used_lds = false;
    // Everything below is enqueued on the context's stream without a host round trip: phase 1,
    // evaluation, the improve_coord_descent gate, phase 2, final evaluation, result copies.
    if (stage == 0 || stage == 1) {
        HIPCHK(c, hipMemsetAsync(c->d_sweeps1, 0, (size_t)c->Rpad * sizeof(int64_t), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_status1, 0, (size_t)c->Rpad * sizeof(int), c->stream));
        if (phase1) {
            CdArgs a1 = a;
            a1.sweeps = c->d_sweeps1;
            a1.status = c->d_status1;
            rc = (c->maxc <= 1) ? launch_cd<1>(c, a1, true, used_lds) : launch_cd<4>(c, a1, true, used_lds);
            if (rc) return rc;
        }
        // gate of improve_coord_descent (qcqp.py:189): phase 2 only if max violation < viol_tol.
        // The evaluation also provides the phase-2 slack (qcqp.py:157) and the running objective.
        if ((rc = launch_eval(c, false))) return rc;
        hipLaunchKernelGGL(gate_kernel, dim3((unsigned)((c->Rpad + 255) / 256)), dim3(256), 0, c->stream, c->d_mv,
                           c->d_status1, c->d_flag, c->R, c->Rpad, viol_tol);
        HIPCHK(c, hipGetLastError());
        c->q_prepared = false;
        if (cd_queue_applies(c, c->profile) && (rc = cd_queue_prepare(c))) return rc;     // queue reset + population published
        if (stage == 1) { c->cd_stage = 1; return 0; }
    }
    if (stage == 0 || stage == 2) {
        if (c->profile) {
            if (c->d_prof) { (void)hipFree(c->d_prof); c->d_prof = nullptr; }
            if ((rc = dev_alloc(c, &c->d_prof, (size_t)(c->Rpad / 16) * 16 + QCQPMI_TRACE_WORDS))) return rc;
            HIPCHK(c, hipMemsetAsync(c->d_prof + (size_t)(c->Rpad / 16) * 16, 0, QCQPMI_TRACE_WORDS * sizeof(long long), c->stream));
            a.prof = c->d_prof;
        }
        // the pipelined kernel leaves objective and max violation of the restarts it ran in place (tracked objective,
        // element-wise violations of the final tile): no evaluation pass afterwards
        bool used_rs = false;
        a.f0out = c->d_f0; a.mvout = c->d_mv;
        rc = (c->maxc <= 1) ? launch_cd<1>(c, a, false, used_lds, &used_rs) : launch_cd<4>(c, a, false, used_lds, &used_rs);
        if (rc) return rc;
        if (!used_rs && (rc = launch_eval(c, false))) return rc;
        if (stage == 2) { c->cd_stage = 2; return 0; }
    }
    c->cd_stage = 0;
    std::vector<int> st, st1;
    if ((rc = fetch_cd_outputs(c, sweeps1, sweeps2, visits2, accepted2, ran_phase2, f0, maxviol, st, st1))) return rc;
    if ((rc = cd_apply_status(c, st, st1, f0, maxviol, 0))) return rc;
    return 0;
}

int qcqpmi_cd_run(qcqpmi_ctx *c, int phase1, int64_t num_iters, double viol_tol, double tol,
                  uint64_t seed, uint64_t first_index, int64_t *sweeps1, int64_t *sweeps2,
                  int64_t *visits2, int64_t *accepted2, uint8_t *ran_phase2, double *f0,
                  double *maxviol) {
    return qcqpmi_cd_run_stage(c, 0, phase1, num_iters, viol_tol, tol, seed, first_index, sweeps1, sweeps2, visits2, accepted2,
                               ran_phase2, f0, maxviol);
}

int qcqpmi_cd_dense_block_step(qcqpmi_ctx *c, int phase, int64_t sweep, int64_t block, int coord_lo, int coord_hi, double viol_tol,
                               double tol, uint64_t seed, uint64_t first_index, const double *slack) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!dense_on(c)) return fail(c, QCQPMI_EUNSUPPORTED, "cd_dense_block_step: the problem does not take the dense-constraint path");
    c->last_cd2_kernel = dense_chain_name(c);
    return cd_dense_block_step(c, phase, sweep, block, coord_lo, coord_hi, viol_tol, tol, seed, first_index, slack);
}

// buffers of cd_life_kernel: the workgroups' X tiles, the staged diagonal blocks (packed once per problem), the watchdog word
static int cd_life2_reserve(qcqpmi_ctx *c, int nmw, int cus) {
    int rc;
    const size_t need = (size_t)cd_life2_max_wgs(nmw, cus, 1) * (size_t)c->n16 * 16;      // (two tiles per workgroup: half the workgroups, the same tiles)
    if (need > c->l2_scratch_cap) {
        if (c->l2_scratch) { HIPCHK(c, spin_sync(c->stream)); (void)hipFree(c->l2_scratch); c->l2_scratch = nullptr; c->l2_scratch_cap = 0; }
        if ((rc = dev_alloc(c, &c->l2_scratch, need))) return rc;
        c->l2_scratch_cap = need;
    }
    if (!c->l2_abort && (rc = dev_alloc(c, &c->l2_abort, 4))) return rc;
    if (!c->l2_cuslot && (rc = dev_alloc(c, &c->l2_cuslot, 4096))) return rc;
    if (!c->l2_D) {
        const size_t NB = (size_t)(c->n16 / 16);
        if ((rc = dev_alloc(c, &c->l2_D, NB * 256))) return rc;
        if ((rc = dev_alloc(c, &c->l2_S, NB * 48))) return rc;
        hipError_t e = (hipError_t)cd_life2_pack(c->dp, c->l2_D, c->l2_S, c->stream);
        if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "cd_life2_pack: %s", hipGetErrorString(e));
    }
    return 0;
}

// ---- lifecycle run: K populations of R restarts through ONE persistent slot-queue launch (cd_queue.h, CdLife)
int qcqpmi_cd_stream_run(qcqpmi_ctx *c, int64_t K, int64_t R, int generate, int phase1, int64_t num_iters, double viol_tol, double tol,
                         uint64_t seed, uint64_t seed_stride, uint64_t first_index, uint64_t first_stride, double select_tol,
                         int64_t *sweeps1, int64_t *sweeps2, int64_t *visits2, int64_t *accepted2, uint8_t *ran_phase2, double *f0,
                         double *maxviol, int64_t *best_index, double *best_f0, double *best_maxviol, double *best_x) {
    int rc = check_ready(c, generate ? false : true);
    if (rc) return rc;
    if (K < 1 || R < 1 || num_iters < 0 || !(tol > 0.0) || K * R >= (1LL << 30)) return fail(c, QCQPMI_EINVAL, "cd_stream_run: bad K / R / num_iters / tol");
    if (!generate && c->R != K * R) return fail(c, QCQPMI_EINVAL, "cd_stream_run: the resident population has %lld points, K R = %lld", (long long)c->R, (long long)(K * R));
    HIPCHK(c, hipSetDevice(c->device));
    // QCQPMI_STREAM_TIMING=1: host-side wall clock of the call's stages on stderr (set-up, kernel, fetch, status, best)
    const bool stt = getenv("QCQPMI_STREAM_TIMING") != nullptr;
    double stm[6] = {0, 0, 0, 0, 0, 0};
    auto stnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    stm[0] = stnow();
    // ---- which kernel (decided BEFORE the resident population is touched: a refused call leaves the context as it was)
    int nmw = 0, cs2 = 0, kind = 0;
    const int factor_rb = (c->lr_RB > 0 && c->life_version != 3 && cd_life2_factor_ok(c->dp, 16 * (int64_t)c->lr_RB)) ? c->lr_RB : 0;
    bool use2 = c->life_version != 1 && !c->force_generic && !(c->dbg & 64) && c->sep &&
                cd_life2_config(c->dp, c->Kreal, c->objclass, c->symcls, factor_rb, &nmw, &cs2, &kind);
    int cs = 0;
    const int queue_switch = c->cd_queue;
    c->cd_queue = 2;                                   // (qcqpmi_cd_queue chooses the phase-2 kernel of qcqpmi_cd_run; it does not apply here)
    const bool eligible = cd_queue_eligible(c, false) && (int)(c->n16 / 16) - 4 <= RQ_NSIMD * RQ_MAXU && c->n16 >= 48;
    c->cd_queue = queue_switch;
    // ONE lifecycle kernel (round 6): cd_life_kernel takes every shape the round-4 kernel (cd_phase2_qs_kernel<CS, lifecycle>, cd_queue.hip)
    // takes and more; the shape rule of round 5 (the round-4 kernel from n = 960 on for runs beyond 8192 restarts, where it is 12 % faster
    // on a FULL-rank objective: profiles/r05_life_vs_round4.md) is gone -- the headline family has a low-rank objective and runs the
    // factored instantiation, 1.3 x faster than either (profiles/r06_summary.md).  qcqpmi_cd_life_version(1) still launches the round-4
    // kernel: a debug switch, kept as the independent implementation the tests compare against.
    // factored objective (qcqpmi_cd_set_objective_factor): cd_life_kernel's products through Y = L^T X -- half the matrix work and less
    // per block interval, a positive diagonal (band / gen kinds), three multiplying waves per tile
    const bool lr = use2 && factor_rb > 0 && nmw == 3 && (kind == L2_KIND_BAND || kind == L2_KIND_GEN);
    if (lr) cs2 = 0;
    if (!use2) {
        if (!eligible || c->life_version != 1)
            return fail(c, QCQPMI_EUNSUPPORTED, "cd_stream_run: the lifecycle kernel needs separable constraints -- at most four classes of coordinates, at most two constraints "
                        "per coordinate --, a diagonal of P0 that is positive everywhere or zero everywhere, and 48 <= n <= 2304: use qcqpmi_cd_run per population");
        const int NBq = (int)(c->n16 / 16);
        cs = (c->dbg & 128) ? ((c->dbg >> 8) & 7) : 4;
        cs = cs > RQ_CSMAX ? RQ_CSMAX : cs;
        if (cs >= NBq) cs = 0;
        cs &= ~1;
        if (NBq - cs > RQ_NSIMD * RQ_MAXU || NBq < 3) return fail(c, QCQPMI_EUNSUPPORTED, "cd_stream_run: n = %lld outside the range of cd_phase2_qs_kernel (n <= 1024, at least 3 blocks of 16)", (long long)c->n);
    }
    if (generate && (rc = pop_reserve(c, K * R))) return rc;
    if (!c->d_qnext && (rc = dev_alloc(c, &c->d_qnext, 16))) return rc;
    if (!c->d_life) HIPCHK(c, hipMalloc((void **)&c->d_life, sizeof(CdLife)));
    HIPCHK(c, hipMemsetAsync(c->d_qnext, 0, 16 * sizeof(int), c->stream));
    CdLife L;
    L.on = 1; L.generate = generate ? 1 : 0; L.phase1 = phase1 ? 1 : 0; L.Rtotal = K * R; L.Rpop = R;
    L.seed = seed; L.seed_stride = seed_stride; L.first_index = first_index; L.first_stride = first_stride; L.viol_tol = viol_tol;
    L.sweeps1 = c->d_sweeps1; L.status1 = c->d_status1; L.ran2 = c->d_flag;
    L.prof = nullptr;
    if (c->profile) {      // qcqpmi_debug_profile: tick sums of the launch (qcqpmi_debug_life_profile)
        if (!c->d_life_prof) HIPCHK(c, hipMalloc((void **)&c->d_life_prof, 24 * sizeof(long long)));
        HIPCHK(c, hipMemsetAsync(c->d_life_prof, 0, 24 * sizeof(long long), c->stream));
        L.prof = c->d_life_prof;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_life, &L, sizeof(L), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, spin_sync(c->stream));       // (L lives on this stack frame)
    int cus = 0;
    HIPCHK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    stm[1] = stnow();
    if (use2) {
        if ((rc = cd_life2_reserve(c, nmw, cus))) return rc;
        if (!lr && (c->dbg & 128)) { const int k2 = (c->dbg >> 8) & 7; if (nmw == 3 && (k2 == 0 || k2 == 2 || k2 == 4) && k2 < (int)(c->n16 / 16)) cs2 = k2; }
        const int tiles = cd_life2_tiles(c->dp, nmw, cs2, K * R, cus, 0, lr ? 1 : 0);
        int64_t wgs = (K * R + 16 * tiles - 1) / (16 * tiles);
        const int maxw = cd_life2_max_wgs(nmw, cus, tiles);
        if (wgs > maxw) wgs = maxw;
        if (c->dbg & 1024) { const int lim = (c->dbg >> 12) & 1023; if (lim > 0 && wgs > lim) wgs = lim; }     // debug knob: workgroups of the launch
        CdLife2Args qa;
        qa.P = c->dp; qa.num_iters = num_iters; qa.tol = tol; qa.life = c->d_life;
        cd_queue_fill_batch(c, qa.b, seed, first_index);
        qa.b.R = K * R;
        qa.scratch = c->l2_scratch; qa.Dpack = c->l2_D; qa.Spack = c->l2_S; qa.abort = c->l2_abort; qa.fbound = c->fbound;
        // which of its CU's two workgroups a workgroup is (arrival counter per CU, HW_ID), and what the second one does with it:
        //   QCQPMI_L2_ROT=2 (the default with the factored objective): it turns its MULTIPLYING roles by one among SIMDs 1-3 -- 16 blocks of Y
        //   are 6 + 5 + 5 over the three waves, and with both six-block waves on SIMD 1 that SIMD's matrix pipe was 83 % busy and late for
        //   one product in ten: the chain's wait for partial tiles 0.76 k -> 0.57 k cycles per block interval, the interval 7.45 k -> 7.29 k
        //   (A / B / A / B on one box: 32.93 / 32.57 / 32.80 / 32.45 ms per 20 x 4096 restarts; same bits);
        //   QCQPMI_L2_ROT=1: it turns ALL roles by two SIMDs (the two chains on different SIMDs) -- measured SLOWER, 33.2 -> 35.7 ms: a chain
        //   beside a product stream waits behind the stream's queued matrix instructions (profiles/r06_summary.md);  0: off
        { const char *ev = getenv("QCQPMI_L2_ROT"); const int rot = ev ? atoi(ev) : (lr ? 2 : 0); qa.cuslot = rot ? c->l2_cuslot : nullptr; qa.rotmode = rot; }
        if (qa.cuslot) HIPCHK(c, hipMemsetAsync(c->l2_cuslot, 0, 4096 * sizeof(int), c->stream));
        qa.dbg = (c->dbg & 2048) ? 1 : 0;
        qa.Gpack = lr ? c->lr_G : nullptr; qa.Upack = lr ? c->lr_U : nullptr; qa.RB = lr ? c->lr_RB : 0;
        qa.nclass = c->Kreal;
        const double l0 = stnow();
        (void)hipEventRecord(c->timers[2].beg, c->stream);
        const double l1 = stnow();
        hipError_t qe = (hipError_t)cd_life2_launch(qa, nmw, cs2, kind, tiles, (int)wgs, c->stream);
        const double l2 = stnow();
        (void)hipEventRecord(c->timers[2].end, c->stream);
        c->timers[2].valid = true;
        if (qe != hipSuccess) return fail(c, QCQPMI_EHIP, "cd_stream_run: %s", hipGetErrorString(qe));
        c->last_cd2_kernel = cd_life2_name(nmw, kind, tiles, lr ? 1 : 0);
        int ab = 0;
        HIPCHK(c, spin_sync(c->stream));
        const double l3 = stnow();
        HIPCHK(c, hipMemcpyAsync(&ab, c->l2_abort, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        const double l4 = stnow();
        HIPCHK(c, spin_sync(c->stream));
        if (stt) fprintf(stderr, "cd_stream_run launch (ms): reserve %.3f, event %.3f, launch call %.3f, event + spin wait %.3f, copy call %.3f, sync %.3f\n", l0 - stm[1], l1 - l0, l2 - l1, l3 - l2, l4 - l3, stnow() - l4);
        if (ab) {
            HIPCHK(c, hipMemsetAsync(c->l2_abort, 0, sizeof(int), c->stream));
            c->R = 0;
            return fail(c, QCQPMI_EHIP, "cd_stream_run: a wait inside cd_life_kernel ran into its watchdog (internal error; no results)");
        }
    } else {
        CdQueueArgs qa;
        qa.P = c->dp; qa.num_iters = num_iters; qa.tol = tol;
        qa.life = c->d_life; qa.life_on = 1;
        cd_queue_fill_batch(c, qa.b, seed, first_index);
        qa.b.R = K * R;
        (void)hipEventRecord(c->timers[2].beg, c->stream);
        hipError_t qe = (hipError_t)cd_queue_launch(qa, cs, cus, c->stream);
        (void)hipEventRecord(c->timers[2].end, c->stream);
        c->timers[2].valid = true;
        if (qe != hipSuccess) return fail(c, QCQPMI_EHIP, "cd_stream_run: %s", hipGetErrorString(qe));
        HIPCHK(c, spin_sync(c->stream));
        c->last_cd2_kernel = "cd_phase2_qs_kernel<lifecycle>";
    }
    c->cd_stage = 0;
    c->evaluated = true;                   // d_f0 / d_mv hold the values of the final points
    if (stt) { (void)spin_sync(c->stream); stm[2] = stnow(); }
    std::vector<int> st, st1;
    if ((rc = fetch_cd_outputs(c, sweeps1, sweeps2, visits2, accepted2, ran_phase2, f0, maxviol, st, st1))) return rc;
    stm[3] = stnow();
    if ((rc = cd_apply_status(c, st, st1, f0, maxviol, 0))) return rc;
    stm[4] = stnow();
    if (best_index || best_f0 || best_maxviol || best_x) {
        // the best restart of every population (QCQPForm.better folded over it, ties -> lowest index)
        if ((rc = cd_bestK_reserve(c, K))) return rc;
        // one workgroup per population, then one gather of the winners' columns: three launches and three copies whatever K
        hipLaunchKernelGGL(select_best_kernel, dim3((unsigned)K), dim3(1024), 0, c->stream, (const double *)c->d_f0, (const double *)c->d_mv,
                           R, select_tol, c->d_bestK_idx, c->d_bestK_key);
        if (best_x) {
            HIPCHK(c, hipMemsetAsync(c->d_bestK_x, 0, (size_t)K * c->n * sizeof(double), c->stream));
            hipLaunchKernelGGL(gather_best_x_kernel, dim3((unsigned)K), dim3(256), 0, c->stream, (const double *)c->X, c->n, c->n16, R,
                               (const int64_t *)c->d_bestK_idx, c->d_bestK_x);
        }
        HIPCHK(c, hipGetLastError());
        std::vector<int64_t> idx((size_t)K * 2);
        std::vector<double> key((size_t)K * 2);
        HIPCHK(c, hipMemcpyAsync(idx.data(), c->d_bestK_idx, idx.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(key.data(), c->d_bestK_key, key.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (best_x) HIPCHK(c, hipMemcpyAsync(best_x, c->d_bestK_x, (size_t)K * c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, spin_sync(c->stream));
        for (int64_t p = 0; p < K; p++) {
            if (best_index) best_index[p] = idx[(size_t)2 * p];            // index WITHIN the population
            if (best_f0) best_f0[p] = key[(size_t)2 * p];
            if (best_maxviol) best_maxviol[p] = key[(size_t)2 * p + 1];
        }
    }
    if (stt) {
        stm[5] = stnow();
        fprintf(stderr, "cd_stream_run timing (ms): set-up %.3f, launch + kernel %.3f, fetch %.3f, status %.3f, best %.3f\n", stm[1] - stm[0], stm[2] - stm[1],
                stm[3] - stm[2], stm[4] - stm[3], stm[5] - stm[4]);
    }
    return 0;
}

int qcqpmi_cd_stream_reserve(qcqpmi_ctx *c, int64_t K, int64_t R) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (K < 1 || R < 1 || K * R >= (1LL << 30)) return fail(c, QCQPMI_EINVAL, "cd_stream_reserve: bad K / R");
    HIPCHK(c, hipSetDevice(c->device));
    if (K * R > c->Rcap) { if ((rc = pop_reserve(c, K * R))) return rc; }       // (a resident population of that size or more stays)
    if ((rc = cd_outputs_reserve(c, K * R))) return rc;
    if ((rc = cd_bestK_reserve(c, K))) return rc;
    if (!c->d_qnext && (rc = dev_alloc(c, &c->d_qnext, 16))) return rc;
    if (!c->d_life) HIPCHK(c, hipMalloc((void **)&c->d_life, sizeof(CdLife)));
    {
        int nmw = 0, cs2 = 0, kind = 0, cus = 0;
        HIPCHK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        if (c->life_version != 1 && c->sep && cd_life2_config(c->dp, c->Kreal, c->objclass, c->symcls, c->lr_RB, &nmw, &cs2, &kind) && (rc = cd_life2_reserve(c, nmw, cus))) return rc;
    }
    return 0;
}

int qcqpmi_debug_life_profile(qcqpmi_ctx *c, int64_t *out16) {
    if (!c || !out16) return QCQPMI_EINVAL;
    for (int k = 0; k < 24; k++) out16[k] = 0;
    if (!c->d_life_prof) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, spin_sync(c->stream));
    HIPCHK(c, hipMemcpy(out16, c->d_life_prof, 24 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

int qcqpmi_cd_status(qcqpmi_ctx *c, int *status1, int *status2) {
    if (!c) return QCQPMI_EINVAL;
    if ((int64_t)c->last_st1.size() != c->R || (int64_t)c->last_st2.size() != c->R)
        return fail(c, QCQPMI_ESTATE, "cd_status: no coordinate-descent run on the resident population");
    if (status1) memcpy(status1, c->last_st1.data(), (size_t)c->R * sizeof(int));
    if (status2) memcpy(status2, c->last_st2.data(), (size_t)c->R * sizeof(int));
    return 0;
}

// -------------------------------------------------------------------------------- best of pop

int qcqpmi_select_best(qcqpmi_ctx *c, double tol, int64_t *best_index, double *best_f0,
                       double *best_maxviol, double *best_x) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->evaluated && (rc = launch_eval(c, false))) return rc;
    hipLaunchKernelGGL(select_best_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_f0, c->d_mv, c->R, tol,
                       c->d_best_idx, c->d_best_key);
    HIPCHK(c, hipGetLastError());
    int64_t idx[2];
    double key[2];
    HIPCHK(c, hipMemcpyAsync(idx, c->d_best_idx, sizeof(idx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(key, c->d_best_key, sizeof(key), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    if (best_index) *best_index = idx[0];
    if (best_f0) *best_f0 = key[0];
    if (best_maxviol) *best_maxviol = key[1];
    if (best_x && idx[0] >= 0) {
        int64_t r = idx[0];
        const double *src = c->X + (r >> 4) * c->n16 * 16 + (r & 15);
        HIPCHK(c, hipMemcpy2DAsync(best_x, sizeof(double), src, 16 * sizeof(double), sizeof(double), (size_t)c->n,
                                   hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, spin_sync(c->stream));
    }
    return 0;
}

const char *qcqpmi_last_cd_kernel(qcqpmi_ctx *c) { return c ? c->last_cd2_kernel : ""; }

int qcqpmi_cd_set_objective_factor(qcqpmi_ctx *c, const double *L, int64_t r) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, spin_sync(c->stream));
    for (double **p : {&c->lr_L, &c->lr_G, &c->lr_U}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    c->lr_RB = 0;
    if (!L || r == 0) return 0;                       // cleared
    if (r < 1 || r > c->n) return fail(c, QCQPMI_EINVAL, "cd_set_objective_factor: r = %lld outside 1 .. n", (long long)r);
    if (!cd_life2_factor_ok(c->dp, r))
        return fail(c, QCQPMI_EUNSUPPORTED, "cd_set_objective_factor: the factored lifecycle kernel takes factors of at most 288 columns and n >= 128");
    const int64_t n = c->n, n16 = c->n16, r16 = (r + 15) / 16 * 16;
    const int RB = (int)(r16 / 16), NB = (int)(n16 / 16);
    if ((rc = dev_alloc(c, &c->lr_L, (size_t)n16 * r16))) return rc;      // (zero-filled)
    if ((rc = dev_alloc(c, &c->lr_G, (size_t)NB * RB * 256, false))) return rc;
    if ((rc = dev_alloc(c, &c->lr_U, (size_t)NB * RB * 256, false))) return rc;
    HIPCHK(c, hipMemcpy2DAsync(c->lr_L, (size_t)r16 * sizeof(double), L, (size_t)r * sizeof(double), (size_t)r * sizeof(double), (size_t)n,
                               hipMemcpyHostToDevice, c->stream));
    hipError_t e = (hipError_t)cd_life2_pack_factor(c->lr_L, c->lr_G, c->lr_U, NB, RB, c->stream);
    if (e != hipSuccess) return fail(c, QCQPMI_EHIP, "cd_life2_pack_factor: %s", hipGetErrorString(e));
    HIPCHK(c, spin_sync(c->stream));      // (the 2-D copy may still be reading the caller's pages)
    c->lr_RB = RB;
    return 0;
}

int qcqpmi_cd_life_version(qcqpmi_ctx *c, int version) {
    if (!c || version < 0 || version > 3) return QCQPMI_EINVAL;      // (3: cd_life_kernel without the objective factor)
    c->life_version = version;
    return 0;
}

int qcqpmi_cd_queue(qcqpmi_ctx *c, int mode) {
    if (!c || mode < 0 || mode > 2) return QCQPMI_EINVAL;
    c->cd_queue = mode;
    return 0;
}

int qcqpmi_cd_reference_order(qcqpmi_ctx *c, int enable) {
    if (!c) return QCQPMI_EINVAL;
    c->cd_ref_order = enable != 0;
    return 0;
}

int qcqpmi_last_kernel_ms(qcqpmi_ctx *c, int which, double *ms) {
    if (!c || which < 0 || which > 4 || !ms) return QCQPMI_EINVAL;
    if (!c->timers[which].valid) return fail(c, QCQPMI_ESTATE, "kernel %d has not been launched", which);
    HIPCHK(c, hipEventSynchronize(c->timers[which].end));
    float f = 0.f;
    HIPCHK(c, hipEventElapsedTime(&f, c->timers[which].beg, c->timers[which].end));
    *ms = (double)f;
    return 0;
}

int qcqpmi_dense_chain_geometry(int64_t m, int *out4) {
    if (m < 0 || m > (1 << 24) || !out4) return QCQPMI_EINVAL;
    const MwGeom g = mw_geometry((int)m + 1);
    out4[0] = g.SL; out4[1] = g.Tc; out4[2] = g.ts; out4[3] = g.T;
    return 0;
}

int qcqpmi_dense_chain_mode(qcqpmi_ctx *c, int mode) {
    if (!c || mode < 0 || mode > 1) return QCQPMI_EINVAL;
    c->dense_chain_mode = mode;
    return 0;
}

int qcqpmi_debug_profile(qcqpmi_ctx *c, int enable, int64_t *sums8) {
    if (!c) return QCQPMI_EINVAL;
    c->profile = (enable & 1) != 0;
    c->force_generic = (enable & 2) != 0;
    c->dbg = enable >> 4;
    if (sums8 && c->d_prof && c->Rpad > 0) {
        std::vector<long long> h((size_t)(c->Rpad / 16) * 16);
        HIPCHK(c, hipMemcpy(h.data(), c->d_prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int k = 0; k < 16; k++) sums8[k] = 0;
        for (size_t t = 0; t < h.size(); t++) sums8[t & 15] += h[t];
    }
    return 0;
}

// debug: event trace of tile 0 written by the profiled phase-2 kernel (8 slices of 256 words, one per wave: entries of
// 4 timestamps per block / product; see cd_phase2_q.h)
int qcqpmi_debug_trace(qcqpmi_ctx *c, int64_t *out, int count) {
    if (!c || !out || count < 0 || count > QCQPMI_TRACE_WORDS) return QCQPMI_EINVAL;
    if (!c->d_prof || c->Rpad <= 0) return fail(c, QCQPMI_EINVAL, "no profiled run to trace");
    HIPCHK(c, hipMemcpy(out, c->d_prof + (size_t)(c->Rpad / 16) * 16, (size_t)count * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

int qcqpmi_sync(qcqpmi_ctx *c) {
    if (!c) return QCQPMI_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------- RCCL

#define NCCLCHK(c, expr)                                                                        \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(c, QCQPMI_ECOMM, "%s failed: %s", #expr,                                \
                        rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "rccl error");    \
    } while (0)

int qcqpmi_comm_unique_id(uint8_t id_out[128]) {
    if (!rccl()) return fail(nullptr, QCQPMI_ECOMM, "librccl.so could not be loaded");
    ncclUniqueId id;
    if (rccl()->GetUniqueId(&id) != ncclSuccess) return fail(nullptr, QCQPMI_ECOMM, "ncclGetUniqueId failed");
    memcpy(id_out, id.internal, 128);
    return 0;
}

int qcqpmi_comm_init(qcqpmi_ctx *c, int rank, int world, const uint8_t id_in[128]) {
    if (!c || world < 1 || rank < 0 || rank >= world) return QCQPMI_EINVAL;
    if (!rccl()) return fail(c, QCQPMI_ECOMM, "librccl.so could not be loaded");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, id_in, 128);
    NCCLCHK(c, rccl()->CommInitRank(&c->comm, world, id, rank));
    c->rank = rank; c->world = world;
    if (!c->d_comm) {
        int rc = dev_alloc(c, &c->d_comm, (size_t)4 * world + 4 + (size_t)c->n);
        if (rc) return rc;
    }
    return 0;
}

int qcqpmi_comm_barrier(qcqpmi_ctx *c) {
    double v = 0.0;
    return qcqpmi_comm_allreduce(c, &v, 1, 0);
}

int qcqpmi_comm_allreduce(qcqpmi_ctx *c, double *values, int64_t count, int op) {
    if (!c || !values || count < 1 || count > ((int64_t)1 << 28) || op < 0 || op > 1) return QCQPMI_EINVAL;
    if (!c->comm) return fail(c, QCQPMI_ESTATE, "comm_init has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    double *buf = c->d_comm;
    if (count > 4) {      // the exchange of a streamed run (keys of all populations, then the winners' points): a buffer of its own
        if (count > c->comm_big_cap) {
            HIPCHK(c, spin_sync(c->stream));
            if (c->d_comm_big) (void)hipFree(c->d_comm_big);
            c->d_comm_big = nullptr; c->comm_big_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_comm_big, (size_t)count * sizeof(double)));
            c->comm_big_cap = count;
        }
        buf = c->d_comm_big;
    }
    HIPCHK(c, hipMemcpyAsync(buf, values, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, rccl()->AllReduce(buf, buf, (size_t)count, ncclDouble, op == 0 ? ncclMax : ncclSum, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(values, buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

// every rank contributes `nbytes` bytes; everyone receives the world x nbytes bytes in rank order (one ncclAllGather)
int qcqpmi_comm_allgather(qcqpmi_ctx *c, const void *send, int64_t nbytes, void *recv) {
    if (!c || !send || !recv || nbytes < 1 || nbytes > ((int64_t)1 << 28)) return QCQPMI_EINVAL;
    if (!c->comm) return fail(c, QCQPMI_ESTATE, "comm_init has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t need = ((int64_t)(c->world + 1) * nbytes + 7) / 8;      // doubles: send area + receive area
    if (need > c->comm_big_cap) {
        HIPCHK(c, spin_sync(c->stream));
        if (c->d_comm_big) (void)hipFree(c->d_comm_big);
        c->d_comm_big = nullptr; c->comm_big_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_comm_big, (size_t)need * sizeof(double)));
        c->comm_big_cap = need;
    }
    char *d_send = (char *)c->d_comm_big, *d_recv = d_send + nbytes;
    HIPCHK(c, hipMemcpyAsync(d_send, send, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, rccl()->AllGather(d_send, d_recv, (size_t)nbytes, ncclInt8, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(recv, d_recv, (size_t)nbytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    return 0;
}

int qcqpmi_comm_select_best(qcqpmi_ctx *c, double tol, int64_t index_offset, int64_t *best_global_index,
                            double *best_f0, double *best_maxviol, double *best_x) {
    if (!c) return QCQPMI_EINVAL;
    if (!c->comm) return fail(c, QCQPMI_ESTATE, "comm_init has not been called");
    int64_t li = -1;
    double lf = 0.0, lv = 0.0;
    int rc = qcqpmi_select_best(c, tol, &li, &lf, &lv, nullptr);
    if (rc) return rc;
    // key record: (bucket, f0, maxviol, global index) as 4 doubles (indices < 2^53 are exact)
    double rec[4] = {std::floor(lv / tol), lf, lv, (double)(index_offset + li)};
    if (!(lv == lv) || !(lf == lf) || li < 0) rec[0] = 9.0e18;
    const int W = c->world;
    double *d_send = c->d_comm, *d_recv = c->d_comm + 4, *d_x = c->d_comm + 4 + 4 * (size_t)W;
    HIPCHK(c, hipMemcpyAsync(d_send, rec, sizeof(rec), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, rccl()->AllGather(d_send, d_recv, 4, ncclDouble, c->comm, c->stream));
    std::vector<double> all((size_t)4 * W);
    HIPCHK(c, hipMemcpyAsync(all.data(), d_recv, all.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    int win = 0;
    for (int w = 1; w < W; w++) {
        const double *a = &all[(size_t)4 * w], *b = &all[(size_t)4 * win];
        if (a[0] < b[0] || (a[0] == b[0] && (a[1] < b[1] || (a[1] == b[1] && a[3] < b[3])))) win = w;
    }
    if (c->rank == win && li >= 0) {
        const double *src = c->X + (li >> 4) * c->n16 * 16 + (li & 15);
        HIPCHK(c, hipMemcpy2DAsync(d_x, sizeof(double), src, 16 * sizeof(double), sizeof(double), (size_t)c->n,
                                   hipMemcpyDeviceToDevice, c->stream));
    }
    NCCLCHK(c, rccl()->Broadcast(d_x, d_x, (size_t)c->n, ncclDouble, win, c->comm, c->stream));
    if (best_x) HIPCHK(c, hipMemcpyAsync(best_x, d_x, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, spin_sync(c->stream));
    if (best_global_index) *best_global_index = (int64_t)all[(size_t)4 * win + 3];
    if (best_f0) *best_f0 = all[(size_t)4 * win + 1];
    if (best_maxviol) *best_maxviol = all[(size_t)4 * win + 2];
    return 0;
}

}  // extern "C"

#include "capi_admm.inc"
#include "capi_units.inc"
