/*
 * mcmc_hip.hip -- C-ABI implementation (include/mcmc_hip.h) over the kernels of
 * mcmc_kernels.hip.h.  Host side: context/ownership, the one-off response-table
 * precompute (AvgDiseaseTrajectory, covid.h:96-282), the step schedule of
 * MetropolisChain::run (mcmc.h:1005-1138) and result copies.  Plain HIP runtime only.
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
 */
#include "mcmc_hip.h"
#include "mcmc_kernels.hip.h"
#include "mcmc_wave_kernel.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#ifndef MCMC_T
#define MCMC_T 8           /* days per register block, see mcmc_kernels.hip.h */
#endif
#ifndef MCMC_STEPS_PER_LAUNCH
#define MCMC_STEPS_PER_LAUNCH 16   /* MH steps one mh_step_kernel launch advances every chain by */
#endif
#ifndef MCMC_WAVE_STEPS_PER_LAUNCH
#define MCMC_WAVE_STEPS_PER_LAUNCH 64   /* ... and one mh_step_wave_kernel launch: its launches are short (0.9 ms per 16 steps on BASELINE
                                           configs[1]), the host's wait for each one and the restaging of its LDS image are 2-3 % of them */
#endif
#ifndef MCMC_WAVE_AUTO_MAX
#define MCMC_WAVE_AUTO_MAX 16384   /* MCMC_HIP_MAPPING_AUTO: ensembles up to this many chains take one chain per wavefront (tools/mapping_sweep.sh) */
#endif
#ifndef MCMC_T_DERIVE
#define MCMC_T_DERIVE 4    /* the derive kernel is not a hot path: small block, short compile */
#endif

/* ---- the one entry point that allocates nothing (mcmc_hip_eval_static) -----------------------
 * Its kernels' scratch -- the tiled copy of the parameter vectors, both per-lane histories, the
 * weighted-deaths rows and the result rows, everything mcmc_hip_eval hipMallocs per call -- lives
 * in this file-scope device array.  Every GPU the library runs on holds its own copy, at its own
 * address, until the process ends.  In host code the array's NAME is only a handle: the address
 * kernels and memsets may use is the one hipGetSymbolAddress returns, fetched once per device in
 * mcmc_hip_create (static_scratch_for) with that device current, and only that pointer is used. */
#define MCMC_STATIC_SCRATCH_BYTES (16u << 20)
static __device__ __align__(256) double g_static_scratch[MCMC_STATIC_SCRATCH_BYTES / sizeof(double)];

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(MCMC_HIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

/* No exception crosses the C ABI: every entry point with a body that can allocate host
 * memory is a function-try-block ending in this handler. */
#define MCMC_ABI_GUARD                                                                           \
    catch (const std::bad_alloc&) { return fail(MCMC_HIP_ENOMEM, "out of host memory"); }        \
    catch (const std::exception& e_) { return fail(MCMC_HIP_EHIP, "unexpected exception: %s", e_.what()); } \
    catch (...) { return fail(MCMC_HIP_EHIP, "unexpected exception"); }

template <typename Tp>
struct DevBuf {
    Tp* p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        release();
        if (count == 0) return 0;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(Tp));
        if (e != hipSuccess) { p = nullptr; return fail(MCMC_HIP_ENOMEM, "hipMalloc(%zu B): %s", count * sizeof(Tp), hipGetErrorString(e)); }
        n = count;
        return 0;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    ~DevBuf() { release(); }
};

}  // namespace

struct mcmc_hip_ctx {
    mcmc_covid_config cfg{};
    std::vector<int32_t> discDays;
    std::vector<uint8_t> discFixed;
    int device = 0, nCU = 0;
    double* staticScratch = nullptr;    /* this device's g_static_scratch (mcmc_hip_eval_static) */
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int size = 0, nB = 0, dim = 0, simDays = 0, simPad = 0, nLag = 0, tabRows = 0, histRows = 0;
    /* multi-region model (extension): dim = nRegions * dimRegion once set_regions has been called */
    bool multi = false;
    int nRegions = 1, dimRegion = 0;
    long long headCounter = 0;          /* random-subspace steps since set_chains: the systematic scan's position */
    DevBuf<double> dRegDeaths, dRegSigma2, dRegPop, dRegCap0, dRegCapRate, dLLProp;
    bool haveTables = false, haveData = false, haveChains = false;
    DevModel M{};
    ChainState S{};
    long long nChains = 0, nP = 0;      /* nP = nTiles*64 */
    int nTiles = 0;
    int nSaved = 0, nSavedCap = 0;
    bool recSamples = false;
    /* model constants */
    DevBuf<double> dRpos, dRneg, dCapacity, dDeaths, dSigma2;
    DevBuf<float> dRF;                  /* fast mode: float tables [size + 3*MCMC_TF][8] and [..][4] */
    int precision = MCMC_HIP_PRECISION_F64;
    int mapping = MCMC_HIP_MAPPING_AUTO;   /* what the caller asked for */
    bool useWave = false;                  /* what the step launches do: decided in set_chains / set_mapping */
    DevBuf<int> dSeg;
    /* chain state */
    DevBuf<double> dParams, dLL, dStepsize, dHn, dHe, dD1s, dMean, dRecLL, dRecSamples;
    DevBuf<uint64_t> dRng;
    DevBuf<int> dNAccept;
    DevBuf<unsigned long long> dPairs;
    /* tempering ladder (extension) */
    int ladderK = 0;
    long long nLadders = 0;
    DevBuf<double> dLadderBeta, dChainBeta;
    DevBuf<int> dChainAt, dRungOf;
    DevBuf<uint64_t> dSwapRng;
    DevBuf<unsigned long long> dSwapAcc;
    unsigned long long swapRounds[2] = {0, 0};
    /* scratch */
    DevBuf<double> dBetas;
    DevBuf<int> dFlags;
    /* timing */
    double adjustS = 0, samplingS = 0, kernelS = 0;
    long long kernelLaunches = 0;
    unsigned long long stepsDone = 0;
};

namespace {

/* ------------------------------------------------------------ host precompute ---- */
struct HostPcg {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;   /* pcg32() */
    uint32_t next_uint() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t x = (uint32_t)(((old >> 18u) ^ old) >> 27u), r = (uint32_t)(old >> 59u);
        return (x >> r) | (x << ((~r + 1u) & 31));
    }
    float next_float() { uint32_t u = (next_uint() >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f; }
    double next_double() { uint64_t u = ((uint64_t)next_uint() << 20) | 0x3ff0000000000000ULL; double d; memcpy(&d, &u, 8); return d - 1.0; }
};

/* enoki::erfinv as used by toNormal (mcmc.h:45-47): Giles' single-precision polynomial */
double erfinv_giles(double x) {
    double w = -std::log((1.0 - x) * (1.0 + x));
    static const double lo[9] = {2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06,
                                 0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941};
    static const double hi[9] = {-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844,
                                 0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682};
    const double* c;
    if (w < 5.0) { w = w - 2.5; c = lo; } else { w = std::sqrt(w) - 3.0; c = hi; }
    double p = c[0];
    for (int i = 1; i < 9; ++i) p = c[i] + p * w;
    return p * x;
}

constexpr size_t kGrid = 100000;      /* AvgDiseaseTrajectory::nGrid, covid.h:205 */
constexpr size_t kDays = MCMC_HIP_TRAJ_NDAYS;

/* AvgDiseaseTrajectory ctor (covid.h:100-203) -> the 11 cumulative curves on the grid */
struct Trajectories {
    enum { INCUB, ASYMP, MILD, INFMILD, INFHIGH, SERIOUS, DEAD, RECOV, SFS, RFS, DFS, N };
    std::vector<double> a[N];
    static size_t slot(double t) {                                   /* t2i, covid.h:273-280 */
        double v = (double)kGrid * (t / (double)kDays);
        return (v < (double)kGrid) ? (size_t)v : kGrid - 1;
    }
    /* sampleTime (covid.h:262-271) receives the generator by value: a peek */
    static double sample_time(HostPcg rnd, double mean, double sig) {
        double s = std::sqrt(std::log(sig * sig / (mean * mean) + 1));
        double expmu = mean * std::exp(-0.5 * s * s);
        double u;
        do { u = rnd.next_double(); } while (u <= 0 || u >= 1);
        double n = M_SQRT2 * erfinv_giles(2.0 * u - 1.0) * s;
        return expmu * std::exp(n);
    }
    void generate(const mcmc_disease_params& p, uint64_t nTraj) {
        for (auto& v : a) v.assign(kGrid, 0.0);
        HostPcg rnd;
        for (uint64_t k = 0; k < nTraj; ++k) {
            a[INCUB][0] += 1;
            const double t = sample_time(rnd, p.timeIncub, p.timeIncubSigma);
            const size_t it = slot(t);
            a[INCUB][it] -= 1;
            if (rnd.next_double() < p.probAsymp) {
                a[ASYMP][it] += 1; a[INFHIGH][it] += 1;
                const size_t i2 = slot(t + sample_time(rnd, p.timeMildDuration, p.timeMildDurationSigma));
                a[ASYMP][i2] -= 1; a[RECOV][i2] += 1; a[INFHIGH][i2] -= 1;
            } else {
                a[MILD][it] += 1; a[INFMILD][it] += 1;
                if ((int)it - 1 > 0) { a[INFHIGH][it - 1] += 1; a[INFHIGH][it] -= 1; }
                const double x = rnd.next_double();
                if (x < 1 - (p.probSerious + p.probLethal) / (1 - p.probAsymp)) {
                    const size_t i2 = slot(t + sample_time(rnd, p.timeMildDuration, p.timeMildDurationSigma));
                    a[MILD][i2] -= 1; a[INFMILD][i2] -= 1; a[RECOV][i2] += 1;
                } else {
                    const double t2 = sample_time(rnd, p.timeMildToSerious, p.timeMildToSerious); /* covid.h:142 */
                    const size_t i2 = slot(t + t2);
                    a[INFMILD][i2] -= 1; a[SERIOUS][i2] += 1; a[MILD][i2] -= 1;
                    double t3;
                    if (x < 1 - p.probLethal / (1 - p.probAsymp)) {
                        t3 = sample_time(rnd, p.timeSeriousToRec, p.timeSeriousToRecSigma);
                        a[RECOV][slot(t + t2 + t3)] += 1;
                    } else {
                        t3 = sample_time(rnd, p.timeSeriousToDeath, p.timeSeriousToDeathSigma);
                        a[DEAD][slot(t + t2 + t3)] += 1;
                    }
                    a[SERIOUS][slot(t + t2 + t3)] -= 1;
                }
            }
            const double x = rnd.next_double();
            a[SFS][0] += 1;
            double t3;
            if (x < p.probSerious / (p.probLethal + p.probSerious)) {
                t3 = sample_time(rnd, p.timeSeriousToRec, p.timeSeriousToRecSigma);
                a[RFS][slot(t3)] += 1;
            } else {
                t3 = sample_time(rnd, p.timeSeriousToDeath, p.timeSeriousToDeathSigma);
                a[DFS][slot(t3)] += 1;
            }
            a[SFS][slot(t3)] -= 1;
        }
        const double nt = (double)nTraj;
        for (auto& v : a) {
            v[0] /= nt;
            for (size_t t = 1; t < kGrid; ++t) v[t] = v[t - 1] + v[t] / nt;
        }
    }
    /* getDay, covid.h:221-235 */
    static double day_delta(const std::vector<double>& ar, size_t day) {
        if (day == 0) return ar[0];
        if (day == kDays) return 0;
        const size_t i1 = (size_t)((double)kGrid * ((double)(day - 1) / (double)kDays));
        const size_t i2 = (size_t)((double)kGrid * ((double)day / (double)kDays));
        if (i2 > kGrid - 1) return 0;
        return ar[i2] - ar[i1];
    }
};

int round_up(int v, int m) { return (v + m - 1) / m * m; }

/* device address of g_static_scratch on `device` (which must be current): asked once per device */
constexpr int kMaxDevices = 64;
double* g_scratch_of_device[kMaxDevices];
std::mutex g_scratch_mu;
/* one lock per device's array: eval_static holds it from its memset to the end of its kernels, so calls from
 * several host threads (contexts have streams of their own) take turns instead of racing on the array */
std::mutex g_scratch_busy[kMaxDevices];
int static_scratch_for(int device, double** out) {
    if (device < 0 || device >= kMaxDevices) return fail(MCMC_HIP_EINVAL, "device ordinal %d beyond %d", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_of_device[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)));
        if (!p) return fail(MCMC_HIP_EHIP, "hipGetSymbolAddress returned a null address for the static scratch");
        g_scratch_of_device[device] = (double*)p;
    }
    *out = g_scratch_of_device[device];
    return 0;
}

double target_beta(const mcmc_covid_config& c, double time) {
    switch (c.target_kind) {
        case MCMC_TARGET_TEMP: return 1 / c.target_T;                         /* mcmc.h:801,806 */
        case MCMC_TARGET_COOLING: return std::exp(time * c.target_slope) / c.target_Tinit; /* mcmc.h:830 */
        default: return 1;                                                    /* mcmc.h:742 */
    }
}

/* One workgroup = 4 waves = 4 chain tiles sharing one LDS image (table + series).  With at
 * most one workgroup per CU's worth of tiles the LDS request is padded past half the CU's
 * 160 KB so that the dispatcher cannot stack two workgroups on one CU while another idles. */
constexpr size_t kLdsMax = 160 * 1024;
constexpr size_t kLdsSolo = 81 * 1024;

template <typename K, typename... Args>
int launch_chain_kernel(mcmc_hip_ctx* c, K kernel, bool derive, int T, Args... args) {
    const unsigned grid = (unsigned)((c->nTiles + MCMC_WAVES_PER_BLOCK - 1) / MCMC_WAVES_PER_BLOCK);
    /* the fast mode on the matrix cores has an LDS image of its own (float tables + transposition buffers) */
    const bool f32Image = !derive && !c->multi && c->precision == MCMC_HIP_PRECISION_F32_MODEL;
    size_t shmem = f32Image ? lds_bytes_f32(c->M) : lds_bytes(c->M, derive, T);
    if (c->multi && !derive)           /* + one region slice per wave (stage_region) */
        shmem = ((shmem + 7) / 8 + (size_t)MCMC_WAVES_PER_BLOCK * region_slice_doubles(c->M)) * sizeof(double);
    if (shmem > kLdsMax)
        return fail(MCMC_HIP_EINVAL, "the series is too long for the LDS image (%zu B > %zu B)", shmem, kLdsMax);
    if (grid <= (unsigned)c->nCU && shmem < kLdsSolo) shmem = kLdsSolo;
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(MCMC_BLOCK), shmem, c->stream, args...);
    HIP_TRY(hipGetLastError());
    return 0;
}

int run_eval_current(mcmc_hip_ctx* c) {
    if (c->multi) {
        if (c->cfg.contract) return launch_chain_kernel(c, eval_kernel<MCMC_T, true, 0, true>, false, MCMC_T, c->M, c->S, (const double*)c->S.params, c->S.ll);
        return launch_chain_kernel(c, eval_kernel<MCMC_T, false, 0, true>, false, MCMC_T, c->M, c->S, (const double*)c->S.params, c->S.ll);
    }
    if (c->precision == MCMC_HIP_PRECISION_F32_MODEL)
        return launch_chain_kernel(c, eval_kernel<MCMC_T, true, 2>, false, MCMC_T, c->M, c->S, (const double*)c->S.params, c->S.ll);
    if (c->precision == MCMC_HIP_PRECISION_F32_PACKED)
        return launch_chain_kernel(c, eval_kernel<MCMC_T, true, 1>, false, MCMC_T, c->M, c->S, (const double*)c->S.params, c->S.ll);
    if (c->cfg.contract) return launch_chain_kernel(c, eval_kernel<MCMC_T, true>, false, MCMC_T, c->M, c->S, (const double*)c->S.params, c->S.ll);
    return launch_chain_kernel(c, eval_kernel<MCMC_T, false>, false, MCMC_T, c->M, c->S, (const double*)c->S.params, c->S.ll);
}

/* Can this context's chains run one per wavefront (mcmc_wave_kernel.hip.h)?  `why` names the obstacle. */
bool wave_mapping_possible(const mcmc_hip_ctx* c, const char** why) {
    const char* w = nullptr;
    if (c->multi) w = "the multi-region model runs one chain per lane only";
    else if (c->precision != MCMC_HIP_PRECISION_F64) w = "the fast mode runs one chain per lane only";
    else if (wave_lds_bytes(c->M, c->dim, 1) > kLdsMax) w = "the series is too long for a chain's slice of LDS";
    if (why) *why = w;
    return w == nullptr;
}
/* chains of one workgroup beside the shared image while every chain has a SIMD of its own: 4, 2 or 1 (0: none fits) */
static int wave_slices_that_fit(const mcmc_hip_ctx* c) {
    int wpb = MCMC_WAVES_PER_BLOCK;
    while (wpb > 1 && wave_lds_bytes(c->M, c->dim, wpb) > kLdsMax) wpb /= 2;
    return wave_lds_bytes(c->M, c->dim, wpb) > kLdsMax ? 0 : wpb;
}
void decide_mapping(mcmc_hip_ctx* c) {
    const bool ok = wave_mapping_possible(c, nullptr);
    /* AUTO: the wave mapping's ceiling is proportional to the chains a CU holds at a time, and so is the ensemble
     * size at which one chain per lane overtakes it (27 k chains with four per workgroup, tools/mapping_sweep.sh) */
    const long long autoMax = ok ? (long long)MCMC_WAVE_AUTO_MAX * wave_slices_that_fit(c) / MCMC_WAVES_PER_BLOCK : 0;
    c->useWave = ok && (c->mapping == MCMC_HIP_MAPPING_WAVE ||
                        (c->mapping == MCMC_HIP_MAPPING_AUTO && c->nChains > 0 && c->nChains <= autoMax));
}

int launch_wave_steps(mcmc_hip_ctx* c, int nSteps, const double* dBetas, const int* dFlags, int slot0) {
    /* As many chains (waves) per workgroup as the LDS holds slices beside the shared image.  Up to one chain per
     * SIMD of the chip: 4, 2 or 1, every chain a SIMD of its own.  Larger ensembles: 8 or 7 where they fit (two
     * waves on a SIMD each run at 0.64 of a lone wave's rate: 8 192 chains x 200 days run 24.6 M steps/s with 8 per
     * workgroup, 20.2 with 7, 19.2 with 4, 17.2 with 6, 15.1 with 5; tools/wpb_sweep.sh; MCMC_WAVE_WPB overrides,
     * for such measurements). */
    int wpb = std::max(wave_slices_that_fit(c), 1);
    const int maxWaves = c->cfg.contract ? MCMC_WV_MAX_WAVES : MCMC_WAVES_PER_BLOCK;      /* the strict kernel's launch bounds */
    if (c->nChains > (long long)MCMC_WAVES_PER_BLOCK * c->nCU && wpb == MCMC_WAVES_PER_BLOCK)
        for (int w = maxWaves; w >= 7; --w)
            if (wave_lds_bytes(c->M, c->dim, w) <= kLdsMax) { wpb = w; break; }
    if (const char* e = getenv("MCMC_WAVE_WPB")) {
        const int w = atoi(e);
        if (w >= 1 && w <= maxWaves && wave_lds_bytes(c->M, c->dim, w) <= kLdsMax) wpb = w;
    }
    const unsigned grid = (unsigned)((c->nChains + wpb - 1) / wpb);
    size_t shmem = wave_lds_bytes(c->M, c->dim, wpb);
    if (shmem > kLdsMax) return fail(MCMC_HIP_EINVAL, "the series is too long for a chain's slice of LDS (%zu B > %zu B)", shmem, kLdsMax);
    if (grid <= (unsigned)c->nCU && shmem < kLdsSolo) shmem = kLdsSolo;      /* one workgroup per CU while CUs are idle */
    if (c->cfg.contract) {
        HIP_TRY(hipFuncSetAttribute((const void*)mh_step_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(mh_step_wave_kernel<true>, dim3(grid), dim3(wpb * MCMC_WAVE), shmem, c->stream, c->M, c->S, nSteps, dBetas, dFlags, slot0);
    } else {
        HIP_TRY(hipFuncSetAttribute((const void*)mh_step_wave_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(mh_step_wave_kernel<false>, dim3(grid), dim3(wpb * MCMC_WAVE), shmem, c->stream, c->M, c->S, nSteps, dBetas, dFlags, slot0);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_steps(mcmc_hip_ctx* c, int nSteps, const double* dBetas, const int* dFlags, int slot0) {
    const long long hc = c->headCounter;
    if (c->useWave) return launch_wave_steps(c, nSteps, dBetas, dFlags, slot0);
    if (c->multi) {
        if (c->cfg.contract)
            return launch_chain_kernel(c, mh_step_kernel<MCMC_T, true, 0, true>, false, MCMC_T, c->M, c->S, nSteps, dBetas, dFlags, slot0, hc);
        return launch_chain_kernel(c, mh_step_kernel<MCMC_T, false, 0, true>, false, MCMC_T, c->M, c->S, nSteps, dBetas, dFlags, slot0, hc);
    }
    if (c->precision == MCMC_HIP_PRECISION_F32_MODEL)
        return launch_chain_kernel(c, mh_step_kernel<MCMC_T, true, 2>, false, MCMC_T, c->M, c->S, nSteps, dBetas, dFlags, slot0, hc);
    if (c->precision == MCMC_HIP_PRECISION_F32_PACKED)
        return launch_chain_kernel(c, mh_step_kernel<MCMC_T, true, 1>, false, MCMC_T, c->M, c->S, nSteps, dBetas, dFlags, slot0, hc);
    if (c->cfg.contract)
        return launch_chain_kernel(c, mh_step_kernel<MCMC_T, true>, false, MCMC_T, c->M, c->S, nSteps, dBetas, dFlags, slot0, hc);
    return launch_chain_kernel(c, mh_step_kernel<MCMC_T, false>, false, MCMC_T, c->M, c->S, nSteps, dBetas, dFlags, slot0, hc);
}

int check_ctx(const mcmc_hip_ctx* c) {
    if (!c) return fail(MCMC_HIP_EINVAL, "null context");
    return 0;
}

/* Runs a prepared schedule in launches of at most `perLaunch` steps, timing the kernels. */
int run_schedule(mcmc_hip_ctx* c, const std::vector<double>& betas, const std::vector<int>& flags,
                 size_t first, size_t last, int perLaunch, double* wallS) {
    if (last <= first) { if (wallS) *wallS = 0; return 0; }
    const size_t n = last - first;
    if (c->dBetas.n < n) { if (int r = c->dBetas.alloc(n)) return r; }
    if (c->dFlags.n < n) { if (int r = c->dFlags.alloc(n)) return r; }
    HIP_TRY(hipMemcpyAsync(c->dBetas.p, betas.data() + first, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->dFlags.p, flags.data() + first, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    int slot = 0;
    for (size_t g = 0; g < first; ++g) if (flags[g] & SF_RECORD) ++slot;
    double total = 0;
    for (size_t off = 0; off < n; off += (size_t)perLaunch) {
        const int cnt = (int)std::min((size_t)perLaunch, n - off);
        HIP_TRY(hipEventRecord(c->ev0, c->stream));
        if (int r = launch_steps(c, cnt, c->dBetas.p + off, c->dFlags.p + off, slot)) return r;
        HIP_TRY(hipEventRecord(c->ev1, c->stream));
        HIP_TRY(hipEventSynchronize(c->ev1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        total += ms * 1e-3;
        c->kernelS += ms * 1e-3;
        c->kernelLaunches += 1;
        for (int g = 0; g < cnt; ++g) {
            if (flags[first + off + g] & SF_RECORD) ++slot;
            if (flags[first + off + g] & SF_RANDOM) ++c->headCounter;
        }
        c->stepsDone += (unsigned long long)cnt;
    }
    if (wallS) *wallS = total;
    return 0;
}

/* uploads a chain-major host field [n][rows] into a tile-major device field [nTiles][rows][64] */
template <typename Tp>
int upload_field(mcmc_hip_ctx* c, const Tp* host, int rows, Tp* dev) {
    const long long n = c->nChains, nP = c->nP;
    DevBuf<Tp> stage;
    if (int r = stage.alloc((size_t)n * rows)) return r;
    HIP_TRY(hipMemcpyAsync(stage.p, host, (size_t)n * rows * sizeof(Tp), hipMemcpyHostToDevice, c->stream));
    const unsigned tb = 256, tg = (unsigned)((nP + tb - 1) / tb);
    hipLaunchKernelGGL(tile_in_kernel<Tp>, dim3(tg), dim3(tb), 0, c->stream, (const Tp*)stage.p, dev, n, rows, nP);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace

extern "C" {

const char* mcmc_hip_last_error(void) { return g_err.c_str(); }

int mcmc_hip_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} MCMC_ABI_GUARD

void mcmc_host_default_disease_params(mcmc_disease_params* p) {
    if (!p) return;
    *p = mcmc_disease_params{4, 2, 0.1, 0.08, 0.004, 0.2, 0.3, 10, 5, 8, 3, 10, 5, 9, 5};
}

int mcmc_host_make_response_tables(const mcmc_disease_params* p, uint64_t n_traj, int32_t n_lag, double* R) try {
    if (!p || !R || n_lag < 1 || n_lag > MCMC_HIP_TRAJ_NDAYS || n_traj == 0)
        return fail(MCMC_HIP_EINVAL, "make_response_tables: bad arguments");
    Trajectories tr;
    tr.generate(*p, n_traj);
    /* curve order of MCMC_HIP_NCURVES */
    const int order[MCMC_HIP_NCURVES] = {Trajectories::INFMILD, Trajectories::INFHIGH, Trajectories::INCUB,
                                         Trajectories::MILD, Trajectories::SERIOUS, Trajectories::RECOV,
                                         Trajectories::DEAD, Trajectories::SFS, Trajectories::RFS, Trajectories::DFS};
    for (int c = 0; c < MCMC_HIP_NCURVES; ++c) {
        double cumsum = 0;                                           /* covid.h:377-382 */
        for (int d = 0; d < n_lag; ++d) {
            cumsum += Trajectories::day_delta(tr.a[order[c]], (size_t)d);
            R[(size_t)c * n_lag + d] = cumsum;
        }
    }
    return 0;
} MCMC_ABI_GUARD

int mcmc_host_perturb_ics(const double* x0, int32_t dim, int64_t n_chains, double a, uint64_t initstate,
                          uint64_t initseq, double* out) try {
    if (!x0 || !out || dim < 1 || n_chains < 0) return fail(MCMC_HIP_EINVAL, "perturb_ics: bad arguments");
    HostPcg r;
    pcg32_seed(r.state, r.inc, initstate, initseq);
    for (int64_t c = 0; c < n_chains; ++c)
        for (int k = 0; k < dim; ++k)
            out[c * dim + k] = x0[k] + (r.next_float() - .5f) * x0[k] * a;   /* mcmc.h:310 */
    return 0;
} MCMC_ABI_GUARD

/* pcg32's stream `delta` draws further on, in O(log delta): the LCG state after delta steps is
 * mult^delta * state + inc * (mult^delta - 1)/(mult - 1), both factors formed by repeated squaring
 * (Brown, "Random number generation with arbitrary strides"; what the PCG library calls advance) */
static void pcg32_jump(HostPcg& r, uint64_t delta) {
    uint64_t curMult = 0x5851f42d4c957f2dULL, curPlus = r.inc, accMult = 1, accPlus = 0;
    while (delta > 0) {
        if (delta & 1) { accMult *= curMult; accPlus = accPlus * curMult + curPlus; }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
        delta >>= 1;
    }
    r.state = accMult * r.state + accPlus;
}

int mcmc_host_perturb_ics_from(const double* x0, int32_t dim, int64_t first_chain, int64_t n_chains, double a,
                               uint64_t initstate, uint64_t initseq, double* out) try {
    if (!x0 || !out || dim < 1 || n_chains < 0 || first_chain < 0) return fail(MCMC_HIP_EINVAL, "perturb_ics_from: bad arguments");
    HostPcg r;
    pcg32_seed(r.state, r.inc, initstate, initseq);
    pcg32_jump(r, (uint64_t)first_chain * (uint64_t)dim);            /* one nextUInt per coordinate drawn so far */
    for (int64_t c = 0; c < n_chains; ++c)
        for (int k = 0; k < dim; ++k)
            out[c * dim + k] = x0[k] + (r.next_float() - .5f) * x0[k] * a;   /* mcmc.h:310 */
    return 0;
} MCMC_ABI_GUARD

int32_t mcmc_hip_n_saved(int32_t n_samples, int32_t n_burnin, int32_t n_adjust, int32_t thinning) {
    if (thinning < 1) return 0;
    int32_t n = 0;
    for (int32_t i = 0; i < n_samples + n_adjust; ++i)
        if (i >= n_burnin && (i % thinning) == 0) ++n;               /* mcmc.h:1037 */
    return n;
}

static int init_ctx(mcmc_hip_ctx* c, const mcmc_covid_config* cfg, int device);

int mcmc_hip_create(const mcmc_covid_config* cfg, int device, mcmc_hip_ctx** out) try {
    if (!cfg || !out) return fail(MCMC_HIP_EINVAL, "create: null argument");
    *out = nullptr;
    if (cfg->n_data < 2 || cfg->max_delay < 1 || cfg->n_predict < 0)
        return fail(MCMC_HIP_EINVAL, "create: need n_data >= 2, max_delay >= 1, n_predict >= 0");
    if (cfg->n_disc < 1 || cfg->n_disc > MCMC_MAXDISC || !cfg->disc_days || !cfg->disc_fixed)
        return fail(MCMC_HIP_EINVAL, "create: n_disc must be 1..%d with disc_days/disc_fixed given", MCMC_MAXDISC);
    if (cfg->n_data - cfg->fix_behavior_in_advance < 3)
        return fail(MCMC_HIP_EINVAL, "create: behavior needs at least 3 points (n_data - fix_behavior_in_advance)");
    if (cfg->pop_size < 1) return fail(MCMC_HIP_EINVAL, "create: pop_size must be positive");
    if (cfg->target_kind < MCMC_TARGET_PLAIN || cfg->target_kind > MCMC_TARGET_COOLING)
        return fail(MCMC_HIP_EINVAL, "create: unknown target_kind %d", cfg->target_kind);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MCMC_HIP_ENODEVICE, "no HIP device visible: the hot path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MCMC_HIP_EINVAL, "create: device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    mcmc_hip_ctx* c = new (std::nothrow) mcmc_hip_ctx();
    if (!c) return fail(MCMC_HIP_ENOMEM, "out of host memory");
    int r;
    try { r = init_ctx(c, cfg, device); } catch (...) { mcmc_hip_destroy(c); throw; }
    if (r) { mcmc_hip_destroy(c); return r; }
    *out = c;
    return 0;
} MCMC_ABI_GUARD

/* second half of mcmc_hip_create: everything that can fail once the context exists */
static int init_ctx(mcmc_hip_ctx* c, const mcmc_covid_config* cfg, int device) {
    c->cfg = *cfg;
    c->discDays.assign(cfg->disc_days, cfg->disc_days + cfg->n_disc);
    c->discFixed.assign(cfg->disc_fixed, cfg->disc_fixed + cfg->n_disc);
    c->cfg.disc_days = c->discDays.data();
    c->cfg.disc_fixed = c->discFixed.data();
    c->device = device;
    HIP_TRY(hipDeviceGetAttribute(&c->nCU, hipDeviceAttributeMultiprocessorCount, device));
    if (int r = static_scratch_for(device, &c->staticScratch)) return r;      /* hipSetDevice(device) was just made */
    c->size = cfg->max_delay + cfg->n_predict + cfg->n_data;          /* covid.h:297 */
    c->nB = cfg->n_data - cfg->fix_behavior_in_advance;               /* covid.h:299 */
    c->dim = c->nB + cfg->n_disc + 3;
    c->dimRegion = c->dim;
    c->simDays = cfg->max_delay + cfg->n_data;                        /* what loglike can see */
    c->simPad = c->size + 2 * MCMC_T;
    c->nLag = std::min(c->size, (int)MCMC_HIP_TRAJ_NDAYS);            /* covid.h:378 */
    c->tabRows = c->size + 3 * MCMC_T;
    c->histRows = c->size + 4 * MCMC_T + MCMC_HOFF(MCMC_T);

    auto cleanup = [](int r) { return r; };            /* the caller destroys the context on failure */
    if (hipStreamCreate(&c->stream) != hipSuccess) return cleanup(fail(MCMC_HIP_EHIP, "hipStreamCreate failed"));
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess)
        return cleanup(fail(MCMC_HIP_EHIP, "hipEventCreate failed"));

    /* per-day constants: capacity (covid.h:314) and the piecewise-constant segment of each
     * day (covid.h:432-442) */
    std::vector<double> cap(c->simPad);
    std::vector<int> seg(c->simPad);
    const int md = cfg->max_delay, nD = cfg->n_disc;
    for (int i = 0; i < c->simPad; ++i) {
        cap[i] = cfg->cap0 + std::max(cfg->cap_incr_rate * (i - md), 0.0);
        int found = -1;
        for (int m = 0; m + 1 < nD; ++m)
            if (c->discDays[m] <= i - md && i - md < c->discDays[m + 1]) { found = m; break; }
        if (found == -1 && i - md >= c->discDays[nD - 1]) found = nD - 1;
        seg[i] = found;
    }
    if (int r = c->dCapacity.alloc(c->simPad)) return cleanup(r);
    if (int r = c->dSeg.alloc(c->simPad)) return cleanup(r);
    if (hipMemcpy(c->dCapacity.p, cap.data(), cap.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->dSeg.p, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(MCMC_HIP_EHIP, "upload of per-day constants failed"));

    float nNotFixed = 0;                                              /* counts the FIXED ones, covid.h:560-563 */
    unsigned mask = 0;
    for (int i = 0; i < nD; ++i) if (c->discFixed[i]) { nNotFixed += 1; mask |= 1u << i; }

    DevModel& M = c->M;
    M.nData = cfg->n_data; M.md = md; M.size = c->size; M.nB = c->nB; M.nDisc = nD;
    M.simDays = c->simDays; M.nLagRows = c->nLag; M.hasPrior = cfg->has_prior ? 1 : 0;
    M.tabRows = c->tabRows; M.simPad = c->simPad;
    M.pop = (double)cfg->pop_size; M.probICU = cfg->prob_icu_if_serious;
    M.probLethal = cfg->prob_lethal_daily_untreated;
    M.smoothLs = cfg->smooth_Ls; M.smoothL = cfg->smooth_L; M.dNData = (double)cfg->n_data;
    M.discThr = 2 / nNotFixed;                                        /* covid.h:581 (+inf if none fixed) */
    M.discFixedMask = mask;
    M.capacity = c->dCapacity.p; M.seg = c->dSeg.p;
    M.nRegions = 1; M.dimRegion = c->dimRegion;
    return 0;
}

int mcmc_hip_destroy(mcmc_hip_ctx* c) try {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_set_tables(mcmc_hip_ctx* c, const double* R, int32_t n_lag) try {
    if (int r = check_ctx(c)) return r;
    if (!R) return fail(MCMC_HIP_EINVAL, "set_tables: null table");
    if (n_lag < c->nLag) return fail(MCMC_HIP_EINVAL, "set_tables: n_lag=%d but the model needs %d lags", n_lag, c->nLag);
    HIP_TRY(hipSetDevice(c->device));
    /* interleave by lag so that one aligned 64-byte scalar load brings all curves of a lag;
     * rows >= nLag stay zero: they stand for the d < nDays cut-off of project() (covid.h:378) */
    std::vector<double> pos((size_t)c->tabRows * 8, 0.0), neg((size_t)c->tabRows * 4, 0.0);
    for (int d = 0; d < c->nLag; ++d) {
        for (int k = 0; k < 7; ++k) pos[(size_t)d * 8 + k] = R[(size_t)k * n_lag + d];
        for (int k = 0; k < 3; ++k) neg[(size_t)d * 4 + k] = R[(size_t)(7 + k) * n_lag + d];
        neg[(size_t)d * 4 + 3] = 1.0;                                /* the flat add() of covid.h:387-391 reaches this lag */
    }
    if (int r = c->dRpos.alloc(pos.size())) return r;
    if (int r = c->dRneg.alloc(neg.size())) return r;
    HIP_TRY(hipMemcpy(c->dRpos.p, pos.data(), pos.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dRneg.p, neg.data(), neg.size() * sizeof(double), hipMemcpyHostToDevice));
    c->M.Rpos = c->dRpos.p;
    c->M.Rneg = c->dRneg.p;
    /* fast-mode tables: per lag (MI HI INC MILD SER REC DEAD 0) and (-SFS -RFS 1-DFS 0), every
     * entry formed in double and rounded once; rows >= nLag stay zero.  One buffer: the
     * positive table first, the correction table behind it. */
    const size_t rowsF = (size_t)c->size + 3 * MCMC_TF + 8;     /* the MFMA form stages up to simDays + MCMC_TM + 7 of them */
    std::vector<float> rf(rowsF * 12, 0.0f);
    for (int d = 0; d < c->nLag; ++d) {
        float* rowp = rf.data() + (size_t)d * 8;
        float* rown = rf.data() + rowsF * 8 + (size_t)d * 4;
        for (int k = 0; k < 7; ++k) rowp[k] = (float)R[(size_t)k * n_lag + d];
        rown[0] = (float)(-R[(size_t)7 * n_lag + d]);
        rown[1] = (float)(-R[(size_t)8 * n_lag + d]);
        rown[2] = (float)(1.0 - R[(size_t)9 * n_lag + d]);
    }
    if (int r = c->dRF.alloc(rf.size())) return r;
    HIP_TRY(hipMemcpy(c->dRF.p, rf.data(), rf.size() * sizeof(float), hipMemcpyHostToDevice));
    c->M.RF = c->dRF.p;
    c->M.RFn = c->dRF.p + rowsF * 8;
    c->haveTables = true;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_set_data(mcmc_hip_ctx* c, const double* deaths, const double* sigma) try {
    if (int r = check_ctx(c)) return r;
    if (!deaths || !sigma) return fail(MCMC_HIP_EINVAL, "set_data: null series");
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->cfg.n_data;
    std::vector<double> s2(n);
    for (int i = 0; i < n; ++i) s2[i] = sigma[i] * sigma[i];          /* covid.h:491 */
    if (int r = c->dDeaths.alloc(n)) return r;
    if (int r = c->dSigma2.alloc(n)) return r;
    HIP_TRY(hipMemcpy(c->dDeaths.p, deaths, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dSigma2.p, s2.data(), n * sizeof(double), hipMemcpyHostToDevice));
    c->M.deaths = c->dDeaths.p;
    c->M.sigma2 = c->dSigma2.p;
    c->haveData = true;
    c->multi = false; c->nRegions = 1; c->dim = c->dimRegion;          /* back to the single-region State */
    c->M.nRegions = 1;
    c->haveChains = false;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_set_regions(mcmc_hip_ctx* c, int32_t n_regions, const int32_t* pop_size, const double* cap0,
                         const double* cap_incr_rate, const double* deaths, const double* sigma) try {
    if (int r = check_ctx(c)) return r;
    if (n_regions < 1 || n_regions > 4096 || !pop_size || !cap0 || !cap_incr_rate || !deaths || !sigma)
        return fail(MCMC_HIP_EINVAL, "set_regions: need 1..4096 regions and all five arrays");
    if (c->precision != MCMC_HIP_PRECISION_F64) return fail(MCMC_HIP_EINVAL, "set_regions: the multi-region model runs in double only");
    for (int r = 0; r < n_regions; ++r)
        if (pop_size[r] < 1) return fail(MCMC_HIP_EINVAL, "set_regions: pop_size[%d] must be positive", r);
    HIP_TRY(hipSetDevice(c->device));
    const int R = n_regions, n = c->cfg.n_data;
    std::vector<double> s2((size_t)R * n), pop(R);
    for (size_t i = 0; i < s2.size(); ++i) s2[i] = sigma[i] * sigma[i];            /* covid.h:491 */
    for (int r = 0; r < R; ++r) pop[r] = (double)pop_size[r];
    if (int r = c->dRegDeaths.alloc((size_t)R * n)) return r;
    if (int r = c->dRegSigma2.alloc((size_t)R * n)) return r;
    if (int r = c->dRegPop.alloc(R)) return r;
    if (int r = c->dRegCap0.alloc(R)) return r;
    if (int r = c->dRegCapRate.alloc(R)) return r;
    HIP_TRY(hipMemcpy(c->dRegDeaths.p, deaths, (size_t)R * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dRegSigma2.p, s2.data(), s2.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dRegPop.p, pop.data(), R * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dRegCap0.p, cap0, R * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dRegCapRate.p, cap_incr_rate, R * sizeof(double), hipMemcpyHostToDevice));
    /* the shared LDS image still wants a series (its slots are overridden per region): region 0's */
    if (int r = c->dDeaths.alloc(n)) return r;
    if (int r = c->dSigma2.alloc(n)) return r;
    HIP_TRY(hipMemcpy(c->dDeaths.p, deaths, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->dSigma2.p, s2.data(), n * sizeof(double), hipMemcpyHostToDevice));
    DevModel& M = c->M;
    M.deaths = c->dDeaths.p; M.sigma2 = c->dSigma2.p;
    M.nRegions = R; M.dimRegion = c->dimRegion;
    M.regDeaths = c->dRegDeaths.p; M.regSigma2 = c->dRegSigma2.p; M.regPop = c->dRegPop.p;
    M.regCap0 = c->dRegCap0.p; M.regCapRate = c->dRegCapRate.p;
    c->multi = true; c->nRegions = R; c->dim = R * c->dimRegion;
    c->haveData = true;
    c->haveChains = false;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_set_precision(mcmc_hip_ctx* c, int32_t mode) try {
    if (int r = check_ctx(c)) return r;
    if (mode != MCMC_HIP_PRECISION_F64 && mode != MCMC_HIP_PRECISION_F32_MODEL && mode != MCMC_HIP_PRECISION_F32_PACKED)
        return fail(MCMC_HIP_EINVAL, "set_precision: unknown mode %d", mode);
    if (mode != MCMC_HIP_PRECISION_F64) {
        if (!c->haveTables) return fail(MCMC_HIP_ESTATE, "set_precision: call set_tables first");
        if (!c->cfg.contract) return fail(MCMC_HIP_EINVAL, "set_precision: the fast mode needs cfg.contract = 1");
        if (c->multi) return fail(MCMC_HIP_EINVAL, "set_precision: the multi-region model runs in double only");
    }
    c->precision = mode;
    c->haveChains = false;             /* log-likelihoods of the other arithmetic are not comparable */
    decide_mapping(c);
    return 0;
} MCMC_ABI_GUARD

static int alloc_work(mcmc_hip_ctx* c, long long nP) {
    /* history + LL scratch for nP lanes; contents are fully rewritten by every evaluation */
    if (int r = c->dHn.alloc((size_t)c->histRows * nP)) return r;
    if (int r = c->dHe.alloc((size_t)c->histRows * nP)) return r;
    if (int r = c->dD1s.alloc((size_t)2 * c->cfg.n_data * nP)) return r;
    HIP_TRY(hipMemsetAsync(c->dHn.p, 0, c->dHn.n * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->dHe.p, 0, c->dHe.n * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(c->dD1s.p, 0, c->dD1s.n * sizeof(double), c->stream));
    c->S.hn = c->dHn.p; c->S.he = c->dHe.p; c->S.d1s = c->dD1s.p;
    c->S.histRows = c->histRows; c->S.nTiles = (int)(nP / MCMC_WAVE);
    return 0;
}

int mcmc_hip_set_chains(mcmc_hip_ctx* c, int64_t n_chains, const double* params, const double* stepsizes,
                        const uint64_t* chain_ids, uint64_t first_chain_id) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveTables || !c->haveData) return fail(MCMC_HIP_ESTATE, "set_chains: call set_tables and set_data first");
    if (n_chains < 1 || !params) return fail(MCMC_HIP_EINVAL, "set_chains: need n_chains >= 1 and params");
    HIP_TRY(hipSetDevice(c->device));
    const long long n = n_chains;
    const long long nP = (n + MCMC_WAVE - 1) / MCMC_WAVE * MCMC_WAVE;
    const int dim = c->dim;
    c->haveChains = false;
    c->nChains = n; c->nP = nP; c->nTiles = (int)(nP / MCMC_WAVE);
    if (int r = c->dParams.alloc((size_t)2 * dim * nP)) return r;
    const int R = c->nRegions;
    if (int r = c->dLL.alloc((size_t)2 * R * nP)) return r;
    if (int r = c->dStepsize.alloc((size_t)R * nP)) return r;
    if (c->multi) { if (int r = c->dLLProp.alloc((size_t)R * nP)) return r; }
    if (int r = c->dRng.alloc(2 * nP)) return r;
    if (int r = c->dNAccept.alloc(nP)) return r;
    if (int r = c->dPairs.alloc(2 * nP)) return r;
    if (int r = c->dMean.alloc((size_t)dim * nP)) return r;
    c->dRecLL.release(); c->dRecSamples.release();
    c->nSaved = 0; c->nSavedCap = 0; c->recSamples = false;
    ChainState& S = c->S;
    S.params = c->dParams.p; S.prop = c->dParams.p + (size_t)dim * nP; S.ll = c->dLL.p; S.stepsize = c->dStepsize.p;
    S.rng = c->dRng.p; S.nAccept = c->dNAccept.p; S.pairs = c->dPairs.p; S.mean = c->dMean.p;
    S.recLL = nullptr; S.recSamples = nullptr; S.nChains = n; S.dim = dim;
    S.chainBeta = nullptr; c->ladderK = 0; c->nLadders = 0;         /* a new set of chains has no ladder */
    S.llProp = c->multi ? c->dLLProp.p : nullptr;
    c->headCounter = 0;
    if (int r = alloc_work(c, nP)) return r;

    /* upload chain-major params and transpose to SoA on the device */
    DevBuf<double> stage;
    if (int r = stage.alloc((size_t)n * dim)) return r;
    HIP_TRY(hipMemcpyAsync(stage.p, params, (size_t)n * dim * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const unsigned tb = 256, tg = (unsigned)((nP + tb - 1) / tb);
    hipLaunchKernelGGL(tile_in_kernel<double>, dim3(tg), dim3(tb), 0, c->stream, (const double*)stage.p, S.params, n, dim, nP);
    HIP_TRY(hipGetLastError());
    if (c->multi && R > 1) {           /* the shared betas are region 0's: mirror them into every block */
        hipLaunchKernelGGL(mirror_betas_kernel, dim3(tg), dim3(tb), 0, c->stream, S.params, nP, R, c->dimRegion, c->nB + c->cfg.n_disc);
        HIP_TRY(hipGetLastError());
    }
    /* MetropolisChain::mean starts as a deep copy of the state (mcmc.h:1002) */
    HIP_TRY(hipMemcpyAsync(c->dMean.p, S.params, (size_t)dim * nP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(S.params + (size_t)dim * nP, 0, (size_t)dim * nP * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(S.nAccept, 0, nP * sizeof(int), c->stream));
    HIP_TRY(hipMemsetAsync(S.pairs, 0, 2 * nP * sizeof(unsigned long long), c->stream));

    std::vector<double> ss((size_t)R * nP, 1.0);                    /* [tile][R][64]: every region starts at the chain's */
    std::vector<uint64_t> rng(2 * nP);
    for (long long i = 0; i < nP; ++i) {
        const long long src = i < n ? i : n - 1;
        if (stepsizes)
            for (int q = 0; q < R; ++q) ss[((size_t)(i / MCMC_WAVE) * R + q) * MCMC_WAVE + (size_t)(i % MCMC_WAVE)] = stepsizes[src];
        const uint64_t id = chain_ids ? chain_ids[src] : first_chain_id + (uint64_t)src;
        const size_t t0 = (size_t)(i / MCMC_WAVE) * 2 * MCMC_WAVE + (size_t)(i % MCMC_WAVE);
        pcg32_seed(rng[t0], rng[t0 + MCMC_WAVE], 0, id);             /* rnd(0, chainid), mcmc.h:999 */
    }
    HIP_TRY(hipMemcpyAsync(S.stepsize, ss.data(), ss.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(S.rng, rng.data(), 2 * nP * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    if (int r = run_eval_current(c)) return r;                       /* State::eval, mcmc.h:385-393 */
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->haveChains = true;
    c->stepsDone = 0;
    decide_mapping(c);
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_set_mapping(mcmc_hip_ctx* c, int32_t mapping) try {
    if (int r = check_ctx(c)) return r;
    if (mapping != MCMC_HIP_MAPPING_AUTO && mapping != MCMC_HIP_MAPPING_LANE && mapping != MCMC_HIP_MAPPING_WAVE)
        return fail(MCMC_HIP_EINVAL, "set_mapping: unknown mapping %d", mapping);
    if (mapping == MCMC_HIP_MAPPING_WAVE) {
        const char* why = nullptr;
        if (!wave_mapping_possible(c, &why)) return fail(MCMC_HIP_EINVAL, "set_mapping: %s", why);
    }
    c->mapping = mapping;
    decide_mapping(c);
    return 0;
} MCMC_ABI_GUARD

int32_t mcmc_hip_get_mapping(const mcmc_hip_ctx* c) {
    if (!c) return MCMC_HIP_EINVAL;
    return c->useWave ? MCMC_HIP_MAPPING_WAVE : MCMC_HIP_MAPPING_LANE;
}

int mcmc_hip_set_state(mcmc_hip_ctx* c, const uint64_t* rng_state, const double* loglike, const double* mean,
                       const int32_t* accept_counts) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "set_state: call set_chains first");
    HIP_TRY(hipSetDevice(c->device));
    if (rng_state) {
        for (long long i = 0; i < c->nChains; ++i)
            if (!(rng_state[2 * i + 1] & 1u)) return fail(MCMC_HIP_EINVAL, "set_state: chain %lld: a pcg32 increment is odd by construction", i);
        if (int r = upload_field<uint64_t>(c, rng_state, 2, c->S.rng)) return r;
    }
    if (loglike) { if (int r = upload_field<double>(c, loglike, 2 * c->nRegions, c->S.ll)) return r; }
    if (mean) { if (int r = upload_field<double>(c, mean, c->dim, c->dMean.p)) return r; }
    if (accept_counts) { if (int r = upload_field<int>(c, accept_counts, 1, c->S.nAccept)) return r; }
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_get_scan_position(mcmc_hip_ctx* c, int64_t* out) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "get_scan_position: call set_chains first");
    if (!out) return fail(MCMC_HIP_EINVAL, "null output buffer");
    *out = (int64_t)c->headCounter;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_set_region_state(mcmc_hip_ctx* c, int64_t scan_position, const double* region_stepsizes) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "set_region_state: call set_chains first");
    if (scan_position < 0) return fail(MCMC_HIP_EINVAL, "set_region_state: the scan position counts steps, it cannot be negative");
    HIP_TRY(hipSetDevice(c->device));
    if (region_stepsizes) {
        const size_t cnt = (size_t)c->nChains * c->nRegions;
        for (size_t i = 0; i < cnt; ++i)
            if (!(region_stepsizes[i] > 0) || !std::isfinite(region_stepsizes[i]))
                return fail(MCMC_HIP_EINVAL, "set_region_state: step size %zu is not a positive finite number", i);
        if (int r = upload_field<double>(c, region_stepsizes, c->nRegions, c->S.stepsize)) return r;
    }
    c->headCounter = (long long)scan_position;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_step(mcmc_hip_ctx* c, int32_t n_steps, double beta, int32_t random_subspace) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "step: call set_chains first");
    if (n_steps < 0) return fail(MCMC_HIP_EINVAL, "step: n_steps < 0");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<double> betas(n_steps, beta);
    std::vector<int> flags(n_steps, random_subspace ? SF_RANDOM : 0);
    return run_schedule(c, betas, flags, 0, (size_t)n_steps, c->useWave ? MCMC_WAVE_STEPS_PER_LAUNCH : MCMC_STEPS_PER_LAUNCH, nullptr);
} MCMC_ABI_GUARD

/* ---- tempering ladder (extension) ---- */
int mcmc_hip_set_ladder(mcmc_hip_ctx* c, int32_t n_temps, const double* temps, uint64_t swap_seed, uint64_t first_ladder) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "set_ladder: call set_chains first");
    if (n_temps == 0) { c->S.chainBeta = nullptr; c->ladderK = 0; c->nLadders = 0; return 0; }
    if (c->multi) return fail(MCMC_HIP_EINVAL, "set_ladder: the tempering ladder takes the single-region State");
    if (n_temps < 2 || !temps) return fail(MCMC_HIP_EINVAL, "set_ladder: need n_temps >= 2 and temps");
    if (c->nChains % n_temps != 0) return fail(MCMC_HIP_EINVAL, "set_ladder: n_chains must be a multiple of n_temps");
    for (int r = 0; r < n_temps; ++r)
        if (!(temps[r] > 0) || !std::isfinite(temps[r])) return fail(MCMC_HIP_EINVAL, "set_ladder: temperatures must be positive and finite");
    HIP_TRY(hipSetDevice(c->device));
    const int K = n_temps;
    const long long n = c->nChains, nP = c->nP, nL = n / K;
    std::vector<double> lb(K), cb(nP, 1.0);
    for (int r = 0; r < K; ++r) lb[r] = 1 / temps[r];                /* TempTarget: bet = 1/T, mcmc.h:797-808 */
    std::vector<int> at(n), rung(nP, 0);
    std::vector<uint64_t> rng(2 * nL);
    for (long long i = 0; i < n; ++i) { at[i] = (int)i; rung[i] = (int)(i % K); cb[i] = lb[i % K]; }
    for (long long L = 0; L < nL; ++L) pcg32_seed(rng[2 * L], rng[2 * L + 1], swap_seed, first_ladder + (uint64_t)L);
    if (int r = c->dLadderBeta.alloc(K)) return r;
    if (int r = c->dChainBeta.alloc(nP)) return r;
    if (int r = c->dChainAt.alloc(n)) return r;
    if (int r = c->dRungOf.alloc(nP)) return r;
    if (int r = c->dSwapRng.alloc(2 * nL)) return r;
    if (int r = c->dSwapAcc.alloc(K)) return r;
    HIP_TRY(hipMemcpyAsync(c->dLadderBeta.p, lb.data(), K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->dChainBeta.p, cb.data(), nP * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->dChainAt.p, at.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->dRungOf.p, rung.data(), nP * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->dSwapRng.p, rng.data(), 2 * nL * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->dSwapAcc.p, 0, K * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->S.chainBeta = c->dChainBeta.p;
    c->ladderK = K; c->nLadders = nL;
    c->swapRounds[0] = c->swapRounds[1] = 0;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_swap(mcmc_hip_ctx* c, int32_t parity) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains || c->ladderK < 2) return fail(MCMC_HIP_ESTATE, "swap: call set_ladder first");
    if (parity != 0 && parity != 1) return fail(MCMC_HIP_EINVAL, "swap: parity must be 0 or 1");
    HIP_TRY(hipSetDevice(c->device));
    const unsigned grid = (unsigned)((c->nLadders + 255) / 256);
    hipLaunchKernelGGL(ladder_swap_kernel, dim3(grid), dim3(256), 0, c->stream, c->S, c->M.hasPrior, c->ladderK, c->nLadders,
                       (const double*)c->dLadderBeta.p, c->dChainAt.p, c->dRungOf.p, c->dSwapRng.p, c->dChainBeta.p,
                       (int)parity, c->dSwapAcc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->swapRounds[parity] += 1;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_run_tempered(mcmc_hip_ctx* c, int32_t n_rounds, int32_t steps_per_round) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains || c->ladderK < 2) return fail(MCMC_HIP_ESTATE, "run_tempered: call set_ladder first");
    if (n_rounds < 0 || steps_per_round < 0) return fail(MCMC_HIP_EINVAL, "run_tempered: negative count");
    for (int k = 0; k < n_rounds; ++k) {
        if (int r = mcmc_hip_step(c, steps_per_round, 1.0, 1)) return r;
        if (int r = mcmc_hip_swap(c, k & 1)) return r;
    }
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_get_rungs(mcmc_hip_ctx* c, int32_t* out) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains || c->ladderK < 2) return fail(MCMC_HIP_ESTATE, "get_rungs: call set_ladder first");
    if (!out) return fail(MCMC_HIP_EINVAL, "null output buffer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(out, c->dRungOf.p, c->nChains * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_get_swap_counts(mcmc_hip_ctx* c, int64_t* accepted, int64_t* proposed) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains || c->ladderK < 2) return fail(MCMC_HIP_ESTATE, "get_swap_counts: call set_ladder first");
    if (!accepted || !proposed) return fail(MCMC_HIP_EINVAL, "null output buffer");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<unsigned long long> h(c->ladderK);
    HIP_TRY(hipMemcpy(h.data(), c->dSwapAcc.p, c->ladderK * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int r = 0; r + 1 < c->ladderK; ++r) {
        accepted[r] = (int64_t)h[r];
        proposed[r] = (int64_t)(c->swapRounds[r & 1] * (unsigned long long)c->nLadders);
    }
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_run(mcmc_hip_ctx* c, int32_t nSamples, int32_t nBurnin, int32_t nAdjust, int32_t thinning,
                 int32_t record_samples, int32_t compute_mean) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "run: call set_chains first");
    if (nSamples < 0 || nBurnin < 0 || nAdjust < 0 || thinning < 1)
        return fail(MCMC_HIP_EINVAL, "run: need n_samples,n_burnin,n_adjust >= 0 and thinning >= 1");
    HIP_TRY(hipSetDevice(c->device));
    const long long nP = c->nP;
    /* the step schedule of MetropolisChain::run (mcmc.h:1023-1127) */
    std::vector<double> betas;
    std::vector<int> flags;
    betas.reserve((size_t)20 * nAdjust + nSamples);
    flags.reserve((size_t)20 * nAdjust + nSamples);
    size_t adjustEnd = 0;
    bool first = true;
    for (int i = 0; i < nSamples + nAdjust; ++i) {
        int pre = 0;
        if (first) { pre |= SF_RESET_ACC; first = false; }            /* int nAccept = 0, mcmc.h:1007 */
        if (i == nAdjust) pre |= SF_RESET_ACC;                         /* mcmc.h:1032 */
        if (i >= nBurnin && (i % thinning) == 0) pre |= SF_RECORD;     /* mcmc.h:1037 */
        if (i > nBurnin && compute_mean) pre |= SF_MEAN;               /* mcmc.h:1064 */
        if (i < nAdjust) {
            const double b = target_beta(c->cfg, (double)i / nAdjust); /* mcmc.h:1091 */
            for (int j = 0; j < 20; ++j) {                             /* nRepeat, mcmc.h:1096 */
                int f = (j == 0) ? (pre | SF_RANDOM | SF_RESET_ACC) : 0;
                if (j == 19) f |= SF_ADJUST_END;
                betas.push_back(b);
                flags.push_back(f);
            }
            adjustEnd = betas.size();
        } else {
            betas.push_back(target_beta(c->cfg, (double)(i - nAdjust) / nSamples)); /* mcmc.h:1126 */
            flags.push_back(pre | SF_RANDOM);
        }
    }
    const int nSaved = mcmc_hip_n_saved(nSamples, nBurnin, nAdjust, thinning);
    c->nSaved = 0;
    c->recSamples = record_samples != 0;
    if (int r = c->dRecLL.alloc((size_t)std::max(nSaved, 1) * nP)) return r;
    c->S.recLL = c->dRecLL.p;
    if (record_samples) {
        if (int r = c->dRecSamples.alloc((size_t)std::max(nSaved, 1) * c->dim * nP)) return r;
        c->S.recSamples = c->dRecSamples.p;
    } else {
        c->dRecSamples.release();
        c->S.recSamples = nullptr;
    }
    c->nSavedCap = nSaved;
    c->S.mean = compute_mean ? c->dMean.p : nullptr;
    c->adjustS = c->samplingS = c->kernelS = 0;
    c->kernelLaunches = 0;
    /* launches are cut so that a single kernel stays well below a second */
    const int perLaunch = c->useWave ? MCMC_WAVE_STEPS_PER_LAUNCH : MCMC_STEPS_PER_LAUNCH;
    if (int r = run_schedule(c, betas, flags, 0, adjustEnd, perLaunch, &c->adjustS)) return r;
    if (int r = run_schedule(c, betas, flags, adjustEnd, betas.size(), perLaunch, &c->samplingS)) return r;
    if (compute_mean) {                                                /* mcmc.h:1133-1134 (1/0 = inf there too) */
        const long long count = (long long)c->dim * nP;
        const double f = 1.0 / (double)(nAdjust + nSamples - nBurnin);
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, c->dMean.p, count, f);
        HIP_TRY(hipGetLastError());
    }
    c->S.mean = c->dMean.p;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->nSaved = nSaved;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_dim(const mcmc_hip_ctx* c) { return c ? c->dim : MCMC_HIP_EINVAL; }
int mcmc_hip_size(const mcmc_hip_ctx* c) { return c ? c->size : MCMC_HIP_EINVAL; }
int64_t mcmc_hip_n_chains(const mcmc_hip_ctx* c) { return c ? c->nChains : MCMC_HIP_EINVAL; }

}  /* extern "C" */

/* copies a tile-major [nTiles][rows][64] field out as [n][rows] */
template <typename Tp>
static int copy_field(mcmc_hip_ctx* c, const Tp* dev, int rows, Tp* out) {
    const long long nP = c->nP, n = c->nChains;
    std::vector<Tp> h((size_t)rows * nP);
    HIP_TRY(hipMemcpy(h.data(), dev, h.size() * sizeof(Tp), hipMemcpyDeviceToHost));
    for (long long i = 0; i < n; ++i)
        for (int r = 0; r < rows; ++r)
            out[i * rows + r] = h[((size_t)(i / MCMC_WAVE) * rows + r) * MCMC_WAVE + (size_t)(i % MCMC_WAVE)];
    return 0;
}
/* copies a flat [rows][nP] field out as [n][rows] */
template <typename Tp>
static int copy_rows(mcmc_hip_ctx* c, const Tp* dev, int rows, Tp* out) {
    const long long nP = c->nP, n = c->nChains;
    std::vector<Tp> h((size_t)rows * nP);
    HIP_TRY(hipMemcpy(h.data(), dev, h.size() * sizeof(Tp), hipMemcpyDeviceToHost));
    for (long long i = 0; i < n; ++i)
        for (int r = 0; r < rows; ++r) out[i * rows + r] = h[(size_t)r * nP + i];
    return 0;
}

extern "C" {

static int need_chains(mcmc_hip_ctx* c, const void* out) {
    if (int r = check_ctx(c)) return r;
    if (!c->haveChains) return fail(MCMC_HIP_ESTATE, "no chains: call set_chains first");
    if (!out) return fail(MCMC_HIP_EINVAL, "null output buffer");
    if (hipSetDevice(c->device) != hipSuccess) return fail(MCMC_HIP_EHIP, "hipSetDevice failed");
    return 0;
}

static int gather_soa(mcmc_hip_ctx* c, const double* src, double* out) {
    const long long n = c->nChains;
    DevBuf<double> stage;
    if (int r = stage.alloc((size_t)n * c->dim)) return r;
    const unsigned tb = 256, tg = (unsigned)((n + tb - 1) / tb);
    hipLaunchKernelGGL(tile_out_kernel<double>, dim3(tg), dim3(tb), 0, c->stream, src, stage.p, n, c->dim);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, stage.p, (size_t)n * c->dim * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int mcmc_hip_get_params(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    return gather_soa(c, c->S.params, out);
} MCMC_ABI_GUARD
int mcmc_hip_get_loglike(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    if (c->nRegions == 1) return copy_field<double>(c, c->S.ll, 2, out);
    std::vector<double> all((size_t)c->nChains * 2 * c->nRegions);     /* summed in State order */
    if (int r = copy_field<double>(c, c->S.ll, 2 * c->nRegions, all.data())) return r;
    for (long long i = 0; i < c->nChains; ++i) {
        double m = 0, p = 0;
        for (int q = 0; q < c->nRegions; ++q) { m += all[((size_t)i * c->nRegions + q) * 2]; p += all[((size_t)i * c->nRegions + q) * 2 + 1]; }
        out[2 * i] = m; out[2 * i + 1] = p;
    }
    return 0;
} MCMC_ABI_GUARD
int mcmc_hip_get_region_loglike(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    return copy_field<double>(c, c->S.ll, 2 * c->nRegions, out);
} MCMC_ABI_GUARD
int mcmc_hip_get_stepsizes(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    if (c->nRegions == 1) return copy_field<double>(c, c->S.stepsize, 1, out);
    std::vector<double> all((size_t)c->nChains * c->nRegions);
    if (int r = copy_field<double>(c, c->S.stepsize, c->nRegions, all.data())) return r;
    for (long long i = 0; i < c->nChains; ++i) out[i] = all[(size_t)i * c->nRegions];      /* region 0's */
    return 0;
} MCMC_ABI_GUARD
int mcmc_hip_get_region_stepsizes(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    return copy_field<double>(c, c->S.stepsize, c->nRegions, out);
} MCMC_ABI_GUARD
int32_t mcmc_hip_n_regions(const mcmc_hip_ctx* c) { return c ? c->nRegions : MCMC_HIP_EINVAL; }
int mcmc_hip_get_accept_counts(mcmc_hip_ctx* c, int32_t* out) try {
    if (int r = need_chains(c, out)) return r;
    return copy_field<int>(c, c->S.nAccept, 1, out);
} MCMC_ABI_GUARD
int mcmc_hip_get_rng_state(mcmc_hip_ctx* c, uint64_t* out) try {
    if (int r = need_chains(c, out)) return r;
    return copy_field<uint64_t>(c, c->S.rng, 2, out);
} MCMC_ABI_GUARD
int mcmc_hip_get_loglikes(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    if (c->nSaved == 0) return fail(MCMC_HIP_ESTATE, "Requested loglikes, but none are present");
    return copy_rows<double>(c, c->S.recLL, c->nSaved, out);
} MCMC_ABI_GUARD
int mcmc_hip_get_samples(mcmc_hip_ctx* c, int64_t chain, int32_t coord_first, int32_t coord_count, double* out) try {
    if (int r = need_chains(c, out)) return r;
    if (!c->recSamples || c->nSaved == 0) return fail(MCMC_HIP_ESTATE, "Requested samples, but none are present");
    if (chain < 0 || chain >= c->nChains || coord_first < 0 || coord_count < 1 || coord_first + coord_count > c->dim)
        return fail(MCMC_HIP_EINVAL, "get_samples: chain/coordinate range out of bounds");
    /* element (s,k) sits at ((s*nTiles + tile)*dim + k)*64 + lane: gathered on the device, one copy out */
    const long long count = (long long)c->nSaved * coord_count;
    DevBuf<double> stage;
    if (int r = stage.alloc((size_t)count)) return r;
    hipLaunchKernelGGL(chain_samples_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream,
                       (const double*)c->S.recSamples, stage.p, c->nSaved, c->nTiles, c->dim, (long long)(chain / MCMC_WAVE),
                       (int)(chain % MCMC_WAVE), (int)coord_first, (int)coord_count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, stage.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
} MCMC_ABI_GUARD
int mcmc_hip_get_sample_all(mcmc_hip_ctx* c, int32_t saved_index, double* out) try {
    if (int r = need_chains(c, out)) return r;
    if (!c->recSamples || saved_index < 0 || saved_index >= c->nSaved)
        return fail(MCMC_HIP_EINVAL, "get_sample_all: no such saved sample");
    return gather_soa(c, c->S.recSamples + (size_t)saved_index * c->dim * c->nP, out);
} MCMC_ABI_GUARD
int mcmc_hip_get_mean(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    return gather_soa(c, c->dMean.p, out);
} MCMC_ABI_GUARD

/* shared by eval and derive: external vectors on their own lane set */
static int with_external(mcmc_hip_ctx* c, int64_t n, const double* params, bool derive, double* out) {
    if (int r = check_ctx(c)) return r;
    if (!c->haveTables || !c->haveData) return fail(MCMC_HIP_ESTATE, "call set_tables and set_data first");
    if (n < 1 || !params || !out) return fail(MCMC_HIP_EINVAL, "need n >= 1, params and an output buffer");
    if (derive && c->nRegions > 1)
        return fail(MCMC_HIP_EINVAL, "derive: one region at a time (a single-region context with that region's data)");
    HIP_TRY(hipSetDevice(c->device));
    const long long nP = (n + MCMC_WAVE - 1) / MCMC_WAVE * MCMC_WAVE;
    const int dim = c->dim;
    const int R = c->nRegions;
    /* a temporary lane set so that chain state is untouched */
    mcmc_hip_ctx tmp;
    tmp.multi = c->multi && !derive; tmp.nRegions = R; tmp.dimRegion = c->dimRegion; tmp.precision = c->precision;
    tmp.cfg = c->cfg; tmp.device = c->device; tmp.nCU = c->nCU; tmp.stream = c->stream; tmp.size = c->size;
    tmp.simPad = c->simPad; tmp.histRows = c->histRows; tmp.dim = dim; tmp.nP = nP; tmp.nChains = n;
    tmp.M = c->M; tmp.nTiles = (int)(nP / MCMC_WAVE);
    int rc = 0;
    DevBuf<double> stage, soa, res;
    do {
        if ((rc = alloc_work(&tmp, nP))) break;
        if ((rc = stage.alloc((size_t)n * dim))) break;
        if ((rc = soa.alloc((size_t)dim * nP))) break;
        if ((rc = res.alloc(derive ? (size_t)9 * c->size * nP : (size_t)2 * R * nP))) break;
        tmp.S.dim = dim; tmp.S.nChains = n;
        hipError_t e = hipMemcpyAsync(stage.p, params, (size_t)n * dim * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = fail(MCMC_HIP_EHIP, "upload failed: %s", hipGetErrorString(e)); break; }
        const unsigned tb = 256, tg = (unsigned)((nP + tb - 1) / tb);
        hipLaunchKernelGGL(tile_in_kernel<double>, dim3(tg), dim3(tb), 0, c->stream, (const double*)stage.p, soa.p, (long long)n, dim, nP);
        e = hipGetLastError();
        if (e != hipSuccess) { rc = fail(MCMC_HIP_EHIP, "tile_in launch failed: %s", hipGetErrorString(e)); break; }
        if (tmp.multi && R > 1) {
            hipLaunchKernelGGL(mirror_betas_kernel, dim3(tg), dim3(tb), 0, c->stream, soa.p, nP, R, c->dimRegion, c->nB + c->cfg.n_disc);
            e = hipGetLastError();
            if (e != hipSuccess) { rc = fail(MCMC_HIP_EHIP, "mirror_betas launch failed: %s", hipGetErrorString(e)); break; }
        }
        if (derive) {
            if (c->cfg.contract) rc = launch_chain_kernel(&tmp, derive_kernel<MCMC_T_DERIVE, true>, true, MCMC_T_DERIVE, tmp.M, tmp.S, (const double*)soa.p, res.p);
            else rc = launch_chain_kernel(&tmp, derive_kernel<MCMC_T_DERIVE, false>, true, MCMC_T_DERIVE, tmp.M, tmp.S, (const double*)soa.p, res.p);
        } else if (tmp.multi) {
            if (c->cfg.contract) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true, 0, true>, false, MCMC_T, tmp.M, tmp.S, (const double*)soa.p, res.p);
            else rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, false, 0, true>, false, MCMC_T, tmp.M, tmp.S, (const double*)soa.p, res.p);
        } else {
            if (c->precision == MCMC_HIP_PRECISION_F32_MODEL) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true, 2>, false, MCMC_T, tmp.M, tmp.S, (const double*)soa.p, res.p);
            else if (c->precision == MCMC_HIP_PRECISION_F32_PACKED) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true, 1>, false, MCMC_T, tmp.M, tmp.S, (const double*)soa.p, res.p);
            else if (c->cfg.contract) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true>, false, MCMC_T, tmp.M, tmp.S, (const double*)soa.p, res.p);
            else rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, false>, false, MCMC_T, tmp.M, tmp.S, (const double*)soa.p, res.p);
        }
        if (rc) break;
        const int rows = derive ? 9 * c->size : 2 * R;
        std::vector<double> h((size_t)rows * nP);
        e = hipMemcpyAsync(h.data(), res.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(MCMC_HIP_EHIP, "kernel or download failed: %s", hipGetErrorString(e)); break; }
        auto at = [&](long long i, int r) { return h[((size_t)(i / MCMC_WAVE) * rows + r) * MCMC_WAVE + (size_t)(i % MCMC_WAVE)]; };
        if (!derive && R > 1) {                      /* (sum of the regions' models, sum of their priors), State order */
            for (long long i = 0; i < n; ++i) {
                double m = 0, p = 0;
                for (int q = 0; q < R; ++q) { m += at(i, 2 * q); p += at(i, 2 * q + 1); }
                out[2 * i] = m; out[2 * i + 1] = p;
            }
        } else {
            for (long long i = 0; i < n; ++i)
                for (int r = 0; r < rows; ++r) out[i * rows + r] = at(i, r);
        }
    } while (0);
    tmp.stream = nullptr;      /* borrowed */
    return rc;
}

int mcmc_hip_eval(mcmc_hip_ctx* c, int64_t n, const double* params, double* ll_out) try {
    return with_external(c, n, params, false, ll_out);
} MCMC_ABI_GUARD
int mcmc_hip_derive(mcmc_hip_ctx* c, int64_t n, const double* params, double* curves_out) try {
    return with_external(c, n, params, true, curves_out);
} MCMC_ABI_GUARD

/* rows of 64-lane doubles one tile of mcmc_hip_eval_static needs in the static scratch */
static size_t static_rows_per_tile(const mcmc_hip_ctx* c) {
    return (size_t)c->dim + 2 * (size_t)c->histRows + 2 * (size_t)c->cfg.n_data + 2;
}

int64_t mcmc_hip_eval_static_max(const mcmc_hip_ctx* c) {
    if (!c) return MCMC_HIP_EINVAL;
    const size_t tiles = (size_t)MCMC_STATIC_SCRATCH_BYTES / (static_rows_per_tile(c) * MCMC_WAVE * sizeof(double));
    return (int64_t)(tiles * MCMC_WAVE);
}

int mcmc_hip_eval_static(mcmc_hip_ctx* c, int64_t n, const double* params_dev, double* ll_out_dev) try {
    if (int r = check_ctx(c)) return r;
    if (!c->haveTables || !c->haveData) return fail(MCMC_HIP_ESTATE, "call set_tables and set_data first");
    if (n < 1 || !params_dev || !ll_out_dev) return fail(MCMC_HIP_EINVAL, "need n >= 1, device params and a device output buffer");
    if (c->multi) return fail(MCMC_HIP_EINVAL, "eval_static: single-region contexts only");
    const int64_t nMax = mcmc_hip_eval_static_max(c);
    if (n > nMax)
        return fail(MCMC_HIP_EINVAL, "eval_static: %lld parameter vectors exceed the static scratch (%u bytes hold at most %lld "
                    "vectors of this model); use mcmc_hip_eval", (long long)n, (unsigned)MCMC_STATIC_SCRATCH_BYTES, (long long)nMax);
    if (!c->staticScratch) return fail(MCMC_HIP_ESTATE, "eval_static: no static scratch address for this device");
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> busy(g_scratch_busy[c->device]);      /* until this call's kernels have finished */
    const long long nP = (n + MCMC_WAVE - 1) / MCMC_WAVE * MCMC_WAVE;
    const int dim = c->dim;
    /* carve this device's array: [soa | hn | he | d1s | res], each tile-major for nP lanes; the
     * array keeps the last call's contents, so everything this call uses starts at zero, as the
     * freshly allocated buffers of mcmc_hip_eval do */
    double* base = c->staticScratch;
    double* soa = base;
    double* hn = soa + (size_t)dim * nP;
    double* he = hn + (size_t)c->histRows * nP;
    double* d1s = he + (size_t)c->histRows * nP;
    double* res = d1s + (size_t)2 * c->cfg.n_data * nP;
    const size_t used = static_rows_per_tile(c) * (size_t)nP * sizeof(double);
    HIP_TRY(hipMemsetAsync(base, 0, used, c->stream));
    ChainState S{};
    S.hn = hn; S.he = he; S.d1s = d1s; S.histRows = c->histRows; S.nTiles = (int)(nP / MCMC_WAVE);
    S.dim = dim; S.nChains = n;
    mcmc_hip_ctx tmp;                      /* launch geometry only: borrows the stream, owns nothing */
    tmp.cfg = c->cfg; tmp.device = c->device; tmp.nCU = c->nCU; tmp.stream = c->stream; tmp.M = c->M;
    tmp.precision = c->precision;
    tmp.nTiles = S.nTiles; tmp.nP = nP; tmp.nChains = n; tmp.dim = dim;
    const unsigned tb = 256, tg = (unsigned)((nP + tb - 1) / tb);
    hipLaunchKernelGGL(tile_in_kernel<double>, dim3(tg), dim3(tb), 0, c->stream, params_dev, soa, (long long)n, dim, nP);
    int rc = 0;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(MCMC_HIP_EHIP, "tile_in launch failed: %s", hipGetErrorString(e));
    if (!rc) {
        if (c->precision == MCMC_HIP_PRECISION_F32_MODEL) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true, 2>, false, MCMC_T, tmp.M, S, (const double*)soa, res);
        else if (c->precision == MCMC_HIP_PRECISION_F32_PACKED) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true, 1>, false, MCMC_T, tmp.M, S, (const double*)soa, res);
        else if (c->cfg.contract) rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, true>, false, MCMC_T, tmp.M, S, (const double*)soa, res);
        else rc = launch_chain_kernel(&tmp, eval_kernel<MCMC_T, false>, false, MCMC_T, tmp.M, S, (const double*)soa, res);
    }
    if (!rc) {
        const unsigned og = (unsigned)((n + tb - 1) / tb);
        hipLaunchKernelGGL(tile_out_kernel<double>, dim3(og), dim3(tb), 0, c->stream, (const double*)res, ll_out_dev, (long long)n, 2);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(MCMC_HIP_EHIP, "eval_static: kernel failed: %s", hipGetErrorString(e));
    }
    tmp.stream = nullptr;                  /* borrowed */
    return rc;
} MCMC_ABI_GUARD

int mcmc_hip_diagnostics(mcmc_hip_ctx* c, double* out) try {
    if (int r = need_chains(c, out)) return r;
    const int len = 4 + 2 * c->dim;
    DevBuf<double> d;
    if (int r = d.alloc(len)) return r;
    HIP_TRY(hipMemsetAsync(d.p, 0, len * sizeof(double), c->stream));
    hipLaunchKernelGGL(diagnostics_kernel, dim3((unsigned)c->nTiles), dim3(MCMC_WAVE), 0, c->stream, c->M, c->S, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d.p, len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    out[1] = (double)c->stepsDone;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_last_timing(mcmc_hip_ctx* c, double* adjust_s, double* sampling_s, double* step_kernel_s,
                         int64_t* step_kernel_launches) try {
    if (int r = check_ctx(c)) return r;
    if (adjust_s) *adjust_s = c->adjustS;
    if (sampling_s) *sampling_s = c->samplingS;
    if (step_kernel_s) *step_kernel_s = c->kernelS;
    if (step_kernel_launches) *step_kernel_launches = c->kernelLaunches;
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_debug_phase_cycles(mcmc_hip_ctx* c, double* out, int32_t reset) try {
    if (int r = check_ctx(c)) return r;
    if (!out) return fail(MCMC_HIP_EINVAL, "null output buffer");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long h[16];
    HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_cycles), sizeof(h)));
    for (int k = 0; k < 16; ++k) out[k] = (double)h[k];
    if (reset) {
        memset(h, 0, sizeof(h));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), h, sizeof(h)));
    }
    return 0;
} MCMC_ABI_GUARD

/* -DMCMC_F32_CHECK builds of the fast mode: out[4] = kind (0: no violation), low row, high row, bound of the
 * first row range that fell outside its buffer (mcmc_model_f32.hip.h) */
int mcmc_hip_debug_f32_check(mcmc_hip_ctx* c, int32_t* out) try {
    if (int r = check_ctx(c)) return r;
    if (!out) return fail(MCMC_HIP_EINVAL, "null output buffer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32_check), 4 * sizeof(int)));
    return 0;
} MCMC_ABI_GUARD

/* -DMCMC_STAMPS builds: per tile of the last launch, the wave's cycles, its (XCC_ID << 32 | HW_ID), its lag chunks
 * with e-terms, the days on which any lane had a capacity correction and the length of their hull */
int mcmc_hip_debug_wave_log(mcmc_hip_ctx* c, double* cycles, uint64_t* hwid, double* e_chunks, double* e_days, double* e_hull, int64_t n) try {
    if (int r = check_ctx(c)) return r;
    if (!cycles || !hwid || !e_chunks || !e_days || !e_hull || n < 0 || n > MCMC_WAVE_LOG) return fail(MCMC_HIP_EINVAL, "wave log: bad arguments (at most %d tiles)", MCMC_WAVE_LOG);
    HIP_TRY(hipSetDevice(c->device));
    std::vector<unsigned long long> h(4 * MCMC_WAVE_LOG);
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_wave_log), h.size() * sizeof(unsigned long long)));
    for (int64_t i = 0; i < n; ++i) {
        cycles[i] = (double)h[4 * i]; hwid[i] = h[4 * i + 1]; e_chunks[i] = (double)h[4 * i + 2];
        e_days[i] = (double)(h[4 * i + 3] >> 32); e_hull[i] = (double)(h[4 * i + 3] & 0xffffffffull);
    }
    return 0;
} MCMC_ABI_GUARD

/* sums one of the two per-chain work counters ([nTiles][2][64]: algorithmic, executed) */
static int work_count(mcmc_hip_ctx* c, int which, double* out, int32_t reset) {
    if (int r = need_chains(c, out)) return r;
    std::vector<unsigned long long> h(2 * c->nP);
    HIP_TRY(hipMemcpy(h.data(), c->S.pairs, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double s = 0;
    for (long long i = 0; i < c->nChains; ++i)
        s += (double)h[(size_t)(i / MCMC_WAVE) * 2 * MCMC_WAVE + (size_t)which * MCMC_WAVE + (size_t)(i % MCMC_WAVE)];
    *out = s;
    if (reset) HIP_TRY(hipMemset(c->S.pairs, 0, h.size() * sizeof(unsigned long long)));
    return 0;
}
int mcmc_hip_pair_count(mcmc_hip_ctx* c, double* pairs_out, int32_t reset) try {
    return work_count(c, 0, pairs_out, reset);
} MCMC_ABI_GUARD
int mcmc_hip_executed_pair_count(mcmc_hip_ctx* c, double* pairs_out, int32_t reset) try {
    return work_count(c, 1, pairs_out, reset);
} MCMC_ABI_GUARD

/* ------------------------------------------------------------------ toy targets ---- */
static int toy_dim(int kind) { return kind == MCMC_TOY_MYLIKE1 ? 3 : (kind == MCMC_TOY_FOURGAUSS || kind == MCMC_TOY_ROSENBROCK) ? 2 : -1; }
static int toy_n_out(int kind) { return kind == MCMC_TOY_MYLIKE1 ? 4 : (kind == MCMC_TOY_FOURGAUSS || kind == MCMC_TOY_ROSENBROCK) ? 2 : -1; }

int mcmc_hip_toy_dim(int32_t kind) try {
    const int d = toy_dim(kind);
    return d > 0 ? d : fail(MCMC_HIP_EINVAL, "unknown toy target %d", kind);
} MCMC_ABI_GUARD

int mcmc_hip_toy_n_out(int32_t kind) try {
    const int d = toy_n_out(kind);
    return d > 0 ? d : fail(MCMC_HIP_EINVAL, "unknown toy target %d", kind);
} MCMC_ABI_GUARD

static int toy_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MCMC_HIP_ENODEVICE, "no HIP device visible: the hot path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MCMC_HIP_EINVAL, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    return 0;
}

int mcmc_hip_toy_eval(int device, int32_t kind, double difficulty, int64_t n, const double* x, double* ll_out) try {
    const int dim = toy_dim(kind);
    if (dim < 0 || n < 1 || !x || !ll_out) return fail(MCMC_HIP_EINVAL, "toy_eval: bad arguments");
    if (int r = toy_device(device)) return r;
    std::vector<double> soa((size_t)dim * n);
    for (int64_t c = 0; c < n; ++c)
        for (int k = 0; k < dim; ++k) soa[(size_t)k * n + c] = x[c * dim + k];
    DevBuf<double> dx, dl;
    if (int r = dx.alloc(soa.size())) return r;
    if (int r = dl.alloc(n)) return r;
    HIP_TRY(hipMemcpy(dx.p, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (kind == MCMC_TOY_MYLIKE1) hipLaunchKernelGGL(toy_eval_kernel<TOY_MYLIKE1>, dim3(grid), dim3(256), 0, 0, (const double*)dx.p, dl.p, (long long)n, difficulty);
    else if (kind == MCMC_TOY_ROSENBROCK) hipLaunchKernelGGL(toy_eval_kernel<TOY_ROSENBROCK>, dim3(grid), dim3(256), 0, 0, (const double*)dx.p, dl.p, (long long)n, difficulty);
    else hipLaunchKernelGGL(toy_eval_kernel<TOY_FOURGAUSS>, dim3(grid), dim3(256), 0, 0, (const double*)dx.p, dl.p, (long long)n, difficulty);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(ll_out, dl.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
} MCMC_ABI_GUARD

int mcmc_hip_toy_run(int device, int32_t kind, double difficulty, int32_t target_kind, double target_T,
                     double target_slope, double target_Tinit, int64_t n, const uint64_t* chain_ids,
                     uint64_t first_chain_id, double* x, double* stepsizes, int32_t nSamples, int32_t nBurnin,
                     int32_t nAdjust, int32_t thinning, double* loglikes_out, double* samples_out,
                     double* mean_out, int32_t* n_accept_out, uint64_t* rng_state, int32_t mean_continue) try {
    const int dim = toy_dim(kind), nOut = toy_n_out(kind);
    if (dim < 0 || n < 1 || !x || !stepsizes) return fail(MCMC_HIP_EINVAL, "toy_run: bad arguments");
    if (nSamples < 0 || nBurnin < 0 || nAdjust < 0 || thinning < 1)
        return fail(MCMC_HIP_EINVAL, "toy_run: need n_samples,n_burnin,n_adjust >= 0 and thinning >= 1");
    if (target_kind < MCMC_TARGET_PLAIN || target_kind > MCMC_TARGET_COOLING) return fail(MCMC_HIP_EINVAL, "toy_run: unknown target_kind");
    if (int r = toy_device(device)) return r;
    mcmc_covid_config tc{};
    tc.target_kind = target_kind; tc.target_T = target_T; tc.target_slope = target_slope; tc.target_Tinit = target_Tinit;
    const int nIter = nSamples + nAdjust;
    std::vector<double> betas((size_t)std::max(nIter, 1), 1.0);
    for (int i = 0; i < nIter; ++i)
        betas[i] = target_beta(tc, i < nAdjust ? (double)i / nAdjust : (double)(i - nAdjust) / nSamples);
    const int nSaved = mcmc_hip_n_saved(nSamples, nBurnin, nAdjust, thinning);
    std::vector<double> soa((size_t)dim * n);
    std::vector<uint64_t> ids(n);
    for (int64_t c = 0; c < n; ++c) {
        ids[c] = chain_ids ? chain_ids[c] : first_chain_id + (uint64_t)c;
        for (int k = 0; k < dim; ++k) soa[(size_t)k * n + c] = x[c * dim + k];
    }
    DevBuf<double> dx, dss, dbeta, dll, dsamp, dmean;
    DevBuf<uint64_t> dids, drng;
    DevBuf<int> dacc;
    if (mean_continue && !mean_out) return fail(MCMC_HIP_EINVAL, "toy_run: mean_continue needs mean_out");
    if (rng_state) {
        for (int64_t c = 0; c < n; ++c)
            if (!(rng_state[2 * c + 1] & 1u)) return fail(MCMC_HIP_EINVAL, "toy_run: chain %lld: a pcg32 increment is odd by construction", (long long)c);
        if (int r = drng.alloc(2 * n)) return r;
        HIP_TRY(hipMemcpy(drng.p, rng_state, 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    if (int r = dx.alloc(soa.size())) return r;
    if (int r = dss.alloc(n)) return r;
    if (int r = dbeta.alloc(betas.size())) return r;
    if (int r = dids.alloc(n)) return r;
    if (int r = dacc.alloc(n)) return r;
    if (loglikes_out && nSaved > 0) { if (int r = dll.alloc((size_t)nSaved * n)) return r; }
    if (samples_out && nSaved > 0) { if (int r = dsamp.alloc((size_t)nSaved * nOut * n)) return r; }
    if (mean_out) {
        if (int r = dmean.alloc((size_t)nOut * n)) return r;
        if (mean_continue) {
            std::vector<double> h((size_t)nOut * n);
            for (int64_t c = 0; c < n; ++c)
                for (int k = 0; k < nOut; ++k) h[(size_t)k * n + c] = mean_out[(size_t)c * nOut + k];
            HIP_TRY(hipMemcpy(dmean.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(hipMemcpy(dx.p, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dss.p, stepsizes, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dbeta.p, betas.data(), betas.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dids.p, ids.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (kind == MCMC_TOY_MYLIKE1)
        hipLaunchKernelGGL(toy_run_kernel<TOY_MYLIKE1>, dim3(grid), dim3(256), 0, 0, dx.p, dss.p, (const uint64_t*)dids.p, (long long)n,
                           difficulty, (const double*)dbeta.p, nSamples, nBurnin, nAdjust, thinning, dll.p, dsamp.p, dmean.p, dacc.p, drng.p, (int)mean_continue);
    else if (kind == MCMC_TOY_ROSENBROCK)
        hipLaunchKernelGGL(toy_run_kernel<TOY_ROSENBROCK>, dim3(grid), dim3(256), 0, 0, dx.p, dss.p, (const uint64_t*)dids.p, (long long)n,
                           difficulty, (const double*)dbeta.p, nSamples, nBurnin, nAdjust, thinning, dll.p, dsamp.p, dmean.p, dacc.p, drng.p, (int)mean_continue);
    else
        hipLaunchKernelGGL(toy_run_kernel<TOY_FOURGAUSS>, dim3(grid), dim3(256), 0, 0, dx.p, dss.p, (const uint64_t*)dids.p, (long long)n,
                           difficulty, (const double*)dbeta.p, nSamples, nBurnin, nAdjust, thinning, dll.p, dsamp.p, dmean.p, dacc.p, drng.p, (int)mean_continue);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(soa.data(), dx.p, soa.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < n; ++c)
        for (int k = 0; k < dim; ++k) x[c * dim + k] = soa[(size_t)k * n + c];
    HIP_TRY(hipMemcpy(stepsizes, dss.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (n_accept_out) HIP_TRY(hipMemcpy(n_accept_out, dacc.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (rng_state) HIP_TRY(hipMemcpy(rng_state, drng.p, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (dll.p) {
        std::vector<double> h((size_t)nSaved * n);
        HIP_TRY(hipMemcpy(h.data(), dll.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t c = 0; c < n; ++c)
            for (int s = 0; s < nSaved; ++s) loglikes_out[c * nSaved + s] = h[(size_t)s * n + c];
    }
    if (dsamp.p) {
        std::vector<double> h((size_t)nSaved * nOut * n);
        HIP_TRY(hipMemcpy(h.data(), dsamp.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t c = 0; c < n; ++c)
            for (int s = 0; s < nSaved; ++s)
                for (int k = 0; k < nOut; ++k) samples_out[((size_t)c * nSaved + s) * nOut + k] = h[((size_t)s * nOut + k) * n + c];
    }
    if (dmean.p) {
        std::vector<double> h((size_t)nOut * n);
        HIP_TRY(hipMemcpy(h.data(), dmean.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t c = 0; c < n; ++c)
            for (int k = 0; k < nOut; ++k) mean_out[(size_t)c * nOut + k] = h[(size_t)k * n + c];
    }
    return 0;
} MCMC_ABI_GUARD

}  // extern "C"
