// Instantiates every kernel for ONE state dimension (compile with -DMF_D=<d>) and both scalar types,
// and exports the launch tables declared in mf_launch.hpp.
#ifndef MF_D
#error "compile with -DMF_D=<state dimension>"
#endif
#include "mf_kernels.hpp"
#include "mf_kf_lds.hpp"
#include "mf_kf_x.hpp"
#include "mf_row.hpp"
#include "mf_row_par.hpp"
#include "mf_row_scan.hpp"
#include "mf_row_grad.hpp"
#include "mf_row_post.hpp"
#include "mf_row_gpr.hpp"
#include "mf_btd_par.hpp"
#include "mf_gpr_fused.hpp"
#include "mf_kl_grad.hpp"
#include "mf_post.hpp"
#include "mf_launch.hpp"

#include <cstdlib>
#include <string>
#include <type_traits>

namespace mf {
namespace {

constexpr int D = MF_D;
// The lane-per-chunk / lane-per-series kernels exist up to d = MF_MAX_D (their state no longer fits a lane's registers beyond it).
// For 10 <= d <= 15 this file is compiled with the ROW kernels only: a call whose plan needs a lane kernel returns -100 and the C
// ABI hands it to the LDS-tile / MFMA engine (mf_api.hip).
constexpr bool LANE = D <= MF_MAX_D;
// outputs of the row log-likelihood kernels: four wherever lane kernels share the entry point, eight in the row-only builds (k
// independent Matern-3/2 outputs are d = 2 k, m = k: five to seven outputs live at d = 10 ... 14)
constexpr int ROW_MAXM = LANE ? MF_MAXM : 8;
#define MF_LANE_LAUNCH(...)                                   \
    do {                                                      \
        if constexpr (LANE) { hipLaunchKernelGGL(__VA_ARGS__); } \
        else return -100;                                     \
    } while (0)
constexpr long RED_CHUNK = 8;    // chunk length of the level-0 floor of the parallel-in-time operators
// reduction levels of the log-likelihood: chunk length, and the size at which the last level is walked serially
// (d >= 7: a reduction step is ~10 k instructions on one lane, so the levels are cut shorter - 64 -> 16 -> 4 -> walk 4 is 12
// dependent steps in three launches instead of 16 in two)
constexpr long RED_DEFAULT = D >= 7 ? 4 : 8;
inline long red_chunk() {
    static const long v = [] { const char* e = mf_knob("MF_RED_CHUNK"); const long x = e ? std::atol(e) : 0; return x >= 2 ? x : RED_DEFAULT; }();
    return v;
}
inline long red_final() {
    static const long v = [] { const char* e = mf_knob("MF_RED_FINAL"); const long x = e ? std::atol(e) : 0; return x >= 1 ? x : RED_DEFAULT; }();
    return v;
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

// elements per reduced block: Dv, GU, F (D*D each), tv, gU (D each), sc (1)
constexpr long RED_ELEMS = 3 * D * D + 2 * D + 1;

template <typename T> size_t red_bytes(long B, long n) { return align_up(size_t(B) * n * RED_ELEMS * sizeof(T)); }

template <typename T> RedSys<T> carve(char*& p, long B, long n) {
    RedSys<T> r;
    T* base = reinterpret_cast<T*>(p);
    const long nb = B * n;
    r.Dv = base;
    r.GU = r.Dv + nb * D * D;
    r.F = r.GU + nb * D * D;
    r.tv = r.F + nb * D * D;
    r.gU = r.tv + nb * D;
    r.sc = r.gU + nb * D;
    r.n = n;
    r.f_stride = n;
    r.f_off = 0;
    p += red_bytes<T>(B, n);
    return r;
}

// Number of level-0 chunks per series: enough sub-problems to put one wavefront on every SIMD
// (256 CUs x 4 SIMDs x 64 lanes), but never chunks shorter than 4 blocks.
inline long auto_chunks(long B, long n) {
    static const long target = [] {
        const char* e = mf_knob("MF_TARGET_LANES");
        return e ? std::atol(e) : 65536L;
    }();
    long P = cdiv(target, B);
    const long maxP = n / 4 > 0 ? n / 4 : 1;
    if (P > maxP) P = maxP;
    if (P < 1) P = 1;
    return P;
}

template <typename T> size_t levels_ws(long B, long P, long rc = 0, long rf = 0) {
    if (rc <= 0) rc = red_chunk();
    if (rf <= 0) rf = red_final();
    size_t total = red_bytes<T>(B, P);
    long n = P;
    while (n > rf) {
        n = cdiv(n, rc);
        total += red_bytes<T>(B, n);
    }
    return total;
}
// reduction levels in row form (mf_row.hpp): a step costs ~450 instructions issued by a single wavefront per SIMD, so the
// levels are cut short: chunks of ROW_RED_CHUNK blocks, the last <= ROW_RED_FINAL blocks walked by one row per series
constexpr long ROW_RED_CHUNK = 6, ROW_RED_FINAL = 4;
inline long row_red_chunk() {
    static const long v = [] { const char* e = mf_knob("MF_ROW_RED_CHUNK"); const long x = e ? std::atol(e) : 0; return x >= 2 ? x : ROW_RED_CHUNK; }();
    return v;
}
inline long row_red_final() {
    static const long v = [] { const char* e = mf_knob("MF_ROW_RED_FINAL"); const long x = e ? std::atol(e) : 0; return x >= 1 ? x : ROW_RED_FINAL; }();
    return v;
}

// State dimensions whose elimination state (with the spike) does not fit a lane's 512 registers take the kernels of
// mf_kf_x.hpp (spike in LDS): d >= 7 in fp64, d = 9 in fp32.  MF_KF_X=0 switches them off (A/B timing).
template <typename T> bool x_path() {
    static const bool on = [] { const char* e = mf_knob("MF_KF_X"); return !(e && e[0] == '0'); }();
    return on && ((sizeof(T) == 8 && D >= 7) || (sizeof(T) == 4 && D >= 9));
}
// H_k, y_k staged in LDS by the spike-in-LDS kernel (several outputs): only when the larger image keeps the waves per CU
template <typename T> bool x_obs_lds(int m) {
    const int a = LdsSpike<T, D>::BYTES, b = a + LdsObs<T, D>::bytes(m);
    return (160 * 1024) / b >= 1 && ((160 * 1024) / b == (160 * 1024) / a || (160 * 1024) / b >= 4);
}
// lanes that fill the chip with the spike of every lane in LDS
template <typename T> long x_target_lanes() {
    int w = (160 * 1024) / LdsSpike<T, D>::BYTES;
    w = w > 4 ? 4 : (w < 1 ? 1 : w);
    return 256L * 64 * w;
}

// Reduce `cur` (already in workspace or user memory) down to a scalar per series.
template <typename T>
int reduce_levels(RedSys<T> cur, long B, char* p, T add_const, T* out, int* info, hipStream_t st) {
    while (cur.n > red_final()) {
        const long P = cdiv(cur.n, red_chunk());
        RedSys<T> nxt = carve<T>(p, B, P);
        const long lanes = B * P;
        constexpr int x_lds = LdsSpike<T, D>::BYTES;
        if (x_path<T>())
            MF_LANE_LAUNCH((red_chunk_x_kernel<T, D>), dim3((unsigned)cdiv(lanes, 64)), dim3(64), x_lds, st, cur, nxt, B,
                               P, info);
        else
            MF_LANE_LAUNCH((red_chunk_kernel<T, D, true>), dim3((unsigned)cdiv(lanes, 64)), dim3(64), 0, st, cur, nxt,
                               B, P, info);
        cur = nxt;
    }
    MF_LANE_LAUNCH((red_final_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, cur, B, add_const, out,
                       info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// The same with the row kernels (used behind the row level-0 kernel - and, for d = 4 ... 6, behind the lane-per-chunk level-0
// kernels too: the levels have few blocks (B x 8, then B) and a row's reduction step is ~300 instructions against a lane's ~2 700).
constexpr bool ROW_RED_SMALL = D >= 4 && D <= 6;
template <typename T>
int reduce_levels_row(RedSys<T> cur, long B, char* p, T add_const, T* out, int* info, hipStream_t st) {
    if constexpr ((D >= 7 || ROW_RED_SMALL) && D + 1 <= 16) {
        while (cur.n > row_red_final()) {
            const long P = cdiv(cur.n, row_red_chunk());
            RedSys<T> nxt = carve<T>(p, B, P);
            hipLaunchKernelGGL((row::red_row_kernel<T, D>), dim3((unsigned)cdiv(B * P, 4)), dim3(64), 0, st, cur, nxt, B, P, info);
            cur = nxt;
        }
        hipLaunchKernelGGL((row::red_row_final_kernel<T, D>), dim3((unsigned)cdiv(B, 4)), dim3(64), 0, st, cur, B, add_const, out,
                           info);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// The LDS-DMA streaming kernel (mf_kf_lds.hpp) covers: up to 4 outputs with a shared observation precision,
// or one output with per-step precisions (sites); matrix rows that are a whole number of 16-B units; 16-B aligned
// tensors; at least one transition.  Everything else takes kf_chunk_kernel (direct loads).
template <typename T, int M, bool RSTEP> constexpr bool lds_supported() { return KfLdsCfg<T, D, M, RSTEP>::SUPPORTED; }
template <typename T> bool use_lds_kernel(long Tn, int m, int rinv_per_step) {
    static const bool force_direct = [] {
        const char* e = mf_knob("MF_KF_IMPL");
        return e && std::string(e) == "direct";
    }();
    if (force_direct || Tn < 2) return false;
    if (rinv_per_step) return m == 1 && lds_supported<T, 1, true>();
    switch (m) {
        case 1: return lds_supported<T, 1, false>();
        case 2: return lds_supported<T, 2, false>();
        case 3: return lds_supported<T, 3, false>();
        case 4: return lds_supported<T, 4, false>() && (sizeof(T) == 4 || D <= 5);   // fp64 d >= 6: spills, no faster than direct loads
        default: return false;
    }
}
// lanes that fill the chip with this variant: one wavefront per SIMD unless the LDS image allows fewer per CU
template <typename T> long lds_target_lanes(int m, int rinv_per_step) {
    auto waves = [](int lds_bytes) { const int w = (160 * 1024) / lds_bytes; return w > 4 ? 4 : (w < 1 ? 1 : w); };
    int w = 4;
    if (rinv_per_step) w = waves(KfLdsCfg<T, D, 1, true>::LDS_TOTAL);
    else if (m == 2) w = waves(KfLdsCfg<T, D, 2, false>::LDS_TOTAL);
    else if (m == 3) w = waves(KfLdsCfg<T, D, 3, false>::LDS_TOTAL);
    else if (m == 4) w = waves(KfLdsCfg<T, D, 4, false>::LDS_TOTAL);
    long lanes = 256L * 64 * w;
    // Small blocks: every lane keeps a few partially used 128-B lines alive between two steps, and with one wave on every
    // SIMD (65 536 lanes x 5 streams x 128 B = 42 MB) they no longer fit the 32 MB of L2 - each line is then fetched twice.
    // Half the lanes, twice the steps per lane: measured at B=1024, T=10000 (scripts/prof_kf.py --chunks): fp64 d = 1, 2, 3:
    // 0.50 -> 0.34, 0.55 -> 0.38, 1.00 -> 0.62 ms; fp32 d = 2, 3: 0.51 -> 0.34, 0.77 -> 0.52 ms; fp32 d = 4, 5 best at 3/4.
    const long small = sizeof(T) == 8 ? (D <= 3 ? 32768L : 65536L) : (D <= 3 ? 32768L : (D <= 5 ? 49152L : 65536L));
    return lanes < small ? lanes : small;
}

// chunks per series and transitions per chunk of the LDS kernel
inline void lds_partition(long B, long Tn, long chunks, long& P, long& L, long target_lanes = 65536) {
    const long nt = Tn - 1;
    long want = chunks > 0 ? chunks : auto_chunks(B, nt);
    if (chunks <= 0 && target_lanes < 65536) {          // fewer resident waves per CU: fewer, longer chunks
        want = cdiv(target_lanes, B);
        const long maxP = nt / 4 > 0 ? nt / 4 : 1;
        if (want > maxP) want = maxP;
    }
    if (want > nt) want = nt;
    if (want < 1) want = 1;
    L = cdiv(nt, want);
    P = cdiv(nt, L);
}

// Which level-0 kernel evaluates a call, and with which time partition.  ONE function decides it for the launcher and for the
// workspace query, so the two can never disagree.
enum KfPath { KF_PATH_ROW, KF_PATH_X, KF_PATH_LDS, KF_PATH_DIRECT };
// The row kernels (mf_row.hpp: one 16-lane DPP row per chunk) take the log-likelihood wherever the spike-in-LDS kernels did.
// MF_KF_ROW=0 switches them off (A/B timing, experiment builds only).
template <typename T> bool row_path() {
    static const bool on = [] { const char* e = mf_knob("MF_KF_ROW"); return !(e && e[0] == '0'); }();
    return on && D + 1 <= 16 && ((sizeof(T) == 8 && D >= 7) || (sizeof(T) == 4 && D >= 9));
}
// rows that fill the chip ONCE: four per wavefront, as many wavefronts per SIMD as the level-0 kernel's registers allow - three up to
// d = 9 (~150 registers per lane); in the row-only builds two (fp64, d <= 12) or one (fp64, d >= 13), three in fp32.  A partial
// second round of rows costs a whole one (B=512, T=1000, fp64, d = 15: 8 chunks per series = 4 096 rows 0.82 ms, 12 chunks 1.01 ms),
// whole multiples are equal within noise (16: 0.83, 24: 0.85 ms; scripts/sweep_row_chunks.py) - the smallest one has the least
// reduction work and workspace.
template <typename T> long row_target_rows() {
    static const long v = [] {
        const char* e = mf_knob("MF_ROW_TARGET");
        const long x = e ? std::atol(e) : 0;
        const int waves = (sizeof(T) == 4 || D <= 9) ? 3 : row::row_waves_per_simd(D);
        return x > 0 ? x : 256L * 4 * waves * 4;
    }();
    return v;
}
struct KfPlan {
    KfPath path;
    long P, L;       // chunks per series; transitions per chunk (LDS kernel)
};
inline bool row_reduction(KfPath path) { return path == KF_PATH_ROW || (ROW_RED_SMALL && path == KF_PATH_LDS); }
template <typename T> size_t plan_ws(long B, const KfPlan& pl) {
    return row_reduction(pl.path) ? levels_ws<T>(B, pl.P, row_red_chunk(), row_red_final()) : levels_ws<T>(B, pl.P);
}
template <typename T> KfPlan kf_plan(long B, long Tn, int m, int rinv_per_step, long chunks, bool aligned16) {
    KfPlan pl{KF_PATH_DIRECT, 1, 0};
    if (row_path<T>() && Tn >= 2 && m >= 1 && m <= ROW_MAXM) {
        long P = chunks > 0 ? (chunks > Tn ? Tn : chunks) : cdiv(row_target_rows<T>(), B);
        if (chunks <= 0) {
            const long maxP = Tn / 4 > 0 ? Tn / 4 : 1;
            if (P > maxP) P = maxP;
        }
        if (row::row_offsets_fit(Tn, P, D, m, (int)sizeof(T))) {
            pl.path = KF_PATH_ROW;
            pl.P = P;
            return pl;
        }
    }
    // (fp64 d = 6 with four outputs: the streaming kernel spills and the plain direct-load kernel runs at 17 % - the spike-in-LDS
    // kernel with its grouped loads is the better home)
    if ((x_path<T>() || (sizeof(T) == 8 && D == 6 && m == 4)) && Tn >= 2) {
        long P = chunks > 0 ? (chunks > Tn ? Tn : chunks) : cdiv(x_target_lanes<T>(), B);
        if (chunks <= 0) {
            const long maxP = Tn / 4 > 0 ? Tn / 4 : 1;
            if (P > maxP) P = maxP;
        }
        if (P > 1) {
            pl.path = KF_PATH_X;
            pl.P = P;
            return pl;
        }
    }
    if (aligned16 && use_lds_kernel<T>(Tn, m, rinv_per_step)) {
        pl.path = KF_PATH_LDS;
        lds_partition(B, Tn, chunks, pl.P, pl.L, lds_target_lanes<T>(m, rinv_per_step));
        return pl;
    }
    pl.P = chunks > 0 ? (chunks > Tn ? Tn : chunks) : auto_chunks(B, Tn);
    return pl;
}

// Workspace of the log-likelihood: the call's m / per-step flag / alignment are not arguments of the query, so it is sized
// for the largest partition any of them can choose.
template <typename T> size_t kf_loglik_ws(long B, long Tn, long chunks) {
    size_t need = levels_ws<T>(B, 1);
    for (int per_step = 0; per_step < 2; ++per_step)
        for (int m = 1; m <= ROW_MAXM; ++m)
            for (int al = 0; al < 2; ++al) {
                const size_t w = plan_ws<T>(B, kf_plan<T>(B, Tn, m, per_step, chunks, al != 0));
                if (w > need) need = w;
            }
    if (Tn >= 2) {                                            // the fused GPR route (gpr_loglik) partitions like this
        long P = 1, L = 1;
        lds_partition(B, Tn, chunks, P, L);
        const size_t w = ROW_RED_SMALL ? levels_ws<T>(B, P, row_red_chunk(), row_red_final()) : levels_ws<T>(B, P);
        if (w > need) need = w;
    }
    return need;
}

template <typename T>
int kf_loglik(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,
              const T* H, const T* y, const T* Rinv, int rinv_per_step, T add_const, T* out, void* ws,
              size_t ws_bytes, int* info, long chunks, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    if (m < 1 || m > ROW_MAXM) return LANE ? -4 : -100;      // (row-only build: more outputs belong to the tile engine)
    if (ws == nullptr) return -15;
    const bool aligned16 = ((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(cholQ)) & 15) == 0;
    const KfPlan pl = kf_plan<T>(B, Tn, m, rinv_per_step, chunks, aligned16);
    const long P = pl.P;
    if constexpr (!LANE) { if (pl.path != KF_PATH_ROW) return -100; }      // 10 <= d <= 15: the row kernel or the tile engine
    if (ws_bytes < plan_ws<T>(B, pl)) return -15;           // checked against the partition that is actually launched
    int dbg = 0;
#ifdef MF_EXPERIMENT
    static const int dbg_env = [] { const char* e = mf_knob("MF_KF_DEBUG"); return e ? std::atoi(e) : 0; }();
    dbg = dbg_env;
#endif
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, P, info, dbg};
    char* p = static_cast<char*>(ws);
    RedSys<T> lvl0 = carve<T>(p, B, P);
    const dim3 grid((unsigned)cdiv(B * P, 64)), block(64);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (pl.path == KF_PATH_ROW) {
        if constexpr (D >= 7 && D + 1 <= 16) {
            const dim3 rgrid((unsigned)cdiv(B * P, 4));
            auto launch = [&](auto mtag) {
                constexpr int M = decltype(mtag)::value;
                if (rinv_per_step) hipLaunchKernelGGL((row::kf_row_kernel<T, D, M, true>), rgrid, block, 0, st, a, lvl0);
                else hipLaunchKernelGGL((row::kf_row_kernel<T, D, M, false>), rgrid, block, 0, st, a, lvl0);
            };
            using std::integral_constant;
            if (m == 1) launch(integral_constant<int, 1>{});
            else if (m == 2) launch(integral_constant<int, 2>{});
            else if (m == 3) launch(integral_constant<int, 3>{});
            else if (m == 4) launch(integral_constant<int, 4>{});
            else if constexpr (!LANE) {
                if (m == 5) launch(integral_constant<int, 5>{});
                else if (m == 6) launch(integral_constant<int, 6>{});
                else if (m == 7) launch(integral_constant<int, 7>{});
                else launch(integral_constant<int, 8>{});
            }
        }
    } else if (pl.path == KF_PATH_X) {
        constexpr int x_lds = LdsSpike<T, D>::BYTES;
        if (m > 1 && x_obs_lds<T>(m)) {
            const int lds = x_lds + LdsObs<T, D>::bytes(m);
            if constexpr (LANE) if (lds > 64 * 1024) {         // past the default dynamic-LDS limit of a kernel: raise it (once per process)
                static const hipError_t raised = hipFuncSetAttribute(
                    reinterpret_cast<const void*>(&kf_chunk_x_kernel<T, D, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                    x_lds + LdsObs<T, D>::bytes(MF_MAXM));
                if (raised != hipSuccess) return -1000;
            }
            MF_LANE_LAUNCH((kf_chunk_x_kernel<T, D, true>), grid, block, lds, st, a, lvl0);
        } else {
            MF_LANE_LAUNCH((kf_chunk_x_kernel<T, D, false>), grid, block, x_lds, st, a, lvl0);
        }
    } else if (pl.path == KF_PATH_LDS) {
        const long L = pl.L;
        auto launch = [&](auto mtag, auto rtag) {
            constexpr int M = decltype(mtag)::value;
            constexpr bool RS = decltype(rtag)::value;
            if constexpr (KfLdsCfg<T, D, M, RS>::SUPPORTED) {
                constexpr int lds = KfLdsCfg<T, D, M, RS>::LDS_TOTAL;
                if (P > 1) MF_LANE_LAUNCH((kf_chunk_lds_kernel<T, D, M, true, RS>), grid, block, lds, st, a, L, lvl0);
                else MF_LANE_LAUNCH((kf_chunk_lds_kernel<T, D, M, false, RS>), grid, block, lds, st, a, L, lvl0);
            }
        };
        using std::integral_constant;
        if (rinv_per_step) launch(integral_constant<int, 1>{}, integral_constant<bool, true>{});
        else if (m == 1) launch(integral_constant<int, 1>{}, integral_constant<bool, false>{});
        else if (m == 2) launch(integral_constant<int, 2>{}, integral_constant<bool, false>{});
        else if (m == 3) launch(integral_constant<int, 3>{}, integral_constant<bool, false>{});
        else launch(integral_constant<int, 4>{}, integral_constant<bool, false>{});
    } else if (P > 1) {
        if (m == 1) MF_LANE_LAUNCH((kf_chunk_kernel<T, D, 1, true>), grid, block, 0, st, a, lvl0);
        else MF_LANE_LAUNCH((kf_chunk_kernel<T, D, 0, true>), grid, block, 0, st, a, lvl0);
    } else {
        if (m == 1) MF_LANE_LAUNCH((kf_chunk_kernel<T, D, 1, false>), grid, block, 0, st, a, lvl0);
        else MF_LANE_LAUNCH((kf_chunk_kernel<T, D, 0, false>), grid, block, 0, st, a, lvl0);
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    if (row_reduction(pl.path)) return reduce_levels_row<T>(lvl0, B, p, add_const, out, info, st);
    return reduce_levels<T>(lvl0, B, p, add_const, out, info, st);
}

template <typename T>
int kf_loglik_plan(long B, long Tn, int m, int rinv_per_step, long chunks, int aligned16, int* path, long* P, long* L) {
    if (B < 1 || Tn < 1 || m < 1 || m > ROW_MAXM) return -1;
    const KfPlan pl = kf_plan<T>(B, Tn, m, rinv_per_step, chunks, aligned16 != 0);
    *path = (int)pl.path;
    *P = pl.P;
    *L = pl.L;
    return 0;
}

template <typename T> size_t btd_logdet_quad_ws(long B, long n, long chunks) {
    (void)chunks;
    long nn = n;
    size_t total = 256;
    while (nn > red_final()) {
        nn = cdiv(nn, red_chunk());
        total += red_bytes<T>(B, nn);
    }
    return total;
}

// out[s] = 0.5 |L^-1 rhs|^2 - log|L|  for the natural-order Cholesky L of (diag, sub); any order gives
// the same number, so the partitioned elimination is used.
template <typename T>
int btd_logdet_quad(long B, long n, const T* diag, const T* sub, const T* rhs, T* out, void* ws, size_t ws_bytes,
                    int* info, long chunks, hipStream_t st) {
    if (ws_bytes < btd_logdet_quad_ws<T>(B, n, chunks) || ws == nullptr) return -9;
    RedSys<T> in;
    in.Dv = const_cast<T*>(diag);
    in.GU = nullptr;
    in.gU = nullptr;
    in.F = const_cast<T*>(sub);
    in.tv = const_cast<T*>(rhs);
    in.sc = nullptr;
    in.n = n;
    in.f_stride = n - 1;
    in.f_off = -1;
    return reduce_levels<T>(in, B, static_cast<char*>(ws), T(0), out, info, st);
}

// ---- parallel-in-time cholesky / solve (mf_btd_par.hpp): chosen when there are too few series to fill the chip ----
constexpr long PAR_MIN_BLOCKS = 64;      // never partition chains shorter than this
constexpr long PAR_MAX_SERIES = 4096;    // with this many series one lane per series already fills the chip

// level-0 chunk length (0 = use the serial one-lane-per-series kernel)
// chunk length of the reduced levels of the parallel-in-time operators (and the size at which the coarsest level is walked
// serially).  Sequential depth is ~ 2 r log_r(n) block steps against one launch per level: 5 measured best on config 3 and
// on B=64, T=10000 (scripts/sweep_radix.sh; 8: 192 / 535 us, 5: 166 / 494 us) - see DESIGN.md 4.3.
constexpr long PAR_RADIX = 5;
inline long par_radix() {
    static const long r = [] { const char* e = mf_knob("MF_BTD_RADIX"); const long v = e ? std::atol(e) : 0; return v >= 2 ? v : PAR_RADIX; }();
    return r;
}
inline long par_len0(long B, long n) {
    static const long force = [] { const char* e = mf_knob("MF_BTD_PAR_LEN"); return e ? std::atol(e) : -1L; }();
    if (force >= 0) return (force > 0 && n >= 2 * force) ? force : 0;
    if constexpr (LANE) { if (B >= PAR_MAX_SERIES || n < PAR_MIN_BLOCKS) return 0; }
    long len = cdiv(B * n, 65536);           // aim at one wavefront per SIMD ...
    if (len < RED_CHUNK) len = RED_CHUNK;     // ... but keep the reduced system at most 1/8 of the input
    if (n >= 2 * len) return len;
    // row-only builds (10 <= d <= 15) have no lane-per-series kernels to fall back on: many series or short chains run the same
    // row kernels with ONE chunk per series (the up-sweep of a whole series has no spike; the emit pass is the serial recursion)
    return (!LANE && n >= 2) ? n : 0;
}

// The parallel-in-time operators in row form (mf_row_par.hpp, mf_row_scan.hpp, mf_row_post.hpp: a 16-lane row per chunk instead
// of a lane).  From d = 7 on always (a lane's state no longer fits its registers).  At d = 5, 6 the lane-per-chunk kernels are
// register resident and carry 64 chunks per wavefront against the row kernels' 4, so the row form only pays where the
// DEPENDENT block steps are the run time, i.e. with few level-0 chunks (`rows0`): measured at d = 6 - one chain of 10^5 blocks
// (12 500 chunks) 167 -> 96 us, B=64 T=10^4 (80 000) 520 -> 430 us, but B=1024 T=2000 (256 000) 0.99 -> 1.08 ms and the
// headline-shape backward 19.5 -> 22.5 ms.  MF_BTD_ROW=0 / 1 forces it off / on (experiment builds).
template <typename T> bool row_par_path(long rows0) {
    static const int force = [] { const char* e = mf_knob("MF_BTD_ROW"); return e ? std::atoi(e) : -1; }();
    if (D + 1 > 16 || D < 2) return false;
    if (force >= 0) return force != 0;
    return D >= 7 || (D >= 5 && rows0 <= 131072);
}
// level-0 chunks of the time partition all of these operators share
inline long par_rows0(long B, long n) {
    const long len0 = par_len0(B, n);
    return len0 > 0 ? B * cdiv(n, len0) : B;
}

// one step of prefetch in the row kernels when the level-0 rows are at most four wavefronts per SIMD
inline bool row_par_prefetch(long rows0) { return rows0 <= 4L * 256 * 4 * 4; }

struct ParPlan {
    int levels;          // number of reduced levels (>= 1)
    long n[24];          // n[0] = T, n[l] = blocks per series on level l
    long len[24];        // chunk length used on level l to form level l+1
};
inline ParPlan par_plan(long n0, long len0) {
    ParPlan pl;
    pl.n[0] = n0;
    pl.len[0] = len0;
    int l = 0;
    do {
        pl.n[l + 1] = cdiv(pl.n[l], pl.len[l]);
        ++l;
        pl.len[l] = par_radix();
    } while (pl.n[l] > par_radix() && l < 22);
    pl.levels = l;
    return pl;
}

// register-resident up-sweep of the parallel-in-time Cholesky; mode 0 / 1: level 0 of the matrix / of the block-reversed
// matrix, 2: a reduced level
template <typename T>
int launch_chol_up(const ParLevel<T>& lv, int mode, long B, long len, long P, T* oDv, T* oGf, T* oGU, T* oF, int* info,
                   hipStream_t st) {
    const dim3 grid((unsigned)cdiv(B * P, 64)), block(64);
    if (mode == 0) MF_LANE_LAUNCH((par_chol_up_kernel<T, D, 0>), grid, block, 0, st, lv, B, len, P, oDv, oGf, oGU, oF, info);
    else if (mode == 1) MF_LANE_LAUNCH((par_chol_up_kernel<T, D, 1>), grid, block, 0, st, lv, B, len, P, oDv, oGf, oGU, oF, info);
    else MF_LANE_LAUNCH((par_chol_up_kernel<T, D, 2>), grid, block, 0, st, lv, B, len, P, oDv, oGf, oGU, oF, info);
    return 0;
}

template <typename T> size_t btd_cholesky_ws(long B, long n) {
    const long len0 = par_len0(B, n);
    if (len0 == 0) return 0;
    const ParPlan pl = par_plan(n, len0);
    size_t total = 0;
    for (int l = 1; l <= pl.levels; ++l) total += 5 * align_up(size_t(B) * pl.n[l] * D * D * sizeof(T));
    return total;
}

template <typename T>
int btd_cholesky(long B, long n, const T* diag, const T* sub, T* ldiag, T* lsub, void* ws, size_t ws_bytes, int* info,
                 hipStream_t st) {
    const long len0 = sub ? par_len0(B, n) : 0;
    if (len0 == 0 || ws == nullptr || ws_bytes < btd_cholesky_ws<T>(B, n)) {
        // one lane per series; with a sub-diagonal the level-0 emit kernel as ONE chunk: same recursion, but the next block's
        // loads are in flight during the current block's arithmetic
        if (sub && n >= 2)
            MF_LANE_LAUNCH((par_chol_emit_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, n, n, 1L, diag,
                               sub, static_cast<const T*>(nullptr), ldiag, lsub, info);
        else if (!sub)
            // block-diagonal matrix: every block is a chain of its own - a lane per BLOCK, not per series (naturals_to_ssm_params
            // factors three of these at [64, 10^4]: 19 ms each with 64 lanes walking 10^4 blocks, profiles/r05_cvi_chain.txt)
            MF_LANE_LAUNCH((btd_cholesky_kernel<T, D>), dim3((unsigned)cdiv(B * n, 64)), dim3(64), 0, st, B * n, 1L, diag, sub,
                               ldiag, lsub, info);
        else
            MF_LANE_LAUNCH((btd_cholesky_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, n, diag, sub,
                               ldiag, lsub, info);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    const ParPlan pl = par_plan(n, len0);
    struct Arr { T *Dv, *Gf, *GU, *F, *Pn; } arr[24];
    char* p = static_cast<char*>(ws);
    for (int l = 1; l <= pl.levels; ++l) {
        const size_t sz = align_up(size_t(B) * pl.n[l] * D * D * sizeof(T));
        T** f[5] = {&arr[l].Dv, &arr[l].Gf, &arr[l].GU, &arr[l].F, &arr[l].Pn};
        for (auto* q : f) { *q = reinterpret_cast<T*>(p); p += sz; }
    }
    auto level = [&](int l) {
        if (l == 0) return ParLevel<T>{diag, nullptr, nullptr, sub, n, n - 1, -1, 0};
        return ParLevel<T>{arr[l].Dv, arr[l].Gf, arr[l].GU, arr[l].F, pl.n[l], pl.n[l], 0, 0};
    };
    if (row_par_path<T>(B * pl.n[1])) {
        if constexpr (D >= 2 && D + 1 <= 16) {
            const dim3 blk(64);
            auto rgrid = [](long rows) { return dim3((unsigned)cdiv(rows, 4)); };
            auto run = [&](auto pf) {
                constexpr bool PF = decltype(pf)::value;
                for (int l = 0; l < pl.levels; ++l) {
                    const long P = pl.n[l + 1];
                    if (l == 0)
                        hipLaunchKernelGGL((row::row_chol_up_kernel<T, D, false, PF>), rgrid(B * P), blk, 0, st, level(l), B, pl.len[l],
                                           P, arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU, arr[l + 1].F, info);
                    else
                        hipLaunchKernelGGL((row::row_chol_up_kernel<T, D, true, PF>), rgrid(B * P), blk, 0, st, level(l), B, pl.len[l],
                                           P, arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU, arr[l + 1].F, info);
                }
                {
                    const int l = pl.levels;
                    hipLaunchKernelGGL((row::row_chol_down_kernel<T, D, PF>), rgrid(B), blk, 0, st, level(l), B, pl.n[l], 1L,
                                       static_cast<const T*>(nullptr), arr[l].Pn, info);
                }
                for (int l = pl.levels - 1; l >= 1; --l) {
                    const long P = pl.n[l + 1];
                    hipLaunchKernelGGL((row::row_chol_down_kernel<T, D, PF>), rgrid(B * P), blk, 0, st, level(l), B, pl.len[l], P,
                                       static_cast<const T*>(arr[l + 1].Pn), arr[l].Pn, info);
                }
                hipLaunchKernelGGL((row::row_chol_emit_kernel<T, D, PF>), rgrid(B * pl.n[1]), blk, 0, st, B, n, len0, pl.n[1], diag,
                                   sub, static_cast<const T*>(arr[1].Pn), ldiag, lsub, info);
            };
            // few rows (one long chain): nothing but a prefetch hides a step's loads; many rows: other wavefronts do, and the
            // second set of step data would only cost occupancy
            if (row_par_prefetch(B * pl.n[1])) run(std::true_type{}); else run(std::false_type{});
        }
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    for (int l = 0; l < pl.levels; ++l) {
        const long P = pl.n[l + 1];
        constexpr int x_lds = LdsSpike<T, D>::BYTES;
        if (x_path<T>())
            MF_LANE_LAUNCH((par_chol_up_x_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), x_lds, st, level(l),
                               B, pl.len[l], P, arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU, arr[l + 1].F, info);
        else
            if (const int rc = launch_chol_up<T>(level(l), l == 0 ? 0 : 2, B, pl.len[l], P, arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU,
                              arr[l + 1].F, info, st)) return rc;
    }
    {   // coarsest level: one lane per series walks it
        const int l = pl.levels;
        MF_LANE_LAUNCH((par_chol_down_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, level(l), B,
                           pl.n[l], 1L, static_cast<const T*>(nullptr), arr[l].Pn, info);
    }
    for (int l = pl.levels - 1; l >= 1; --l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_chol_down_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, level(l), B,
                           pl.len[l], P, static_cast<const T*>(arr[l + 1].Pn), arr[l].Pn, info);
    }
    MF_LANE_LAUNCH((par_chol_emit_kernel<T, D>), dim3((unsigned)cdiv(B * pl.n[1], 64)), dim3(64), 0, st, B, n, len0,
                       pl.n[1], diag, sub, static_cast<const T*>(arr[1].Pn), ldiag, lsub, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T> size_t btd_solve_ws(long Bl, long Br, long n) {
    (void)Bl;
    const long len0 = par_len0(Br, n);
    if (len0 == 0) return 0;
    const ParPlan pl = par_plan(n, len0);
    size_t total = 0;
    for (int l = 1; l <= pl.levels; ++l)
        total += align_up(size_t(Br) * pl.n[l] * D * D * sizeof(T)) + 2 * align_up(size_t(Br) * pl.n[l] * D * sizeof(T));
    return total;
}

template <typename T>
int btd_solve(long Bl, long Br, long n, const T* ldiag, const T* lsub, const T* rhs, T* out, int transpose, void* ws,
              size_t ws_bytes, hipStream_t st) {
    const long len0 = lsub ? par_len0(Br, n) : 0;
    if (len0 == 0 || ws == nullptr || ws_bytes < btd_solve_ws<T>(Bl, Br, n)) {
        if (lsub && n >= 2)     // one lane per right-hand side: the level-0 emit kernel as ONE chunk (prefetched loads)
            MF_LANE_LAUNCH((par_solve_emit_kernel<T, D>), dim3((unsigned)cdiv(Br, 64)), dim3(64), 0, st, Bl, Br, n, n, 1L,
                               ldiag, lsub, rhs, static_cast<const T*>(nullptr), transpose, out);
        else if (!lsub)         // block-diagonal factor: a lane per (right-hand side, block); flat (r n + k) mod (Bl n) is the factor's block
            MF_LANE_LAUNCH((btd_solve_kernel<T, D>), dim3((unsigned)cdiv(Br * n, 64)), dim3(64), 0, st, Bl * n, Br * n, 1L, ldiag,
                               lsub, rhs, out, transpose);
        else
            MF_LANE_LAUNCH((btd_solve_kernel<T, D>), dim3((unsigned)cdiv(Br, 64)), dim3(64), 0, st, Bl, Br, n, ldiag,
                               lsub, rhs, out, transpose);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    const ParPlan pl = par_plan(n, len0);
    struct Arr { T *M, *c, *Z; } arr[24];
    char* p = static_cast<char*>(ws);
    for (int l = 1; l <= pl.levels; ++l) {
        arr[l].M = reinterpret_cast<T*>(p); p += align_up(size_t(Br) * pl.n[l] * D * D * sizeof(T));
        arr[l].c = reinterpret_cast<T*>(p); p += align_up(size_t(Br) * pl.n[l] * D * sizeof(T));
        arr[l].Z = reinterpret_cast<T*>(p); p += align_up(size_t(Br) * pl.n[l] * D * sizeof(T));
    }
    if (row_par_path<T>(Br * pl.n[1])) {
        if constexpr (D >= 2 && D + 1 <= 16) {
            const dim3 blk(64);
            auto rgrid = [](long rows) { return dim3((unsigned)cdiv(rows, 4)); };
            auto run = [&](auto pf) {
                constexpr bool PF = decltype(pf)::value;
                hipLaunchKernelGGL((row::row_solve_up0_kernel<T, D, PF>), rgrid(Br * pl.n[1]), blk, 0, st, Bl, Br, n, len0, pl.n[1],
                                   ldiag, lsub, rhs, transpose, arr[1].M, arr[1].c);
                for (int l = 1; l < pl.levels; ++l) {
                    const long P = pl.n[l + 1];
                    hipLaunchKernelGGL((row::row_affine_up_kernel<T, D, PF>), rgrid(Br * P), blk, 0, st, Br, pl.n[l], pl.len[l], P,
                                       static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c), arr[l + 1].M, arr[l + 1].c);
                }
                {
                    const int l = pl.levels;
                    hipLaunchKernelGGL((row::row_affine_down_kernel<T, D, PF>), rgrid(Br), blk, 0, st, Br, pl.n[l], pl.n[l], 1L,
                                       static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                                       static_cast<const T*>(nullptr), arr[l].Z);
                }
                for (int l = pl.levels - 1; l >= 1; --l) {
                    const long P = pl.n[l + 1];
                    hipLaunchKernelGGL((row::row_affine_down_kernel<T, D, PF>), rgrid(Br * P), blk, 0, st, Br, pl.n[l], pl.len[l], P,
                                       static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                                       static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
                }
                hipLaunchKernelGGL((row::row_solve_emit_kernel<T, D, PF>), rgrid(Br * pl.n[1]), blk, 0, st, Bl, Br, n, len0, pl.n[1],
                                   ldiag, lsub, rhs, static_cast<const T*>(arr[1].Z), transpose, out);
            };
            if (row_par_prefetch(Br * pl.n[1])) run(std::true_type{}); else run(std::false_type{});
        }
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    MF_LANE_LAUNCH((par_solve_up0_kernel<T, D>), dim3((unsigned)cdiv(Br * pl.n[1], 64)), dim3(64), 0, st, Bl, Br, n,
                       len0, pl.n[1], ldiag, lsub, rhs, transpose, arr[1].M, arr[1].c);
    for (int l = 1; l < pl.levels; ++l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_affine_up_kernel<T, D>), dim3((unsigned)cdiv(Br * P, 64)), dim3(64), 0, st, Br, pl.n[l],
                           pl.len[l], P, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c), arr[l + 1].M,
                           arr[l + 1].c);
    }
    {
        const int l = pl.levels;
        MF_LANE_LAUNCH((par_affine_down_kernel<T, D>), dim3((unsigned)cdiv(Br, 64)), dim3(64), 0, st, Br, pl.n[l],
                           pl.n[l], 1L, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                           static_cast<const T*>(nullptr), arr[l].Z);
    }
    for (int l = pl.levels - 1; l >= 1; --l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_affine_down_kernel<T, D>), dim3((unsigned)cdiv(Br * P, 64)), dim3(64), 0, st, Br, pl.n[l],
                           pl.len[l], P, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                           static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
    }
    MF_LANE_LAUNCH((par_solve_emit_kernel<T, D>), dim3((unsigned)cdiv(Br * pl.n[1], 64)), dim3(64), 0, st, Bl, Br, n,
                       len0, pl.n[1], ldiag, lsub, rhs, static_cast<const T*>(arr[1].Z), transpose, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int btd_matvec(long Bl, long Br, long n, const T* diag, const T* sub, const T* x, T* out, int mode, hipStream_t st) {
    MF_LANE_LAUNCH((btd_matvec_kernel<T, D>), dim3((unsigned)cdiv(Br * n, 256)), dim3(256), 0, st, Bl, Br, n, diag,
                       sub, x, out, mode);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T> int btd_logdet(long B, long n, const T* ldiag, T* out, hipStream_t st) {
    MF_LANE_LAUNCH((btd_logdet_kernel<T, D>), dim3((unsigned)B), dim3(64), 0, st, B, n, ldiag, out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T> size_t btd_diag_of_inverse_ws(long B, long n) {
    const long len0 = par_len0(B, n);
    if (len0 == 0) return 0;
    const ParPlan pl = par_plan(n, len0);
    size_t total = 0;
    for (int l = 1; l <= pl.levels; ++l) total += 3 * align_up(size_t(B) * pl.n[l] * D * D * sizeof(T));
    return total;
}

// Congruence scan Sigma(p) = N_p + G_p^T Sigma(p-1) G_p over the positions of every series (mf_btd_par.hpp: TakSrc):
// SRC 0 = block Takahashi on a Cholesky factor, SRC 1 = marginal covariances of a state space model.
template <typename T, int SRC>
int tak_scan(long B, long n, TakSrc<T> src, T* odiag, T* osub, void* ws, size_t ws_bytes, hipStream_t st,
             const T** up_only = nullptr,        // up_only: stop before the emit kernel and hand out the chunk-start values
             TakMeanUp<T> mup = TakMeanUp<T>{}) {  // mup.oc != NULL (SRC 1): the level-0 up-sweep of the means rides along
    const long len0 = par_len0(B, n);
    if (len0 == 0 || ws == nullptr || ws_bytes < btd_diag_of_inverse_ws<T>(B, n)) {
        if (up_only) return -16;
        // one lane per series: the level-0 emit kernel as ONE chunk (prefetched loads)
        MF_LANE_LAUNCH((par_tak_emit_kernel<T, D, SRC>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, n, n, 1L, src,
                           static_cast<const T*>(nullptr), odiag, osub);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    const ParPlan pl = par_plan(n, len0);
    struct Arr { T *G, *N, *Z; } arr[24];
    char* p = static_cast<char*>(ws);
    for (int l = 1; l <= pl.levels; ++l) {
        const size_t sz = align_up(size_t(B) * pl.n[l] * D * D * sizeof(T));
        arr[l].G = reinterpret_cast<T*>(p); p += sz;
        arr[l].N = reinterpret_cast<T*>(p); p += sz;
        arr[l].Z = reinterpret_cast<T*>(p); p += sz;
    }
    if constexpr (D >= 2 && D + 1 <= 16) {
        // the covariance (and mean) scan / its adjoint in row form (mf_row_scan.hpp); the Takahashi recursion on a factor (SRC 0)
        // from d = 7 on (below, the register-resident lane kernels carry 64 chunks per wavefront)
        if (row_par_path<T>(B * pl.n[1]) && (SRC != 0 || D >= 7)) {
            const dim3 blk(64);
            auto rgrid = [](long rows) { return dim3((unsigned)cdiv(rows, 4)); };
            if (SRC == 1 && mup.oc != nullptr)
                hipLaunchKernelGGL((row::row_cov_up0_kernel<T, D, 1, true>), rgrid(B * pl.n[1]), blk, 0, st, B, n, len0, pl.n[1], src,
                                   arr[1].G, arr[1].N, mup);
            else
                hipLaunchKernelGGL((row::row_cov_up0_kernel<T, D, SRC, false>), rgrid(B * pl.n[1]), blk, 0, st, B, n, len0, pl.n[1],
                                   src, arr[1].G, arr[1].N, TakMeanUp<T>{});
            for (int l = 1; l < pl.levels; ++l) {
                const long P = pl.n[l + 1];
                hipLaunchKernelGGL((row::row_cov_up_kernel<T, D>), rgrid(B * P), blk, 0, st, B, pl.n[l], pl.len[l], P,
                                   static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N), arr[l + 1].G, arr[l + 1].N);
            }
            {
                const int l = pl.levels;
                hipLaunchKernelGGL((row::row_cov_down_kernel<T, D>), rgrid(B), blk, 0, st, B, pl.n[l], pl.n[l], 1L,
                                   static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N), static_cast<const T*>(nullptr),
                                   arr[l].Z);
            }
            for (int l = pl.levels - 1; l >= 1; --l) {
                const long P = pl.n[l + 1];
                hipLaunchKernelGGL((row::row_cov_down_kernel<T, D>), rgrid(B * P), blk, 0, st, B, pl.n[l], pl.len[l], P,
                                   static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N),
                                   static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
            }
            if (up_only) { *up_only = arr[1].Z; return hipGetLastError() == hipSuccess ? 0 : -1000; }
            hipLaunchKernelGGL((row::row_cov_emit_kernel<T, D, SRC, false>), rgrid(B * pl.n[1]), blk, 0, st, B, n, len0, pl.n[1], src,
                               static_cast<const T*>(arr[1].Z), odiag, osub, TakMean<T>{});
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
    }
    constexpr int g_lds = D * D * 64 * (int)sizeof(T);      // x path: the composed G of a run lives in LDS
    if (SRC == 1 && mup.oc != nullptr) {
        if (x_path<T>())
            MF_LANE_LAUNCH((par_tak_up0_x_kernel<T, D, 1, true>), dim3((unsigned)cdiv(B * pl.n[1], 64)), dim3(64), g_lds, st, B,
                               n, len0, pl.n[1], src, arr[1].G, arr[1].N, mup);
        else
            MF_LANE_LAUNCH((par_tak_up0_kernel<T, D, 1, true>), dim3((unsigned)cdiv(B * pl.n[1], 64)), dim3(64), 0, st, B, n,
                               len0, pl.n[1], src, arr[1].G, arr[1].N, mup);
    } else if (x_path<T>())
        MF_LANE_LAUNCH((par_tak_up0_x_kernel<T, D, SRC>), dim3((unsigned)cdiv(B * pl.n[1], 64)), dim3(64), g_lds, st, B, n,
                           len0, pl.n[1], src, arr[1].G, arr[1].N, TakMeanUp<T>{});
    else
        MF_LANE_LAUNCH((par_tak_up0_kernel<T, D, SRC>), dim3((unsigned)cdiv(B * pl.n[1], 64)), dim3(64), 0, st, B, n, len0,
                           pl.n[1], src, arr[1].G, arr[1].N, TakMeanUp<T>{});
    for (int l = 1; l < pl.levels; ++l) {
        const long P = pl.n[l + 1];
        if (x_path<T>())
            MF_LANE_LAUNCH((par_tak_up_x_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), g_lds, st, B, pl.n[l],
                               pl.len[l], P, static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N),
                               arr[l + 1].G, arr[l + 1].N);
        else
            MF_LANE_LAUNCH((par_tak_up_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, B, pl.n[l],
                               pl.len[l], P, static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N), arr[l + 1].G,
                               arr[l + 1].N);
    }
    {
        const int l = pl.levels;
        MF_LANE_LAUNCH((par_tak_down_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, pl.n[l], pl.n[l],
                           1L, static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N),
                           static_cast<const T*>(nullptr), arr[l].Z);
    }
    for (int l = pl.levels - 1; l >= 1; --l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_tak_down_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, B, pl.n[l],
                           pl.len[l], P, static_cast<const T*>(arr[l].G), static_cast<const T*>(arr[l].N),
                           static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
    }
    if (up_only) { *up_only = arr[1].Z; return hipGetLastError() == hipSuccess ? 0 : -1000; }
    MF_LANE_LAUNCH((par_tak_emit_kernel<T, D, SRC>), dim3((unsigned)cdiv(B * pl.n[1], 64)), dim3(64), 0, st, B, n, len0,
                       pl.n[1], src, static_cast<const T*>(arr[1].Z), odiag, osub);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int btd_diag_of_inverse(long B, long n, const T* ldiag, const T* lsub, T* odiag, T* osub, void* ws, size_t ws_bytes,
                        hipStream_t st) {
    if (!lsub || n < 2) {       // block-diagonal factor: nothing to scan - a lane per block
        const long nb = lsub ? B : B * n, nn = lsub ? n : 1L;
        MF_LANE_LAUNCH((btd_diag_of_inverse_kernel<T, D>), dim3((unsigned)cdiv(nb, 64)), dim3(64), 0, st, nb, nn, ldiag,
                           lsub, odiag, osub);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    return tak_scan<T, 0>(B, n, TakSrc<T>{ldiag, lsub, nullptr}, odiag, osub, ws, ws_bytes, st);
}

// Reverse mode of SymmetricBlockTriDiagonal.cholesky / LowerTriangularBlockTriDiagonal.block_diagonal_of_inverse (mf_btd_par.hpp:
// a local kernel, a congruence scan, a local kernel).  Workspace: the scan's + three / two tensors of blocks.
template <typename T> size_t btd_grad_ws(long B, long n) {
    return align_up(btd_diag_of_inverse_ws<T>(B, n)) + 3 * align_up(size_t(B) * n * D * D * sizeof(T));
}
template <typename T>
int btd_cholesky_grad(long B, long n, const T* ldiag, const T* lsub, const T* g_ldiag, const T* g_lsub, T* g_diag, T* g_sub, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    if (ws == nullptr || ws_bytes < btd_grad_ws<T>(B, n)) return -9;
    const size_t blk = align_up(size_t(B) * n * D * D * sizeof(T));
    char* p = static_cast<char*>(ws);
    T* C = reinterpret_cast<T*>(p); p += blk;
    T* G = reinterpret_cast<T*>(p); p += blk;
    T* Zs = reinterpret_cast<T*>(p); p += blk;            // -Z_{k+1} G_k of the scan
    const size_t scan_bytes = ws_bytes - 3 * blk;
    const bool chain = lsub != nullptr && n > 1;
    // (no coupling: the local kernel's C IS the answer, written straight into g_diag)
    MF_LANE_LAUNCH((btd_chol_grad_local_kernel<T, D>), dim3((unsigned)cdiv(B * n, 64)), dim3(64), 0, st, B, n, ldiag,
                   chain ? lsub : static_cast<const T*>(nullptr), g_ldiag, chain ? g_lsub : static_cast<const T*>(nullptr),
                   chain ? C : g_diag, G, g_sub);
    if (!chain) return hipGetLastError() == hipSuccess ? 0 : -1000;
    const int rc = tak_scan<T, 2>(B, n, TakSrc<T>{C, G, nullptr}, g_diag, Zs, p, scan_bytes, st);
    if (rc != 0) return rc;
    const long cnt = B * (n - 1) * D * D;
    hipLaunchKernelGGL((axpy_kernel<T>), dim3((unsigned)cdiv(cnt, 256)), dim3(256), 0, st, cnt, T(2), static_cast<const T*>(Zs), g_sub);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <typename T>
int btd_diag_of_inverse_grad(long B, long n, const T* ldiag, const T* lsub, const T* sigma, const T* g_diag, const T* g_sub,
                             T* g_ldiag, T* g_lsub, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws == nullptr || ws_bytes < btd_grad_ws<T>(B, n)) return -11;
    const size_t blk = align_up(size_t(B) * n * D * D * sizeof(T));
    char* p = static_cast<char*>(ws);
    T* Q = reinterpret_cast<T*>(p); p += blk;
    T* G = reinterpret_cast<T*>(p); p += blk;
    T* A = reinterpret_cast<T*>(p); p += blk;
    const size_t scan_bytes = ws_bytes - 3 * blk;
    const bool chain = lsub != nullptr && n > 1;
    const T* ls = chain ? lsub : static_cast<const T*>(nullptr);
    const T* gs = chain ? g_sub : static_cast<const T*>(nullptr);
    MF_LANE_LAUNCH((btd_inv_grad_pre_kernel<T, D>), dim3((unsigned)cdiv(B * n, 64)), dim3(64), 0, st, B, n, ldiag, ls, g_diag, gs,
                   chain ? Q : A, G);
    if (chain) {
        const int rc = tak_scan<T, 3>(B, n, TakSrc<T>{Q, G, nullptr}, A, static_cast<T*>(nullptr), p, scan_bytes, st);
        if (rc != 0) return rc;
    }
    MF_LANE_LAUNCH((btd_inv_grad_post_kernel<T, D>), dim3((unsigned)cdiv(B * n, 64)), dim3(64), 0, st, B, n, ldiag, ls, sigma,
                   static_cast<const T*>(A), gs, static_cast<const T*>(G), g_ldiag, g_lsub);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// StateSpaceModel.marginal_covariances (+ subsequent_covariances) by the forward recursion; n = T >= 2 blocks
template <typename T>
int ssm_marginal_covs(long B, long n, const T* cholP0, const T* A, const T* cholQ, T* ocov, T* osub, void* ws, size_t ws_bytes,
                      hipStream_t st) {
    return tak_scan<T, 1>(B, n, TakSrc<T>{cholQ, A, cholP0}, ocov, osub, ws, ws_bytes, st);
}

template <typename T> size_t btd_udl_ws(long B, long n) {
    const long len0 = par_len0(B, n);
    if (len0 == 0) return 0;
    const ParPlan pl = par_plan(n, len0);
    size_t total = 0;
    for (int l = 1; l <= pl.levels; ++l)
        total += 6 * align_up(size_t(B) * pl.n[l] * D * D * sizeof(T)) + 2 * align_up(size_t(B) * pl.n[l] * D * sizeof(T));
    return total;
}

template <typename T>
int btd_udl(long B, long n, const T* diag, const T* sub, T* ut, T* chol_d, const T* eta, T* m_post, T* chol_dinv,
            int chain, void* ws, size_t ws_bytes, int* info, hipStream_t st) {
    const long len0 = sub ? par_len0(B, n) : 0;
    if (len0 == 0 || ws == nullptr || ws_bytes < btd_udl_ws<T>(B, n)) {
        MF_LANE_LAUNCH((btd_udl_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, n, diag, sub, ut,
                           chol_d, eta, m_post, chol_dinv, chain, info);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    // Delta_k = natural-order pivots of the REVERSED matrix: the Cholesky hierarchy with rev = 1 on level 0
    const ParPlan pl = par_plan(n, len0);
    struct Arr { T *Dv, *Gf, *GU, *F, *Pn, *M, *c, *Z; } arr[24];
    char* p = static_cast<char*>(ws);
    for (int l = 1; l <= pl.levels; ++l) {
        const size_t sz = align_up(size_t(B) * pl.n[l] * D * D * sizeof(T)), sv = align_up(size_t(B) * pl.n[l] * D * sizeof(T));
        T** f[6] = {&arr[l].Dv, &arr[l].Gf, &arr[l].GU, &arr[l].F, &arr[l].Pn, &arr[l].M};
        for (auto* q : f) { *q = reinterpret_cast<T*>(p); p += sz; }
        arr[l].c = reinterpret_cast<T*>(p); p += sv;
        arr[l].Z = reinterpret_cast<T*>(p); p += sv;
    }
    auto level = [&](int l) {
        if (l == 0) return ParLevel<T>{diag, nullptr, nullptr, sub, n, n - 1, -1, 1};
        return ParLevel<T>{arr[l].Dv, arr[l].Gf, arr[l].GU, arr[l].F, pl.n[l], pl.n[l], 0, 0};
    };
    if constexpr (D >= 2 && D + 1 <= 16) {
        if (row_par_path<T>(B * pl.n[1]) && (chain || eta == nullptr)) {
            // everything in row form (mf_row_par.hpp, mf_row_post.hpp): reversed up-sweep, down-sweep, emit, the offsets' affine scan
            const dim3 blk(64);
            auto rgrid = [](long rows) { return dim3((unsigned)cdiv(rows, 4)); };
            for (int l = 0; l < pl.levels; ++l) {
                const long P = pl.n[l + 1];
                if (l == 0)
                    hipLaunchKernelGGL((row::row_chol_up_kernel<T, D, false, false>), rgrid(B * P), blk, 0, st, level(l), B, pl.len[l], P,
                                       arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU, arr[l + 1].F, info);
                else
                    hipLaunchKernelGGL((row::row_chol_up_kernel<T, D, true, false>), rgrid(B * P), blk, 0, st, level(l), B, pl.len[l], P,
                                       arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU, arr[l + 1].F, info);
            }
            {
                const int l = pl.levels;
                hipLaunchKernelGGL((row::row_chol_down_kernel<T, D, false>), rgrid(B), blk, 0, st, level(l), B, pl.n[l], 1L,
                                   static_cast<const T*>(nullptr), arr[l].Pn, info);
            }
            for (int l = pl.levels - 1; l >= 1; --l) {
                const long P = pl.n[l + 1];
                hipLaunchKernelGGL((row::row_chol_down_kernel<T, D, false>), rgrid(B * P), blk, 0, st, level(l), B, pl.len[l], P,
                                   static_cast<const T*>(arr[l + 1].Pn), arr[l].Pn, info);
            }
            hipLaunchKernelGGL((row::row_udl_emit_kernel<T, D>), rgrid(B * pl.n[1]), blk, 0, st, B, n, len0, pl.n[1], diag, sub,
                               static_cast<const T*>(arr[1].Pn), ut, chol_d, chol_dinv, chain, info);
            if (eta) {
                hipLaunchKernelGGL((row::row_means_up0_kernel<T, D, true>), rgrid(B * pl.n[1]), blk, 0, st, B, B, n, len0, pl.n[1],
                                   static_cast<const T*>(ut), eta, arr[1].M, arr[1].c, T(1));      // ut = -U^T: x = eta + ut^T x'
                for (int l = 1; l < pl.levels; ++l) {
                    const long P = pl.n[l + 1];
                    hipLaunchKernelGGL((row::row_affine_up_kernel<T, D, false>), rgrid(B * P), blk, 0, st, B, pl.n[l], pl.len[l], P,
                                       static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c), arr[l + 1].M, arr[l + 1].c);
                }
                {
                    const int l = pl.levels;
                    hipLaunchKernelGGL((row::row_affine_down_kernel<T, D, false>), rgrid(B), blk, 0, st, B, pl.n[l], pl.n[l], 1L,
                                       static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                                       static_cast<const T*>(nullptr), arr[l].Z);
                }
                for (int l = pl.levels - 1; l >= 1; --l) {
                    const long P = pl.n[l + 1];
                    hipLaunchKernelGGL((row::row_affine_down_kernel<T, D, false>), rgrid(B * P), blk, 0, st, B, pl.n[l], pl.len[l], P,
                                       static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                                       static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
                }
                hipLaunchKernelGGL((row::row_post_emit_kernel<T, D>), rgrid(B * pl.n[1]), blk, 0, st, B, n, len0, pl.n[1],
                                   static_cast<const T*>(ut), eta, static_cast<const T*>(arr[1].Z), m_post,
                                   static_cast<const T*>(chol_dinv));
            }
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
    }
    for (int l = 0; l < pl.levels; ++l) {
        const long P = pl.n[l + 1];
        constexpr int x_lds = LdsSpike<T, D>::BYTES;
        if (x_path<T>())
            MF_LANE_LAUNCH((par_chol_up_x_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), x_lds, st, level(l),
                               B, pl.len[l], P, arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU, arr[l + 1].F, info);
        else
            if (const int rc = launch_chol_up<T>(level(l), l == 0 ? 1 : 2, B, pl.len[l], P, arr[l + 1].Dv, arr[l + 1].Gf, arr[l + 1].GU,
                              arr[l + 1].F, info, st)) return rc;
    }
    {
        const int l = pl.levels;
        MF_LANE_LAUNCH((par_chol_down_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, level(l), B,
                           pl.n[l], 1L, static_cast<const T*>(nullptr), arr[l].Pn, info);
    }
    for (int l = pl.levels - 1; l >= 1; --l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_chol_down_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, level(l), B,
                           pl.len[l], P, static_cast<const T*>(arr[l + 1].Pn), arr[l].Pn, info);
    }
    const dim3 g0((unsigned)cdiv(B * pl.n[1], 64));
    MF_LANE_LAUNCH((par_udl_emit_kernel<T, D>), g0, dim3(64), 0, st, B, n, len0, pl.n[1], diag, sub,
                       static_cast<const T*>(arr[1].Pn), ut, chol_d, chol_dinv, chain, info);
    if (eta) {
        // x_k = eta_k - U_k x_{k+1}: affine scan over the reversed positions, then the per-block finish
        MF_LANE_LAUNCH((par_post_up0_kernel<T, D>), g0, dim3(64), 0, st, B, n, len0, pl.n[1], static_cast<const T*>(ut),
                           eta, arr[1].M, arr[1].c, chain);
        for (int l = 1; l < pl.levels; ++l) {
            const long P = pl.n[l + 1];
            MF_LANE_LAUNCH((par_affine_up_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, B, pl.n[l],
                               pl.len[l], P, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                               arr[l + 1].M, arr[l + 1].c);
        }
        {
            const int l = pl.levels;
            MF_LANE_LAUNCH((par_affine_down_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, pl.n[l],
                               pl.n[l], 1L, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                               static_cast<const T*>(nullptr), arr[l].Z);
        }
        for (int l = pl.levels - 1; l >= 1; --l) {
            const long P = pl.n[l + 1];
            MF_LANE_LAUNCH((par_affine_down_kernel<T, D>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, B, pl.n[l],
                               pl.len[l], P, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                               static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
        }
        MF_LANE_LAUNCH((par_post_emit_kernel<T, D>), g0, dim3(64), 0, st, B, n, len0, pl.n[1], ut,
                           static_cast<const T*>(chol_d), eta, static_cast<const T*>(arr[1].Z), m_post, chol_dinv, chain, info);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int ssm_precision(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ,
                  const T* H, const T* y, const T* Rinv, int rinv_per_step, T* diag, T* sub, T* eta, hipStream_t st) {
    if (H && (m < 1 || m > MF_MAXM)) return LANE ? -3 : -100;
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, 1, nullptr, 0};
    if constexpr (!LANE) {                         // 10 <= d <= 15: a row per (series, block)
        const dim3 rgrid((unsigned)cdiv(B * Tn, 4));
        auto launch = [&](auto mtag) {
            constexpr int M = decltype(mtag)::value;
            hipLaunchKernelGGL((row::row_ssm_precision_kernel<T, D, M>), rgrid, dim3(64), 0, st, a, diag, sub, eta);
        };
        using std::integral_constant;
        if (!H || m == 1) launch(integral_constant<int, 1>{});
        else if (m == 2) launch(integral_constant<int, 2>{});
        else if (m == 3) launch(integral_constant<int, 3>{});
        else launch(integral_constant<int, 4>{});
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    const dim3 grid((unsigned)cdiv(B * Tn, 256)), block(256);
    if (m == 1) MF_LANE_LAUNCH((ssm_precision_kernel<T, D, 1>), grid, block, 0, st, a, diag, sub, eta);
    else MF_LANE_LAUNCH((ssm_precision_kernel<T, D, 0>), grid, block, 0, st, a, diag, sub, eta);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T, bool REV = false>
int ssm_means(long Bl, long Br, long Tn, const T* A, const T* offs, T* out, void* ws, size_t ws_bytes, hipStream_t st,
              const T** up_only = nullptr,        // up_only: stop before the emit kernel and hand out the chunk-end means
              bool have_up0 = false) {            // level 1 (M, c at the head of ws) was already filled by the caller
    const long n = Tn;
    const long len0 = par_len0(Br, n);
    if (len0 == 0 || ws == nullptr || ws_bytes < btd_solve_ws<T>(Bl, Br, n)) {
        if (up_only) return -16;
        if ((A && n >= 2) || REV)   // one lane per series: the level-0 emit kernel as ONE chunk (loads a group of steps ahead)
            MF_LANE_LAUNCH((par_means_emit_kernel<T, D, REV>), dim3((unsigned)cdiv(Br, 64)), dim3(64), 0, st, Bl, Br, n, n, 1L, A,
                               offs, static_cast<const T*>(nullptr), out);
        else
            MF_LANE_LAUNCH((ssm_means_kernel<T, D>), dim3((unsigned)cdiv(Br, 64)), dim3(64), 0, st, Bl, Br, Tn, A, offs,
                               out);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    // the mean recursion is the affine scan of the triangular solve with M_p = A_{p-1}, c_p = offs_p (same workspace)
    const ParPlan pl = par_plan(n, len0);
    struct Arr { T *M, *c, *Z; } arr[24];
    char* p = static_cast<char*>(ws);
    for (int l = 1; l <= pl.levels; ++l) {
        arr[l].M = reinterpret_cast<T*>(p); p += align_up(size_t(Br) * pl.n[l] * D * D * sizeof(T));
        arr[l].c = reinterpret_cast<T*>(p); p += align_up(size_t(Br) * pl.n[l] * D * sizeof(T));
        arr[l].Z = reinterpret_cast<T*>(p); p += align_up(size_t(Br) * pl.n[l] * D * sizeof(T));
    }
    bool row_levels = false;
    if constexpr (D >= 2 && D + 1 <= 16) row_levels = row_par_path<T>(Br * pl.n[1]);
    if (!have_up0) {
        if (row_levels) {
            if constexpr (D >= 2 && D + 1 <= 16)
                hipLaunchKernelGGL((row::row_means_up0_kernel<T, D, REV>), dim3((unsigned)cdiv(Br * pl.n[1], 4)), dim3(64), 0, st, Bl, Br,
                                   n, len0, pl.n[1], A, offs, arr[1].M, arr[1].c, T(1));
        } else
        MF_LANE_LAUNCH((par_means_up0_kernel<T, D, REV>), dim3((unsigned)cdiv(Br * pl.n[1], 64)), dim3(64), 0, st, Bl, Br, n,
                           len0, pl.n[1], A, offs, arr[1].M, arr[1].c);
    }
    if (row_levels) {
        if constexpr (D >= 2 && D + 1 <= 16) {     // the reduced levels of the affine scan in row form (mf_row_par.hpp)
            const dim3 blk(64);
            auto rgrid = [](long rows) { return dim3((unsigned)cdiv(rows, 4)); };
            for (int l = 1; l < pl.levels; ++l) {
                const long P = pl.n[l + 1];
                hipLaunchKernelGGL((row::row_affine_up_kernel<T, D, false>), rgrid(Br * P), blk, 0, st, Br, pl.n[l], pl.len[l], P,
                                   static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c), arr[l + 1].M, arr[l + 1].c);
            }
            {
                const int l = pl.levels;
                hipLaunchKernelGGL((row::row_affine_down_kernel<T, D, false>), rgrid(Br), blk, 0, st, Br, pl.n[l], pl.n[l], 1L,
                                   static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                                   static_cast<const T*>(nullptr), arr[l].Z);
            }
            for (int l = pl.levels - 1; l >= 1; --l) {
                const long P = pl.n[l + 1];
                hipLaunchKernelGGL((row::row_affine_down_kernel<T, D, false>), rgrid(Br * P), blk, 0, st, Br, pl.n[l], pl.len[l], P,
                                   static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                                   static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
            }
        }
    } else {
    for (int l = 1; l < pl.levels; ++l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_affine_up_kernel<T, D>), dim3((unsigned)cdiv(Br * P, 64)), dim3(64), 0, st, Br, pl.n[l],
                           pl.len[l], P, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c), arr[l + 1].M,
                           arr[l + 1].c);
    }
    {
        const int l = pl.levels;
        MF_LANE_LAUNCH((par_affine_down_kernel<T, D>), dim3((unsigned)cdiv(Br, 64)), dim3(64), 0, st, Br, pl.n[l],
                           pl.n[l], 1L, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                           static_cast<const T*>(nullptr), arr[l].Z);
    }
    for (int l = pl.levels - 1; l >= 1; --l) {
        const long P = pl.n[l + 1];
        MF_LANE_LAUNCH((par_affine_down_kernel<T, D>), dim3((unsigned)cdiv(Br * P, 64)), dim3(64), 0, st, Br, pl.n[l],
                           pl.len[l], P, static_cast<const T*>(arr[l].M), static_cast<const T*>(arr[l].c),
                           static_cast<const T*>(arr[l + 1].Z), arr[l].Z);
    }
    }
    if (up_only) { *up_only = arr[1].Z; return hipGetLastError() == hipSuccess ? 0 : -1000; }
    if (row_levels) {
        if constexpr (D >= 2 && D + 1 <= 16)
            hipLaunchKernelGGL((row::row_means_emit_kernel<T, D, REV>), dim3((unsigned)cdiv(Br * pl.n[1], 4)), dim3(64), 0, st, Bl, Br, n,
                               len0, pl.n[1], A, offs, static_cast<const T*>(arr[1].Z), out, T(1));
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    MF_LANE_LAUNCH((par_means_emit_kernel<T, D, REV>), dim3((unsigned)cdiv(Br * pl.n[1], 64)), dim3(64), 0, st, Bl, Br, n,
                       len0, pl.n[1], A, offs, static_cast<const T*>(arr[1].Z), out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int ssm_means_entry(long Bl, long Br, long Tn, const T* A, const T* offs, T* out, void* ws, size_t ws_bytes, hipStream_t st) {
    return ssm_means<T, false>(Bl, Br, Tn, A, offs, out, ws, ws_bytes, st);
}

// `marginals` (means + covariances [+ Cov(x_{k+1}, x_k)]) in ONE sweep per series; only where one lane per series is the
// chosen decomposition anyway (many series or a short chain) - otherwise -101 and the caller runs the two scans in time
// With few series both recursions are scans in time whose up / down sweeps stay separate (congruence and affine maps), but ONE
// emit kernel restarts both from the chunk boundaries (workspace: marginals_ws; -101 without it: the caller's two scans).
template <typename T> size_t marginals_ws(long B, long n) {
    if (n < 2 || par_len0(B, n) == 0) return 0;
    return align_up(btd_diag_of_inverse_ws<T>(B, n)) + btd_solve_ws<T>(B, B, n);
}
template <typename T>
int ssm_marginals(long B, long n, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, T* omean, T* ocov,
                  T* osub, void* ws, size_t ws_bytes, hipStream_t st, const T** boundaries = nullptr) {
    if (n < 2) return -101;
    const TakSrc<T> src{cholQ, A, cholP0};
    const long len0 = par_len0(B, n);
    if (len0 == 0) {
        MF_LANE_LAUNCH((par_tak_emit_kernel<T, D, 1, true>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, n, n, 1L, src,
                           static_cast<const T*>(nullptr), ocov, osub, TakMean<T>{mu0, b, omean, nullptr});
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (ws == nullptr || ws_bytes < marginals_ws<T>(B, n)) return -101;
    char* p = static_cast<char*>(ws);
    char* ws_cov = p; p += align_up(btd_diag_of_inverse_ws<T>(B, n));
    char* ws_mean = p;
    const long P = par_plan(n, len0).n[1];
    // level 1 of the means' hierarchy (M, c at the head of its workspace: the carve of ssm_means) is filled by the level-0
    // up-sweep of the covariances, which has every transition in registers anyway
    T* m1 = reinterpret_cast<T*>(ws_mean);
    T* c1 = reinterpret_cast<T*>(ws_mean + align_up(size_t(B) * P * D * D * sizeof(T)));
    const T *up_cov = nullptr, *up_mean = nullptr;
    int rc = tak_scan<T, 1>(B, n, src, ocov, osub, ws_cov, btd_diag_of_inverse_ws<T>(B, n), st, &up_cov,
                            TakMeanUp<T>{mu0, b, m1, c1});
    if (rc != 0) return rc;
    rc = ssm_means<T, false>(B, B, n, A, static_cast<const T*>(nullptr), omean, ws_mean, btd_solve_ws<T>(B, B, n), st, &up_mean,
                             true);
    if (rc != 0) return rc;
    if (boundaries) {          // the caller runs its own level-0 emit (kl_value: the fused KL emit)
        boundaries[0] = up_cov;
        boundaries[1] = up_mean;
        return 0;
    }
    if constexpr (D >= 2 && D + 1 <= 16) {
        if (row_par_path<T>(B * P)) {
            hipLaunchKernelGGL((row::row_cov_emit_kernel<T, D, 1, true>), dim3((unsigned)cdiv(B * P, 4)), dim3(64), 0, st, B, n, len0, P,
                               src, up_cov, ocov, osub, TakMean<T>{mu0, b, omean, up_mean});
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
    }
    MF_LANE_LAUNCH((par_tak_emit_kernel<T, D, 1, true>), dim3((unsigned)cdiv(B * P, 64)), dim3(64), 0, st, B, n, len0, P, src,
                       up_cov, ocov, osub, TakMean<T>{mu0, b, omean, up_mean});
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int ssm_marginals_entry(long B, long n, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, T* omean, T* ocov,
                        T* osub, void* ws, size_t ws_bytes, hipStream_t st) {
    return ssm_marginals<T>(B, n, mu0, cholP0, A, b, cholQ, omean, ocov, osub, ws, ws_bytes, st);
}

template <typename T>
int block_matmul(long B, long n, const T* X, long xs, const T* Y, long ys, T* out, hipStream_t st) {
    MF_LANE_LAUNCH((block_matmul_kernel<T, D>), dim3((unsigned)cdiv(B * n, 256)), dim3(256), 0, st, B, n, X, xs, Y, ys,
                       out);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// GPR log-likelihood with the Matern kernel -> SSM generation fused into the sweep (mf_gpr_fused.hpp).  This translation
// unit (state dimension D) holds the component signatures whose sizes add up to D; others return -101 and the caller
// materialises the state space model instead.
template <typename T, int O0, int O1>
int gpr_launch(const GprArgs<T>& a, RedSys<T> lvl0, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(a.B * a.P, 64)), block(64);
    if (a.P > 1) MF_LANE_LAUNCH((gpr_chunk_kernel<T, O0, O1, true>), grid, block, 0, st, a, lvl0);
    else MF_LANE_LAUNCH((gpr_chunk_kernel<T, O0, O1, false>), grid, block, 0, st, a, lvl0);
    return 0;
}
template <typename T>
int gpr_loglik(long B, long Tn, int ncomp, const int* orders, const T* lam, const T* var, int per_series, const T* t,
               const T* y, int m, int multi, const T* rinv, T jitter, T add_const, T* out, void* ws, size_t ws_bytes, int* info,
               long chunks, hipEvent_t ev0, hipEvent_t ev1, hipStream_t st) {
    if constexpr (D >= 7 && D + 1 <= 16) {
        // 7 <= d <= 15: any concatenation of components, one output (Sum) or one per component (up to four), generated inside the
        // row kernel (mf_row_gpr.hpp)
        if (ncomp > row::GPR_MAX_COMP || m < 1 || m > ROW_MAXM || (multi ? m != ncomp : m != 1)) return -101;
        if (ws == nullptr) return -15;
        const KfPlan pl = kf_plan<T>(B, Tn, m, 0, chunks, true);
        if (pl.path != KF_PATH_ROW) return -101;
        if (ws_bytes < plan_ws<T>(B, pl)) return -15;
        row::GprRowArgs<T> a{B, Tn, ncomp, multi, {}, lam, var, per_series ? (long)ncomp : 0L, t, y, rinv, jitter, pl.P, info};
        for (int c = 0; c < ncomp; ++c) a.order[c] = orders[c];
        char* p = static_cast<char*>(ws);
        RedSys<T> lvl0 = carve<T>(p, B, pl.P);
        const dim3 rgrid((unsigned)cdiv(B * pl.P, 4)), block(64);
        if (ev0) (void)hipEventRecord(ev0, st);
        if (m == 1) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 1>), rgrid, block, 0, st, a, lvl0);
        else if (m == 2) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 2>), rgrid, block, 0, st, a, lvl0);
        else if (m == 3) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 3>), rgrid, block, 0, st, a, lvl0);
        else if (m == 4) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 4>), rgrid, block, 0, st, a, lvl0);
        else if constexpr (!LANE) {
            if (m == 5) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 5>), rgrid, block, 0, st, a, lvl0);
            else if (m == 6) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 6>), rgrid, block, 0, st, a, lvl0);
            else if (m == 7) hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 7>), rgrid, block, 0, st, a, lvl0);
            else hipLaunchKernelGGL((row::gpr_row_kernel<T, D, 8>), rgrid, block, 0, st, a, lvl0);
        }
        if (ev1) (void)hipEventRecord(ev1, st);
        return reduce_levels_row<T>(lvl0, B, p, add_const, out, info, st);
    }
    const int o0 = orders[0], o1 = ncomp > 1 ? orders[1] : 0;
    if (ncomp > 2 || m != 1 || multi) return -101;
    if (ws == nullptr) return -15;
    long P = 1, L = 1;
    if (Tn >= 2) lds_partition(B, Tn, chunks, P, L);
    const size_t need = ROW_RED_SMALL ? levels_ws<T>(B, P, row_red_chunk(), row_red_final()) : levels_ws<T>(B, P);
    if (ws_bytes < need) return -15;                        // checked against the partition that is actually launched
    GprArgs<T> a{B, Tn, lam, var, per_series ? (long)ncomp : 0L, t, y, rinv, jitter, P, L, info};
    char* p = static_cast<char*>(ws);
    RedSys<T> lvl0 = carve<T>(p, B, P);
    if (ev0) (void)hipEventRecord(ev0, st);
    int rc = -101;
    if constexpr (D == 1) { if (o0 == 1 && o1 == 0) rc = gpr_launch<T, 1, 0>(a, lvl0, st); }
    if constexpr (D == 2) { if (o0 == 3 && o1 == 0) rc = gpr_launch<T, 3, 0>(a, lvl0, st); }
    if constexpr (D == 3) { if (o0 == 5 && o1 == 0) rc = gpr_launch<T, 5, 0>(a, lvl0, st); }
    if constexpr (D == 4) { if (o0 == 3 && o1 == 3) rc = gpr_launch<T, 3, 3>(a, lvl0, st); }
    if constexpr (D == 5) {
        if (o0 == 5 && o1 == 3) rc = gpr_launch<T, 5, 3>(a, lvl0, st);
        else if (o0 == 3 && o1 == 5) rc = gpr_launch<T, 3, 5>(a, lvl0, st);
    }
    if constexpr (D == 6) { if (o0 == 5 && o1 == 5) rc = gpr_launch<T, 5, 5>(a, lvl0, st); }
    if (rc != 0) return rc;
    if (ev1) (void)hipEventRecord(ev1, st);
    if (ROW_RED_SMALL) return reduce_levels_row<T>(lvl0, B, p, add_const, out, info, st);
    return reduce_levels<T>(lvl0, B, p, add_const, out, info, st);
}

template <typename T>
int sde_predict(long B, long N, long Np, const long long* idx, const T* Amt, const T* Qmt, const T* Atp, const T* Qtp,
                const T* means, const T* covs, const T* subseq, const T* m0, const T* P0, T* omean, T* ocov, int* info,
                hipStream_t st) {
    MF_LANE_LAUNCH((sde_predict_kernel<T, D>), dim3((unsigned)cdiv(B * Np, 64)), dim3(64), 0, st, B, N, Np, idx, Amt, Qmt,
                       Atp, Qtp, means, covs, subseq, m0, P0, omean, ocov, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int sde_cond_stats(long n, const T* Amt, const T* Qmt, const T* Atp, const T* Qtp, T* proj, T* cov, int* info, hipStream_t st) {
    MF_LANE_LAUNCH((sde_cond_stats_kernel<T, D>), dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, n, Amt, Qmt, Atp, Qtp, proj, cov, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int kf_grad(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
            const T* y, const T* Rinv, const T* pm, const T* pS, const T* pX, T* gmu0, T* gC0, T* gA, T* gb, T* gC, T* gH,
            T* gy, T* gOm, const T* weights, int rinv_per_step, int* info, hipStream_t st) {
    if (m < 1 || m > MF_MAXM) return LANE ? -3 : -100;
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, 1, info, 0, weights};
    const dim3 grid((unsigned)cdiv(B * Tn, 64)), block(64);
    // from d = 7 on (where the lane-per-point kernel spills) a 16-lane row per (series, time point): mf_row_grad.hpp
    static const int row_force = [] { const char* e = mf_knob("MF_GRAD_ROW"); return e ? std::atoi(e) : -1; }();
    if constexpr (D >= 2 && D + 1 <= 16) {
        if (row_force >= 0 ? row_force != 0 : D >= 7) {
            const dim3 rgrid((unsigned)cdiv(B * Tn, 4));
            auto launch = [&](auto mtag) {
                constexpr int M = decltype(mtag)::value;
                hipLaunchKernelGGL((row::row_kf_grad_kernel<T, D, M>), rgrid, block, 0, st, a, pm, pS, pX, gmu0, gC0, gA, gb, gC, gH, gy,
                                   gOm);
            };
            using std::integral_constant;
            if (m == 1) launch(integral_constant<int, 1>{});
            else if (m == 2) launch(integral_constant<int, 2>{});
            else if (m == 3) launch(integral_constant<int, 3>{});
            else launch(integral_constant<int, 4>{});
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
    }
    if (m == 1) MF_LANE_LAUNCH((kf_grad_kernel<T, D, 1>), grid, block, 0, st, a, pm, pS, pX, gmu0, gC0, gA, gb, gC, gH, gy, gOm);
    else MF_LANE_LAUNCH((kf_grad_kernel<T, D, 0>), grid, block, 0, st, a, pm, pS, pX, gmu0, gC0, gA, gb, gC, gH, gy, gOm);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// workspace of the adjoint sweeps: N, n, M, lam for every (series, block) + the scratch of the two scans in time
template <typename T> size_t adjoint_scan_ws(long B, long Tn) {
    const size_t a = btd_diag_of_inverse_ws<T>(B, Tn), b = btd_solve_ws<T>(B, B, Tn);
    return a > b ? a : b;
}
template <typename T> size_t adjoint_ws(long B, long Tn) {
    return 2 * align_up(size_t(B) * Tn * D * D * sizeof(T)) + 2 * align_up(size_t(B) * Tn * D * sizeof(T)) +
           adjoint_scan_ws<T>(B, Tn);
}
template <typename T> AdjointWs<T, D> carve_adjoint(void* ws, long B, long Tn) {
    char* p = static_cast<char*>(ws);
    AdjointWs<T, D> w;
    w.N = reinterpret_cast<T*>(p); p += align_up(size_t(B) * Tn * D * D * sizeof(T));
    w.M = reinterpret_cast<T*>(p); p += align_up(size_t(B) * Tn * D * D * sizeof(T));
    w.n = reinterpret_cast<T*>(p); p += align_up(size_t(B) * Tn * D * sizeof(T));
    w.lam = reinterpret_cast<T*>(p);
    return w;
}
// M_k = N_k + A_k^T M_{k+1} A_k and lam_k = n_k + A_k^T lam_{k+1} from the workspace inputs: few series -> the congruence and
// affine scans in time (mf_btd_par.hpp, SRC 2 / REV), many series -> one lane per series
template <typename T>
int adjoint_scan(long B, long Tn, const T* A, const AdjointWs<T, D>& w, void* ws, hipStream_t st) {
    if (par_len0(B, Tn) == 0 || Tn < 2) {
        MF_LANE_LAUNCH((ssm_adjoint_scan_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, Tn, A,
                           static_cast<const T*>(nullptr), static_cast<const T*>(nullptr), 1, w);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    char* scratch = static_cast<char*>(ws) + (adjoint_ws<T>(B, Tn) - adjoint_scan_ws<T>(B, Tn));
    const size_t bytes = adjoint_scan_ws<T>(B, Tn);
    int rc = tak_scan<T, 2>(B, Tn, TakSrc<T>{w.N, A, nullptr}, w.M, static_cast<T*>(nullptr), scratch, bytes, st);
    if (rc != 0) return rc;
    return ssm_means<T, true>(B, B, Tn, A, static_cast<const T*>(w.n), w.lam, scratch, bytes, st);
}

template <typename T>
int kl_grad(long B, long Tn, const T* mu0_1, const T* C0_1, const T* A_1, const T* b_1, const T* C_1, const T* mu0_2,
            const T* C0_2, const T* A_2, const T* b_2, const T* C_2, const T* pm, const T* pS, const T* weights,
            const T* in_N, const T* in_n, T* gmu0, T* gC0, T* gA, T* gb, T* gC, void* ws, size_t ws_bytes, int* info,
            hipStream_t st) {
    if (ws == nullptr || ws_bytes < adjoint_ws<T>(B, Tn)) return -23;
    if ((in_N == nullptr) != (in_n == nullptr)) return -17;
    AdjointWs<T, D> w = carve_adjoint<T>(ws, B, Tn);
    const dim3 per_step((unsigned)cdiv(B * Tn, 64)), per_series((unsigned)cdiv(B, 64)), block(64);
    if (in_N) {       // the forward sweep kept the inputs of the recursion (mf_ssm_kl_divergence: out_N, out_n); only read here
        w.N = const_cast<T*>(in_N);
        w.n = const_cast<T*>(in_n);
    } else {
        MF_LANE_LAUNCH((ssm_kl_adjoint_inputs_kernel<T, D>), per_step, block, 0, st, B, Tn, A_1, b_1, A_2, b_2, C_2, pm, w, info);
    }
    if (int rc = adjoint_scan<T>(B, Tn, A_1, w, ws, st)) return rc;
    AdjointLocalArgs<T, D> a{B, Tn, mu0_1, C0_1, A_1, b_1, C_1, mu0_2, C0_2, A_2, b_2, C_2, pm, pS, weights, gmu0, gC0, gA, gb, gC, info};
    if constexpr (D >= 7 && D + 1 <= 16)
        hipLaunchKernelGGL((row::row_adjoint_local_kernel<T, D, true>), dim3((unsigned)cdiv(B * Tn, 4)), block, 0, st, a, w);
    else
        MF_LANE_LAUNCH((ssm_adjoint_local_kernel<T, D, true>), per_step, block, 0, st, a, w);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int posterior_chain(long B, long Tn, int m, const T* mu0, const T* cholP0, const T* A, const T* b, const T* cholQ, const T* H,
                    const T* y, const T* Rinv, int rinv_per_step, T* a_post, T* mu0_post, T* b_post, T* cp0_post, T* cq_post,
                    int* info, hipStream_t st) {
    if (m < 1 || m > MF_MAXM) return LANE ? -4 : -100;       // (row-only build: more outputs belong to the tile engine)
    KfArgs<T> a{B, Tn, m, mu0, cholP0, A, b, cholQ, H, y, Rinv, rinv_per_step, 1, info, 0};
    const dim3 grid((unsigned)cdiv(B, 64)), block(64);
    if (m == 1) MF_LANE_LAUNCH((kf_posterior_chain_kernel<T, D, 1>), grid, block, 0, st, a, a_post, mu0_post, b_post, cp0_post, cq_post);
    else MF_LANE_LAUNCH((kf_posterior_chain_kernel<T, D, 0>), grid, block, 0, st, a, a_post, mu0_post, b_post, cp0_post, cq_post);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// kl_divergence with few series: workspace of the route over q1's marginals (0: one lane per series is the route)
template <typename T> size_t kl_ws(long B, long Tn) {
    if (Tn < 2 || par_len0(B, Tn) == 0) return 0;
    return align_up(size_t(B) * Tn * D * D * sizeof(T)) + align_up(size_t(B) * Tn * D * sizeof(T)) +
           align_up(size_t(B) * Tn * sizeof(T)) + marginals_ws<T>(B, Tn);
}

template <typename T>
int kl_value(long B, long Tn, const T* mu0_1, const T* C0_1, const T* A_1, const T* b_1, const T* C_1, const T* mu0_2,
             const T* C0_2, const T* A_2, const T* b_2, const T* C_2, T* out, T* out_means, T* out_covs, T* out_cross,
             T* out_N, T* out_n, void* ws, size_t ws_bytes, int* info, hipStream_t st) {
    if ((out_N == nullptr) != (out_n == nullptr)) return -18;
    const size_t need = kl_ws<T>(B, Tn);
    if (need != 0 && ws != nullptr && ws_bytes >= need) {
        // few series: q1's marginals by the scans in time, then every term is local (mf_kl_grad.hpp)
        char* p = static_cast<char*>(ws);
        T* pS = reinterpret_cast<T*>(p); p += align_up(size_t(B) * Tn * D * D * sizeof(T));
        T* pm = reinterpret_cast<T*>(p); p += align_up(size_t(B) * Tn * D * sizeof(T));
        T* part = reinterpret_cast<T*>(p); p += align_up(size_t(B) * Tn * sizeof(T));
        if (out_covs) pS = out_covs;
        if (out_means) pm = out_means;
        bool row_local = false;
        if constexpr (D >= 2 && D + 1 <= 16) row_local = row_par_path<T>(par_rows0(B, Tn));
        if (row_local) {
            if constexpr (D >= 2 && D + 1 <= 16) {
                // row form: the scans up to the chunk boundaries, then ONE level-0 kernel that restarts q1's moments and evaluates
                // the divergence's terms from them (mf_row_scan.hpp: row_kl_emit_kernel); moments only written when asked for
                const long len0 = par_len0(B, Tn);
                const long P = par_plan(Tn, len0).n[1];
                const T* bnd[2] = {nullptr, nullptr};
                const int rcb = ssm_marginals<T>(B, Tn, mu0_1, C0_1, A_1, b_1, C_1, pm, pS, out_cross, p, marginals_ws<T>(B, Tn), st, bnd);
                if (rcb != 0) return rcb;
                const row::RowKlChains<T, D> ch{mu0_1, C0_1, A_1, b_1, C_1, mu0_2, C0_2, A_2, b_2, C_2};
                hipLaunchKernelGGL((row::row_kl_emit_kernel<T, D>), dim3((unsigned)cdiv(B * P, 4)), dim3(64), 0, st, B, Tn, len0, P, ch,
                                   bnd[0], bnd[1], out_means, out_covs, out_cross, out_N, out_n, part, info);
                hipLaunchKernelGGL((row_sum_kernel<T>), dim3((unsigned)B), dim3(64), 0, st, P, static_cast<const T*>(part), out);
                return hipGetLastError() == hipSuccess ? 0 : -1000;
            }
        }
        const int rc = ssm_marginals<T>(B, Tn, mu0_1, C0_1, A_1, b_1, C_1, pm, pS, out_cross, p, marginals_ws<T>(B, Tn), st);
        if (rc != 0) return rc;
        if (row_local) {
            if constexpr (D >= 2 && D + 1 <= 16)
                hipLaunchKernelGGL((row::row_kl_local_kernel<T, D>), dim3((unsigned)cdiv(B * Tn, 4)), dim3(64), 0, st, B, Tn, mu0_1,
                                   C0_1, A_1, b_1, C_1, mu0_2, C0_2, A_2, b_2, C_2, static_cast<const T*>(pm),
                                   static_cast<const T*>(pS), part, out_N, out_n, info);
        } else
        MF_LANE_LAUNCH((ssm_kl_local_kernel<T, D>), dim3((unsigned)cdiv(B * Tn, 64)), dim3(64), 0, st, B, Tn, mu0_1, C0_1,
                           A_1, b_1, C_1, mu0_2, C0_2, A_2, b_2, C_2, static_cast<const T*>(pm), static_cast<const T*>(pS),
                           part, out_N, out_n, info);
        hipLaunchKernelGGL((row_sum_kernel<T>), dim3((unsigned)B), dim3(64), 0, st, Tn, static_cast<const T*>(part), out);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if ((out_means != nullptr) != (out_covs != nullptr) || (out_cross && !out_means)) return -15;
    if (Tn < 2 && out_cross) out_cross = nullptr;
    MF_LANE_LAUNCH((ssm_kl_kernel<T, D>), dim3((unsigned)cdiv(B, 64)), dim3(64), 0, st, B, Tn, mu0_1, C0_1, A_1, b_1, C_1,
                       mu0_2, C0_2, A_2, b_2, C_2, out, out_N, out_n, out_means, out_covs, out_cross, info);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T>
int marginals_grad(long B, long Tn, const T* C0, const T* A, const T* C, const T* pm, const T* pS, const T* gm, const T* gS,
                   T* gmu0, T* gC0, T* gA, T* gb, T* gC, void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws == nullptr || ws_bytes < adjoint_ws<T>(B, Tn)) return -16;
    const AdjointWs<T, D> w = carve_adjoint<T>(ws, B, Tn);
    const dim3 per_step((unsigned)cdiv(B * Tn, 64)), per_series((unsigned)cdiv(B, 64)), block(64);
    if (par_len0(B, Tn) == 0 || Tn < 2) {
        MF_LANE_LAUNCH((ssm_adjoint_scan_kernel<T, D>), per_series, block, 0, st, B, Tn, A, gm, gS, 0, w);
    } else {
        if constexpr (D >= 2 && D + 1 <= 16)
            hipLaunchKernelGGL((row::row_adjoint_sym_inputs_kernel<T, D>), dim3((unsigned)cdiv(B * Tn, 4)), block, 0, st, B * Tn, gm, gS,
                               w.N, w.n);
        else
            MF_LANE_LAUNCH((ssm_adjoint_sym_inputs_kernel<T, D>), per_step, block, 0, st, B * Tn, gm, gS, w);
        if (int rc = adjoint_scan<T>(B, Tn, A, w, ws, st)) return rc;
    }
    AdjointLocalArgs<T, D> a{B, Tn, nullptr, C0, A, nullptr, C, nullptr, nullptr, nullptr, nullptr, nullptr, pm, pS, nullptr,
                             gmu0, gC0, gA, gb, gC, nullptr};
    if constexpr (D >= 7 && D + 1 <= 16)
        hipLaunchKernelGGL((row::row_adjoint_local_kernel<T, D, false>), dim3((unsigned)cdiv(B * Tn, 4)), block, 0, st, a, w);
    else
        MF_LANE_LAUNCH((ssm_adjoint_local_kernel<T, D, false>), per_step, block, 0, st, a, w);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

template <typename T> const OpsTable<T>* table() {
    static const OpsTable<T> t = {
        &kf_loglik_ws<T>, &kf_loglik<T>, &btd_logdet_quad_ws<T>, &btd_logdet_quad<T>, &btd_cholesky_ws<T>, &btd_cholesky<T>,
        &btd_solve_ws<T>, &btd_solve<T>,    &btd_matvec<T>, &btd_logdet<T>,        &btd_diag_of_inverse_ws<T>, &btd_diag_of_inverse<T>, &ssm_marginal_covs<T>, &btd_udl_ws<T>, &btd_udl<T>,
        &ssm_precision<T>, &ssm_means_entry<T>, &block_matmul<T>, &gpr_loglik<T>, &sde_predict<T>, &kf_grad<T>, &kl_grad<T>, &posterior_chain<T>, &kl_value<T>, &marginals_grad<T>, &adjoint_ws<T>, &ssm_marginals_entry<T>, &kl_ws<T>, &marginals_ws<T>,
        &kf_loglik_plan<T>, &sde_cond_stats<T>, &btd_grad_ws<T>, &btd_cholesky_grad<T>, &btd_diag_of_inverse_grad<T>,
    };
    return &t;
}

}  // namespace

#define MF_CAT2(a, b) a##b
#define MF_CAT(a, b) MF_CAT2(a, b)
const OpsTable<float>* MF_CAT(ops_f32_d, MF_D)() { return table<float>(); }
const OpsTable<double>* MF_CAT(ops_f64_d, MF_D)() { return table<double>(); }

}  // namespace mf
