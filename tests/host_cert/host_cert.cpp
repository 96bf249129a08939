// host_cert.cpp -- TEST INFRASTRUCTURE: the lane-level certificates of kernel family 3 (toppra_amd/csrc/tpr_cert_lane.hip.inc,
// the product's own source, compiled here as plain C++ through tests/host_cert/hip_shim) driven over whole backward scans on
// the CPU, against the CPU restatement of the reference (oracle/seidel_oracle.c) stage LP by stage LP.
//
// For every stage of every trajectory the harness does what cert_solve_kernel's stage block does -- proposal + pair
// certificate for the upper-bound LP, the lower-bound certificate, the equality certificate of the first stage -- and what
// the certificates do not answer is taken from the restatement (the role of the cooperative batches' full iteration).  The
// restatement solves BOTH LPs of every stage in any case, so every certified answer is checked: it must be the reference's
// bits -- u, x, the active pair -- and the reference must not have failed where a certificate answered.  Nothing of this
// can be observed on the GPU more cheaply: a fast-mode certificate that answers an LP on which the reference exits
// "infeasible" at an intermediate pivot shows up here as a mismatch on the stage it happens.
//
//   g++ -O2 -ffp-contract=off -Itests/host_cert/hip_shim -o host_cert tests/host_cert/host_cert.cpp oracle_obj.o -lm
//   ./host_cert workload.bin      (tests/test_host_cert.py and tools/host_cert_hunt.py write the workloads)
#define TPR_CERT_WHY 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/toppra_hip.h"
#include "../../toppra_amd/csrc/tpr_device.hpp"
#include "../../toppra_amd/csrc/tpr_cert_lane.hip.inc"

extern "C" {
struct orc_wrapper;
orc_wrapper *orc_wrapper_new(int d, int nseg, int N, const double *coef, const double *breaks, const double *grid,
                             const double *vlim, const double *alim, int flags, int solve_lp1d);
void orc_wrapper_free(orc_wrapper *w);
void orc_solve_stagewise_optim(orc_wrapper *w, int i, const double *g, double x_min, double x_max, double x_next_min,
                               double x_next_max, double *out);
const double *orc_wrapper_a(const orc_wrapper *w);
const double *orc_wrapper_b(const orc_wrapper *w);
const double *orc_wrapper_c(const orc_wrapper *w);
const double *orc_wrapper_low(const orc_wrapper *w);
const double *orc_wrapper_high(const orc_wrapper *w);
int orc_wrapper_nC(const orc_wrapper *w);
void orc_wrapper_active(const orc_wrapper *w, long *out4);
}

struct Stats {
    long stages = 0, upper = 0, upper_cert = 0, upper_moved = 0, upper_moved_cert = 0, lower = 0, lower_cert = 0;
    long eq_upper = 0, eq_upper_cert = 0, eq_lower = 0, eq_lower_cert = 0;
    long mismatch = 0, ref_failed_cert_answered = 0, feas_upper = 0, feas_upper_cert = 0, feas_lower = 0, feas_lower_cert = 0;
    long ref_infeasible = 0;
    long why[16] = {0}, why_pair[16] = {0}, why_low[16] = {0}, why_fu[16] = {0}, why_fl[16] = {0};
};
static Stats G;
static int g_verbose = 0;
static int g_legacy = 0;  // 1: the round-2/3 FAST certificates (cert_propose2 / cert_lower / cert_equality) instead of the sound ones
static int g_minform = 0; // 1: the slides' verdicts as a running minimum (the TOPPRAsd kernels' form) instead of sign bits (the other kernels')

static bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b) || (isnan(a) && isnan(b)); }

template <int D>
struct Runner {
    using C = tpr::CertStage<D, 1>;
    static constexpr int nC = 2 + 4 * D;
    // one lane's "LDS columns" (BS = 1): q', q'' of the two gridpoints (two parities) and the row constants
    double q[2][2][D];
    double lim[6 + 2 * D];
    double alim_lane[2 * D];
    C S;

    void report(const char *what, int b, int i, const tpr::Lp2dOut &o, const double *ref, long r0, long r1) {
        if (G.mismatch < 20 || g_verbose)
            std::printf("MISMATCH %s traj %d stage %d: certificate ok=%d (u %.17g x %.17g pair %d,%d) reference (u %.17g x %.17g pair %ld,%ld)\n",
                        what, b, i, (int)o.ok, o.u, o.x, C::id_of(o.ac0), C::id_of(o.ac1), ref[0], ref[1], r0, r1);
        G.mismatch++;
    }

    // mode 0: compute_controllable_sets (the backward scan); mode 1: compute_feasible_sets (ascending, stages 0 .. N-1)
    void run(int b, int nseg, int N, const double *coef, const double *breaks, const double *grid, const double *vlim,
             const double *alim, int flags, double sd_end, int mode) {
        orc_wrapper *w = orc_wrapper_new(D, nseg, N, coef, breaks, grid, vlim, alim, flags, 1);
        const bool interp = (flags >> 2) & 1;
        const int wnC = orc_wrapper_nC(w);
        const double *WA = orc_wrapper_a(w), *WB = orc_wrapper_b(w), *WC = orc_wrapper_c(w), *WL = orc_wrapper_low(w), *WH = orc_wrapper_high(w);
        for (int k = 0; k < D; ++k) {
            S.cpos[k] = WC[2 + k];          // -amax
            S.cneg[k] = WC[2 + D + k];      // +amin
        }
        S.lim = lim;
        for (int k = 0; k < D; ++k) { alim_lane[2 * k] = S.cneg[k]; alim_lane[2 * k + 1] = -S.cpos[k]; }
        S.alim_lane = alim_lane;
        lim[0] = tpr::kVarMin; lim[1] = -tpr::kVarMax;
        S.cmax = 0.0; S.min_range = 3.0e300;
        for (int k = 0; k < D; ++k) {
            lim[6 + k] = S.cpos[k]; lim[6 + D + k] = S.cneg[k];
            S.cmax = fmax(S.cmax, fmax(fabs(S.cpos[k]), fabs(S.cneg[k])));
            S.min_range = fmin(S.min_range, -S.cpos[k] - S.cneg[k]);
        }
        int up0 = 4, up1 = 4, dn0 = 4, dn1 = 4;
        auto load = [&](int i, int par) {
            for (int k = 0; k < D; ++k) { q[par][0][k] = WA[(size_t)i * wnC + 2 + k]; q[par][1][k] = WB[(size_t)i * wnC + 2 + k]; }
        };
        auto bind = [&](int i) {
            const int pc = i & 1, pn = pc ^ 1;
            S.cur1 = q[pc][0]; S.cur2 = q[pc][1]; S.nxt1 = q[pn][0]; S.nxt2 = q[pn][1];
            for (int k = 0; k < D; ++k) {
                S.rc1[k] = q[pc][0][k]; S.rc2[k] = q[pc][1][k];
                S.rn1[k] = interp ? q[pn][0][k] : 0.0; S.rn2[k] = interp ? q[pn][1][k] : 0.0;
            }
        };
        auto state_to_internal = [&](int &u0, int &u1, int &d0, int &d1) {
            long st[4];
            orc_wrapper_active(w, st);
            u0 = C::vi_of((int)st[0]); u1 = C::vi_of((int)st[1]); d0 = C::vi_of((int)st[2]); d1 = C::vi_of((int)st[3]);
        };
        if (mode == 0) {
            double kn0 = sd_end * sd_end, kn1 = kn0;
            bool failed = false;
            load(N, N & 1);
            for (int i = N - 1; i >= 0 && !failed; --i) {
                load(i, i & 1);
                bind(i);
                S.two_delta = 2 * (grid[i + 1] - grid[i]); S.n0 = kn0; S.n1 = kn1;
                S.low1 = WL[2 * i + 1]; S.high1 = WH[2 * i + 1];
                S.publish_special(lim);
                double nmax;
                const bool nok = S.norms(nmax);
                tpr::Lp2dOut su, sl;
                int up_p, up_q; bool up_ok; double prow[3], qrow[3];
                const int wdn0 = dn0, wdn1 = dn1;
                tpr::tpr_cert_why = 0;
                if (g_legacy) tpr::cert_propose2<D, 1>(S, -1e-9, 1.0, dn0, dn1, dn0 != dn1, false, nmax, up_p, up_q, up_ok, prow, qrow);
                else if (g_minform) tpr::cert_propose_sound<D, 1, 0, false>(S, -1e-9, 1.0, dn0, dn1, dn0 != dn1, nmax, up_p, up_q, up_ok, prow, qrow);
                else tpr::cert_propose_sound<D, 1, 0, true>(S, -1e-9, 1.0, dn0, dn1, dn0 != dn1, nmax, up_p, up_q, up_ok, prow, qrow);
                const int why = tpr::tpr_cert_why;
                bool need_u = g_legacy ? !tpr::cert_pair_rows<D, 1, false, false>(S, -1e-9, 1.0, up_p, up_q, nok & up_ok, prow[0], prow[1], prow[2],
                                                                                  qrow[0], qrow[1], qrow[2], dn0, dn1, nmax, su)
                                       : !tpr::cert_pair_rows<D, 1, false, true>(S, -1e-9, 1.0, up_p, up_q, nok & up_ok, prow[0], prow[1], prow[2],
                                                                                 qrow[0], qrow[1], qrow[2], dn0, dn1, nmax, su);
                tpr::tpr_cert_why = 0;
                bool need_l = g_legacy ? !tpr::cert_lower<D, 1>(S, nok, nmax, sl)
                                       : (g_minform ? !tpr::cert_lower_sound<D, 1, true, false>(S, nok, nmax, up0, up1, sl) : !tpr::cert_lower_sound<D, 1, true, true>(S, nok, nmax, up0, up1, sl));
                if (need_l && !(kn0 == kn1)) G.why_low[tpr::tpr_cert_why & 15]++;
                const bool eq = kn0 == kn1;
                if (eq) {
                    tpr::Lp2dOut eu, el;
                    tpr::tpr_cert_why = 0;
                    if (g_legacy) tpr::cert_equality<D, 1>(S, nok, nmax, dn0, dn1, up0, up1, eu, el);
                    else tpr::cert_equality_sound<D, 1>(S, nok, nmax, dn0, dn1, up0, up1, eu, el);
                    if (g_verbose && !eu.ok && b < 4) std::printf("EQ upper refused: why %d (pair %d,%d)\n", tpr::tpr_cert_why, eu.ac0, eu.ac1);
                    if (need_u & eu.ok) { su = eu; need_u = false; }
                    if (need_l & el.ok) { sl = el; need_l = false; }
                }
                // the reference: both LPs of the stage
                const double g_upper[2] = {1e-9, -1}, g_lower[2] = {-1e-9, 1};
                double ru[2], rl[2];
                long st[4];
                orc_solve_stagewise_optim(w, i, g_upper, NAN, NAN, kn0, kn1, ru);
                orc_wrapper_active(w, st);
                G.stages++;
                (eq ? G.eq_upper : G.upper)++;
                const bool ref_u_ok = !isnan(ru[0]);
                if (!ref_u_ok) G.ref_infeasible++;
                const bool moved = ref_u_ok && !((C::vi_of((int)st[2]) == wdn0 && C::vi_of((int)st[3]) == wdn1) || (C::vi_of((int)st[2]) == wdn1 && C::vi_of((int)st[3]) == wdn0));
                if (!eq && moved) G.upper_moved++;
                if (!eq && need_u) { G.why[why & 15]++; if (why == 0) G.why_pair[(nok ? 0 : 1)]++; }
                if (!need_u) {
                    (eq ? G.eq_upper_cert : G.upper_cert)++;
                    if (!eq && moved) G.upper_moved_cert++;
                    if (!ref_u_ok) { G.ref_failed_cert_answered++; report("upper (reference failed)", b, i, su, ru, st[2], st[3]); }
                    else if (!(same_bits(su.u, ru[0]) && same_bits(su.x, ru[1]) && C::id_of(su.ac0) == st[2] && C::id_of(su.ac1) == st[3]))
                        report("upper", b, i, su, ru, st[2], st[3]);
                }
                orc_solve_stagewise_optim(w, i, g_lower, NAN, NAN, kn0, kn1, rl);
                orc_wrapper_active(w, st);
                (eq ? G.eq_lower : G.lower)++;
                const bool ref_l_ok = !isnan(rl[0]);
                if (!ref_l_ok) G.ref_infeasible++;
                if (!need_l) {
                    (eq ? G.eq_lower_cert : G.lower_cert)++;
                    if (!ref_l_ok) { G.ref_failed_cert_answered++; report("lower (reference failed)", b, i, sl, rl, st[0], st[1]); }
                    else if (!(same_bits(sl.u, rl[0]) && same_bits(sl.x, rl[1]) && C::id_of(sl.ac0) == st[0] && C::id_of(sl.ac1) == st[1]))
                        report("lower", b, i, sl, rl, st[0], st[1]);
                }
                state_to_internal(up0, up1, dn0, dn1);
                double lo = rl[1], hi = ru[1];
                if (lo < 0) lo = 0;
                if (isnan(lo) || isnan(hi)) failed = true;
                kn0 = lo; kn1 = hi;
            }
        } else {
            load(0, 0);
            for (int i = 0; i < N; ++i) {
                load(i + 1, (i + 1) & 1);
                bind(i);
                double low1 = WL[2 * i + 1], high1 = WH[2 * i + 1];
                low1 = low1 > -tpr::kFeasMaxX ? low1 : -tpr::kFeasMaxX;
                high1 = high1 < tpr::kFeasMaxX ? high1 : tpr::kFeasMaxX;
                S.two_delta = 2 * (grid[i + 1] - grid[i]); S.n0 = -tpr::kFeasMaxX; S.n1 = tpr::kFeasMaxX; S.low1 = low1; S.high1 = high1;
                S.publish_special(lim);
                double nmax;
                const bool nok = S.norms(nmax);
                tpr::Lp2dOut su, sl;
                int up_p, up_q; bool up_ok; double prow[3], qrow[3];
                tpr::tpr_cert_why = 0;
                if (g_minform) tpr::cert_propose_sound<D, 1, 0, false>(S, 1e-9, 1.0, dn0, dn1, dn0 != dn1, nmax, up_p, up_q, up_ok, prow, qrow);
                else tpr::cert_propose_sound<D, 1, 0, true>(S, 1e-9, 1.0, dn0, dn1, dn0 != dn1, nmax, up_p, up_q, up_ok, prow, qrow);
                const int why_u = tpr::tpr_cert_why;
                const bool need_u = !tpr::cert_pair_rows<D, 1, false, true>(S, 1e-9, 1.0, up_p, up_q, nok & up_ok, prow[0], prow[1], prow[2],
                                                                             qrow[0], qrow[1], qrow[2], dn0, dn1, nmax, su);
                if (need_u) G.why_fu[why_u & 15]++;
                tpr::tpr_cert_why = 0;
                const bool need_l = g_minform ? !tpr::cert_lower_sound<D, 1, false, false>(S, nok, nmax, up0, up1, sl) : !tpr::cert_lower_sound<D, 1, false, true>(S, nok, nmax, up0, up1, sl);
                if (need_l) G.why_fl[tpr::tpr_cert_why & 15]++;
                // reachability_algorithm.py:149-157: min x first (g = (1e-9, 1): state active_c_up), then max x
                const double g_lo[2] = {1e-9, 1}, g_hi[2] = {-1e-9, -1};
                double rl[2], ru[2];
                long st[4];
                orc_solve_stagewise_optim(w, i, g_lo, -tpr::kFeasMaxX, tpr::kFeasMaxX, -tpr::kFeasMaxX, tpr::kFeasMaxX, rl);
                orc_wrapper_active(w, st);
                G.feas_lower++;
                if (!need_l) {
                    G.feas_lower_cert++;
                    if (isnan(rl[0])) { G.ref_failed_cert_answered++; report("feasible min-x (reference failed)", b, i, sl, rl, st[0], st[1]); }
                    else if (!(same_bits(sl.u, rl[0]) && same_bits(sl.x, rl[1]) && C::id_of(sl.ac0) == st[0] && C::id_of(sl.ac1) == st[1]))
                        report("feasible min-x", b, i, sl, rl, st[0], st[1]);
                }
                orc_solve_stagewise_optim(w, i, g_hi, -tpr::kFeasMaxX, tpr::kFeasMaxX, -tpr::kFeasMaxX, tpr::kFeasMaxX, ru);
                orc_wrapper_active(w, st);
                G.feas_upper++;
                if (!need_u) {
                    G.feas_upper_cert++;
                    if (isnan(ru[0])) { G.ref_failed_cert_answered++; report("feasible max-x (reference failed)", b, i, su, ru, st[2], st[3]); }
                    else if (!(same_bits(su.u, ru[0]) && same_bits(su.x, ru[1]) && C::id_of(su.ac0) == st[2] && C::id_of(su.ac1) == st[3]))
                        report("feasible max-x", b, i, su, ru, st[2], st[3]);
                }
                state_to_internal(up0, up1, dn0, dn1);
            }
        }
        orc_wrapper_free(w);
    }
};

template <int D>
static void run_all(int B, int nseg, int N, const double *coef, const double *breaks, const double *grid, const double *vlim,
                    const double *alim, const double *sd_end, int flags, int mode, int breaks_per_traj) {
    Runner<D> *R = new Runner<D>();
    for (int b = 0; b < B; ++b)
        R->run(b, nseg, N, coef + (size_t)b * 4 * nseg * D, breaks + (breaks_per_traj ? (size_t)b * (nseg + 1) : 0), grid,
               vlim + (size_t)b * 2 * D, alim + (size_t)b * 2 * D, flags, sd_end ? sd_end[b] : 0.0, mode);
    delete R;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s workload.bin [-v]\n", argv[0]); return 2; }
    g_verbose = argc > 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int hdr[8];
    if (std::fread(hdr, sizeof(int), 8, f) != 8) return 2;
    const int B = hdr[0], d = hdr[1], nseg = hdr[2], N = hdr[3], flags = hdr[4], mode = hdr[5], has_sd_end = hdr[6];
    g_legacy = hdr[7] & 1;
    g_minform = (hdr[7] >> 1) & 1;
    std::vector<double> coef((size_t)B * 4 * nseg * d), breaks(nseg + 1), grid(N + 1), vlim((size_t)B * 2 * d), alim((size_t)B * 2 * d), sd_end(B, 0.0);
    size_t got = std::fread(coef.data(), 8, coef.size(), f) + std::fread(breaks.data(), 8, breaks.size(), f) + std::fread(grid.data(), 8, grid.size(), f) +
                 std::fread(vlim.data(), 8, vlim.size(), f) + std::fread(alim.data(), 8, alim.size(), f);
    if (has_sd_end) got += std::fread(sd_end.data(), 8, sd_end.size(), f);
    std::fclose(f);
    (void)got;
#define RUN(DD) case DD: run_all<DD>(B, nseg, N, coef.data(), breaks.data(), grid.data(), vlim.data(), alim.data(), has_sd_end ? sd_end.data() : nullptr, flags, mode, 0); break
    switch (d) {
        RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); RUN(8); RUN(9); RUN(12); RUN(13);
        default: std::fprintf(stderr, "dof %d not instantiated\n", d); return 2;
    }
    std::printf("{\"B\": %d, \"d\": %d, \"N\": %d, \"mode\": %d, \"stages\": %ld, \"upper\": %ld, \"upper_cert\": %ld, \"upper_moved\": %ld, "
                "\"upper_moved_cert\": %ld, \"lower\": %ld, \"lower_cert\": %ld, \"eq_upper\": %ld, \"eq_upper_cert\": %ld, \"eq_lower\": %ld, "
                "\"eq_lower_cert\": %ld, \"feas_upper\": %ld, \"feas_upper_cert\": %ld, \"feas_lower\": %ld, \"feas_lower_cert\": %ld, "
                "\"ref_infeasible\": %ld, \"ref_failed_cert_answered\": %ld, \"mismatch\": %ld}\n",
                B, d, N, mode, G.stages, G.upper, G.upper_cert, G.upper_moved, G.upper_moved_cert, G.lower, G.lower_cert, G.eq_upper,
                G.eq_upper_cert, G.eq_lower, G.eq_lower_cert, G.feas_upper, G.feas_upper_cert, G.feas_lower, G.feas_lower_cert,
                G.ref_infeasible, G.ref_failed_cert_answered, G.mismatch);
    std::fprintf(stderr, "refused upper LPs by reason (0 = the pair certificate itself):");
    for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d:%ld", k, G.why[k]);
    std::fprintf(stderr, "\nrefused lower LPs:");
    for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d:%ld", k, G.why_low[k]);
    std::fprintf(stderr, "\nrefused feasible max-x LPs:");
    for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d:%ld", k, G.why_fu[k]);
    std::fprintf(stderr, "\nrefused feasible min-x LPs:");
    for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %d:%ld", k, G.why_fl[k]);
    std::fprintf(stderr, "\n");
    return G.mismatch ? 1 : 0;
}
