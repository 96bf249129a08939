/* pivot_trace.c -- ANALYSIS TOOL (test infrastructure, CPU only): what does the reference's Seidel run do on the
 * backward upper-bound LPs whose active pair moved?  Includes the oracle with its trace hooks defined and classifies
 * every 2-D LP of compute_controllable_sets by its pivot sequence.  Built and driven by tools/pivot_trace.py.
 *
 *   gcc -O2 -ffp-contract=off -o /tmp/pivot_trace tools/pivot_trace.c -lm
 *   /tmp/pivot_trace <workload.bin>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

struct tr_pivot { unsigned k; long row; double before[2]; double cmin, cmax; int amin, amax; unsigned n1d; double gap_best; int feas; };
static struct {
    const double *v, *a, *b, *c, *low, *high; const long *index_map; int nrows; long warm[2];
    int np; struct tr_pivot p[64];
} T;

static void tr_begin(const double *v, int nrows, const double *a, const double *b, const double *c, const double *low,
                     const double *high, const long *index_map, const long *active_c) {
    T.v = v; T.nrows = nrows; T.a = a; T.b = b; T.c = c; T.low = low; T.high = high; T.index_map = index_map;
    T.warm[0] = active_c[0]; T.warm[1] = active_c[1]; T.np = 0;
}
static void tr_pivot_fn(unsigned k, long row, const double *before, unsigned n1d, const double *a1, const double *b1, int feas) {
    if (T.np >= 64) return;
    struct tr_pivot *p = &T.p[T.np++];
    p->k = k; p->row = row; p->before[0] = before[0]; p->before[1] = before[1]; p->n1d = n1d; p->feas = feas;
    double cmin = -1e10, cmax = 1e10, min2 = -1e10, max2 = 1e10; int amin = -1, amax = -2;
    for (unsigned j = 0; j < n1d; ++j) {
        if (a1[j] > 1e-10) { double x = -b1[j] / a1[j]; if (x < cmax) { max2 = cmax; cmax = x; amax = (int)j; } else if (x < max2) max2 = x; }
        else if (a1[j] < -1e-10) { double x = -b1[j] / a1[j]; if (x > cmin) { min2 = cmin; cmin = x; amin = (int)j; } else if (x > min2) min2 = x; }
    }
    p->cmin = cmin; p->cmax = cmax; p->amin = amin; p->amax = amax;
    (void)min2; (void)max2;
}
static void tr_end(int result, const int *active);

#define ORC_TRACE_BEGIN(v, nrows, a, b, c, low, high, index_map, active_c) tr_begin(v, nrows, a, b, c, low, high, index_map, active_c)
#define ORC_TRACE_PIVOT(k, row, before, nrows_1d, a_1d, b_1d, sol_1d) tr_pivot_fn(k, row, before, nrows_1d, a_1d, b_1d, (sol_1d).result)
#define ORC_TRACE_END(sol) tr_end((sol).result, (sol).active_c)
#include "../oracle/seidel_oracle.c"

/* ---- statistics ------------------------------------------------------------------------------------------------- */
static long n_lp, n_upper, n_upper_warm, n_kept, n_moved, n_infeasible;
static long cls_seq[8];       /* moved: 0 = [0,1,+1], 1 = [0,1,+2], 2 = [0,1,+3..], 3 = no pivot at k=1, 4 = no pivot at k=0, 5 = other */
static long loose_hist[8][16];/* per class: rows not holding with the margin at z_w (beyond the two warm rows) */
static long viol_hist[8][16]; /* per class: rows violated (> 1e-9) at z_w */
static long extra_hist[32];
static double min_rel_seg = 1e300;
static long seg_small[12];    /* pivots after the warm pair: relative segment length < 10^-j */
static long first_is_most;    /* single/multi: the first violated row in order is also the most violated */
static long emu_ok[8], emu_total[8];
static long stage_moved_hist[64];
static int cur_stage_moved;

static double tol_at(double u, double x, int nrows) {
    double nmax = 0, cmax = 0;
    for (int r = 0; r < nrows; ++r) { double n = fabs(T.a[r]) + fabs(T.b[r]); if (n > nmax) nmax = n; if (fabs(T.c[r]) > cmax) cmax = fabs(T.c[r]); }
    return 1e-8 + 1e-6 * (nmax * (fabs(u) + fabs(x)) + cmax);
}

/* margin-emulation of the run after the warm pair: at every visited point every row either holds or is violated with the
 * margin, every 1-D problem has a clear winner and a long segment */
static int emulate_ok(int jfirst) {
    /* uses the trace itself: for each pivot after the first two, check margins at `before` */
    for (int j = jfirst; j < T.np; ++j) {
        const struct tr_pivot *p = &T.p[j];
        const double u = p->before[0], x = p->before[1];
        const double tol = tol_at(u, x, T.nrows);
        /* rows visited between the previous pivot and this one held (by margin?), this one violated by margin */
        unsigned kprev = T.p[j - 1].k;
        for (unsigned kk = kprev + 1; kk <= p->k; ++kk) {
            long r = T.index_map[kk];
            double res = T.a[r] * u + T.b[r] * x + T.c[r];
            if (kk < p->k) { if (!(res < -tol)) return 0; }
            else if (!(res > tol)) return 0;
        }
        const double len = p->cmax - p->cmin, sc = fmax(fabs(p->cmax), fabs(p->cmin)) + 1e-300;
        if (!(len > 1e-6 * sc)) return 0;
    }
    /* rows after the last pivot hold by margin at the final point: that is certificate (A), checked elsewhere */
    return 1;
}

static char cold_sigs[256][160]; static long cold_cnt[256]; static int cold_nsig = 0;
static void tr_end(int result, const int *active) {
    n_lp++;
    if (!(T.v[1] > 0)) return;  /* upper-bound LP of the backward scan: v = (-1e-9, 1) */
    n_upper++;
    const int warm_ok = T.warm[0] >= 0 && T.warm[1] >= 0 && T.warm[0] < T.nrows && T.warm[1] < T.nrows && T.warm[0] != T.warm[1];
    if (!warm_ok) {
        /* cold order (identity): signature of the run */
        char sig[160]; int n = 0;
        n += sprintf(sig + n, "cold(%ld,%ld) res=%d: ", T.warm[0], T.warm[1], result);
        for (int j = 0; j < T.np && n < 140; ++j) {
            const struct tr_pivot *p = &T.p[j];
            const double v1d = -T.b[p->row] * T.v[0] + T.a[p->row] * T.v[1];
            const int pick_min = fabs(v1d) < 1e-10 || v1d < 0;
            const int act = pick_min ? p->amin : p->amax;
            long lim = act < (int)p->k ? T.index_map[act] : -1 - (act - (int)p->k);
            char rc = p->row == 0 ? '0' : (p->row == 1 ? '1' : 'A');
            char lc = lim < 0 ? (lim == -4 ? 'H' : (lim == -3 ? 'L' : 'B')) : (lim == 0 ? '0' : (lim == 1 ? '1' : 'A'));
            sig[n++] = rc; sig[n++] = lc; sig[n++] = ' ';
        }
        sig[n] = 0;
        int f = -1;
        for (int q = 0; q < cold_nsig; ++q) if (!strcmp(cold_sigs[q], sig)) f = q;
        if (f < 0 && cold_nsig < 256) { f = cold_nsig++; strcpy(cold_sigs[f], sig); }
        if (f >= 0) cold_cnt[f]++;
        return;
    }
    n_upper_warm++;
    if (!result) { n_infeasible++; return; }
    const int kept = (active[0] == T.warm[0] && active[1] == T.warm[1]) || (active[0] == T.warm[1] && active[1] == T.warm[0]);
    if (kept) { n_kept++; return; }
    n_moved++;
    cur_stage_moved++;
    int cls, jw = -1;
    for (int j = 0; j < T.np && j < 2; ++j) {
        /* pivot on a warm row (k = 0: w1, k = 1: w0) whose 1-D optimum is limited by the other warm row (index 0 in the 1-D problem) */
        const struct tr_pivot *p = &T.p[j];
        const double v1d = -T.b[p->row] * T.v[0] + T.a[p->row] * T.v[1];
        const int pick_min = fabs(v1d) < 1e-10 || v1d < 0;
        const int act = pick_min ? p->amin : p->amax;
        if (p->k == 1 && act == 0) jw = j;
    }
    if (jw < 0) cls = T.np > 0 && T.p[0].k <= 1 ? 3 : 4;
    else { const int extra = T.np - 1 - jw; cls = extra == 1 ? 0 : (extra == 2 ? 1 : (extra >= 3 ? 2 : 5)); }
    if (1) {
        /* signature of the run: per pivot (row class, limiter class); classes: W warm row, 0 / 1 the x_next rows, A another row, B a box row */
        char sig[128]; int n = 0;
        for (int j = 0; j < T.np && n < 120; ++j) {
            const struct tr_pivot *p = &T.p[j];
            const double v1d = -T.b[p->row] * T.v[0] + T.a[p->row] * T.v[1];
            const int pick_min = fabs(v1d) < 1e-10 || v1d < 0;
            const int act = pick_min ? p->amin : p->amax;
            long lim = act < (int)p->k ? T.index_map[act] : -1 - (act - (int)p->k);
            char rc = p->k < 2 ? 'W' : (p->row == 0 ? '0' : (p->row == 1 ? '1' : 'A'));
            char lc = lim < 0 ? (lim == -4 ? 'H' : 'B') : (lim == T.warm[0] ? 'P' : (lim == T.warm[1] ? 'Q' : (lim == 0 ? '0' : (lim == 1 ? '1' : 'A'))));
            if (p->k < 2) rc = p->k == 0 ? 'q' : 'p';
            sig[n++] = rc; sig[n++] = lc; sig[n++] = ' ';
        }
        sig[n] = 0;
        static char sigs[256][128]; static long cnt[256]; static int nsig = 0;
        int f = -1;
        for (int q = 0; q < nsig; ++q) if (!strcmp(sigs[q], sig)) f = q;
        if (f < 0 && nsig < 256) { f = nsig++; strcpy(sigs[f], sig); }
        if (f >= 0) cnt[f]++;
        if (getenv("TRACE_SIGS") && n_moved == atol(getenv("TRACE_SIGS"))) { for (int q = 0; q < nsig; ++q) if (cnt[q] > 3) printf("%8ld  %s\n", cnt[q], sigs[q]); }
    }
    static int shown = 0;
    if (cls == 3 && shown < 25 && getenv("TRACE_SHOW")) {
        shown++;
        printf("LP warm=(%ld,%ld) final=(%d,%d) corner: ", T.warm[0], T.warm[1], active[0], active[1]);
        for (int j = 0; j < T.np; ++j) {
            const struct tr_pivot *p = &T.p[j];
            const double v1d = -T.b[p->row] * T.v[0] + T.a[p->row] * T.v[1];
            const int pick_min = fabs(v1d) < 1e-10 || v1d < 0;
            const int act = pick_min ? p->amin : p->amax;
            long lim = act < (int)p->k ? T.index_map[act] : -1 - (act - (int)p->k);
            printf(" [k=%u row=%ld from(%.3g,%.3g) lim=%ld seg=%.2g]", p->k, p->row, p->before[0], p->before[1], lim, (p->cmax - p->cmin) / (fmax(fabs(p->cmax), fabs(p->cmin)) + 1e-300));
        }
        printf("\n");
    }
    cls_seq[cls]++;
    extra_hist[T.np < 31 ? T.np : 31]++;
    if (jw >= 0 && T.np > jw + 1) {
        /* z_w = the point before the next pivot */
        const double u = T.p[jw + 1].before[0], x = T.p[jw + 1].before[1];
        const double tol = tol_at(u, x, T.nrows);
        int loose = 0, viol = 0; double worst = 1e-9; long most = -1, first = -1;
        for (int kk = 2; kk < T.nrows; ++kk) {
            long r = T.index_map[kk];
            double res = T.a[r] * u + T.b[r] * x + T.c[r];
            if (!(res < -tol)) loose++;
            if (res > 1e-9) { viol++; if (first < 0) first = r; }
            if (res > worst) { worst = res; most = r; }
        }
        /* box rows */
        if (!(T.low[1] - x < -tol)) loose++;
        if (!(x - T.high[1] < -tol)) loose++;
        loose_hist[cls][loose < 15 ? loose : 15]++;
        viol_hist[cls][viol < 15 ? viol : 15]++;
        if (first == most) first_is_most++;
        for (int j = jw + 1; j < T.np; ++j) {
            const double len = T.p[j].cmax - T.p[j].cmin, sc = fmax(fabs(T.p[j].cmax), fabs(T.p[j].cmin)) + 1e-300;
            const double rel = len / sc;
            if (rel < min_rel_seg) min_rel_seg = rel;
            for (int e = 0; e < 12; ++e) if (rel < pow(10.0, -e)) seg_small[e]++;
        }
        emu_total[cls]++;
        if (emulate_ok(jw + 1)) emu_ok[cls]++;
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s workload.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int B = hdr[0], d = hdr[1], nseg = hdr[2], N = hdr[3];
    double *coef = malloc(sizeof(double) * (size_t)B * 4 * nseg * d), *breaks = malloc(sizeof(double) * (nseg + 1)),
           *grid = malloc(sizeof(double) * (N + 1)), *vlim = malloc(sizeof(double) * (size_t)B * 2 * d),
           *alim = malloc(sizeof(double) * (size_t)B * 2 * d), *K = malloc(sizeof(double) * 2 * (N + 1));
    size_t ok = fread(coef, sizeof(double), (size_t)B * 4 * nseg * d, f) + fread(breaks, sizeof(double), nseg + 1, f) +
                fread(grid, sizeof(double), N + 1, f) + fread(vlim, sizeof(double), (size_t)B * 2 * d, f) +
                fread(alim, sizeof(double), (size_t)B * 2 * d, f);
    (void)ok;
    fclose(f);
    for (int b = 0; b < B; ++b) {
        orc_wrapper *w = orc_wrapper_new(d, nseg, N, coef + (size_t)b * 4 * nseg * d, breaks, grid, vlim + (size_t)b * 2 * d,
                                         alim + (size_t)b * 2 * d, 7, 1);
        orc_compute_controllable_sets(w, 0.0, 0.0, K);
        orc_wrapper_free(w);
    }
    printf("B=%d d=%d N=%d: 2-D LPs %ld, upper-bound %ld (valid warm pair %ld): kept %ld, moved %ld, infeasible %ld\n", B, d, N,
           n_lp, n_upper, n_upper_warm, n_kept, n_moved, n_infeasible);
    printf("moved pairs per trajectory: %.2f\n", (double)n_moved / B);
    const char *names[] = {"z_w + 1 pivot", "z_w + 2 pivots", "z_w + >=3 pivots", "z_w not visited (warm pivot)", "z_w not visited (no warm pivot)", "other (z_w final?)"};
    for (int c = 0; c < 6; ++c) {
        printf("  %-22s %8ld (%.2f / traj)  margin-emulation ok %ld of %ld\n", names[c], cls_seq[c], (double)cls_seq[c] / B, emu_ok[c], emu_total[c]);
        printf("      loose rows at z_w:"); for (int j = 0; j < 16; ++j) printf(" %ld", loose_hist[c][j]); printf("\n");
        printf("      violated at z_w:  "); for (int j = 0; j < 16; ++j) printf(" %ld", viol_hist[c][j]); printf("\n");
    }
    if (getenv("TRACE_COLD")) for (int q = 0; q < cold_nsig; ++q) printf("%8ld  %s\n", cold_cnt[q], cold_sigs[q]);
    printf("pivots per moved LP:"); for (int j = 0; j < 32; ++j) if (extra_hist[j]) printf(" %d:%ld", j, extra_hist[j]); printf("\n");
    printf("first violated row in order == most violated: %ld\n", first_is_most);
    printf("smallest relative 1-D segment after the warm pair: %.3g; below 1e-j:", min_rel_seg);
    for (int e = 0; e < 12; ++e) printf(" %ld", seg_small[e]); printf("\n");
    return 0;
}
