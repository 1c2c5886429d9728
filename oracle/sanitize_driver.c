/* sanitize_driver.c -- the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan; tests/test_sanitizers_cpu.py).
 * TEST INFRASTRUCTURE like the oracle itself.  One executable, oracle compiled in: a small synthetic bundle adjustment (cameras on a ring, some of them
 * and some points unobserved, ragged tracks) through sfmba_oracle_solve / _build_reduced / the evaluation entry points, and Powell's function through the
 * dense model of the same trust-region loop. */
#include "sfmba_oracle.c"

static unsigned long long lcg_state = 88172645463325252ULL;
static double urand(void) { lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(lcg_state >> 11) / 9007199254740992.0; }

int main(void) {
    enum { NC = 9, NP = 260 };
    double cam[NC * 6], cam0[NC * 6], pt[NP * 3], pt0[NP * 3];
    for (int j = 0; j < NC; ++j) {
        const double a = 0.35 * j;
        double* c = cam + 6 * j;
        c[0] = 0.0; c[1] = a; c[2] = 0.0;                        /* rotation about y: the optical axis (row 3 of R) points at the origin */
        const double R[9] = { cos(a), 0, sin(a), 0, 1, 0, -sin(a), 0, cos(a) };
        const double centre[3] = { 5.0 * sin(a), 0.2 * (j % 3), -5.0 * cos(a) };
        for (int r = 0; r < 3; ++r) c[3 + r] = -(R[3 * r] * centre[0] + R[3 * r + 1] * centre[1] + R[3 * r + 2] * centre[2]);   /* t = -R c */
    }
    cam[0] = cam[1] = cam[2] = 0.0;                               /* camera 0: theta = 0, the first-order branch */
    for (int i = 0; i < 3 * NP; ++i) pt[i] = 2.0 * urand() - 1.0;
    /* observations: ragged tracks of 2 .. 6 views; camera NC - 1 and the last ten points are never observed */
    static int32_t oc[NP * 6], op[NP * 6];
    static double oxy[NP * 12];
    int64_t n_obs = 0;
    const double focal_true = 800.0;
    for (int i = 0; i < NP - 10; ++i) {
        const int k = 2 + (int)(urand() * 5.0), first = (int)(urand() * (NC - 1));
        for (int v = 0; v < k; ++v) {
            const int j = (first + v) % (NC - 1);
            double r[2];
            sfmba_oracle_residual(cam + 6 * j, pt + 3 * i, focal_true, 0.0, 0.0, r);
            if (!isfinite(r[0]) || !isfinite(r[1])) continue;
            oc[n_obs] = j; op[n_obs] = i;
            oxy[2 * n_obs] = r[0] + 0.3 * (urand() - 0.5); oxy[2 * n_obs + 1] = r[1] + 0.3 * (urand() - 0.5);
            ++n_obs;
        }
    }
    memcpy(cam0, cam, sizeof(cam)); memcpy(pt0, pt, sizeof(pt));
    for (int e = 6; e < 6 * NC; ++e) cam0[e] += 0.01 * (urand() - 0.5);
    for (int e = 0; e < 3 * NP; ++e) pt0[e] += 0.02 * (urand() - 0.5);
    int bad = 0;
    for (int variant = 0; variant < 3; ++variant) {
        double c2[NC * 6], p2[NP * 3], f = 1.02 * focal_true;
        memcpy(c2, cam0, sizeof(c2)); memcpy(p2, pt0, sizeof(p2));
        sfmba_options opt; sfmba_oracle_options_default(&opt); opt.max_seconds = 0.0;
        sfmba_summary sum; sfmba_iteration trace[64]; int tl = 0;
        sfmba_oracle_set_minimizer_variant(variant);
        const int rc = sfmba_oracle_solve(NC, c2, NP, p2, n_obs, oc, op, oxy, &f, &opt, &sum, trace, 64, &tl);
        printf("BA variant %d: rc %d termination %d iterations %d cost %.6e -> %.6e focal %.3f\n", variant, rc, sum.termination, sum.iterations, sum.initial_cost, sum.final_cost, f);
        bad |= rc != 0 || sum.termination != SFMBA_CONVERGENCE || !(sum.final_cost < sum.initial_cost);
        for (int e = 0; e < 6; ++e) bad |= c2[6 * (NC - 1) + e] != cam0[6 * (NC - 1) + e];       /* the unobserved camera is untouched */
    }
    sfmba_oracle_set_minimizer_variant(0);
    {   /* the reduced system at the initial point */
        const int dim = 6 * (NC - 1) + 1;
        double* S = (double*)malloc(sizeof(double) * (size_t)dim * dim); double* rhs = (double*)malloc(sizeof(double) * dim); double* sc = (double*)malloc(sizeof(double) * dim);
        const int info = sfmba_oracle_build_reduced(NC, cam0, NP, pt0, n_obs, oc, op, oxy, 1.02 * focal_true, NULL, 1e4, S, rhs, sc);
        double asym = 0.0;
        for (int a = 0; a < dim; ++a) for (int b = 0; b < a; ++b) asym = fmax(asym, fabs(S[a * dim + b] - S[b * dim + a]));
        printf("reduced system: info %d dim %d max asymmetry %.2e\n", info, dim, asym);
        bad |= info != 0 || !(asym < 1e-6 * fabs(S[0]));
        free(S); free(rhs); free(sc);
    }
    {   /* Powell's function: the dense model of the same loop */
        double x[4] = { 3.0, -1.0, 0.0, 1.0 };
        sfmba_options opt; sfmba_oracle_options_default(&opt); opt.max_seconds = 0.0; opt.max_iters = 100;
        sfmba_summary sum; sfmba_iteration trace[64]; int tl = 0;
        const int rc = sfmba_oracle_solve_dense(0, 4, x, &opt, &sum, trace, 64, &tl);
        printf("Powell: rc %d termination %d iterations %d final cost %.6e\n", rc, sum.termination, sum.iterations, sum.final_cost);
        bad |= rc != 0 || sum.termination != SFMBA_CONVERGENCE || sum.iterations != 14;
    }
    printf("oracle sanitize driver: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
