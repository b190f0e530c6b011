// twin_counted.cpp -- oracle/redmax_tensorfree.c compiled with counting doubles (counted.h) + the entry tests/flop_count.py calls.
#include "counted.h"
long long fc_add, fc_mul, fc_div, fc_trans;
#define double cd
#include "../../oracle/redmax_tensorfree.c"

// counts[k][4] = {add, mul, div, trans} of k = 0: one residual-only evaluation (the line search's nargout == 1 path), 1: one (g, H)
// evaluation, 2: one solve dx = -H\g, 3: the Newton bookkeeping of one iteration with one trial point (norms, compensated update)
extern "C" void fc_counts(const orc_desc* d, const cd* q, const cd* qdot, cd h, long long* counts) {
    tf_model* m = tf_create(d);
    const int nr = m->nr;
    cd* x = (cd*)malloc(sizeof(cd) * (nr + 1));
    cd* xB = (cd*)malloc(sizeof(cd) * (nr + 1));
    cd* g = (cd*)malloc(sizeof(cd) * (nr + 1));
    cd* dx = (cd*)malloc(sizeof(cd) * (nr + 1));
    cd* lo = (cd*)calloc(nr + 1, sizeof(cd));
    cd* H = (cd*)malloc(sizeof(cd) * ((size_t)nr * nr + 1));
    for (int i = 0; i < nr; i++) x[i] = xB[i] = q[i].v + h.v * qdot[i].v;
    auto take = [&](int k) {
        counts[4 * k + 0] = fc_add; counts[4 * k + 1] = fc_mul; counts[4 * k + 2] = fc_div; counts[4 * k + 3] = fc_trans;
        fc_add = fc_mul = fc_div = fc_trans = 0;
    };
    fc_add = fc_mul = fc_div = fc_trans = 0;
    tf_eval_lo(m, x, lo, q, xB, h, g, NULL);
    take(0);
    tf_eval_lo(m, x, lo, q, xB, h, g, H);
    take(1);
    tf_solve_neg(nr, H, g, dx);
    take(2);
    {   // what newton() itself adds per iteration with one trial point (tf_newton): |dx|^2, |g|^2 twice, the TwoSum update
        cd dn = 0.0, f0 = 0.0, gn = 0.0;
        for (int i = 0; i < nr; i++) { dn += dx[i] * dx[i]; f0 += g[i] * g[i]; }
        (void)sqrt(dn);
        f0 *= 0.5;
        for (int i = 0; i < nr; i++) {
            const cd a = x[i], b = lo[i] + cd(1.0) * dx[i];
            const cd s = a + b, bb = s - a;
            lo[i] = (a - (s - bb)) + (b - bb);
        }
        for (int i = 0; i < nr; i++) gn += g[i] * g[i];
        (void)sqrt(gn);
        (void)(cd(0.5) * gn < f0);
    }
    take(3);
    free(x); free(xB); free(g); free(dx); free(lo); free(H);
    tf_destroy(m);
}
