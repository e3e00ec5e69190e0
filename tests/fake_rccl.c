/* fake_rccl.c - a stand-in librccl.so for tests/test_multi_cpu.py: communicators are small heap objects with a state, and every
 * call on a communicator that has been freed (ncclCommAbort and ncclCommDestroy both free theirs) is counted as a violation.
 * TEST INFRASTRUCTURE: what host/qvz_multi.c must never do on its error path is exactly what RCCL would crash or hang on. */
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

typedef struct fake_comm { int id, state; } fake_comm;      /* state: 1 live, 2 aborted, 3 destroyed */
static fake_comm *all[64];
static int made, aborts, destroys, violations, reduces;

int ncclCommInitAll(fake_comm **comms, int ndev, const int *devlist) {
    (void)devlist;
    for (int i = 0; i < ndev && made < 64; ++i) {
        fake_comm *c = (fake_comm *)malloc(sizeof(*c));
        c->id = made; c->state = 1;
        all[made++] = c;                                     /* kept (never freed for real) so that a late call can be recognised */
        comms[i] = c;
    }
    return 0;
}
int ncclCommAbort(fake_comm *c) { if (!c || c->state != 1) { ++violations; return 1; } c->state = 2; ++aborts; return 0; }
int ncclCommDestroy(fake_comm *c) { if (!c || c->state != 1) { ++violations; return 1; } c->state = 3; ++destroys; return 0; }
int ncclAllReduce(const void *s, void *r, size_t n, int dt, int op, fake_comm *c, void *st) {
    (void)s; (void)r; (void)n; (void)dt; (void)op; (void)st;
    if (!c || c->state != 1) { ++violations; return 1; }
    ++reduces;
    usleep(20000);                                           /* long enough for another part to fail meanwhile */
    return c->state == 1 ? 0 : 1;
}
const char *ncclGetErrorString(int e) { return e ? "fake error" : "ok"; }
void fake_rccl_report(int *o_made, int *o_aborts, int *o_destroys, int *o_violations, int *o_live) {
    int live = 0;
    for (int i = 0; i < made; ++i) live += all[i]->state == 1;
    *o_made = made; *o_aborts = aborts; *o_destroys = destroys; *o_violations = violations; *o_live = live;
}
