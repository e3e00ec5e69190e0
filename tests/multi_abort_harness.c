/* multi_abort_harness.c - host/qvz_multi.c's error path against a fake RCCL (tests/fake_rccl.c) and a device layer in which
 * every call fails (tests/fuzz_stubs.c): a two-device encode must come back with an error, and every communicator must have
 * been freed exactly once - aborted OR destroyed, never both, never twice. TEST INFRASTRUCTURE (tests/test_multi_cpu.py). */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qvz_encode.h"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "w");
    for (int i = 0; i < 64; ++i) fprintf(f, "%s\n", i & 1 ? "IIIIHHHH##" : "##IIIIHHHH");
    fclose(f);
    qvz_options_t o;
    qvz_options_default(&o);
    o.seed_mode = QVZ_SEED_DEBUG;
    const int devices[2] = {0, 1};
    qvz_result_t res;
    void *h = dlopen("librccl.so", RTLD_NOW);                  /* held over the call: the driver's dlclose must not unload the counters */
    if (!h) { printf("no fake rccl: %s\n", dlerror()); return 2; }
    const int rc = qvz_encode_file_devices(argv[1], argv[2], &o, devices, 2, 2, &res);
    void (*report)(int *, int *, int *, int *, int *) = (void (*)(int *, int *, int *, int *, int *))dlsym(h, "fake_rccl_report");
    int made, aborts, destroys, violations, live;
    report(&made, &aborts, &destroys, &violations, &live);
    printf("rc=%d made=%d aborts=%d destroys=%d violations=%d live=%d error=%s\n", rc, made, aborts, destroys, violations, live, qvz_encode_last_error());
    return 0;
}
