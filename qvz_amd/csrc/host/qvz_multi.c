/* qvz_multi.c - one file encoded on SEVERAL devices of one process (include/qvz_encode.h, qvz_encode_file_devices).
 *
 * The reference is one thread on one core; its loops over reads (cluster.c:65-131, codebook.c:193-205,
 * qv_compressor.c:76-135) are independent per read, so reads shard over devices:
 *
 *   partition    device d holds the reads of shards [d * S / D, (d + 1) * S / D) of qvz_shard_bounds(N, S): a contiguous range
 *                that starts on a WELL word boundary. The output is the SAME shard set a single device writes for S shards,
 *                byte for byte.
 *   centroids    the k reads libc rand() picks (cluster.c:198-201) are drawn once, before any HIP call, and every part is
 *                handed their rows (in one process the mapped input is visible to all host threads; between processes
 *                this is the broadcast from the owning rank, qvz_amd/pipeline.py fetch_centroids)
 *   all-reduce   per k-means iteration the accumulators u64[k][C] and the counts [k]; once the PMF counters [k][C][72][72] -
 *                every part's u32 counts widened to 64 bits first, so that a file of 2^32 reads or more (which the
 *                reference's u32 counters cannot hold, pmf.h:35-36) sums without wrapping. RCCL (ncclAllReduce over xGMI, the communicator of ncclCommInitAll) when the ordinals
 *                are distinct; a host-staged sum otherwise (the same ordinal listed twice: two parts sharing one GPU,
 *                which is how the partition logic is tested on a one-GPU box - RCCL refuses duplicate devices)
 *   codebooks    N-independent and deterministic: generated once (part 0) from the reduced counters, shared as host structs
 *   quantize / code   per part, embarrassingly parallel: the part's WELL draws start first_read * C draws into the stream
 *   distortion   per-read error sums of all parts, added in file order on the host: bit-identical to the single-device value
 *
 * One host thread per part; every stage is the single-device stage of qvz_encode.c (qvz_job_*), with the all-reduce hooked in. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "qvz_coder.h"
#include "qvz_encode.h"
#include "qvz_job.h"
#include "qvz_shardset.h"

#define A QVZ_A

/* ---- RCCL, bound at run time (librccl.so is part of ROCm; the single-device paths must not depend on it) ---- */
typedef struct ncclComm *ncclComm_t;
typedef int (*fn_comm_init_all)(ncclComm_t *comms, int ndev, const int *devlist);
typedef int (*fn_comm_destroy)(ncclComm_t comm);
typedef int (*fn_all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t comm, void *stream);
typedef const char *(*fn_error_string)(int);
typedef int (*fn_comm_abort)(ncclComm_t comm);
enum { NCCL_SUM = 0, NCCL_UINT32 = 3, NCCL_UINT64 = 5 };      /* ncclSum, ncclUint32, ncclUint64 (rccl.h: ncclRedOp_t, ncclDataType_t) */

typedef struct {
    void *lib;
    fn_comm_init_all comm_init_all;
    fn_comm_destroy comm_destroy;
    fn_all_reduce all_reduce;
    fn_error_string error_string;
    fn_comm_abort comm_abort;
} rccl_t;

static int rccl_open(rccl_t *r) {
    memset(r, 0, sizeof(*r));
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", NULL};
    for (int i = 0; names[i] && !r->lib; ++i) r->lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r->lib) return qvz_job_fail(QVZ_E_HIP, "librccl.so not found: %s", dlerror());
    r->comm_init_all = (fn_comm_init_all)dlsym(r->lib, "ncclCommInitAll");
    r->comm_destroy = (fn_comm_destroy)dlsym(r->lib, "ncclCommDestroy");
    r->all_reduce = (fn_all_reduce)dlsym(r->lib, "ncclAllReduce");
    r->error_string = (fn_error_string)dlsym(r->lib, "ncclGetErrorString");
    r->comm_abort = (fn_comm_abort)dlsym(r->lib, "ncclCommAbort");
    if (!r->comm_init_all || !r->comm_destroy || !r->all_reduce) return qvz_job_fail(QVZ_E_HIP, "librccl.so lacks ncclCommInitAll / ncclAllReduce");
    return QVZ_OK;
}

/* ---- what the parts share ---- */
typedef struct {
    uint32_t parts;
    /* a barrier the parts can be released from when one of them fails (a pthread barrier would keep the others waiting) */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    uint32_t arrived, generation;
    int use_rccl;
    rccl_t rccl;
    ncclComm_t *comms;            /* [parts]; an entry is NULL once its communicator has been aborted (ncclCommAbort frees it) */
    uint8_t busy[64];             /* part d is inside a host call on comms[d]: nobody else may abort it meanwhile */
    /* host-staged reduction: every part copies its buffer here, part 0 adds them up */
    uint8_t *stage;               /* [parts][stage_bytes] */
    size_t stage_bytes;
    int failed;                   /* set by any part: the others stop at the next barrier */
} shared_t;

typedef struct {
    shared_t *sh;
    uint32_t part;
    int device;
    qvz_job_t job;
    const qvz_options_t *opts;
    const uint8_t *text;          /* the part's first row */
    uint64_t first_read, reads;
    uint32_t columns, shards;     /* the part's shard count */
    const uint64_t *bounds;       /* [shards + 1], relative to first_read */
    const uint8_t *centroid_rows;
    const uint32_t *well_state;
    qvz_codebook_t ***cbs_slot;   /* part 0 publishes the codebooks here */
    uint64_t stream_bytes;
    int rc;
    char err[512];
} part_t;

/* waits for all parts; returns nonzero (at once, for everybody) when some part has failed */
static int bar_wait(shared_t *sh) {
    int failed;
    pthread_mutex_lock(&sh->mu);
    if (!sh->failed && ++sh->arrived == sh->parts) {
        sh->arrived = 0;
        sh->generation += 1;
        pthread_cond_broadcast(&sh->cv);
    } else {
        const uint32_t gen = sh->generation;
        while (!sh->failed && gen == sh->generation) pthread_cond_wait(&sh->cv, &sh->mu);
    }
    failed = sh->failed;
    pthread_mutex_unlock(&sh->mu);
    return failed;
}

/* Marks the encode as failed and releases everybody: the host barrier at once, and - with RCCL - the ranks parked in a
 * collective on the device, by aborting the communicators. ncclCommAbort frees its communicator, so each one is aborted
 * AT MOST ONCE: it is taken out of sh->comms under the lock before the call (several parts may fail at the same time), a
 * communicator whose owner is inside ncclAllReduce right now is left to that owner (part_allreduce aborts it when the call
 * returns), and whatever is NULL here is neither aborted again nor destroyed by qvz_encode_file_devices. */
static void bar_abort(shared_t *sh) {
    ncclComm_t mine[64];
    uint32_t n = 0;
    pthread_mutex_lock(&sh->mu);
    sh->failed = 1;
    pthread_cond_broadcast(&sh->cv);
    if (sh->use_rccl && sh->comms && sh->rccl.comm_abort)
        for (uint32_t d = 0; d < sh->parts && d < 64; ++d)
            if (sh->comms[d] && !sh->busy[d]) { mine[n++] = sh->comms[d]; sh->comms[d] = NULL; }
    pthread_mutex_unlock(&sh->mu);
    for (uint32_t i = 0; i < n; ++i) sh->rccl.comm_abort(mine[i]);
}

static int part_fail(part_t *p, int rc) {
    p->rc = rc;
    snprintf(p->err, sizeof(p->err), "device %d (part %u): %s", p->device, p->part, qvz_encode_last_error());
    bar_abort(p->sh);
    return rc;
}

/* sum over the parts, in place on every part's device buffer; called by all parts with the same arguments */
static int part_allreduce(void *ctx, void *d_buf, uint64_t count, int width, void *stream) {
    part_t *p = (part_t *)ctx;
    shared_t *sh = p->sh;
    if (sh->use_rccl) {
        ncclComm_t comm, late = NULL;
        pthread_mutex_lock(&sh->mu);
        comm = sh->failed ? NULL : sh->comms[p->part];
        if (comm) sh->busy[p->part] = 1;
        pthread_mutex_unlock(&sh->mu);
        if (!comm) { qvz_job_fail(QVZ_E_HIP, "ncclAllReduce: another part failed, communicator aborted"); return QVZ_E_HIP; }
        const int e = sh->rccl.all_reduce(d_buf, d_buf, (size_t)count, width == 8 ? NCCL_UINT64 : NCCL_UINT32, NCCL_SUM, comm, stream);
        pthread_mutex_lock(&sh->mu);
        sh->busy[p->part] = 0;
        if (sh->failed && sh->comms[p->part] && sh->rccl.comm_abort) { late = sh->comms[p->part]; sh->comms[p->part] = NULL; }   /* bar_abort left it to us */
        pthread_mutex_unlock(&sh->mu);
        if (late) sh->rccl.comm_abort(late);
        if (e) { qvz_job_fail(QVZ_E_HIP, "ncclAllReduce: %s", sh->rccl.error_string ? sh->rccl.error_string(e) : "error"); return QVZ_E_HIP; }
        if (late) { qvz_job_fail(QVZ_E_HIP, "ncclAllReduce: another part failed, communicator aborted"); return QVZ_E_HIP; }
        return QVZ_OK;
    }
    const size_t bytes = (size_t)count * (size_t)width;
    int rc = QVZ_OK;
    if (bytes > sh->stage_bytes) { qvz_job_fail(QVZ_E_ARG, "reduction of %zu bytes exceeds the staging area", bytes); rc = QVZ_E_ARG; }
    if (!rc && (qvz_hip_copy_d2h(sh->stage + (size_t)p->part * sh->stage_bytes, d_buf, bytes, stream) || qvz_hip_stream_sync(stream))) rc = QVZ_E_HIP;
    if (rc) { bar_abort(sh); return rc; }
    if (bar_wait(sh)) return QVZ_E_HIP;
    if (p->part == 0) {
        if (width == 8) {
            uint64_t *acc = (uint64_t *)sh->stage;
            for (uint32_t q = 1; q < sh->parts; ++q) {
                const uint64_t *src = (const uint64_t *)(sh->stage + (size_t)q * sh->stage_bytes);
                for (uint64_t i = 0; i < count; ++i) acc[i] += src[i];
            }
        } else {
            uint32_t *acc = (uint32_t *)sh->stage;
            for (uint32_t q = 1; q < sh->parts; ++q) {
                const uint32_t *src = (const uint32_t *)(sh->stage + (size_t)q * sh->stage_bytes);
                for (uint64_t i = 0; i < count; ++i) acc[i] += src[i];
            }
        }
    }
    if (bar_wait(sh)) return QVZ_E_HIP;
    if (qvz_hip_copy_h2d(d_buf, sh->stage, bytes, stream) || qvz_hip_stream_sync(stream)) { bar_abort(sh); return QVZ_E_HIP; }
    if (bar_wait(sh)) return QVZ_E_HIP;                          /* the sum is read by everyone before the area is reused */
    return QVZ_OK;
}

static void *part_main(void *arg) {
    part_t *p = (part_t *)arg;
    shared_t *sh = p->sh;
    qvz_options_t o = *p->opts;
    int rc;
    o.device = p->device;
    o.seed_mode = QVZ_SEED_EXPLICIT;                             /* one WELL state for the whole file, drawn by the caller */
    memcpy(o.well_state, p->well_state, sizeof(o.well_state));
    o.verbose = p->part == 0 ? p->opts->verbose : 0;             /* the k-means lines are the same everywhere: printed once */
    o.uncompressed_name = NULL;
    FILE *fu = NULL;
    if ((rc = qvz_job_init_part(&p->job, p->text, p->reads, p->columns, &o, p->first_read, p->centroid_rows))) part_fail(p, rc);
    p->job.allreduce = part_allreduce;
    p->job.allreduce_ctx = p;
    p->job.shard_bounds = p->bounds;
    p->job.collect = 1;
    /* The stages contain collectives: a part that fails releases the others from whatever barrier they wait at
     * (bar_abort) and everyone stops at the stage boundary. */
#define STAGE(call)                                                                               \
    do {                                                                                          \
        if (!p->rc && (rc = (call))) part_fail(p, rc);                                            \
        if (bar_wait(sh)) goto out;                                                               \
    } while (0)
    STAGE(qvz_job_upload(&p->job));
    STAGE(qvz_job_kmeans(&p->job, 0));
    STAGE(qvz_job_stats(&p->job));
    if (!p->rc && (rc = qvz_job_start_draws(&p->job))) part_fail(p, rc);        /* this part's WELL draws, beside the codebooks */
    if (p->part == 0) {                                          /* identical on every device: generated once */
        if (!p->rc && (rc = qvz_job_codebooks(&p->job))) part_fail(p, rc);
        *p->cbs_slot = p->job.cbs;
    }
    if (bar_wait(sh)) goto out;
    if (p->part != 0) p->job.cbs = *p->cbs_slot;                 /* shared, read-only from here on; part 0 frees them */
    {
        double part_distortion = 0.0;
        if (p->opts->uncompressed_name) {                        /* -u: every part writes its own rows of the text (main.c:81 opened it) */
            fu = fopen(p->opts->uncompressed_name, "r+b");
            if (!fu || fseeko(fu, (off_t)(p->first_read * ((uint64_t)p->columns + 1)), SEEK_SET)) { qvz_job_fail(QVZ_E_ARG, "Unable to open uncompressed file"); part_fail(p, QVZ_E_ARG); goto out; }
        }
        if ((rc = qvz_job_compress_shardset(&p->job, NULL, fu, p->shards, &p->stream_bytes, NULL, &part_distortion))) part_fail(p, rc);
    }
out:
    if (fu) fclose(fu);
    if (p->part != 0) p->job.cbs = NULL;                         /* not this part's to free */
    return NULL;
#undef STAGE
}

int qvz_encode_file_devices(const char *in_path, const char *out_path, const qvz_options_t *opts, const int *devices,
                            uint32_t ndev, uint32_t shards, qvz_result_t *res) {
    int rc = QVZ_OK;
    const double t_start = (double)time(NULL);
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_REALTIME, &ts0);
    (void)t_start;
    if (!in_path || !out_path || !opts || !devices || ndev == 0 || ndev > 64) return qvz_job_fail(QVZ_E_ARG, "bad multi-device arguments");
    if (shards < ndev) return qvz_job_fail(QVZ_E_ARG, "%u shards cannot be dealt to %u devices", shards, ndev);
    if (opts->uncompressed_name) {                                             /* main.c:81; the parts fill in their rows */
        FILE *f = fopen(opts->uncompressed_name, "w");
        if (!f) return qvz_job_fail(QVZ_E_ARG, "Unable to open uncompressed file");
        fclose(f);
    }
    if (opts->input_format == QVZ_INPUT_FASTQ) return qvz_job_fail(QVZ_E_ARG, "the multi-device driver reads quality files (extract FASTQ qualities first)");
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return qvz_job_fail(QVZ_E_ARG, "load_file returned error: 1");
    struct stat st;
    if (fstat(fd, &st) || st.st_size == 0) { close(fd); return qvz_job_fail(QVZ_E_DATA, "empty or unreadable input"); }
    const uint8_t *text = (const uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (text == MAP_FAILED) { close(fd); return qvz_job_fail(QVZ_E_NOMEM, "mmap failed"); }

    shared_t sh;
    part_t *parts = NULL;
    pthread_t *th = NULL;
    uint64_t *first = NULL, *rel = NULL;
    uint8_t *centroid_rows = NULL;
    qvz_codebook_t **cbs = NULL;
    FILE *fout = NULL;
    memset(&sh, 0, sizeof(sh));
    pthread_mutex_init(&sh.mu, NULL);
    pthread_cond_init(&sh.cv, NULL);

    /* load_file (lines.c:44-54): columns from the first line, lines = size / (columns + 1) */
    const uint8_t *nl = (const uint8_t *)memchr(text, '\n', (size_t)st.st_size < 1024 ? (size_t)st.st_size : 1024);
    if (!nl) { rc = qvz_job_fail(QVZ_E_DATA, "first line longer than 1022 symbols or missing newline (LF_ERROR_TOO_LONG)"); goto done; }
    const uint32_t C = (uint32_t)(nl - text);
    const uint64_t stride = (uint64_t)C + 1, N = C ? (uint64_t)st.st_size / stride : 0;
    const uint32_t K = (uint32_t)opts->clusters;
    if (C == 0 || N == 0) { rc = qvz_job_fail(QVZ_E_DATA, "no complete line in the input"); goto done; }
    if (K < 1 || K > 255) { rc = qvz_job_fail(QVZ_E_ARG, "clusters must be 1..255"); goto done; }
    if (shards > qvz_shard_limit(N)) { rc = qvz_job_fail(QVZ_E_ARG, "%u shards for %llu reads: a shard needs at least 4 reads", shards, (unsigned long long)N); goto done; }
    /* N itself may exceed 2^32 - 1 here (the shard set's header holds 64 bits, and the counters are summed 64 bits wide):
     * only a single part may not (its u32 counters and the u32 line count of a reference-format shard file) */
    if ((N + ndev - 1) / ndev + 4 >= (1ull << 32)) { rc = qvz_job_fail(QVZ_E_ARG, "%llu lines over %u devices: a device's part must stay below 2^32 lines (pmf.h:35-36)", (unsigned long long)N, ndev); goto done; }

    /* centroid reads (cluster.c:198-204): libc rand() in the process's state, BEFORE the first HIP call (HIP start-up
     * consumes rand() values) - exactly what qvz_job_init does for a whole file */
    centroid_rows = (uint8_t *)malloc((size_t)K * stride);
    first = (uint64_t *)malloc(((size_t)shards + 1) * sizeof(uint64_t));
    rel = (uint64_t *)malloc(((size_t)shards + ndev + 1) * sizeof(uint64_t));
    parts = (part_t *)calloc(ndev, sizeof(part_t));
    th = (pthread_t *)calloc(ndev, sizeof(pthread_t));
    if (!centroid_rows || !first || !rel || !parts || !th) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    {
        uint32_t block_count = (uint32_t)(N / QVZ_LINES_PER_BLOCK);            /* lines.c:93-96 */
        if ((uint64_t)block_count * QVZ_LINES_PER_BLOCK != N) block_count += 1;
        for (uint32_t j = 0; j < K; ++j) {
            const uint32_t block_id = (uint32_t)rand() % block_count;
            const uint64_t left = N - (uint64_t)block_id * QVZ_LINES_PER_BLOCK;
            const uint32_t count = left > QVZ_LINES_PER_BLOCK ? (uint32_t)QVZ_LINES_PER_BLOCK : (uint32_t)left;
            const uint32_t line_id = (uint32_t)rand() % count;
            if (opts->verbose) printf("Chose block %d, line %d.\n", block_id, line_id);
            memcpy(centroid_rows + (size_t)j * stride, text + ((uint64_t)block_id * QVZ_LINES_PER_BLOCK + line_id) * stride, stride);
        }
    }
    uint32_t well_state[32];
    if (opts->seed_mode == QVZ_SEED_EXPLICIT) memcpy(well_state, opts->well_state, sizeof(well_state));
    else {
        srand((uint32_t)time(0));                                              /* qv_stream.c:77-83 */
        for (int i = 0; i < 32; ++i) well_state[i] = opts->seed_mode == QVZ_SEED_DEBUG ? 0x55555555u : (uint32_t)rand();
    }

    /* partition: device d takes shards [d * S / D, (d + 1) * S / D) of the file's cut */
    qvz_shard_bounds(N, shards, first);
    sh.parts = ndev;
    sh.use_rccl = 1;
    for (uint32_t a = 0; a < ndev; ++a)
        for (uint32_t b = a + 1; b < ndev; ++b)
            if (devices[a] == devices[b]) sh.use_rccl = 0;                    /* parts sharing a GPU: RCCL refuses duplicate devices */
    if (getenv("QVZ_MULTI_REDUCE") && strcmp(getenv("QVZ_MULTI_REDUCE"), "host") == 0) sh.use_rccl = 0;
    /* one part has nothing to sum; QVZ_MULTI_REDUCE=rccl sends it through a one-rank communicator all the same (what a
     * one-GPU box can test of the RCCL binding: library, symbols, data-type and operator codes) */
    if (ndev == 1 && !(getenv("QVZ_MULTI_REDUCE") && strcmp(getenv("QVZ_MULTI_REDUCE"), "rccl") == 0)) sh.use_rccl = 0;
    if (sh.use_rccl) {
        if ((rc = rccl_open(&sh.rccl))) goto done;
        sh.comms = (ncclComm_t *)calloc(ndev, sizeof(ncclComm_t));
        if (!sh.comms) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        const int e = sh.rccl.comm_init_all(sh.comms, (int)ndev, devices);
        if (e) { rc = qvz_job_fail(QVZ_E_HIP, "ncclCommInitAll: %s", sh.rccl.error_string ? sh.rccl.error_string(e) : "error"); goto done; }
    } else {
        sh.stage_bytes = (size_t)K * C * A * A * sizeof(uint64_t);            /* the largest reduction: the PMF counters, 64 bits wide */
        sh.stage = (uint8_t *)malloc(sh.stage_bytes * ndev);
        if (!sh.stage) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
    }
    {
        uint64_t *relp = rel;
        for (uint32_t d = 0; d < ndev; ++d) {
            const uint32_t s0 = (uint32_t)((uint64_t)shards * d / ndev), s1 = (uint32_t)((uint64_t)shards * (d + 1) / ndev);
            part_t *p = &parts[d];
            p->sh = &sh; p->part = d; p->device = devices[d]; p->opts = opts;
            p->first_read = first[s0];
            p->reads = first[s1] - first[s0];
            p->text = text + p->first_read * stride;
            p->columns = C;
            p->shards = s1 - s0;
            p->bounds = relp;
            for (uint32_t g = s0; g <= s1; ++g) *relp++ = first[g] - first[s0];
            p->centroid_rows = centroid_rows;
            p->well_state = well_state;
            p->cbs_slot = &cbs;
        }
    }
    for (uint32_t d = 1; d < ndev; ++d)
        if (pthread_create(&th[d], NULL, part_main, &parts[d])) { rc = qvz_job_fail(QVZ_E_NOMEM, "pthread_create"); bar_abort(&sh); th[d] = 0; }
    if (!rc) part_main(&parts[0]);          /* a missing thread would leave the barriers short: the others were released above */
    for (uint32_t d = 1; d < ndev; ++d) if (th[d]) pthread_join(th[d], NULL);
    if (rc) goto done;
    for (uint32_t d = 0; d < ndev; ++d)
        if (parts[d].rc) { rc = qvz_job_fail(parts[d].rc, "%s", parts[d].err); goto done; }

    /* assemble: one shard set of all parts' streams in order; -u text and distortion in file order */
    {
        uint64_t total = 0, at = 0;
        uint64_t *sizes = (uint64_t *)malloc((size_t)shards * sizeof(uint64_t));
        uint8_t *all = NULL;
        double distortion = 0.0;
        qvz_shardset_info_t si;
        if (!sizes) { rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        for (uint32_t d = 0, g = 0; d < ndev; ++d)
            for (uint32_t i = 0; i < parts[d].shards; ++i, ++g) { sizes[g] = parts[d].job.out_sizes[i]; total += sizes[g]; }
        all = (uint8_t *)malloc(total ? total : 1);
        if (!all) { free(sizes); rc = qvz_job_fail(QVZ_E_NOMEM, "host alloc"); goto done; }
        for (uint32_t d = 0; d < ndev; ++d) {
            memcpy(all + at, parts[d].job.out_streams, parts[d].stream_bytes);
            at += parts[d].stream_bytes;
            for (uint64_t i = 0; i < parts[d].reads; ++i) distortion += parts[d].job.out_err[i] / ((double)C);   /* qv_compressor.c:127 */
        }
        distortion /= (double)N;                                               /* qv_compressor.c:139-140 */
        fout = fopen(out_path, "wb");
        if (!fout) { free(sizes); free(all); rc = qvz_job_fail(QVZ_E_ARG, "Unable to open output file"); goto done; }
        rc = qvz_shardset_write(fout, K, C, N, shards, well_state, (const qvz_codebook_t *const *)cbs, all, 0, sizes, &si);
        free(sizes); free(all);
        if (rc) goto done;
        clock_gettime(CLOCK_REALTIME, &ts1);
        const double t_total = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
        if (res) {
            memset(res, 0, sizeof(*res));
            res->lines = N; res->columns = C; res->stream_bytes = si.stream_bytes; res->file_bytes = si.file_bytes;
            res->distortion = distortion; res->kmeans_iterations = parts[0].job.kmeans_iterations;
            res->deferred_decisions = parts[0].job.deferred_decisions;
            res->t_upload = parts[0].job.t_upload; res->t_cluster = parts[0].job.t_cluster; res->t_stats = parts[0].job.t_stats;
            res->t_codebook = parts[0].job.t_codebook; res->t_encode = parts[0].job.t_encode; res->t_total = t_total;
        }
        if (opts->stats)                                                       /* main.c:124-126 */
            printf("rate, %.4f, distortion, %.4f, time, %.4f, size, %llu \n", (si.stream_bytes * 8.) / ((double)N * C), distortion, t_total,
                   (unsigned long long)si.stream_bytes);
    }
done:
    if (fout) fclose(fout);
    if (parts)
        for (uint32_t d = 0; d < ndev; ++d) {
            free(parts[d].job.out_streams); free(parts[d].job.out_sizes); free(parts[d].job.out_err);
            parts[d].job.out_streams = NULL; parts[d].job.out_sizes = NULL; parts[d].job.out_err = NULL;
            if (parts[d].job.C) { qvz_hip_set_device(parts[d].device); qvz_job_release(&parts[d].job); }
        }
    /* all threads are joined: what is still in sh.comms was never aborted (an aborted communicator is already freed) */
    if (sh.comms) { for (uint32_t d = 0; d < ndev; ++d) if (sh.comms[d]) sh.rccl.comm_destroy(sh.comms[d]); free(sh.comms); }
    if (sh.rccl.lib) dlclose(sh.rccl.lib);
    pthread_mutex_destroy(&sh.mu);
    pthread_cond_destroy(&sh.cv);
    free(sh.stage); free(parts); free(th); free(first); free(rel); free(centroid_rows);
    munmap((void *)text, (size_t)st.st_size);
    close(fd);
    return rc;
}
